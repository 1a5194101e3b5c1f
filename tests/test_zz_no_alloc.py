"""ivae_corr_static: the entry point that allocates nothing (scratch inside the library, include/ivae_amd.h).

Called through the C ABI (ctypes) and checked against the existing entry point ivae_corr -- same kernels, same
summation order, so bit-identical (the bar for ivae_corr itself is 1e-12 vs np.corrcoef in float64) -- on two calls in
a row, the second with a SMALLER input, so that whatever the first call left in the shared scratch cannot leak into it.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from ivae_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _static(x, y, cc):
    dev = x.device
    with torch.cuda.device(dev):
        return _lib.lib().ivae_corr_static(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), x.shape[0], x.shape[1],
                                           C.c_void_p(cc.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))


def test_corr_static_equals_corr_on_two_calls_second_smaller():
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(7)
    # (n, d): first a 256-block problem at the largest d the built-in scratch takes, then smaller ones (fewer blocks,
    # fewer columns: most of the scratch keeps the first call's contents), then a ragged tiny one
    for n, d in ((40000, 40), (1000, 10), (37, 3)):
        s = rng.randn(n, d).astype(np.float32)
        x = torch.from_numpy(s @ rng.randn(d, d).astype(np.float32) + 3.0).to(dev)
        y = torch.from_numpy(s + 0.5 * rng.randn(n, d).astype(np.float32)).to(dev)
        ref = ops.corr(x, y)                       # ivae_corr, caller-provided workspace
        cc = torch.full((d, d), float("nan"), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(dev)
        rc = _static(x, y, cc)
        assert rc == 0, (rc, _lib.lib().ivae_last_hip_error())
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(dev) == before
        # torch's counter only sees torch's allocator: the driver's free-memory figure would show a hipMalloc (or
        # hipMallocAsync) inside the library.  Compared around a SECOND call (the first one on a device resolves the
        # scratch symbol and may make the runtime load the code object's data segment).
        free_before = torch.cuda.mem_get_info(dev)[0]
        assert _static(x, y, cc) == 0
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free_before
        assert torch.equal(cc, ref), (n, d, float((cc - ref).abs().max()))
        npref = np.corrcoef(x.cpu().numpy().astype(np.float64), y.cpu().numpy().astype(np.float64), rowvar=False)[:d, d:]
        assert np.abs(cc.cpu().numpy() - npref).max() < 1e-12


def test_corr_static_rejects_sizes_beyond_its_scratch():
    dev = torch.device("cuda:0")
    x = torch.randn(128, 41, device=dev)
    cc = torch.zeros(41, 41, dtype=torch.float64, device=dev)
    rc = _static(x, x, cc)
    assert rc == -4                                # IVAE_E_TOOLARGE
    assert b"ivae_corr_static" in _lib.lib().ivae_last_hip_error()
    assert float(cc.abs().sum()) == 0.0            # nothing was launched
    assert _static(x[:1, :5].contiguous(), x[:1, :5].contiguous(), cc) == -1     # n < 2: IVAE_E_BADDIMS


def test_corr_static_runs_under_graph_capture():
    """The callers it exists for: under HIP graph capture an allocation fails the capture; ivae_corr_static captures and
    replays (after one call outside the capture that resolves the scratch symbol), and the replay equals ivae_corr."""
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.randn(5000, 10).astype(np.float32)).to(dev)
    y = torch.from_numpy(rng.randn(5000, 10).astype(np.float32)).to(dev) + x
    cc = torch.zeros(10, 10, dtype=torch.float64, device=dev)
    assert _static(x, y, cc) == 0
    torch.cuda.synchronize()
    cc.zero_()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            rc = _static(x, y, cc)
    assert rc == 0, (rc, _lib.lib().ivae_last_hip_error())
    torch.cuda.synchronize()
    x.mul_(2.0).add_(y)                             # new inputs in the captured buffers
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(cc, ops.corr(x, y))
