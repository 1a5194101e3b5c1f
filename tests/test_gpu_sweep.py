"""Packed sweep driver (BASELINE configs[4]) on the real kernels: a small seeds x dims grid, one pack on the fused
kernel (d=5) and one on the generic MFMA kernels (d=12), GPU-side correlation + host assignment for the MCC."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_small_sweep_trains_and_scores_every_run():
    from ivae_amd import ops
    from ivae_amd.sweep import run_sweep
    dev = torch.device("cuda:0")
    logs = []
    (lo, hi), mcc, t_train, work = run_sweep([5, 12], n_seeds=3, epochs=6, n_seg=40, n_per_seg=48, batch=64,
                                             device=dev, log=logs.append)
    assert (lo, hi) == (0, 6) and mcc.shape == (6,)
    assert np.all(np.isfinite(mcc)) and np.all(mcc > 0.2) and np.all(mcc <= 1.0)
    assert work == 2 * 3 * 6 * (40 * 48 // 64)
    assert any("d=5" in s and "fused" in s for s in logs) and any("d=12" in s and "generic" in s for s in logs)
    assert ops.train_steps_supported(ops.ModelShape(5, 5, 40, 50, 3, "lrelu", .1), 64)


def test_two_rank_split_covers_the_grid_once():
    """Rank shares of the run list are disjoint and complete (no GPU work: the partition rule only)."""
    from ivae_amd.dist import model_range
    runs = [(d, s) for d in (5, 10, 20, 40) for s in range(64)]
    got = []
    for r in range(8):
        lo, hi = model_range(len(runs), r, 8)
        got += runs[lo:hi]
    assert got == runs
