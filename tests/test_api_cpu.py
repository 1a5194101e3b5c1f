"""CPU-side checks: the C-ABI library loads and exports what include/ivae_amd.h declares,
host-only entry points work, and the nn.Module surface matches the reference's
(state_dict keys, init streams, constructor signatures).  No GPU compute."""
import ctypes as C
import inspect
import os
import re

import numpy as np
import pytest
import torch

import ivae_amd
from ivae_amd import _lib, ops
from tests.util import Case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ivae_amd.h")).read()
    declared = set(re.findall(r"\b(ivae_[a-z_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.EXPORTS)
    assert b"gfx950" in L.ivae_version()


@pytest.mark.parametrize("name", ["ivae_plain_cfg2", "vae_plain_cfg2", "cleanivae_cfg1", "cleanvae_bvae",
                                  "ivae_plain_l1", "ivae_plain_l4_none"])
def test_param_count_matches_reference_models(name):
    c = Case(name)
    s = c.spec
    shape = ops.ModelShape(s.data_dim, s.latent_dim, s.aux_dim, s.hidden_dim, s.n_layers, s.activation, s.slope)
    assert shape.n_params() == sum(v.size for v in c.params.values())


def test_bad_shapes_return_error_codes():
    L = _lib.lib()
    n = C.c_int64()
    assert L.ivae_param_count(C.byref(_lib.IvaeDims(0, 10, 10, 40, 50, 3, 1, .1, 1)), C.byref(n)) == -1
    assert L.ivae_param_count(C.byref(_lib.IvaeDims(64, 10, 10, 40, 50, 3, 9, .1, 1)), C.byref(n)) == -2
    assert L.ivae_param_count(C.byref(_lib.IvaeDims(64, 10, 100, 40, 50, 3, 1, .1, 1)), C.byref(n)) == -4
    assert L.ivae_param_count(C.byref(_lib.IvaeDims(64, 10, 10, 40, 50, 3, 1, .1, 1)), None) == -3
    nb = C.c_size_t()
    assert L.ivae_workspace_bytes(C.byref(_lib.IvaeDims(64, 10, 10, 40, 50, 3, 1, .1, 1)), 1, C.byref(nb)) == 0
    assert nb.value > 64 * 400 * 4
    assert L.ivae_corr_workspace_bytes(1, 5, C.byref(nb)) == -1
    assert L.ivae_adam_step(None, None, None, None, 10, .01, .9, .999, 1e-8, 1, None) == -3
    # the many-steps entry point validates before it launches anything: null buffers, step counts, dims
    good = _lib.IvaeDims(64, 10, 10, 40, 50, 3, 1, .1, 1)
    coef = ops.make_coef(global_batch=64)
    args = lambda dims, n, t0: (C.byref(dims), C.byref(coef), None, None, None, None, None, None, None, n, t0, .01, None, .9, .999,
                                1e-8, None, None, None, None, 0, None)
    assert L.ivae_elbo_adam_steps(*args(good, 3, 0)) == -3
    assert L.ivae_elbo_adam_steps(*args(good, 0, 0)) == -1
    assert L.ivae_elbo_adam_steps(*args(good, 3, -1)) == -1
    assert L.ivae_elbo_adam_steps(*args(_lib.IvaeDims(64, 10, 10, 40, 50, 3, 9, .1, 1), 3, 0)) == -2
    # every row tiling / wave count the tuning fields can pin yields a layout (and a workspace size) for the yaml shapes
    for rows, waves, wrows in ((16, 4, 16), (64, 16, 256), (32, 8, 64), (0, 0, 0)):
        dims = _lib.IvaeDims(1024, 5, 5, 0, 100, 3, 2, .1, 1, rows, waves, wrows)
        assert L.ivae_workspace_bytes(C.byref(dims), 1, C.byref(nb)) == 0 and nb.value > 0
    for bad in ((17, 0, 0), (0, 5, 0), (0, 0, 100), (-16, 0, 0)):
        assert L.ivae_workspace_bytes(C.byref(_lib.IvaeDims(1024, 5, 5, 0, 100, 3, 2, .1, 1, *bad)), 1, C.byref(nb)) == -1
    # a batch whose 32-bit byte offsets would overflow is refused, not mis-addressed
    assert L.ivae_workspace_bytes(C.byref(_lib.IvaeDims(600000, 10, 10, 40, 1024, 3, 1, .1, 1)), 0, C.byref(nb)) == -4


def test_plain_workspace_is_linear_in_the_batch():
    """Without the pairwise term nothing in the workspace is quadratic in the batch: a full-dataset forward
    (runners/ivae.py:118-122, N = 40000 ... 1e6 rows in one call) must not pay for the pairwise kernels' scratch."""
    L = _lib.lib()
    size = lambda B, pw: (lambda n: (L.ivae_workspace_bytes(C.byref(_lib.IvaeDims(B, 10, 10, 40, 50, 3, 1, .1, 1)), pw,
                                                            C.byref(n)), n.value))(C.c_size_t())
    (rc1, a), (rc2, b), (rc3, c) = size(4096, 0), size(40960, 0), size(409600, 0)
    assert rc1 == rc2 == rc3 == 0
    # (at most linear: a small batch uses shorter weight-gradient row tiles, i.e. relatively more gradient slabs)
    assert b / a < 10.2 and abs(c / b - 10) < 0.2, (a, b, c)
    rc, q = size(40960, 1)
    assert rc == 0 and q > b + 100e6            # the pairwise scratch is what grows quadratically
    lay0, lay1 = _lib.IvaeWsLayout(), _lib.IvaeWsLayout()
    d = _lib.IvaeDims(4096, 10, 10, 40, 50, 3, 1, .1, 1)
    assert L.ivae_workspace_layout(C.byref(d), 0, C.byref(lay0)) == 0 and L.ivae_workspace_layout(C.byref(d), 1, C.byref(lay1)) == 0
    for f, _ in _lib.IvaeWsLayout._fields_:     # row-tensor offsets do not depend on the flag, only the model stride does
        if f != "per_model":
            assert getattr(lay0, f) == getattr(lay1, f), f
    assert lay1.per_model > lay0.per_model


def _build(c):
    m = c.meta
    torch.manual_seed(0)
    if c.family == "iVAE":
        return ivae_amd.iVAE(m["latent_dim"], m["data_dim"], m["aux_dim"], n_layers=m["n_layers"],
                             hidden_dim=m["hidden_dim"], activation=m["activation"], slope=m["slope"])
    if c.family == "VAE":
        return ivae_amd.VAE(m["latent_dim"], m["data_dim"], n_layers=m["n_layers"], hidden_dim=m["hidden_dim"],
                            activation=m["activation"], slope=m["slope"])
    if c.family == "cleanIVAE":
        return ivae_amd.cleanIVAE(m["data_dim"], m["latent_dim"], m["aux_dim"], n_layers=m["n_layers"],
                                  activation=m["activation"], hidden_dim=m["hidden_dim"], slope=m["slope"])
    return ivae_amd.cleanVAE(m["data_dim"], m["latent_dim"], n_layers=m["n_layers"], activation=m["activation"],
                             hidden_dim=m["hidden_dim"], slope=m["slope"])


@pytest.mark.parametrize("name", ["ivae_plain_tiny", "vae_plain_tiny", "cleanivae_tiny", "cleanvae_tiny",
                                  "ivae_plain_cfg2", "cleanivae_cfg1"])
def test_same_seed_gives_reference_initialisation(name):
    """Fixtures were built with torch.manual_seed(0) right before the reference constructor:
    the drop-in module must consume the RNG identically (nets.py:10-12, 99-106, 432)."""
    c = Case(name)
    model = _build(c)
    sd = model.state_dict()
    assert list(sd.keys()) == list(c.params.keys())
    for k, v in sd.items():
        assert v.shape == c.params[k].shape
        assert np.array_equal(v.numpy(), c.params[k]), k


def test_parameters_alias_one_arena_and_survive_module_ops():
    torch.manual_seed(1)
    m = ivae_amd.iVAE(10, 10, 40)
    flat = m._flat_params()
    assert flat.numel() == 19940
    off = 0
    for p in m.parameters():
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    m.to("cpu")
    m.float()
    sd = {k: torch.randn_like(v) for k, v in before.items()}
    m.load_state_dict(sd)
    flat = m._flat_params()
    off = 0
    for k, p in m.named_parameters():
        assert torch.equal(flat[off:off + p.numel()].view(p.shape), sd[k])
        off += p.numel()
    # Adam (stock torch) updates the views in place -> the arena follows
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    assert torch.allclose(m._flat_params()[:5], sd["logl.fc.0.weight"].view(-1)[:5] - 1e-2, atol=1e-6)


def test_constructor_signatures_match_reference():
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(ivae_amd.cleanIVAE.__init__)[1:] == ["data_dim", "latent_dim", "aux_dim", "n_layers", "activation",
                                                    "hidden_dim", "slope"]
    assert sig(ivae_amd.cleanVAE.__init__)[1:] == ["data_dim", "latent_dim", "n_layers", "activation", "hidden_dim",
                                                   "slope"]
    assert sig(ivae_amd.iVAE.__init__)[1:] == ["latent_dim", "data_dim", "aux_dim", "prior", "decoder", "encoder",
                                               "n_layers", "hidden_dim", "activation", "slope", "device", "anneal"]
    assert sig(ivae_amd.VAE.__init__)[1:] == ["latent_dim", "data_dim", "decoder", "encoder", "n_layers", "hidden_dim",
                                              "activation", "slope", "device"]
    assert sig(ivae_amd.cleanIVAE.elbo)[1:8] == ["x", "u", "N", "a", "b", "c", "d"]
    assert sig(ivae_amd.iVAE.elbo)[1:3] == ["x", "u"]
    assert sig(ivae_amd.iVAE.anneal)[1:] == ["N", "max_iter", "it"]


def test_anneal_schedule_matches_reference_fixture():
    c = Case("ivae_anneal_switch")
    torch.manual_seed(0)
    m = ivae_amd.iVAE(3, 3, 4, hidden_dim=8, anneal=True)
    for k, it in enumerate((623, 624, 625, 626)):
        m.anneal(40000, 1000, it)
        ref = c.meta["steps"][k]
        assert bool(m.anneal_params) == ref["pairwise"]
        if ref["pairwise"]:
            a, b, cc, d, N = m._training_hyperparams
            assert (a, b, cc, d, N) == (ref["a"], ref["b"], ref["c"], ref["d"], ref["N"])


def test_no_cpu_fallback():
    m = ivae_amd.iVAE(3, 3, 4, hidden_dim=8)
    with pytest.raises(RuntimeError):
        m.elbo(torch.randn(5, 3), torch.randn(5, 4))
    # the reference's autograd would return dLoss/dx; the fused path forms parameter gradients only and says so
    with pytest.raises(NotImplementedError):
        m.elbo(torch.randn(5, 3, requires_grad=True), torch.randn(5, 4))
    with pytest.raises(NotImplementedError):
        class Lap:
            name = 'laplace'
        ivae_amd.iVAE(3, 3, 4, prior=Lap())


def test_pairwise_workspace_size_is_reported_without_a_gpu():
    """ivae_pairwise_workspace_bytes is pure host arithmetic: positive for d <= 32 (sliced kernels), zero above
    (wave-per-row kernels need none), error codes for bad shapes."""
    import ctypes as C
    L = _lib.lib()
    n = C.c_size_t()
    assert L.ivae_pairwise_workspace_bytes(1024, 1024, 5, C.byref(n)) == 0 and n.value > 0
    small = n.value
    assert L.ivae_pairwise_workspace_bytes(4096, 4096, 20, C.byref(n)) == 0 and n.value > small
    assert L.ivae_pairwise_workspace_bytes(37, 512, 12, C.byref(n)) == 0 and n.value > 0
    assert L.ivae_pairwise_workspace_bytes(64, 64, 40, C.byref(n)) == 0 and n.value == 0
    assert L.ivae_pairwise_workspace_bytes(0, 64, 5, C.byref(n)) == -1
    assert L.ivae_pairwise_workspace_bytes(64, 64, 65, C.byref(n)) == -4
    assert L.ivae_pairwise_workspace_bytes(64, 64, 5, None) == -3


def test_pack_plateau_scheduler_equals_torch_per_model():
    """ivae_amd.trainer.PlateauLR (one instance for a whole pack) takes the decisions torch's ReduceLROnPlateau takes when
    run once per model (runners/ivae.py:39: factor .1, patience 0; wrappers.py:45: patience 3)."""
    import numpy as np
    import torch
    from ivae_amd.trainer import PlateauLR
    rng = np.random.RandomState(0)
    for patience in (0, 3):
        losses = -500 - np.cumsum(rng.randn(40, 5) * 2 + 0.5, 0)          # noisy descents, negative like the ELBO losses
        losses[25:] += np.linspace(0, 30, 15)[:, None]                   # ... that turn upwards
        ours = PlateauLR(5, 1e-2, factor=0.1, patience=patience)
        ref_opt = [torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=1e-2) for _ in range(5)]
        ref = [torch.optim.lr_scheduler.ReduceLROnPlateau(o, factor=0.1, patience=patience) for o in ref_opt]
        for e in range(40):
            got = ours.step(losses[e])
            for m in range(5):
                ref[m].step(float(losses[e, m]))
                assert got[m] == ref_opt[m].param_groups[0]["lr"], (patience, e, m)
        assert len(set(got)) > 1 or got[0] < 1e-2
