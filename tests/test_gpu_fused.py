"""GPU parity of the fused persistent train-step kernel (ivae_train_steps) against the CPU oracle:
same initial parameters, same batch indices, same eps -> same per-step losses and the same
parameters / Adam moments after K steps."""
import numpy as np
import pytest
import torch

from ivae_amd import ops
from oracle import np_ref
from tests.util import Case, relmax

pytestmark = pytest.mark.gpu


def oracle_run(spec, P0, X, U, idx, eps, lr, coef, step0=0, M0=None, V0=None):
    names = spec.param_names()
    P = {k: v.astype(np.float32).copy() for k, v in P0.items()}
    M = {k: (np.zeros_like(v) if M0 is None else M0[k].copy()) for k, v in P.items()}
    V = {k: (np.zeros_like(v) if V0 is None else V0[k].copy()) for k, v in P.items()}
    losses = []
    for s in range(idx.shape[0]):
        x, u = X[idx[s]], (U[idx[s]] if U is not None else None)
        fw = np_ref.elbo_forward(P, x, u, eps[s], spec, coef, np.float64)
        G = np_ref.elbo_backward(P, x, u, fw, spec, coef, np.float64)
        losses.append(float(fw["loss"]))
        for n in names:
            P[n], M[n], V[n] = np_ref.adam_step(P[n], G[n].astype(np.float32), M[n], V[n], step0 + s + 1, lr)
    return P, M, V, np.array(losses), fw["z"]


def flat(d, names):
    return np.concatenate([d[n].reshape(-1) for n in names])


def make_problem(spec, n_models, n_steps, N=300, seed=0):
    rng = np.random.RandomState(seed)
    X = (rng.randn(n_models, N, spec.data_dim) * 1.5).astype(np.float32)
    U = None
    if spec.aux_dim:
        U = np.zeros((n_models, N, spec.aux_dim), np.float32)
        lab = rng.randint(0, spec.aux_dim, (n_models, N))
        for m in range(n_models):
            U[m, np.arange(N), lab[m]] = 1
    idx = rng.randint(0, N, (n_models, n_steps, 64)).astype(np.int32)
    eps = rng.randn(n_models, n_steps, 64, spec.latent_dim).astype(np.float32)
    return X, U, idx, eps


@pytest.mark.parametrize("name,n_steps", [("ivae_plain_cfg2", 1), ("ivae_plain_cfg2", 5), ("vae_plain_cfg2", 3)])
def test_fused_steps_match_oracle(name, n_steps):
    c = Case(name)
    spec, names = c.spec, c.spec.param_names()
    shape = ops.ModelShape(spec.data_dim, spec.latent_dim, spec.aux_dim, spec.hidden_dim, spec.n_layers,
                           spec.activation, spec.slope)
    assert ops.train_steps_supported(shape, 64)
    n_models = 3
    X, U, idx, eps = make_problem(spec, n_models, n_steps)
    rng = np.random.RandomState(1)
    P0s = [{k: (v * (1 + 0.05 * m) + 0.01 * m * rng.randn(*v.shape)).astype(np.float32) for k, v in c.params.items()}
           for m in range(n_models)]
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    params = t(np.stack([flat(P, names) for P in P0s]))
    m1, v1 = torch.zeros_like(params), torch.zeros_like(params)
    coef = ops.make_coef(decoder_var=spec.decoder_var, global_batch=64)
    loss, z = ops.train_steps(shape, coef, params, m1, v1, t(X), t(U) if U is not None else None, t(idx), 0, 1e-2,
                              eps=t(eps), want_z=True)
    torch.cuda.synchronize()
    ocoef = np_ref.Coef(pairwise=False)
    for m in range(n_models):
        P, M, V, losses, zl = oracle_run(spec, P0s[m], X[m], U[m] if U is not None else None, idx[m], eps[m], 1e-2, ocoef)
        got = loss[m].cpu().numpy()
        # step 0 sees identical parameters: 1e-5; later steps inherit Adam's sensitivity (see test_gpu_parity)
        assert abs(got[0] - losses[0]) <= 1e-5 * abs(losses[0]), (m, got[0], losses[0])
        assert np.all(np.abs(got - losses) <= 2e-3 * np.abs(losses)), (m, got, losses)
        perr = np.abs(params[m].cpu().numpy() - flat(P, names))
        assert perr.max() < 2e-2 * 1e-2 and perr.mean() < (2e-7 if n_steps == 1 else 1e-4), (m, perr.max(), perr.mean())
        if n_steps == 1:
            # exp_avg = 0.1 * grad after one step: EVERY gradient tensor to 1e-5 of its own max-norm (gradients of
            # 3e-2 and 5e+1 coexist in one model, SURVEY.md §7: a whole-arena norm would leave the small ones unguarded)
            gm, gv, off = m1[m].cpu().numpy(), v1[m].cpu().numpy(), 0
            for n in names:
                k = M[n].size
                assert relmax(gm[off:off + k], M[n].reshape(-1)) < 1e-5, (m, n)
                assert relmax(gv[off:off + k], V[n].reshape(-1)) < 2e-5, (m, n)
                off += k
            assert relmax(z[m].cpu().numpy(), zl) < 1e-5


@pytest.mark.parametrize("dd,dl,M", [(5, 5, 40), (5, 5, 0), (4, 4, 40), (6, 6, 40), (7, 7, 40), (7, 7, 0), (8, 8, 40),
                                      (8, 8, 0), (9, 9, 40), (9, 9, 0)])
def test_fused_odd_dims_match_oracle(dd, dl, M):
    """d = 5: u sits behind a hole column of the encoder input tile, heads are 5 wide (the d = 5 end of the
    BASELINE configs[4] sweep).  No reference fixture at this shape: parity is against the fp64 oracle, which the
    fixtures pin at other shapes."""
    from oracle import torch_ref
    spec = np_ref.Spec(dd, dl, M)
    names = spec.param_names()
    shape = ops.ModelShape(dd, dl, M, 50, 3, "lrelu", .1)
    assert ops.train_steps_supported(shape, 64)
    n_models, n_steps = 2, 3
    X, U, idx, eps = make_problem(spec, n_models, n_steps, seed=11)
    P0s = [{k: v.numpy() for k, v in torch_ref.xavier_params(spec, seed=m).items()} for m in range(n_models)]
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    params = t(np.stack([flat(P, names) for P in P0s]))
    m1, v1 = torch.zeros_like(params), torch.zeros_like(params)
    coef = ops.make_coef(decoder_var=spec.decoder_var, global_batch=64)
    loss, z = ops.train_steps(shape, coef, params, m1, v1, t(X), t(U) if U is not None else None, t(idx), 0, 1e-2,
                              eps=t(eps), want_z=True)
    for m in range(n_models):
        P, Mo, V, losses, zl = oracle_run(spec, P0s[m], X[m], U[m] if U is not None else None, idx[m], eps[m], 1e-2,
                                          np_ref.Coef(pairwise=False))
        got = loss[m].cpu().numpy()
        assert abs(got[0] - losses[0]) <= 1e-5 * abs(losses[0]), (m, got[0], losses[0])
        assert np.all(np.abs(got - losses) <= 2e-3 * np.abs(losses)), (m, got, losses)
        perr = np.abs(params[m].cpu().numpy() - flat(P, names))
        assert perr.max() < 2e-2 * 1e-2 and perr.mean() < 1e-4, (m, perr.max(), perr.mean())
        assert relmax(z[m].cpu().numpy(), zl) < 1e-3


@pytest.mark.parametrize("aux", [40, 0])
@pytest.mark.parametrize("dim", [4, 5, 6, 7, 8, 9, 10])
def test_fused_first_step_moments_every_dim(dim, aux):
    """One step at d = 4..10, iVAE (40 segments) and VAE (no u): exp_avg = 0.1 * grad, so every gradient tensor (hole
    column included) is checked to 1e-5 of its own max-norm."""
    from oracle import torch_ref
    spec = np_ref.Spec(dim, dim, aux)
    names = spec.param_names()
    shape = ops.ModelShape(dim, dim, aux, 50, 3, "lrelu", .1)
    X, U, idx, eps = make_problem(spec, 1, 1, seed=12)
    P0 = {k: v.numpy() for k, v in torch_ref.xavier_params(spec, seed=3).items()}
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    params = t(flat(P0, names)[None])
    m1, v1 = torch.zeros_like(params), torch.zeros_like(params)
    coef = ops.make_coef(decoder_var=spec.decoder_var, global_batch=64)
    ops.train_steps(shape, coef, params, m1, v1, t(X), t(U) if aux else None, t(idx), 0, 1e-2, eps=t(eps))
    P, Mo, V, losses, zl = oracle_run(spec, P0, X[0], U[0] if aux else None, idx[0], eps[0], 1e-2, np_ref.Coef(pairwise=False))
    got, off = m1[0].cpu().numpy(), 0
    for n in names:
        k = Mo[n].size
        assert relmax(got[off:off + k], Mo[n].reshape(-1)) < 1e-5, n
        off += k


def test_fused_continues_from_saved_state_and_chunks():
    """200 steps in one call (internally two launches of <=128) == 120 + 80 steps in two calls."""
    c = Case("ivae_plain_cfg2")
    spec, names = c.spec, c.spec.param_names()
    shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
    X, U, idx, eps = make_problem(spec, 2, 200, N=1000, seed=3)
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    P0 = np.stack([flat(c.params, names)] * 2)
    coef = ops.make_coef(decoder_var=.01, global_batch=64)
    pa, ma, va = t(P0), torch.zeros(2, P0.shape[1], device=d), torch.zeros(2, P0.shape[1], device=d)
    la, _ = ops.train_steps(shape, coef, pa, ma, va, t(X), t(U), t(idx), 0, 1e-2, eps=t(eps))
    pb, mb, vb = t(P0), torch.zeros_like(ma), torch.zeros_like(va)
    l1, _ = ops.train_steps(shape, coef, pb, mb, vb, t(X), t(U), t(idx[:, :120]), 0, 1e-2, eps=t(eps[:, :120]))
    l2, _ = ops.train_steps(shape, coef, pb, mb, vb, t(X), t(U), t(idx[:, 120:]), 120, 1e-2, eps=t(eps[:, 120:]))
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.equal(la, torch.cat([l1, l2], 1))
    assert torch.isfinite(la).all() and la[:, -20:].mean() < la[:, :20].mean()      # it trains


def test_fused_philox_eps_is_standard_normal_and_trains():
    shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
    spec = np_ref.Spec(10, 10, 40)
    X, U, idx, _ = make_problem(spec, 4, 64, N=2000, seed=5)
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    from ivae_amd.trainer import init_packed_params
    p = init_packed_params(shape, 4, 0, d)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    coef = ops.make_coef(decoder_var=.01, global_batch=64)
    # lr = 0: parameters frozen, z - g = eps * sqrt(v) -> recover eps statistics over many steps
    zs = []
    for s in range(40):
        _, z = ops.train_steps(shape, coef, p, m, v, t(X), t(U), t(idx[:, :1]), s, 0.0, seed=7, want_z=True)
        fz = ops.forward(shape, coef, p[0], t(X[0][idx[0, 0]]), t(U[0][idx[0, 0]]),
                         torch.zeros(64, 10, device=d), ops.Workspace())
        ws = ops.Workspace()
        ops.forward(shape, coef, p[0], t(X[0][idx[0, 0]]), t(U[0][idx[0, 0]]), torch.zeros(64, 10, device=d), ws)
        g, vv = ws.view("g", 64, 10), ws.view("v", 64, 10)
        zs.append(((z[0] - g) / vv.sqrt()).cpu().numpy())
    e = np.stack(zs).reshape(-1)
    assert abs(e.mean()) < 0.03 and abs(e.std() - 1) < 0.03
    assert abs(np.mean(e ** 3)) < 0.1 and abs(np.mean(e ** 4) - 3) < 0.2
    assert len(np.unique(e)) > 0.99 * e.size                        # draws differ across steps / elements
    loss, _ = ops.train_steps(shape, coef, p, m, v, t(X), t(U), t(idx), 40, 1e-2, seed=7)
    assert torch.isfinite(loss).all() and loss[:, -8:].mean() < loss[:, :8].mean()


def test_out_of_range_row_indices_are_clamped_not_faulted():
    """batch_idx entries outside [0, n_data) read a clamped (valid) row: same result as the clamped table."""
    spec = np_ref.Spec(10, 10, 40)
    shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
    X, U, idx, eps = make_problem(spec, 1, 2, N=100, seed=9)
    bad = idx.copy()
    bad[0, 0, :8] = 100 + np.arange(8)          # past the end
    bad[0, 1, :8] = -1 - np.arange(8)           # negative
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    from ivae_amd.trainer import init_packed_params
    coef = ops.make_coef(decoder_var=.01, global_batch=64)
    out = []
    for table in (bad, np.clip(bad, 0, 99)):
        p = init_packed_params(shape, 1, 0, d)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        loss, _ = ops.train_steps(shape, coef, p, m, v, t(X), t(U), t(table), 0, 1e-2, eps=t(eps))
        out.append((loss.clone(), p.clone()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_unsupported_shape_is_refused():
    shape = ops.ModelShape(5, 5, 40, 100, 3, "xtanh", .1)
    assert not ops.train_steps_supported(shape, 64)
    assert not ops.train_steps_supported(ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1), 32)


def test_missed_wave_handoff_is_reported_as_nan_loss():
    """The forward stage's waves hand tiles to each other through LDS flags with a bounded wait.  A wait that runs out (here
    forced: ivae_debug_fault_inject makes the variance chain skip its flag) must not pass as training: the model's
    losses are NaN from that step on."""
    from ivae_amd import _lib
    spec = np_ref.Spec(10, 10, 40)
    shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
    X, U, idx, eps = make_problem(spec, 2, 3, N=100, seed=4)
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    from ivae_amd.trainer import init_packed_params
    coef = ops.make_coef(decoder_var=.01, global_batch=64)
    p = init_packed_params(shape, 2, 0, d)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    try:
        assert _lib.lib().ivae_debug_fault_inject(1) == 0
        loss, _ = ops.train_steps(shape, coef, p, m, v, t(X), t(U), t(idx), 0, 1e-2, eps=t(eps))
        torch.cuda.synchronize()
    finally:
        _lib.lib().ivae_debug_fault_inject(0)
    assert torch.isnan(loss).all(), loss
    p = init_packed_params(shape, 2, 0, d)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    loss, _ = ops.train_steps(shape, coef, p, m, v, t(X), t(U), t(idx), 0, 1e-2, eps=t(eps))
    assert torch.isfinite(loss).all()


def test_per_model_learning_rates():
    """model_lr: every packed model steps with its own rate -- model 1 at lr/10 equals a separate call at lr/10 bit for bit
    up to the rate factor's rounding (1e-6), model 0 equals the shared-rate call bit for bit."""
    spec = np_ref.Spec(10, 10, 40)
    shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
    X, U, idx, eps = make_problem(spec, 2, 6, N=200, seed=8)
    d = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    from ivae_amd.trainer import init_packed_params
    coef = ops.make_coef(decoder_var=.01, global_batch=64)

    def run(lr, models=slice(0, 2)):
        p = init_packed_params(shape, 2, 0, d)[models].contiguous()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        loss, _ = ops.train_steps(shape, coef, p, m, v, t(X[models]), t(U[models]), t(idx[models]), 0, lr, eps=t(eps[models]))
        return loss, p

    l_mix, p_mix = run([1e-2, 1e-3])
    l_hi, p_hi = run(1e-2)
    l_lo, p_lo = run(1e-3, slice(1, 2))
    assert torch.equal(l_mix[0], l_hi[0]) and torch.equal(p_mix[0], p_hi[0])
    assert torch.allclose(l_mix[1], l_lo[0], rtol=1e-5) and torch.allclose(p_mix[1], p_lo[0], rtol=1e-4, atol=1e-6)
    assert not torch.allclose(p_mix[1], p_hi[1], atol=1e-4)
