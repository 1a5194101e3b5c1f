"""GPU parity tests: HIP path (through the C ABI) vs golden vectors from the reference and
vs the CPU oracle.  Tolerances: 1e-5 relative for loss and forward tensors; gradients 1e-5
per-tensor max-norm relative (north_star), written next to each check."""
import numpy as np
import pytest
import torch

import ivae_amd
from ivae_amd import ops
from oracle import np_ref
from tests.test_api_cpu import _build
from tests.util import Case, case_names, relmax

pytestmark = pytest.mark.gpu

CASES = case_names()
TOL = 1e-5


def dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


def shape_of(c):
    s = c.spec
    return ops.ModelShape(s.data_dim, s.latent_dim, s.aux_dim, s.hidden_dim, s.n_layers, s.activation, s.slope)


def flat(group, names):
    return np.concatenate([group[n].reshape(-1) for n in names])


def coef_of(c, step, B):
    co = c.coef(step)
    return ops.make_coef(co.a, co.b, co.c, co.d, co.N, c.spec.decoder_var, co.pairwise, B)


def run_case(c, step=0, params=None):
    d = dev()
    names = c.spec.param_names()
    x, u, eps = c.inputs(step)
    P = flat(c.params, names) if params is None else params
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    ws = ops.Workspace()
    loss, z, grads = ops.elbo_fwd_bwd(shape_of(c), coef_of(c, step, x.shape[0]), t(P), t(x),
                                      t(u) if c.spec.aux_dim else None, t(eps), ws)
    torch.cuda.synchronize()
    return loss, z, grads, ws


@pytest.mark.parametrize("name", CASES)
def test_elbo_and_grads_match_reference_fixture(name):
    c = Case(name)
    names = c.spec.param_names()
    loss, z, grads, ws = run_case(c)
    B = c.meta["batch"]
    assert abs(float(loss[0]) - c.loss()) <= TOL * abs(c.loss())
    assert relmax(z.cpu().numpy(), c.z["z"]) < TOL
    for k, w in (("g", c.spec.latent_dim), ("v", c.spec.latent_dim), ("l", c.spec.latent_dim), ("f", c.spec.data_dim)):
        assert relmax(ws.view(k, B, w).cpu().numpy(), c.z[k]) < TOL, k
    for k in ("lpx", "lqc", "lp"):
        assert relmax(ws.view(k, B).cpu().numpy(), c.z[k]) < TOL, k
    g = grads.cpu().numpy()
    off = 0
    for n in names:
        ref = c.grads[n]
        got = g[off:off + ref.size].reshape(ref.shape)
        off += ref.size
        assert relmax(got, ref) < TOL, (n, relmax(got, ref))
    assert off == g.size


@pytest.mark.parametrize("name", ["ivae_plain_cfg2", "cleanivae_cfg1", "cleanvae_btcvae_b200", "ivae_anneal_cfg2"])
def test_against_fp64_oracle(name):
    """Same inputs through oracle/np_ref in float64: tighter than the fp32 fixture."""
    c = Case(name)
    x, u, eps = c.inputs()
    fw = np_ref.elbo_forward(c.params, x, u, eps, c.spec, c.coef(), np.float64)
    G = np_ref.elbo_backward(c.params, x, u, fw, c.spec, c.coef(), np.float64)
    loss, z, grads, ws = run_case(c)
    assert abs(float(loss[0]) - float(fw["loss"])) <= 3e-6 * abs(float(fw["loss"]))
    g = grads.cpu().numpy()
    off = 0
    for n in c.spec.param_names():
        ref = G[n]
        assert relmax(g[off:off + ref.size].reshape(ref.shape), ref) < 5e-6, n
        off += ref.size
    if c.coef().pairwise:
        B = c.meta["batch"]
        assert relmax(ws.view("lq", B).cpu().numpy(), fw["lq"]) < 3e-6
        assert relmax(ws.view("lqi", B).cpu().numpy(), fw["lqi"]) < 3e-6


def test_results_are_bitwise_reproducible():
    c = Case("cleanivae_lrelu_b70")
    a = run_case(c)
    b = run_case(c)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])


@pytest.mark.parametrize("name", ["ivae_plain_cfg2", "ivae_plain_tiny", "cleanvae_tiny"])
def test_adam_kernel_matches_torch_adam(name):
    """ivae_adam_step on the fixture's own gradient == torch.optim.Adam's first step."""
    c = Case(name)
    d = dev()
    names = c.spec.param_names()
    p = torch.from_numpy(flat(c.params, names)).to(d)
    g = torch.from_numpy(flat(c.grads, names)).to(d)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    ops.adam_step(p, m, v, g, 1e-2, 1)
    assert np.abs(p.cpu().numpy() - flat(c.group("adam1.p."), names)).max() < 1e-7
    assert relmax(m.cpu().numpy(), flat(c.group("adam1.m."), names)) < 1e-6
    assert relmax(v.cpu().numpy(), flat(c.group("adam1.v."), names)) < 1e-6


@pytest.mark.parametrize("name", CASES)
def test_module_training_trajectory(name):
    """Drop-in module + FusedAdam reproduce the reference's 4-step trajectory
    (3 steps at lr 1e-2, lr drop to 1e-3, 1 step): elbo -> backward -> step."""
    c = Case(name)
    d = dev()
    model = _build(c)
    if c.family == "iVAE" and any(s["pairwise"] for s in c.meta["steps"]):
        model.anneal_params = True
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c.params.items()})
    model = model.to(d)
    opt = ivae_amd.FusedAdam(model.parameters(), lr=1e-2)
    # (fixtures of round 3 carry their annealing iteration; the first ones are listed here)
    anneal_it = c.meta.get("anneal_it", {"ivae_anneal_tiny": 300, "ivae_anneal_cfg2": 1, "ivae_anneal_switch": 623}.get(name))
    for k in range(4):
        if k == 3:
            for grp in opt.param_groups:
                grp["lr"] = 1e-3
        x, u, eps = (torch.from_numpy(t).to(d) for t in c.inputs(k))
        co = c.coef(k)
        opt.zero_grad()
        if c.family == "iVAE":
            if anneal_it is not None:
                model.anneal(c.meta["N"], c.meta.get("max_iter", 1000), anneal_it + k)
            elbo, z = model.elbo(x, u, eps=eps)
            loss = elbo.mul(-1)
        elif c.family == "VAE":
            elbo, z = model.elbo(x, eps=eps)
            loss = elbo.mul(-1)
        else:
            loss, z = model.elbo(x, u, c.meta["N"], a=co.a, b=co.b, c=co.c, d=co.d, eps=eps)
        loss.backward()
        # losses after the first step see parameters that already differ at the 1e-6 level
        assert abs(loss.item() - c.loss(k)) <= (TOL if k == 0 else 2e-3) * abs(c.loss(k)), k
        if k == 0:
            for n, p in model.named_parameters():
                assert relmax(p.grad.cpu().numpy(), c.grads[n]) < TOL, n
        opt.step()
        if c.has_adam(k + 1):
            for n, p in model.named_parameters():
                ref = c.z[f"adam{k+1}.p.{n}"]
                # Adam normalises every element to a step of ~lr, so an element whose gradient sits
                # at the rounding floor of its tensor may move differently: bound the worst element
                # by a fraction of lr and the typical element tightly.
                err = np.abs(p.detach().cpu().numpy() - ref)
                assert err.max() < 2e-2 * c.meta["lrs"][k], (k, n, err.max())
                assert err.mean() < (1e-7 if k == 0 else 1e-2 * c.meta["lrs"][k]), (k, n, err.mean())
    sd = opt.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"}
    assert int(sd["state"][0]["step"]) == 4


def test_module_works_with_stock_torch_adam_and_forward():
    """runners/ivae.py:38 uses torch.optim.Adam on model.parameters(): must keep working, and
    forward() must return the reference's tuple structure."""
    c = Case("ivae_plain_cfg2")
    d = dev()
    model = _build(c)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in c.params.items()})
    model = model.to(d)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    x, u, eps = (torch.from_numpy(t).to(d) for t in c.inputs(0))
    elbo, z = model.elbo(x, u, eps=eps)
    elbo.mul(-1).backward()
    opt.step()
    for n, p in model.named_parameters():
        err = np.abs(p.detach().cpu().numpy() - c.z[f"adam1.p.{n}"])
        assert err.max() < 2e-4 and err.mean() < 1e-7, n      # see test_module_training_trajectory
    (f, dv), (g, v), z2, (pm, l) = model(x, u, eps=eps)
    assert f.shape == (64, 10) and g.shape == v.shape == z2.shape == l.shape == (64, 10)
    assert float(dv) == pytest.approx(0.01)
    # without eps the module draws its own noise
    e1, z1 = model.elbo(x, u)
    assert torch.isfinite(e1) and z1.shape == (64, 10)


@pytest.mark.parametrize("nr,nc,d", [(37, 101, 6), (300, 1111, 20), (513, 64, 32), (70, 130, 40), (1, 1, 1),
                                     (2100, 2309, 7), (2050, 2048, 20), (2048, 2100, 24)])
def test_pairwise_rectangular_matches_oracle(nr, nc, d):
    """Rows != columns (the data-parallel shard shape): ivae_pairwise vs explicit numpy, for the sliced kernels
    (d <= 32: several column / row slices, ragged last slice, d a multiple of 4 or not; also at extents > 2048, where
    every slice count is at its chip-filling value) and the wave-per-row kernels (d = 40); the degenerate 1 x 1 x 1
    problem too."""
    rng = np.random.RandomState(3)
    z = rng.randn(nr, d).astype(np.float32)
    g = rng.randn(nc, d).astype(np.float32)
    v = np.exp(0.3 * rng.randn(nc, d)).astype(np.float32)
    co = ops.make_coef(2., 3., 1.5, .5, 40000., .1, True, nc)
    out = ops.pairwise(co, *(torch.from_numpy(t).to(dev()) for t in (z, g, v)))
    z64, g64, v64 = (t.astype(np.float64) for t in (z, g, v))
    T = np_ref.log_normal(z64[:, None, :], g64[None], v64[None], np.log(2 * np.pi))
    S = T.sum(-1)
    lse_s = np_ref.logsumexp(S, 1)
    lse_k = np_ref.logsumexp(T, 1)
    logMN = np.log(nc * 40000.)
    Wt = (3. - 1.5) * np.exp(S - lse_s[:, None])[:, :, None] + (1.5 - .5) * np.exp(T - lse_k[:, None, :])
    diff = z64[:, None, :] - g64[None]
    rij = diff / v64[None]
    ref = dict(lq=lse_s - logMN, lqi=(lse_k - logMN).sum(-1), lse_s=lse_s, lse_k=lse_k, rz=(Wt * -rij).sum(1),
               cg=(Wt * rij).sum(0), cv=(Wt * -0.5 * (1 - diff * rij)).sum(0))
    for k, r in ref.items():
        assert relmax(out[k].cpu().numpy(), r) < 5e-6, k


@pytest.mark.parametrize("case", ["ivae_plain_b70", "cleanivae_lrelu_b70"])
def test_packed_models_equal_individual_runs(case):
    """n_models=3 in one launch == three single-model launches (config-5 packing); the pairwise case runs the
    model-batched B x B x d kernels."""
    c = Case(case)
    d = dev()
    names = c.spec.param_names()
    rng = np.random.RandomState(0)
    P0 = flat(c.params, names)
    Ps = np.stack([P0, P0 + 0.01 * rng.randn(P0.size).astype(np.float32), P0 * 0.9])
    xs = np.stack([c.inputs(k)[0] for k in range(3)])
    us = np.stack([c.inputs(k)[1] for k in range(3)])
    es = np.stack([c.inputs(k)[2] for k in range(3)])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    co = coef_of(c, 0, 70)
    loss, z, grads = ops.elbo_fwd_bwd(shape_of(c), co, t(Ps), t(xs), t(us), t(es), ops.Workspace(), n_models=3)
    for m in range(3):
        l1, z1, g1 = ops.elbo_fwd_bwd(shape_of(c), co, t(Ps[m]), t(xs[m]), t(us[m]), t(es[m]), ops.Workspace())
        assert torch.equal(l1[0], loss[m]) and torch.equal(z1, z[m]) and torch.equal(g1, grads[m])


def test_large_pack_through_the_row_chain_agrees_with_small_packs():
    """96 packed models of the configs[1] shape go through k_chain and a one-slab k_wgrad, whose first workgroup per
    model also sums the loss (1152 weight-gradient jobs fill the chip with one 128-row tile per model); 3 models go
    through k_fwd_enc / k_dec / k_bwd_enc and four 16-row gradient slabs, where k_reduce sums it.  z is row-local and
    bit-equal; the losses agree to 1e-6 (the two paths add a row's per-feature density terms in different orders) and
    so do the gradients (their batch sums are split differently)."""
    from ivae_amd.trainer import init_packed_params
    d = dev()
    shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
    K, B = 96, 64
    g = torch.Generator(device=d).manual_seed(5)
    P = init_packed_params(shape, K, 3, d)
    x = torch.randn(K, B, 10, generator=g, device=d)
    lab = torch.randint(0, 40, (K, B), generator=g, device=d)
    u = torch.zeros(K, B, 40, device=d).scatter_(2, lab.unsqueeze(-1), 1.)
    eps = torch.randn(K, B, 10, generator=g, device=d)
    co = ops.make_coef(N=40000., global_batch=B)
    loss, z, grads = ops.elbo_fwd_bwd(shape, co, P, x, u, eps, ops.Workspace(), n_models=K)
    grads = grads.view(K, -1)
    for m0 in (0, 45, 93):
        l3, z3, g3 = ops.elbo_fwd_bwd(shape, co, P.view(K, -1)[m0:m0 + 3].contiguous().view(-1), x[m0:m0 + 3].contiguous(),
                                      u[m0:m0 + 3].contiguous(), eps[m0:m0 + 3].contiguous(), ops.Workspace(), n_models=3)
        assert torch.equal(z3, z[m0:m0 + 3])
        assert relmax(l3.cpu().numpy(), loss[m0:m0 + 3].cpu().numpy()) < 1e-6
        assert relmax(g3.view(3, -1).cpu().numpy(), grads[m0:m0 + 3].cpu().numpy()) < 1e-6


def test_data_parallel_shards_sum_to_full_batch():
    """Row shards with global_batch = B: partial losses / grads add up to the full-batch result."""
    c = Case("ivae_plain_cfg4s")
    d = dev()
    names = c.spec.param_names()
    x, u, eps = c.inputs()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    P = t(flat(c.params, names))
    B = x.shape[0]
    co = coef_of(c, 0, B)
    full = ops.elbo_fwd_bwd(shape_of(c), co, P, t(x), t(u), t(eps), ops.Workspace())
    acc_l, acc_g = 0., 0.
    for lo, hi in ((0, 40), (40, 96)):
        l, z, g = ops.elbo_fwd_bwd(shape_of(c), co, P, t(x[lo:hi]), t(u[lo:hi]), t(eps[lo:hi]), ops.Workspace())
        acc_l, acc_g = acc_l + l, acc_g + g
        assert torch.equal(z, full[1][lo:hi])
    assert abs(float(acc_l[0]) - float(full[0][0])) <= 1e-6 * abs(float(full[0][0]))
    assert relmax(acc_g.cpu().numpy(), full[2].cpu().numpy()) < 1e-6


def test_corr_kernel_matches_numpy_corrcoef(golden_dir):
    z = np.load(golden_dir + "/mcc.npz")
    for i in range(int(z["n_cases"])):
        x, y = z[f"x{i}"], z[f"y{i}"]
        cc = ivae_amd.mcc.corr_matrix_gpu(x, y).cpu().numpy()
        assert np.abs(cc - z[f"cc{i}"]).max() < 1e-12, i      # float64 arithmetic, as numpy
        assert abs(ivae_amd.mean_corr_coef(x, y) - float(z[f"mcc{i}"])) < 1e-12
    with pytest.raises(ValueError):
        ivae_amd.mean_corr_coef(z["x0"], torch.from_numpy(z["y0"]))


def test_corr_at_the_widest_supported_d():
    """d = 64: four column tiles a side in ivae_corr and in ivae_corr_batched (16 accumulator tiles per wave)."""
    rng = np.random.RandomState(3)
    x = rng.randn(700, 64).astype(np.float32)
    y = (x @ rng.randn(64, 64).astype(np.float32) * 0.1 + rng.randn(700, 64)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev())
    ref = np.corrcoef(x.astype(np.float64), y.astype(np.float64), rowvar=False)[:64, 64:]
    assert np.abs(ops.corr(t(x), t(y)).cpu().numpy() - ref).max() < 1e-12
    assert np.abs(ops.corr_batched(t(x)[None], t(y)[None]).cpu().numpy()[0] - ref).max() < 1e-12


@pytest.mark.parametrize("n,d", [(2, 1), (3, 5), (5, 16), (63, 17), (64, 20), (65, 32), (257, 33), (1000, 40), (1027, 48),
                                 (4099, 49), (70001, 10), (300007, 20)])
def test_corr_fragment_tiles_and_row_tails(n, d):
    """The fp64-MFMA moments kernel at every column-tile count (d = 1..64 -> 1..4 tiles, full and ragged), at row counts
    that are no multiple of the 4-row fragment or of the waves' stride, from fewer rows than one workgroup has waves up to
    many rounds of all 256 workgroups; non-zero means (the centring is what keeps float64 exact enough)."""
    rng = np.random.RandomState(n + d)
    x = (rng.randn(n, d) * rng.uniform(.5, 3, d) + rng.uniform(-50, 50, d)).astype(np.float32)
    y = (x @ (rng.randn(d, d) * 0.3) + rng.randn(n, d) + 7).astype(np.float32)
    ref = np.corrcoef(x.astype(np.float64), y.astype(np.float64), rowvar=False)[:d, d:]
    got = ops.corr(torch.from_numpy(x).to(dev()), torch.from_numpy(y).to(dev())).cpu().numpy()
    assert np.abs(got - ref).max() < 1e-12, (n, d)


def test_corr_large_n_property():
    """Full-size property check (N = 40000 x d = 10, the final-eval shape of runners/ivae.py:123):
    corr(x, x) has unit diagonal, corr is invariant to affine maps, symmetric under swap."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(40000, 10, generator=g).to(dev())
    y = (x @ torch.randn(10, 10, generator=g).to(dev())).tanh()
    cxx = ops.corr(x, x)
    assert (cxx.diagonal() - 1).abs().max() < 1e-12
    cxy, cyx = ops.corr(x, y), ops.corr(y, x)
    assert (cxy - cyx.T).abs().max() < 1e-12
    c2 = ops.corr(x * 3 + 2, y * -0.5 + 1)
    assert (c2 + cxy).abs().max() < 1e-6       # sign flips, magnitude kept (fp32 affine map rounding)


def test_bad_inputs_raise():
    c = Case("ivae_plain_tiny")
    d = dev()
    x, u, eps = (torch.from_numpy(t).to(d) for t in c.inputs())
    P = torch.from_numpy(flat(c.params, c.spec.param_names())).to(d)
    bad = ops.ModelShape(3, 100, 4, 8, 3, "lrelu", .1)       # latent_dim beyond the kernels' limit
    with pytest.raises(RuntimeError):
        ops.elbo_fwd_bwd(bad, ops.make_coef(global_batch=5), P, x, u, eps, ops.Workspace())
    with pytest.raises(TypeError):
        ops.elbo_fwd_bwd(shape_of(c), ops.make_coef(global_batch=5), P, x.double(), u, eps, ops.Workspace())
    with pytest.raises(RuntimeError):
        ops.elbo_fwd_bwd(shape_of(c), ops.make_coef(global_batch=5), P, x.cpu(), u, eps, ops.Workspace())


@pytest.mark.parametrize("waves", [4, 8, 16])
@pytest.mark.parametrize("tile_rows", [16, 32, 64])
@pytest.mark.parametrize("name", ["ivae_plain_b70", "cleanivae_cfg1", "cleanvae_btcvae_b200", "ivae_plain_cfg4s",
                                  "ivae_plain_l4_none", "ivae_plain_l1"])
def test_every_row_tiling_matches_reference_fixture(name, tile_rows, waves, monkeypatch):
    """The generic kernels pick 16, 32 or 64 batch rows and 4, 8 or 16 waves per workgroup from the launch geometry
    (each pair is its own wave -> tile map and chunk width); IvaeDims.tile_rows / .waves pin the choice so that
    every variant meets the same 1e-5 bar on ragged batches, H = 100 xtanh, 1- and 4-layer MLPs."""
    monkeypatch.setitem(ops.TUNING, "tile_rows", tile_rows)
    monkeypatch.setitem(ops.TUNING, "waves", waves)
    c = Case(name)
    loss, z, grads, ws = run_case(c)
    assert abs(float(loss[0]) - c.loss()) <= TOL * abs(c.loss())
    assert relmax(z.cpu().numpy(), c.z["z"]) < TOL
    g = grads.cpu().numpy()
    off = 0
    for n in c.spec.param_names():
        ref = c.grads[n]
        assert relmax(g[off:off + ref.size].reshape(ref.shape), ref) < TOL, (n, tile_rows)
        off += ref.size


@pytest.mark.parametrize("family,B,dims", [("plain", 4096, (20, 20, 200, 50)), ("pairwise", 1024, (5, 5, 0, 100)),
                                           ("pairwise", 777, (12, 12, 40, 50))])
def test_large_batch_against_fp64_oracle(family, B, dims):
    """BASELINE configs[3] (d=20, 200 segments, batch 4096) and configs[2] (beta-TCVAE, batch 1024) shapes, plus a
    ragged pairwise batch with d not a multiple of 4: no reference fixture at these sizes (said here), so the bar is
    the fp64 oracle: loss 3e-6, every gradient tensor 1e-5 max-norm relative."""
    dd, dl, M, H = dims
    act = "lrelu" if M else "xtanh"
    spec = np_ref.Spec(data_dim=dd, latent_dim=dl, aux_dim=M, hidden_dim=H, n_layers=3, activation=act, slope=.1,
                       decoder_var=.01 if M else .1)
    rng = np.random.RandomState(B)
    P = {}
    for m in spec.mlps:
        for li, (i, o) in enumerate(spec.layer_dims(m)):
            P[f"{m}.fc.{li}.weight"] = (rng.uniform(-1, 1, (o, i)) / np.sqrt(i)).astype(np.float32)
            P[f"{m}.fc.{li}.bias"] = (rng.uniform(-1, 1, (o,)) / np.sqrt(i)).astype(np.float32)
    x = rng.randn(B, dd).astype(np.float32)
    u = np.zeros((B, M), np.float32)
    if M:
        u[np.arange(B), rng.randint(0, M, B)] = 1
    eps = rng.randn(B, dl).astype(np.float32)
    coef = np_ref.Coef(a=1., b=1., c=1., d=1., N=1e6, pairwise=False) if family == "plain" else \
        np_ref.Coef(a=1000., b=1., c=15., d=1., N=40000., pairwise=True)
    fw = np_ref.elbo_forward(P, x, u, eps, spec, coef, np.float64)
    G = np_ref.elbo_backward(P, x, u, fw, spec, coef, np.float64)
    d = dev()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)
    names = spec.param_names()
    shape = ops.ModelShape(dd, dl, M, H, 3, act, .1)
    co = ops.make_coef(coef.a, coef.b, coef.c, coef.d, coef.N, spec.decoder_var, coef.pairwise, B)
    loss, z, grads = ops.elbo_fwd_bwd(shape, co, t(flat(P, names)), t(x), t(u) if M else None, t(eps), ops.Workspace())
    assert abs(float(loss[0]) - float(fw["loss"])) <= 3e-6 * abs(float(fw["loss"]))
    g = grads.cpu().numpy()
    off = 0
    for n in names:
        ref = G[n]
        assert relmax(g[off:off + ref.size].reshape(ref.shape), ref) < TOL, (n, relmax(g[off:off + ref.size].reshape(ref.shape), ref))
        off += ref.size


@pytest.mark.parametrize("nb,n,d", [(7, 64, 5), (3, 70, 20), (2, 300, 64), (1, 2, 1)])
def test_corr_batched_matches_numpy_corrcoef(nb, n, d):
    """ivae_corr_batched: every block equals np.corrcoef(x, y, rowvar=False)[:d, d:] (metrics/mcc.py:328) in float64,
    and the single-problem entry point ivae_corr, to 1e-12."""
    rng = np.random.RandomState(nb * 100 + d)
    x = rng.randn(nb, n, d).astype(np.float32)
    y = (0.5 * x[..., ::-1] + rng.randn(nb, n, d)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    cc = ops.corr_batched(t(x), t(y)).cpu().numpy()
    for b in range(nb):
        ref = np.corrcoef(x[b].astype(np.float64), y[b].astype(np.float64), rowvar=False)[:d, d:]
        assert np.abs(cc[b] - ref).max() < 1e-12
        one = ops.corr(t(x[b]), t(y[b])).cpu().numpy()
        assert np.abs(cc[b] - one).max() < 1e-12


@pytest.mark.parametrize("n,d", [(65, 5), (64, 15), (40000, 16)])
def test_corr_of_a_constant_column_is_nan_like_numpy(n, d):
    """A column of zero variance: np.corrcoef divides 0 by 0 and np.clip keeps the nan (metrics/mcc.py:328), the
    reference's Hungarian step then raises and its runner scores the batch 0 (runners/ivae.py:85-88).  The kernels must
    hand back the same nan pattern -- a clip by fmin / fmax would turn it into -1, a perfect |correlation| -- in all three
    entry points; the host side then scores such a block 0 (runners._assignment_scores)."""
    from ivae_amd.runners import _assignment_scores
    rng = np.random.RandomState(d)
    x = rng.randn(n, d).astype(np.float32)
    y = (x @ rng.randn(d, d) + rng.randn(n, d)).astype(np.float32)
    x[:, 0] = 3.25
    y[:, d - 1] = -7.5
    with np.errstate(invalid="ignore", divide="ignore"):
        ref = np.corrcoef(x.astype(np.float64), y.astype(np.float64), rowvar=False)[:d, d:]
    assert np.isnan(ref[0]).all() and np.isnan(ref[:, d - 1]).all()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    import ctypes as C
    from ivae_amd import _lib
    xs, ys = t(x), t(y)
    cs = torch.empty(d, d, dtype=torch.float64, device=dev())
    assert _lib.lib().ivae_corr_static(C.c_void_p(xs.data_ptr()), C.c_void_p(ys.data_ptr()), n, d, C.c_void_p(cs.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream(dev()).cuda_stream)) == 0
    outs = [ops.corr(xs, ys).cpu().numpy(), cs.cpu().numpy()]
    if n <= 300:
        outs.append(ops.corr_batched(t(x[None]), t(y[None])).cpu().numpy()[0])
    for got in outs:
        assert np.array_equal(np.isnan(got), np.isnan(ref))
        ok = ~np.isnan(ref)
        assert np.abs(got[ok] - ref[ok]).max() < 1e-12
        assert _assignment_scores([np.abs(got)]) == [0.]


@pytest.mark.parametrize("wgrad_rows", [16, 32, 64, 128, 256])
@pytest.mark.parametrize("name", ["ivae_plain_b70", "cleanvae_btcvae_b200", "ivae_plain_cfg4s"])
def test_every_wgrad_row_tiling_matches_reference_fixture(name, wgrad_rows, monkeypatch):
    """k_wgrad tiles the batch on its own (its rows are the contraction length of a weight gradient);
    IvaeDims.wgrad_rows pins the tile so that every contraction length (one to four k chunks, ragged last tiles, tiles
    longer than the batch) meets the same 1e-5 bar."""
    monkeypatch.setitem(ops.TUNING, "wgrad_rows", wgrad_rows)
    c = Case(name)
    loss, z, grads, ws = run_case(c)
    assert abs(float(loss[0]) - c.loss()) <= TOL * abs(c.loss())
    g = grads.cpu().numpy()
    off = 0
    for n in c.spec.param_names():
        ref = c.grads[n]
        assert relmax(g[off:off + ref.size].reshape(ref.shape), ref) < TOL, (n, wgrad_rows)
        off += ref.size
