"""Randomised shapes through ivae_elbo_fwd_bwd against the fp64 oracle: odd dimensions (1-wide heads, hidden widths
that are not tile multiples, more than 64 / 128 hidden units -> several column and k chunks), 1..5 layers, every
activation, ragged and single-row batches, with and without the pairwise terms, one and two packed models.
No reference fixture exists at these shapes (said here); the oracle is pinned by the fixtures at others."""
import numpy as np
import pytest
import torch

from ivae_amd import ops
from oracle import np_ref
from tests.util import relmax

pytestmark = pytest.mark.gpu

ACTS = ["lrelu", "xtanh", "sigmoid", "none"]


def _case(seed):
    r = np.random.RandomState(1000 + seed)
    dd = int(r.choice([1, 2, 3, 5, 10, 20, 33, 70]))
    dl = int(r.choice([1, 2, 4, 5, 10, 12, 20, 33]))
    M = int(r.choice([0, 1, 7, 40, 130]))
    H = int(r.choice([1, 3, 17, 50, 64, 65, 100, 130, 200]))
    L = int(r.choice([1, 2, 3, 5]))
    act = ACTS[int(r.randint(4))]
    B = int(r.choice([1, 2, 15, 16, 17, 63, 64, 65, 130, 300]))
    pairwise = bool(r.randint(2))
    n_models = int(r.choice([1, 2]))
    return dd, dl, M, H, L, act, B, pairwise, n_models


WIDE = {100: (10, 10, 40, 600, 3, "lrelu", 70, False, 1),      # hidden widths whose tiles leave the LDS no room for the
        101: (5, 5, 40, 896, 2, "xtanh", 20, True, 1),         # widest workgroups: the layout falls back to fewer waves
        102: (33, 12, 130, 300, 3, "sigmoid", 300, False, 2)}


# launches of at least 128 row tiles: the fused row-chain kernel (k_chain) instead of the three chain kernels -- ragged
# last tile, the two-group (no prior) form with the pairwise split, 32-row tiles, packed models, a single-layer model
CHAIN = {200: ((10, 10, 40, 50, 3, "lrelu", 2100, False, 1), 0),
         201: ((5, 5, 0, 100, 3, "xtanh", 2048, True, 1), 16),
         202: ((20, 20, 200, 50, 3, "lrelu", 4103, False, 1), 32),
         203: ((7, 3, 9, 33, 2, "sigmoid", 1030, False, 2), 0),
         204: ((6, 4, 5, 8, 1, "none", 2100, False, 1), 0),
         205: ((10, 10, 40, 50, 3, "lrelu", 2050, True, 1), 0)}


@pytest.mark.parametrize("seed", list(range(100)) + sorted(WIDE) + sorted(CHAIN) + list(range(300, 800)))
def test_random_shape_matches_fp64_oracle(seed, monkeypatch):
    if seed in CHAIN:
        (dd, dl, M, H, L, act, B, pairwise, n_models), rows = CHAIN[seed]
        monkeypatch.setitem(ops.TUNING, "tile_rows", rows)
    else:
        dd, dl, M, H, L, act, B, pairwise, n_models = WIDE[seed] if seed in WIDE else _case(seed)
    spec = np_ref.Spec(data_dim=dd, latent_dim=dl, aux_dim=M, hidden_dim=H, n_layers=L, activation=act, slope=.1,
                       decoder_var=.1)
    rng = np.random.RandomState(seed)
    names = spec.param_names()
    Ps, xs, us, es = [], [], [], []
    for m in range(n_models):
        P = {}
        for mlp in spec.mlps:
            for li, (i, o) in enumerate(spec.layer_dims(mlp)):
                P[f"{mlp}.fc.{li}.weight"] = (rng.uniform(-1, 1, (o, i)) / np.sqrt(i)).astype(np.float32)
                P[f"{mlp}.fc.{li}.bias"] = (rng.uniform(-1, 1, (o,)) / np.sqrt(i)).astype(np.float32)
        Ps.append(P)
        xs.append(rng.randn(B, dd).astype(np.float32))
        us.append((rng.randn(B, M) * (rng.rand(B, M) < .3)).astype(np.float32))      # dense, not one-hot
        es.append(rng.randn(B, dl).astype(np.float32))
    coef = np_ref.Coef(a=3., b=2., c=1.5 if pairwise else 2., d=.5, N=5000., pairwise=pairwise)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    shape = ops.ModelShape(dd, dl, M, H, L, act, .1)
    co = ops.make_coef(coef.a, coef.b, coef.c, coef.d, coef.N, spec.decoder_var, coef.pairwise, B)
    flat = np.stack([np.concatenate([P[n].reshape(-1) for n in names]) for P in Ps])
    loss, z, grads = ops.elbo_fwd_bwd(shape, co, t(flat), t(np.stack(xs)), t(np.stack(us)) if M else None,
                                      t(np.stack(es)), ops.Workspace(), n_models=n_models)
    g = grads.cpu().numpy()
    for m in range(n_models):
        fw = np_ref.elbo_forward(Ps[m], xs[m], us[m], es[m], spec, coef, np.float64)
        G = np_ref.elbo_backward(Ps[m], xs[m], us[m], fw, spec, coef, np.float64)
        ref = float(fw["loss"])
        assert abs(float(loss[m]) - ref) <= 1e-5 * max(abs(ref), 1.0), (seed, m, float(loss[m]), ref)
        assert relmax(z[m].cpu().numpy(), fw["z"]) < 1e-5
        off = 0
        for n in names:
            r_ = G[n]
            got = g[m, off:off + r_.size].reshape(r_.shape)
            off += r_.size
            scale = np.abs(r_).max()
            if r_.size < 4:     # a one- or two-element gradient (a bias, a weight between 1-wide layers) is a sum over the batch
                # that may cancel to far below its terms: the largest gradient of its MLP sets the scale
                scale = max(scale, max(np.abs(G[k]).max() for k in names if k.split(".")[0] == n.split(".")[0]))
            assert np.abs(got - r_).max() <= 2e-5 * max(scale, 1e-3), (seed, m, n, np.abs(got - r_).max(), np.abs(r_).max())
