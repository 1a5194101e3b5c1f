#!/usr/bin/env python3
"""A seed of tests/test_gpu_fuzz.py in detail: per-tensor gradient error of ivae_elbo_fwd_bwd against the fp64 oracle, the
same for the oracle evaluated in fp32 (what ANY fp32 implementation's rounding does to this case), and -- for lrelu -- the
smallest |pre-activation| of every hidden layer (a value within fp32 rounding of the kink falls on either side of it).
usage: dbg_fuzz_case.py seed [seed ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops
from oracle import np_ref
from tests import test_gpu_fuzz as F

for seed in (int(a) for a in sys.argv[1:]):
    dd, dl, M, H, L, act, B, pairwise, n_models = F._case(seed)
    print("seed", seed, dict(dd=dd, dl=dl, M=M, H=H, L=L, act=act, B=B, pairwise=pairwise, n_models=n_models))
    spec = np_ref.Spec(data_dim=dd, latent_dim=dl, aux_dim=M, hidden_dim=H, n_layers=L, activation=act, slope=.1, decoder_var=.1)
    rng = np.random.RandomState(seed)
    names = spec.param_names()
    Ps, xs, us, es = [], [], [], []
    for m in range(n_models):           # the draws of the test, in its order
        P = {}
        for mlp in spec.mlps:
            for li, (i, o) in enumerate(spec.layer_dims(mlp)):
                P[f"{mlp}.fc.{li}.weight"] = (rng.uniform(-1, 1, (o, i)) / np.sqrt(i)).astype(np.float32)
                P[f"{mlp}.fc.{li}.bias"] = (rng.uniform(-1, 1, (o,)) / np.sqrt(i)).astype(np.float32)
        Ps.append(P)
        xs.append(rng.randn(B, dd).astype(np.float32))
        us.append((rng.randn(B, M) * (rng.rand(B, M) < .3)).astype(np.float32))
        es.append(rng.randn(B, dl).astype(np.float32))
    coef = np_ref.Coef(a=3., b=2., c=1.5 if pairwise else 2., d=.5, N=5000., pairwise=pairwise)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    shape = ops.ModelShape(dd, dl, M, H, L, act, .1)
    co = ops.make_coef(coef.a, coef.b, coef.c, coef.d, coef.N, spec.decoder_var, coef.pairwise, B)
    flat = np.stack([np.concatenate([P[n].reshape(-1) for n in names]) for P in Ps])
    loss, z, grads = ops.elbo_fwd_bwd(shape, co, t(flat), t(np.stack(xs)), t(np.stack(us)) if M else None, t(np.stack(es)),
                                      ops.Workspace(), n_models=n_models)
    g = grads.cpu().numpy()
    for m in range(n_models):
        fw = np_ref.elbo_forward(Ps[m], xs[m], us[m], es[m], spec, coef, np.float64)
        G = np_ref.elbo_backward(Ps[m], xs[m], us[m], fw, spec, coef, np.float64)
        fw32 = np_ref.elbo_forward(Ps[m], xs[m], us[m], es[m], spec, coef, np.float32)
        G32 = np_ref.elbo_backward(Ps[m], xs[m], us[m], fw32, spec, coef, np.float32)
        print(" model", m, "loss fp64 %.6f  fp32 oracle %.6f  GPU %.6f" % (fw["loss"], fw32["loss"], float(loss[m])))
        gmax = max(np.abs(G[n]).max() for n in names)
        off = 0
        for n in names:
            r_ = G[n]
            got = g[m, off:off + r_.size].reshape(r_.shape); off += r_.size
            sc = np.abs(r_).max()
            e_gpu, e_32 = np.abs(got - r_).max(), np.abs(G32[n].astype(np.float64) - r_).max()
            if e_gpu > 1e-5 * max(sc, 1e-3):
                print("   %-18s |g|max %.3e (model max %.3e)  GPU err %.2e  fp32-oracle err %.2e" % (n, sc, gmax, e_gpu, e_32))
        if act == "lrelu":
            xin = np.concatenate([xs[m], us[m]], 1) if M else xs[m]
            for mlp, inp in (("g", xin), ("logv", xin), ("f", fw["z"])) + ((("logl", us[m]),) if M else ()):
                h = inp.astype(np.float64)
                for l in range(L - 1):
                    p = h @ Ps[m][f"{mlp}.fc.{l}.weight"].T.astype(np.float64) + Ps[m][f"{mlp}.fc.{l}.bias"]
                    i = np.unravel_index(np.abs(p).argmin(), p.shape)
                    if np.abs(p).min() < 1e-6 * np.abs(p).mean():
                        print(f"   {mlp} layer {l}: min |pre-activation| {np.abs(p).min():.2e} (row {i[0]}, unit {i[1]}), typical {np.abs(p).mean():.2f}")
                    h = np.where(p > 0, p, .1 * p)
