#!/usr/bin/env python3
"""A case of tests/test_gpu_differential.py against the fp64 oracle: first-step gradients of the fused and of the generic
kernels, per parameter tensor.  usage: dbg_diff_case.py case"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops
from ivae_amd.trainer import init_packed_params
from oracle import np_ref

case = int(sys.argv[1])
dev = torch.device("cuda:0")
r = np.random.RandomState(case)
d = int(r.randint(4, 11)); M = int(r.choice([0, 40])); slope = float(r.choice([0., .1, .5, 1.])); dvar = float(r.choice([.01, .1, 1.]))
a, b, dd = (float(t) for t in r.uniform(.3, 3., 3))
K = int(r.choice([1, 2, 5, 9])); step0 = int(r.choice([0, 1, 17, 1000]))
nm, N = 3, 300
lrs = [float(t) for t in r.choice([1e-2, 1e-3, 3e-4], nm)]
print(dict(d=d, M=M, slope=slope, dvar=dvar, a=a, b=b, dd=dd, K=K, step0=step0))
shape = ops.ModelShape(d, d, M, 50, 3, "lrelu", slope)
g = torch.Generator(device=dev).manual_seed(case)
P0 = init_packed_params(shape, nm, case, dev)
if step0:
    torch.rand(P0.shape, generator=g, device=dev); torch.rand(P0.shape, generator=g, device=dev)
X = torch.randn(nm, N, d, generator=g, device=dev)
U = None
if M:
    lab = torch.randint(0, M, (nm, N), generator=g, device=dev)
    U = torch.zeros(nm, N, M, device=dev).scatter_(2, lab.unsqueeze(-1), 1.) * (1. + .1 * torch.randn(nm, N, 1, generator=g, device=dev))
idx = torch.randint(0, N, (nm, K, 64), generator=g, device=dev, dtype=torch.int32)
eps = torch.randn(nm, K, 64, d, generator=g, device=dev)
coef = ops.make_coef(a, b, 1., dd, 40000., dvar, False, 64)
z = torch.zeros_like(P0)
Pf, mf, vf = P0.clone(), z.clone(), z.clone()
ops.train_steps(shape, coef, Pf, mf, vf, X, U, idx[:, :1].contiguous(), 0, 1e-3, eps=eps[:, :1].contiguous())
spec = np_ref.Spec(data_dim=d, latent_dim=d, aux_dim=M, hidden_dim=50, n_layers=3, activation="lrelu", slope=slope, decoder_var=dvar)
names = spec.param_names()
oc = np_ref.Coef(a=a, b=b, c=1., d=dd, N=40000., pairwise=False)
for m in range(nm):
    li = idx[m, 0].long()
    x = X[m][li]; u = None if U is None else U[m][li]; e = eps[m, 0]
    loss, zz, gg = ops.elbo_fwd_bwd(shape, coef, P0[m].contiguous(), x, u, e, ops.Workspace())
    flat = P0[m].cpu().numpy().astype(np.float64)
    P, off = {}, 0
    for n in names:
        mlp, _, l, kind = n.split(".")
        i, o = spec.layer_dims(mlp)[int(l)]
        shp = (o, i) if kind == "weight" else (o,)
        P[n] = flat[off:off + int(np.prod(shp))].reshape(shp); off += int(np.prod(shp))
    fw = np_ref.elbo_forward(P, x.cpu().numpy(), None if u is None else u.cpu().numpy(), e.cpu().numpy(), spec, oc, np.float64)
    G = np_ref.elbo_backward(P, x.cpu().numpy(), None if u is None else u.cpu().numpy(), fw, spec, oc, np.float64)
    gen = gg.cpu().numpy().astype(np.float64); fus = (mf[m] / 0.1).cpu().numpy().astype(np.float64)
    off = 0
    print("model", m, "loss oracle %.6f generic %.6f" % (fw["loss"], float(loss[0])))
    for n in names:
        ref = G[n].reshape(-1); k = ref.size
        sc = np.abs(ref).max()
        eg, ef = np.abs(gen[off:off + k] - ref).max() / sc, np.abs(fus[off:off + k] - ref).max() / sc
        flag = " <-- " if max(eg, ef) > 1e-5 else ""
        print("  %-18s |g|max %.3e  generic %.1e  fused %.1e%s" % (n, sc, eg, ef, flag))
        off += k
    # smallest |pre-activation| of every hidden layer (fp64): a value within fp32 rounding of the kink of lrelu can fall
    # on either side in an fp32 implementation, which flips one row's derivative there from 1 to the slope
    for mlp, inp in (("g", np.concatenate([x.cpu().numpy()] + ([] if u is None else [u.cpu().numpy()]), 1)), ("logv", None), ("f", fw["z"])):
        if inp is None:
            inp = np.concatenate([x.cpu().numpy()] + ([] if u is None else [u.cpu().numpy()]), 1)
        h = inp.astype(np.float64)
        for l in range(2):
            p = h @ P[f"{mlp}.fc.{l}.weight"].T + P[f"{mlp}.fc.{l}.bias"]
            i = np.unravel_index(np.abs(p).argmin(), p.shape)
            print(f"    {mlp} layer {l}: min |pre-activation| {np.abs(p).min():.2e} (row {i[0]}, unit {i[1]}), typical {np.abs(p).mean():.2f}")
            h = np.where(p > 0, p, slope * p)
