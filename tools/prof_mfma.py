#!/usr/bin/env python3
"""Phase anatomy of k_fwd_enc / k_dec / k_bwd_enc from the s_memtime stamps of the -DMF_PROF diagnostic build.
usage: IVAE_AMD_LIB=/root/repo/build/libivae_mfprof.so [PROF_MODELS=K] prof_mfma.py cfg [fwd|bwd|step] [part]
PROF_MODELS=K packs K copies of the model (the stamps are those of workgroup (0, 0) in the loaded launch)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import _lib, ops
from time_generic import CFG

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1g"
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
dev = torch.device("cuda:0")
shape, B, ck, N = CFG[name]
ck = dict(ck, pairwise=False)
g = torch.Generator(device=dev).manual_seed(0)
K = int(os.environ.get("PROF_MODELS", "1"))
params = (torch.rand(K * shape.n_params(), generator=g, device=dev) - .5) * .2
x = torch.randn(K, B, shape.data_dim, generator=g, device=dev)
u = None
if shape.aux_dim:
    lab = torch.randint(0, shape.aux_dim, (K, B), generator=g, device=dev)
    u = torch.zeros(K, B, shape.aux_dim, device=dev).scatter_(2, lab.unsqueeze(-1), 1.)
eps = torch.randn(K, B, shape.latent_dim, generator=g, device=dev)
if K == 1:
    x, eps, u = x[0], eps[0], None if u is None else u[0]
coef = ops.make_coef(N=N, global_batch=B, **ck)
ws = ops.Workspace()
for _ in range(3):
    ops.forward(shape, coef, params, x, u, eps, ws, n_models=K)
    ops.backward(shape, coef, params, x, u, eps, ws, n_models=K)
buf = torch.zeros(4000, dtype=torch.int64, device=dev)
L = _lib.lib()
L.ivae_debug_set_prof_mf.argtypes = [C.c_void_p]
torch.cuda.synchronize()
if which == "step":
    for _ in range(3):
        ops.elbo_fwd_bwd(shape, coef, params, x, u, eps, ws, n_models=K)
    torch.cuda.synchronize()
    L.ivae_debug_set_prof_mf(C.c_void_p(buf.data_ptr()))
    ops.elbo_fwd_bwd(shape, coef, params, x, u, eps, ws, n_models=K)
elif which == "fwd":
    L.ivae_debug_set_prof_mf(C.c_void_p(buf.data_ptr()))
    ops.forward(shape, coef, params, x, u, eps, ws, n_models=K)
else:
    ops.forward(shape, coef, params, x, u, eps, ws, n_models=K)
    torch.cuda.synchronize()
    L.ivae_debug_set_prof_mf(C.c_void_p(buf.data_ptr()))
    ops.backward(shape, coef, params, x, u, eps, ws, n_models=K)
torch.cuda.synchronize()
L.ivae_debug_set_prof_mf(C.c_void_p(0))
b = buf.cpu().view(-1, 2).tolist()
part = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # 0: slots 0..999 (k_fwd_enc / k_dec), 1: slots 1000.. (k_bwd_enc)
b = b[1000 * part:1000 * part + 1000]
b = [(i, t) for i, t in b if t]
t0 = b[0][1]
prev = t0
tot = {}
for i, t in b:
    print(f"{i:4d} {t - t0:9d} +{t - prev:7d}")
    tot[i] = tot.get(i, 0) + (t - prev)
    prev = t
print("sum by stamp id (ticks before it):", {k: v for k, v in sorted(tot.items())}, "total", prev - t0)
