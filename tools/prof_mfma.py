#!/usr/bin/env python3
"""Phase anatomy of k_fwd_enc / k_dec / k_bwd_enc from the s_memtime stamps of the -DMF_PROF diagnostic build.
usage: IVAE_AMD_LIB=build/libivae_mfprof.so prof_mfma.py cfg"""
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
params = (torch.rand(shape.n_params(), generator=g, device=dev) - .5) * .2
x = torch.randn(B, shape.data_dim, generator=g, device=dev)
u = None
if shape.aux_dim:
    u = torch.zeros(B, shape.aux_dim, device=dev)
    u[torch.arange(B, device=dev), torch.randint(0, shape.aux_dim, (B,), generator=g, device=dev)] = 1
eps = torch.randn(B, shape.latent_dim, generator=g, device=dev)
coef = ops.make_coef(N=N, global_batch=B, **ck)
ws = ops.Workspace()
for _ in range(3):
    ops.forward(shape, coef, params, x, u, eps, ws)
    ops.backward(shape, coef, params, x, u, eps, ws)
buf = torch.zeros(4000, dtype=torch.int64, device=dev)
L = _lib.lib()
L.ivae_debug_set_prof_mf.argtypes = [C.c_void_p]
torch.cuda.synchronize()
if which == "fwd":
    L.ivae_debug_set_prof_mf(C.c_void_p(buf.data_ptr()))
    ops.forward(shape, coef, params, x, u, eps, ws)
else:
    ops.forward(shape, coef, params, x, u, eps, ws)
    torch.cuda.synchronize()
    L.ivae_debug_set_prof_mf(C.c_void_p(buf.data_ptr()))
    ops.backward(shape, coef, params, x, u, eps, ws)
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
