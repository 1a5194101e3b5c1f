#!/usr/bin/env python3
"""Stamps of one k_wgrad job (-DMF_PROF build, job blockIdx = (0, 0, MF_PROF_Z)): cycles between
0 kernel entry, 1 job set-up done, 2 first loads issued, 3 they arrived, 4 contraction done, 5 stores issued.
usage: IVAE_AMD_LIB=/root/repo/build/libivae_mfprof.so [PROF_MODELS=K] prof_wgrad.py cfg"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import _lib, ops
from time_generic import CFG

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
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
coef = ops.make_coef(N=N, global_batch=B, **ck)
ws = ops.Workspace()
for _ in range(3):
    ops.forward(shape, coef, params, x, u, eps, ws, n_models=K)
    ops.backward(shape, coef, params, x, u, eps, ws, n_models=K)
buf = torch.zeros(4000, dtype=torch.int64, device=dev)
L = _lib.lib()
L.ivae_debug_set_prof_mf.argtypes = [C.c_void_p]
for rep in range(3):
    ops.forward(shape, coef, params, x, u, eps, ws, n_models=K)
    torch.cuda.synchronize()
    buf.zero_()
    L.ivae_debug_set_prof_mf(C.c_void_p(buf.data_ptr()))
    ops.backward(shape, coef, params, x, u, eps, ws, n_models=K)
    torch.cuda.synchronize()
    L.ivae_debug_set_prof_mf(C.c_void_p(0))
    b = buf.cpu().view(-1, 2)[1500:1509, 1].tolist()
    print("k_wgrad job stamps (cycles from entry; 6 = decoded, 7 = job descriptor filled):", {i: t - b[0] for i, t in zip((0, 1, 2, 3, 4, 5, 6, 7, 8), b)})
