#!/usr/bin/env python3
"""GPU time of ivae_forward and ivae_backward (+k_reduce) per call, by HIP events, for one of time_generic's configs.
usage: IVAE_AMD_LIB=<so> time_kernels.py cfg [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from time_generic import CFG

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
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
tf, tb = [], []
for i in range(n + 5):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    e[0].record()
    ops.forward(shape, coef, params, x, u, eps, ws)
    e[1].record()
    loss, grads = ops.backward(shape, coef, params, x, u, eps, ws)
    e[2].record()
    torch.cuda.synchronize()
    if i >= 5:
        tf.append(e[0].elapsed_time(e[1]))
        tb.append(e[1].elapsed_time(e[2]))
tf.sort(); tb.sort()
print(f"{os.environ.get('IVAE_AMD_LIB', 'default')} {name}: forward {1e3 * tf[len(tf) // 2]:.1f} us  backward+reduce {1e3 * tb[len(tb) // 2]:.1f} us  "
      f"loss {float(loss[0]):.4f} |g| {float(grads.norm()):.4f}")
