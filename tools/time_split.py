#!/usr/bin/env python3
"""One pack of K models against the same K models as S sub-packs advanced concurrently (a host thread and a HIP stream
each): do the sub-packs' kernels overlap (one pack's HBM-bound Adam under another's latency-bound row chains)?
usage: time_split.py [cfg1|d20|d40] K S"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from ivae_amd.trainer import init_packed_params

cfg, K, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda:0")
d = {"cfg1": 10, "d20": 20, "d40": 40}[cfg]
shape, B = ops.ModelShape(d, d, 40, 50, 3, "lrelu", .1), 64
ck = dict(a=1., b=1., c=1., d=1., pairwise=False, decoder_var=.01)
steps = 100
coef = ops.make_coef(N=40000., global_batch=B, **ck)


def make(k, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    P = init_packed_params(shape, k, 0, dev, xavier=False)
    m, v = torch.zeros_like(P), torch.zeros_like(P)
    xs = torch.randn(steps, k, B, d, generator=g, device=dev)
    lab = torch.randint(0, 40, (steps, k, B), generator=g, device=dev)
    us = torch.zeros(steps, k, B, 40, device=dev).scatter_(3, lab.unsqueeze(-1), 1.)
    eps = torch.randn(steps, k, B, d, generator=g, device=dev)
    return dict(k=k, P=P, m=m, v=v, xs=xs, us=us, eps=eps, ws=ops.Workspace(), z=torch.empty_like(eps), st=torch.cuda.Stream())


def run(p, n, t0):
    with torch.cuda.stream(p["st"]):
        ops.elbo_adam_steps(shape, coef, p["P"], p["m"], p["v"], p["xs"][:n], p["us"][:n], p["eps"][:n], p["ws"], t0, 1e-3,
                            n_models=p["k"], z_out=p["z"][:n])


packs = [make(K // S, i) for i in range(S)]
for p in packs:
    run(p, 10, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(p, steps, 10)) for p in packs]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{cfg} K={K} as {S} sub-packs: {1e6 * dt:8.1f} us/step  {K * B / dt / 1e6:8.2f} M samples/s", flush=True)
