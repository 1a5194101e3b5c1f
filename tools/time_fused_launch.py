#!/usr/bin/env python3
"""Duration of ONE k_train_steps launch (HIP events) as a function of its number of steps: the fixed cost per launch
(state load / store, LDS set-up, first batch) and the cost per step.  usage: time_fused_launch.py [models]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from ivae_amd.trainer import FusedPackedTrainer

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
tr = FusedPackedTrainer(shape, M, 64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, 40000, 10, generator=g, device=dev)
u = torch.zeros(M, 40000, 40, device=dev)
u[:, torch.arange(40000), torch.arange(40000) % 40] = 1
tr.set_data(x, u)
tr.run(250)
torch.cuda.synchronize()
res = {}
for n in (1, 2, 5, 10, 20, 40, 125):
    ts = []
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tr.run(n)
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1))
    ts.sort()
    res[n] = ts[len(ts) // 2]
    print(f"{n:4d} steps: {res[n]:8.1f} us per launch  ({res[n] / n:7.2f} us/step)", flush=True)
# the library call alone (index slice prepared beforehand): what of the fixed cost is the kernel's own
for n in (1, 20):
    ts = []
    for rep in range(8):
        idx = tr._perm[:, :n * 64].reshape(M, n, 64).contiguous()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.train_steps(shape, tr.coef, tr.params, tr.exp_avg, tr.exp_avg_sq, tr.x, tr.u, idx, tr.t, tr.lr, seed=tr.seed, want_z=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1))
    ts.sort()
    print(f"{n:4d} steps, ivae_train_steps alone: {ts[len(ts) // 2]:8.1f} us", flush=True)
per = (res[125] - res[20]) / 105
print(f"per step {per:.2f} us, fixed {res[20] - 20 * per:.1f} us (from the 20- and 125-step launches)")
