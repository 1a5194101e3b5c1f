#!/usr/bin/env python3
"""Host time of one ops.train_steps call (argument checks, ctypes, output allocation, the launch itself) -- what stands between a
caller and the GPU at the driver's 20-step launches.  usage: time_call_host.py"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from ivae_amd.trainer import FusedPackedTrainer

dev = torch.device("cuda:0")
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
tr = FusedPackedTrainer(shape, n_models=4, batch=64, lr=1e-2, seed=0, device=dev)
x = torch.randn(4, 2000, 10, device=dev)
u = torch.zeros(4, 2000, 40, device=dev); u[:, :, 0] = 1
tr.set_data(x, u)
tr.run(5)
torch.cuda.synchronize()
ts = []
for _ in range(300):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.run(1)
    ts.append(time.perf_counter() - t0)
ts.sort()
print("FusedPackedTrainer.run(1): host time per call median %.1f us, min %.1f us" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    tr.run(1)
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
