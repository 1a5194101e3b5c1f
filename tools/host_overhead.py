#!/usr/bin/env python3
"""Host cost of one FusedPackedTrainer.run(20) (what the driver's 20-step timed region is): enqueue time per call,
wall time with the device, and where the Python time goes."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ivae_amd import ops
from ivae_amd.trainer import FusedPackedTrainer
dev = torch.device("cuda:0")
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
M = 256
tr = FusedPackedTrainer(shape, M, 64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, 40000, 10, generator=g, device=dev)
u = torch.zeros(M, 40000, 40, device=dev); u[:, torch.arange(40000), torch.arange(40000) % 40] = 1
tr.set_data(x, u)
tr.run(40); torch.cuda.synchronize()
for timed in (False, True):
    t0 = time.perf_counter()
    for _ in range(20): tr.run(20, timed=timed)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"timed={timed}: host enqueue {1e6*(t1-t0)/20:.1f} us per run(20); with device {1e6*(t2-t0)/20:.1f} us")
# one launch, sync each time (what the driver's 20-step region is)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); tr.run(20, timed=True); torch.cuda.synchronize(); print(f"single run(20)+sync: {1e6*(time.perf_counter()-t0):.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50): tr.run(20, timed=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
