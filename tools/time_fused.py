#!/usr/bin/env python3
"""Time k_train_steps (us per step per model set) for a given library build: IVAE_AMD_LIB=<so> time_fused.py [models] [steps]."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from ivae_amd.trainer import FusedPackedTrainer

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
tr = FusedPackedTrainer(shape, M, 64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, 40000, 10, generator=g, device=dev)
u = torch.zeros(M, 40000, 40, device=dev)
u[:, torch.arange(40000), torch.arange(40000) % 40] = 1
tr.set_data(x, u)
tr.run(125)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{os.environ.get('IVAE_AMD_LIB', 'default')}: {1e6 * dt / steps:.2f} us/step  {M * 64 * steps / dt / 1e6:.1f} M samples/s")
