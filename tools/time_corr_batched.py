#!/usr/bin/env python3
"""ivae_corr_batched at the sizes the loops call it with once per epoch: (blocks, rows, d) = (625, 64, 10) one configs[1]
model, (40000, 64, 5) 64 packed ivae.yaml seeds, (160000, 64, 10) a 256-model pack.  usage: time_corr_batched.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops

dev = torch.device("cuda:0")
for nb, n, d in ((625, 64, 10), (40000, 64, 5), (160000, 64, 10), (40000, 64, 40)):
    x = torch.randn(nb, n, d, device=dev)
    y = torch.randn(nb, n, d, device=dev)
    for _ in range(3):
        ops.corr_batched(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.corr_batched(x, y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"ivae_corr_batched blocks={nb} rows={n} d={d}: {us:.1f} us per call, {2 * nb * n * d * 4 / us / 1e6:.3f} TB/s of input", flush=True)
