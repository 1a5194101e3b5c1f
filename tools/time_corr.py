#!/usr/bin/env python3
"""Time ivae_corr (the MCC correlation block of a full-data evaluation) at N = 10^6, d = 20 (BASELINE configs[3]'s eval)
and at N = 40000, d = 10; prints microseconds per call and the HBM rate that reading x and y once amounts to."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops

dev = torch.device("cuda:0")
for n, d in ((1_000_000, 20), (40_000, 10), (1_000_000, 40)):
    x = torch.randn(n, d, device=dev)
    y = torch.randn(n, d, device=dev) + x
    for _ in range(3):
        ops.corr(x, y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        ops.corr(x, y)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print(f"ivae_corr n={n} d={d}: {us:.1f} us per call, {2 * n * d * 4 / us / 1e6:.2f} TB/s (x and y are read once)")
