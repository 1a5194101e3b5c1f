#!/usr/bin/env python3
"""ivae_corr on arrays of 4 GB and more (n = 2^24 + 5, d = 64: the walk's 64-bit masked path from the first row on; the
buffer-descriptor path covers arrays below 4 GB) and at degenerate sizes (n = 2 .. 9, d = 1 .. 3), against float64 numpy
(chunked two-pass moments for the large one).  usage (GPU box): python tools/chk_corr_huge.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops

dev = torch.device("cuda:0")
worst = 0.
for n in (2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 1023, 4099):
    for d in (1, 2, 3, 16, 17):
        r = np.random.RandomState(n * 100 + d)
        x = (r.randn(n, d) * r.uniform(.1, 10, d) + r.uniform(-100, 100, d)).astype(np.float32)
        y = (x @ r.randn(d, d) + r.randn(n, d)).astype(np.float32)
        ref = np.corrcoef(x.astype(np.float64), y.astype(np.float64), rowvar=False)[:d, d:]
        got = ops.corr(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)).cpu().numpy()
        if n == 2:          # two rows: every correlation is +-1 (or nan): compare where numpy is finite
            m = np.isfinite(ref)
            err = np.abs(got[m] - ref[m]).max() if m.any() else 0.
        else:
            err = np.abs(got - ref).max()
        worst = max(worst, err)
        assert err < 1e-9, (n, d, err)
print("small sizes: worst abs error %.2e" % worst, flush=True)

n, d = (1 << 24) + 5, 64
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, d, generator=g, device=dev) * 3 + 50
y = torch.empty(n, d, device=dev)
mix = torch.randn(d, d, generator=g, device=dev) / 8
for a in range(0, n, 1 << 20):
    y[a:a + (1 << 20)] = x[a:a + (1 << 20)] @ mix + torch.randn(min(1 << 20, n - a), d, generator=g, device=dev)
assert n * d * 4 >= 1 << 32
got = ops.corr(x, y).cpu().numpy()
sx = np.zeros(d); sy = np.zeros(d)
for a in range(0, n, 1 << 20):
    sx += x[a:a + (1 << 20)].double().sum(0).cpu().numpy(); sy += y[a:a + (1 << 20)].double().sum(0).cpu().numpy()
mx, my = torch.from_numpy(sx / n).to(dev), torch.from_numpy(sy / n).to(dev)
cxy = np.zeros((d, d)); cxx = np.zeros(d); cyy = np.zeros(d)
for a in range(0, n, 1 << 20):
    xc, yc = x[a:a + (1 << 20)].double() - mx, y[a:a + (1 << 20)].double() - my
    cxy += (xc.T @ yc).cpu().numpy(); cxx += (xc * xc).sum(0).cpu().numpy(); cyy += (yc * yc).sum(0).cpu().numpy()
ref = cxy / np.sqrt(cxx)[:, None] / np.sqrt(cyy)[None, :]
print("n = %d, d = %d (%.2f GB per array): max abs error %.2e" % (n, d, n * d * 4 / 1e9, np.abs(got - ref).max()), flush=True)
assert np.abs(got - ref).max() < 1e-10
