#!/usr/bin/env python3
"""One-off: ivae_corr and ivae_corr_batched against np.corrcoef (float64) at random (n, d), with large column offsets
(mean 1e3 .. 1e4 against unit spread: the centred two-pass moments must not lose the variance), correlated columns, tiny
and huge scales, and a constant column (np.corrcoef gives nan there: 0 / 0).  usage: fuzz_corr.py n_cases"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops

warnings.filterwarnings("ignore")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
worst, bad = 0., 0
for case in range(n_cases):
    r = np.random.RandomState(case)
    n = int(r.choice([2, 3, 5, 63, 64, 65, 255, 1000, 4097, 40000, 100003, 300000]))
    d = int(r.choice([1, 2, 3, 5, 8, 10, 15, 16, 17, 20, 31, 32, 33, 40, 50, 64]))
    off = float(r.choice([0., 1., 1e3, 1e4]))
    scale = float(r.choice([1e-3, 1., 1e3]))
    mix = r.randn(d, d)
    x = ((r.randn(n, d) @ mix) * scale + off * r.randn(d)).astype(np.float32)
    y = ((x.astype(np.float64) - off) @ r.randn(d, d) / scale + .5 * r.randn(n, d) + off).astype(np.float32)
    const = r.rand() < .15 and d > 1
    if const:
        x[:, 0] = 3.25
    ref = np.corrcoef(x.astype(np.float64), y.astype(np.float64), rowvar=False)[:d, d:]
    got = ops.corr(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)).cpu().numpy()
    ok = np.isfinite(ref)
    err = np.abs(got[ok] - ref[ok]).max() if ok.any() else 0.
    nan_same = bool(np.array_equal(np.isnan(got), np.isnan(ref)))
    worst = max(worst, err)
    if err > 1e-9 or not nan_same:
        bad += 1
        print("FAIL corr", case, (n, d), dict(off=off, scale=scale, const=const), "err %.2e" % err, "nan pattern equal:", nan_same, flush=True)
    if n <= 255 and d <= 40:
        nb = 7
        xb = np.stack([np.roll(x, s, 0) * (1 + .1 * s) for s in range(nb)]).astype(np.float32)
        yb = np.stack([np.roll(y, 2 * s, 0) for s in range(nb)]).astype(np.float32)
        gb = ops.corr_batched(torch.from_numpy(xb).to(dev), torch.from_numpy(yb).to(dev)).cpu().numpy()
        for s in range(nb):
            rb = np.corrcoef(xb[s].astype(np.float64), yb[s].astype(np.float64), rowvar=False)[:d, d:]
            okb = np.isfinite(rb)
            eb = np.abs(gb[s][okb] - rb[okb]).max() if okb.any() else 0.
            worst = max(worst, eb)
            if eb > 1e-9 or not np.array_equal(np.isnan(gb[s]), np.isnan(rb)):
                bad += 1
                print("FAIL corr_batched", case, s, (n, d), "err %.2e" % eb, flush=True)
print("cases", n_cases, "failures", bad, "worst abs error %.2e" % worst)
