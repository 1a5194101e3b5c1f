#!/usr/bin/env python3
"""ivae_corr_batched against np.corrcoef over random (blocks, rows, d): rows 2 .. 300 (partial 4-row groups, fewer rows than
a group), d 1 .. 64 (1 .. 4 column tiles), column offsets and scales, a constant column now and then (numpy: nan in its
row / column).  usage: fuzz_corr_batched.py n_cases"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
worst, bad = 0., 0
for case in range(n_cases):
    r = np.random.RandomState(case)
    nb = int(r.choice([1, 3, 4, 5, 17, 64]))
    n = int(r.choice([2, 3, 4, 5, 7, 31, 32, 33, 63, 64, 65, 100, 256, 300]))
    d = int(r.choice([1, 2, 5, 10, 16, 17, 20, 32, 33, 40, 48, 49, 64]))
    x = (r.randn(nb, n, d) * r.uniform(1e-3, 1e3, d) + r.uniform(-1, 1, d) * r.choice([0., 1e2, 1e4])).astype(np.float32)
    y = (np.einsum("bnd,de->bne", x, r.randn(d, d) / d) + r.randn(nb, n, d) * r.uniform(1e-3, 1e3, d)).astype(np.float32)
    const = r.rand() < .2
    if const:
        y[0, :, d // 2] = 3.25
    got = ops.corr_batched(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)).cpu().numpy()
    with np.errstate(all="ignore"):
        ref = np.stack([np.corrcoef(x[b].astype(np.float64), y[b].astype(np.float64), rowvar=False)[:d, d:] for b in range(nb)])
    if n == 2:          # two rows: +-1 up to rounding (or nan); numpy's own value sits on a knife edge
        m = np.isfinite(ref) & np.isfinite(got)
        err = np.abs(np.abs(got[m]) - 1).max() if m.any() else 0.
    else:
        assert np.array_equal(np.isnan(ref), np.isnan(got)), (case, nb, n, d)
        m = ~np.isnan(ref)
        err = np.abs(got[m] - ref[m]).max() if m.any() else 0.
    worst = max(worst, err)
    if err > 1e-10:
        bad += 1
        print("FAIL", case, (nb, n, d), err, flush=True)
print("cases", n_cases, "failures", bad, "worst abs error %.2e" % worst)
