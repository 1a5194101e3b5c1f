#!/usr/bin/env python3
"""One-off, development container only (imports the reference, like tools/make_golden.py; listed in .gpurunignore): the host
data generator ivae_amd.data.generate_data against the reference's data/data.py::generate_data, bit for bit, over
random arguments (sizes, layers, seeds, priors, activations, uncentered, noisy, d_data > d_sources).
usage: fuzz_data_vs_reference.py n_cases"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
if not hasattr(np, "int"):
    np.int = int                      # data/data.py:32 needs the alias on numpy 2
from data.data import generate_data as ref_generate          # noqa: E402  (the reference)
from ivae_amd.data import generate_data                      # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for case in range(n_cases):
    r = np.random.RandomState(case)
    ds = int(r.choice([1, 2, 3, 5, 8]))
    kw = dict(n_per_seg=int(r.choice([1, 7, 50, 200])), n_seg=int(r.choice([1, 2, 5, 13])), d_sources=ds,
              d_data=int(ds + r.choice([0, 0, 1, 3])), n_layers=int(r.choice([1, 2, 3, 5])), seed=int(r.randint(0, 1000)),
              prior=str(r.choice(["gauss", "lap", "hs"])), activation=str(r.choice(["lrelu", "sigmoid", "xtanh", "none"])),
              uncentered=bool(r.randint(2)), noisy=float(r.choice([0., 0., .1])))
    try:
        want = ref_generate(batch_size=0, repeat_linearity=True, **kw)
    except Exception as e:
        print("reference raises", type(e).__name__, str(e)[:80], kw)
        try:
            generate_data(batch_size=0, repeat_linearity=True, **kw)
            print("   ... ours does not")
            bad += 1
        except Exception as e2:
            print("   ours raises", type(e2).__name__)
        continue
    got = generate_data(batch_size=0, repeat_linearity=True, **kw)
    for name, a, b in zip("SXUML", got, want):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape or a.dtype != b.dtype or a.tobytes() != b.tobytes():
            bad += 1
            print("DIFF", name, kw, a.shape, b.shape, a.dtype, b.dtype, float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()) if a.shape == b.shape else "")
print("cases", n_cases, "differences", bad)
