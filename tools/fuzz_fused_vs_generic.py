#!/usr/bin/env python3
"""One-off extension of tests/test_gpu_differential.py (fused kernel against the generic kernels on random settings):
further cases, worst errors printed.  usage: fuzz_fused_vs_generic.py first last"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_differential import run_case

lo, hi = int(sys.argv[1]), int(sys.argv[2])
worst = {}
for case in range(lo, hi):
    e = run_case(case)
    for k in ("loss1", "lossK", "p", "m", "v"):
        worst[k] = max(worst.get(k, 0.), float(e[k]))
    if not e["finite"] or e["loss1"] > 1e-5 or (e["K"] == 1 and max(e["m"], e["v"]) > 2e-5) or e["lossK"] > 1e-3:
        print("FAIL", case, e, flush=True)
print("cases", hi - lo, "worst:", {k: "%.1e" % v for k, v in worst.items()})
