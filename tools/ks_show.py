#!/usr/bin/env python3
"""ks_show.py tag...: our kernels' call counts and average durations from gpurun_out/ks_<tag>.csv (tools/prof_one.sh)."""
import csv
import os
import sys

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
for t in sys.argv[1:]:
    log = open(os.path.join(root, f"prof_{t}.log")).read().splitlines()
    print(f"== {t}:", next((l[:70] for l in log if "ms/step" in l), "?"))
    tot = 0.
    for r in csv.reader(open(os.path.join(root, f"ks_{t}.csv"))):
        if r and r[0].startswith(("void k_", "k_")):
            print("   ", r[0][:24].ljust(24), r[1].rjust(4), r[3])
            tot += float(r[3])
    print("    sum of averages", round(tot, 2))
