#!/usr/bin/env python3
"""One-off extension of tests/test_gpu_fuzz.py: the same check (ivae_elbo_fwd_bwd against the fp64 oracle at random odd
shapes) for a range of further seeds; failures are listed, not raised.  usage: fuzz_more.py first last"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fuzz as F

lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []
for seed in range(lo, hi):
    try:
        F.test_random_shape_matches_fp64_oracle(seed, None)
    except AssertionError as e:
        bad.append(seed)
        print("FAIL", seed, F._case(seed), str(e)[:300], flush=True)
    except Exception:
        bad.append(seed)
        print("ERROR", seed, F._case(seed), traceback.format_exc()[-600:], flush=True)
    if seed % 50 == 0:
        print("... seed", seed, "failures so far", len(bad), flush=True)
print("checked", hi - lo, "shapes; failures:", bad)
