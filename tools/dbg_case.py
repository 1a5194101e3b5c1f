#!/usr/bin/env python3
"""dbg_case.py <fixture>: per-tensor gradient error of one golden case (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_parity import run_case
from tests.util import Case, relmax
c = Case(sys.argv[1])
names = c.spec.param_names()
loss, z, grads, ws = run_case(c)
g = grads.cpu().numpy(); off = 0
np.set_printoptions(linewidth=200, precision=5)
for n in names:
    ref = c.grads[n]; got = g[off:off + ref.size].reshape(ref.shape); off += ref.size
    e = relmax(got, ref)
    print(n, ref.shape, "relmax %.3g" % e)
    if e > 1e-5 and ref.ndim == 2:
        bad = np.abs(got - ref) > 1e-5 * np.abs(ref).max()
        rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
        print("  bad rows", rows[:40], "bad cols", cols[:40])
        r = rows[0]
        print("  got", got[r, :12]); print("  ref", ref[r, :12])
        # does the got row equal some other ref row?
        for rr in range(ref.shape[0]):
            if np.allclose(got[r], ref[rr], rtol=1e-4, atol=1e-7): print("  got row", r, "== ref row", rr)
