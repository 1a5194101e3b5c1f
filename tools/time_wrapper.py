#!/usr/bin/env python3
"""Wall time of the library flow (models/wrappers.py:14-86 == ivae_amd.runners.IVAE_wrapper) at its default shape
(hidden 200, batch 256) and at the headline shape (hidden 50, batch 64: the fused kernel), with a host-side profile."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd.data import generate_data
from ivae_amd.runners import IVAE_wrapper

S, X, U, _, _ = generate_data(1000, 40, 10, n_layers=3, seed=1, prior='gauss', activation='lrelu', repeat_linearity=True)
for kw in (dict(batch_size=256, hidden_dim=200), dict(batch_size=64, hidden_dim=50)):
    for rep in range(2):
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        out = IVAE_wrapper(X, U, S=S, max_iter=2e4, seed=0, lr=1e-2, ckpt_folder="/tmp/ckpt_tw/", log_folder="/tmp/log_tw/", **kw)
        torch.cuda.synchronize()
        pr.disable()
        dt = time.perf_counter() - t0
    print(kw, "20000 iterations: %.2f s wall = %.1f us per iteration" % (dt, dt / 2e4 * 1e6), flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
    print("\n".join(l[:170] for l in s.getvalue().splitlines()[4:26]))
