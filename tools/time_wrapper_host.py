#!/usr/bin/env python3
"""Wall time of `IVAE_wrapper` (family B) at BASELINE configs[1] against its kernels' time: what the host adds per epoch
(batch-score assignments, logging).  usage: time_wrapper_host.py [max_iter]"""
import contextlib
import io
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import runners
from ivae_amd.data import generate_data

max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
S, X, U, _, _ = generate_data(n_per_seg=1000, n_seg=40, d_sources=10, n_layers=3, seed=1, activation="lrelu", batch_size=0,
                              repeat_linearity=True)
t_assign = [0.]
orig = runners._assignment_scores


def timed(cc):
    t0 = time.perf_counter()
    out = orig(cc)
    t_assign[0] += time.perf_counter() - t0
    return out


runners._assignment_scores = timed
for it in (max_iter // 10, max_iter):          # a short run first: library load, first launches
    t_assign[0] = 0.
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        z, model, params, logger = runners.IVAE_wrapper(X, U, S, batch_size=64, max_iter=it, seed=1, hidden_dim=50, lr=1e-2)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    n_it = -(-it // 625) * 625
    print(f"max_iter {it}: {n_it} iterations, wall {wall:.2f} s = {wall / n_it * 1e6:.1f} us per iteration "
          f"(host assignment {t_assign[0]:.2f} s); {n_it * 64 / wall / 1e6:.2f} M samples/s; final perf {logger.get_last('perf'):.3f}", flush=True)
