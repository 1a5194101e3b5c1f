#!/usr/bin/env python3
"""Where an epoch of `runner` (family A, ivae.yaml, K packed seeds) spends its wall time: the GPU's launches against the host's
per-step assignment problems (runners/ivae.py:86 scores every batch: K x 625 Hungarian solves per epoch).
usage: time_runner_host.py [n_sims] [epochs]"""
import argparse
import contextlib
import io
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import yaml
from ivae_amd import runners

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
E = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(runners.__file__), "..", "configs", "ivae.yaml")))
cfg["epochs"] = E
config = argparse.Namespace(**cfg)
config.device = torch.device("cuda:0")
t_assign = [0.]
orig = runners._assignment_scores


def timed(cc):
    t0 = time.perf_counter()
    out = orig(cc)
    t_assign[0] += time.perf_counter() - t0
    return out


runners._assignment_scores = timed
with tempfile.TemporaryDirectory() as tmp:
    args = argparse.Namespace(data_path=tmp + "/", seed=0, n_sims=K)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        np.random.seed(0); torch.manual_seed(0)
        t0 = time.perf_counter()
        perfs, lh, ph = runners.runner(args, config)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    rt = [l for l in buf.getvalue().splitlines() if "total runtime" in l]
print(f"n_sims {K}, {E} epochs: wall {wall:.2f} s ({rt}), of which host assignment {t_assign[0]:.2f} s; "
      f"{K * E * 625 * 64 / wall / 1e6:.1f} M samples/s end to end")
