#!/usr/bin/env python3
"""One-off: ivae_pairwise against explicit fp64 numpy at random (rows, columns, latent) shapes and posterior widths
(broad: v ~ 1; sharp: v ~ 1e-3, where a row lies dozens of standard deviations from most columns and most
exponentials underflow).  Same check as tests/test_gpu_parity.py::test_pairwise_rectangular_matches_oracle.
usage: fuzz_pairwise.py n_cases"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops
from oracle import np_ref

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
worst = {}
bad = []
for case in range(n_cases):
    r = np.random.RandomState(case)
    nr = int(r.choice([1, 2, 3, 31, 63, 64, 65, 100, 255, 256, 257, 700, 1024, 1500]))
    nc = int(r.choice([1, 2, 5, 32, 63, 64, 65, 129, 256, 511, 1000, 1024, 1400]))
    d = int(r.choice([1, 2, 3, 4, 5, 7, 8, 9, 10, 12, 13, 16, 20, 23, 32, 33, 40, 64]))
    sharp = bool(r.randint(2))
    z = r.randn(nr, d).astype(np.float32)
    g = r.randn(nc, d).astype(np.float32)
    v = np.exp((-6. if sharp else 0.) + 0.5 * r.randn(nc, d)).astype(np.float32)
    a, b, c, dd = (float(t) for t in r.uniform(.2, 5., 4))
    co = ops.make_coef(a, b, c, dd, 40000., .1, True, nc)
    out = ops.pairwise(co, *(torch.from_numpy(t).to(dev) for t in (z, g, v)))
    z64, g64, v64 = (t.astype(np.float64) for t in (z, g, v))
    T = np_ref.log_normal(z64[:, None, :], g64[None], v64[None], np.log(2 * np.pi))
    S = T.sum(-1)
    lse_s, lse_k = np_ref.logsumexp(S, 1), np_ref.logsumexp(T, 1)
    logMN = np.log(nc * 40000.)
    Wt = (b - c) * np.exp(S - lse_s[:, None])[:, :, None] + (c - dd) * np.exp(T - lse_k[:, None, :])
    diff = z64[:, None, :] - g64[None]
    rij = diff / v64[None]
    ref = dict(lq=lse_s - logMN, lqi=(lse_k - logMN).sum(-1), lse_s=lse_s, lse_k=lse_k, rz=(Wt * -rij).sum(1), cg=(Wt * rij).sum(0),
               cv=(Wt * -0.5 * (1 - diff * rij)).sum(0))
    for k, rf in ref.items():
        got = out[k].cpu().numpy().astype(np.float64)
        err = np.abs(got - rf).max() / max(np.abs(rf).max(), 1e-30)
        if not np.isfinite(got).all():
            err = float("inf")
        worst[k] = max(worst.get(k, 0.), err)
        if err >= 5e-6:
            bad.append((case, nr, nc, d, sharp, k, err))
            print("FAIL", case, (nr, nc, d), "sharp" if sharp else "broad", k, "%.2e" % err, flush=True)
print("cases", n_cases, "failures", len(bad), "worst relative error per output:", {k: "%.1e" % e for k, e in worst.items()})
