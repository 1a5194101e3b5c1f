#!/usr/bin/env python3
"""One-off: ivae_adam_step against torch.optim.Adam (CPU, the same fp32 tensors) over random sizes, step numbers, rates,
betas, eps and gradient magnitudes (zeros, 1e-20 .. 1e10), several consecutive steps each.  usage: fuzz_adam.py n_cases"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
worst = dict(p=0., m=0., v=0.)
bad = 0
for case in range(n_cases):
    r = np.random.RandomState(case)
    n = int(r.choice([1, 2, 63, 64, 65, 255, 1000, 19940, 100003, 1 << 20]))
    lr = float(r.choice([0., 1e-5, 1e-3, 1e-2, .3]))
    b1 = float(r.choice([0., .5, .9, .99])); b2 = float(r.choice([.9, .99, .999, .9999])); eps = float(r.choice([1e-8, 1e-6, 1e-12]))
    mag = float(r.choice([1e-20, 1e-8, 1e-3, 1., 1e4, 1e10]))
    steps = int(r.choice([1, 2, 5]))
    start = int(r.choice([0, 1, 9, 999, 123456]))
    p0 = torch.from_numpy(r.randn(n).astype(np.float32))
    tp = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([tp], lr=lr, betas=(b1, b2), eps=eps)
    m0 = torch.from_numpy((r.randn(n) * mag * .1).astype(np.float32)) if start else torch.zeros(n)
    v0 = torch.from_numpy((r.rand(n) * mag * mag * .01).astype(np.float32)) if start else torch.zeros(n)
    if start:
        opt.state[tp] = dict(step=torch.tensor(float(start)), exp_avg=m0.clone(), exp_avg_sq=v0.clone())
    pg, mg, vg = p0.clone().to(dev), m0.clone().to(dev), v0.clone().to(dev)
    for s in range(steps):
        g = (r.randn(n) * mag * (r.rand(n) > .2)).astype(np.float32)         # a fifth of the entries exactly zero
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        ops.adam_step(pg, mg, vg, torch.from_numpy(g).to(dev), lr, start + s + 1, b1, b2, eps)
    st = opt.state[tp]
    rel = lambda a, b: float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))
    ep = float((pg.cpu().double() - tp.detach().double()).abs().max())
    em, ev = rel(mg, st["exp_avg"]), rel(vg, st["exp_avg_sq"])
    fin = bool(torch.isfinite(pg).all()) == bool(torch.isfinite(tp).all())
    worst["p"] = max(worst["p"], ep / max(lr, 1e-30)); worst["m"] = max(worst["m"], em); worst["v"] = max(worst["v"], ev)
    if ep > 2e-6 * max(lr, 1e-12) * steps + 1e-7 * float(tp.detach().abs().max()) or em > 1e-6 or ev > 1e-6 or not fin:
        bad += 1
        print("FAIL", case, dict(n=n, lr=lr, b1=b1, b2=b2, eps=eps, mag=mag, steps=steps, start=start), "p abs %.2e (%.2e lr) m %.1e v %.1e finite-equal %s" % (ep, ep / max(lr, 1e-30), em, ev, fin), flush=True)
print("cases", n_cases, "failures", bad, "worst: p error / lr %.1e, m %.1e, v %.1e" % (worst["p"], worst["m"], worst["v"]))
