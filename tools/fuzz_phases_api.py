#!/usr/bin/env python3
"""One-off: the phase entry points (ivae_forward -> [ivae_pairwise] -> ivae_backward, what a data-parallel caller runs)
against ivae_elbo_fwd_bwd over the random shapes of tests/test_gpu_fuzz.py, single model: loss, z and gradients to 1e-6
(the one-call form may run the decoder's backward inside its forward launch and the fused row-chain kernel: same
arithmetic, the tolerance is for the in-workgroup pairwise kernel's different partial-sum grouping).
usage: fuzz_phases_api.py first last"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops
from tests.test_gpu_fuzz import _case

lo, hi = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
bad, worst = 0, 0.
for seed in range(lo, hi):
    dd, dl, M, H, L, act, B, pairwise, _ = _case(seed)
    shape = ops.ModelShape(dd, dl, M, H, L, act, .1)
    g = torch.Generator(device=dev).manual_seed(seed)
    p = (torch.rand(shape.n_params(), generator=g, device=dev) - .5) * .4
    x = torch.randn(B, dd, generator=g, device=dev)
    u = torch.randn(B, M, generator=g, device=dev) if M else None
    e = torch.randn(B, dl, generator=g, device=dev)
    co = ops.make_coef(2., 3., 1.5, .5, 5000., .1, pairwise, B)
    l1, z1, g1 = ops.elbo_fwd_bwd(shape, co, p, x, u, e, ops.Workspace())
    ws = ops.Workspace()
    z2 = ops.forward(shape, co, p, x, u, e, ws)
    if pairwise:
        into = {k: ws.view(k, B) for k in ("lq", "lqi", "lse_s")}
        into.update({k: ws.view(k, B, dl) for k in ("lse_k", "rz", "cg", "cv")})
        ops.pairwise(co, z2, ws.view("g", B, dl).clone(), ws.view("v", B, dl).clone(), out=into)
    l2, g2 = ops.backward(shape, co, p, x, u, e, ws)
    torch.cuda.synchronize()
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    el, ez, eg = rel(l2, l1), rel(z2, z1), rel(g2, g1)
    worst = max(worst, el, ez, eg)
    if max(el, ez, eg) > 1e-6:
        bad += 1
        print("DIFF", seed, (dd, dl, M, H, L, act, B, pairwise), "loss %.1e z %.1e grads %.1e" % (el, ez, eg), flush=True)
print("cases", hi - lo, "beyond 1e-6:", bad, "worst %.1e" % worst)
