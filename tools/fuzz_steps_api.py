#!/usr/bin/env python3
"""One-off: ivae_elbo_adam_steps (one call for K steps; per-step loss weights, per-model rates) against the explicit
sequence ivae_elbo_fwd_bwd -> ivae_adam_step per step over random shapes (the shapes of tests/test_gpu_fuzz.py),
packs of 1-3 models, pairwise terms switching on and off between steps, random starting steps and rates.  Both run the
same kernels, so parameters, moments, losses and z must be bit-identical.  usage: fuzz_steps_api.py first last"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ivae_amd import ops
from tests.test_gpu_fuzz import _case

lo, hi = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
bad = 0
for seed in range(lo, hi):
    dd, dl, M, H, L, act, B, pairwise, _ = _case(seed)
    r = np.random.RandomState(7000 + seed)
    nm = int(r.choice([1, 2, 3])); K = int(r.choice([1, 2, 4])); step0 = int(r.choice([0, 3, 500]))
    shape = ops.ModelShape(dd, dl, M, H, L, act, .1)
    P = shape.n_params()
    g = torch.Generator(device=dev).manual_seed(seed)
    p0 = (torch.rand(nm, P, generator=g, device=dev) - .5) * .4
    m0 = torch.rand(nm, P, generator=g, device=dev) * 1e-3 * (step0 > 0)
    v0 = torch.rand(nm, P, generator=g, device=dev) * 1e-5 * (step0 > 0)
    xs = torch.randn(K, nm, B, dd, generator=g, device=dev)
    us = torch.randn(K, nm, B, M, generator=g, device=dev) if M else None
    es = torch.randn(K, nm, B, dl, generator=g, device=dev)
    lrs = [float(t) for t in r.choice([1e-2, 1e-3, 2e-4], nm)]
    per_model_lr = bool(r.randint(2)) and nm > 1
    coefs = []
    for s in range(K):
        pw = bool(r.randint(2)) if pairwise else False
        a, b, c, d = (float(t) for t in r.uniform(.3, 4., 4))
        coefs.append(ops.make_coef(a, b, c, d, 5000., .1, pw, B))
    same_coef = bool(r.randint(2))
    if same_coef:
        coefs = [coefs[0]] * K
    # one call
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    la, za = ops.elbo_adam_steps(shape, coefs[0] if same_coef else coefs, pa, ma, va, xs, us, es, ops.Workspace(), step0,
                                 lrs if per_model_lr else lrs[0], n_models=nm)
    # explicit sequence
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    ws = ops.Workspace()
    lb, zb = [], []
    for s in range(K):
        loss, z, grads = ops.elbo_fwd_bwd(shape, coefs[s], pb, xs[s], None if us is None else us[s], es[s], ws, n_models=nm)
        for m in range(nm):
            ops.adam_step(pb[m], mb[m], vb[m], grads.view(nm, P)[m], lrs[m] if per_model_lr else lrs[0], step0 + s + 1)
        lb.append(loss); zb.append(z)
    lb, zb = torch.stack(lb), torch.stack(zb)
    torch.cuda.synchronize()
    same = [torch.equal(pa, pb), torch.equal(ma, mb), torch.equal(va, vb), torch.equal(la, lb), torch.equal(za.view_as(zb), zb)]
    if not all(same):
        bad += 1
        rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
        print("DIFF", seed, dict(shape=(dd, dl, M, H, L, act), B=B, nm=nm, K=K, step0=step0, per_model_lr=per_model_lr, same_coef=same_coef,
                                 pw=[c.pairwise for c in coefs]), same, "p %.1e m %.1e v %.1e loss %.1e" % (rel(pa, pb), rel(ma, mb), rel(va, vb), rel(la, lb)), flush=True)
print("cases", hi - lo, "not bit-identical:", bad)
