#!/usr/bin/env python3
"""Time ivae_pairwise alone (HIP events around repeated calls) at the shapes the BASELINE configs produce, and check the
results against an fp64 torch evaluation of the same sums.
usage: time_pairwise.py [B,d ...]      default: 4096,20 1024,5 64,5 64,10 512,20(rows of a DP shard: 512 x 4096)"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops


def ref(coef, z, g, v, MN):
    z, g, v = z.double(), g.double(), v.double()
    T = -0.5 * (math.log(2 * math.pi) + v.log()[None] + (z[:, None] - g[None]) ** 2 / v[None])      # (R, C, d)
    lse_s = torch.logsumexp(T.sum(2), 1)
    lse_k = torch.logsumexp(T, 1)
    P = torch.softmax(T.sum(2), 1)[:, :, None] * (coef.b - coef.c) + torch.softmax(T, 1) * (coef.c - coef.d)
    rr = (z[:, None] - g[None]) / v[None]
    return dict(lse_s=lse_s, lse_k=lse_k, lq=lse_s - math.log(MN), lqi=(lse_k - math.log(MN)).sum(1), rz=(-P * rr).sum(1),
                cg=(P * rr).sum(0), cv=(-0.5 * P * (1 - (z[:, None] - g[None]) * rr)).sum(0))


def main():
    dev = torch.device("cuda:0")
    cases = [tuple(int(t) for t in a.split(",")) for a in sys.argv[1:]] or [(4096, 4096, 20), (1024, 1024, 5), (64, 64, 5), (64, 64, 10),
                                                                           (512, 4096, 20)]
    cases = [(c[0], c[0], c[1]) if len(c) == 2 else c for c in cases]
    # every shape twice: broad posteriors (variance ~ 1) and sharp ones (variance ~ 1e-3: a trained encoder; a row is then
    # dozens of standard deviations away from most columns)
    for R, Cc, d, vs in [c + (vs,) for c in cases for vs in (1., 1e-3)]:
        gen = torch.Generator(device=dev).manual_seed(1)
        g = torch.randn(Cc, d, generator=gen, device=dev)
        v = (torch.rand(Cc, d, generator=gen, device=dev) * 1.5 + .25) * vs
        z = (g + v.sqrt() * torch.randn(Cc, d, generator=gen, device=dev))[:R].contiguous()
        coef = ops.make_coef(a=1., b=1., c=2., d=1., N=1e6, pairwise=True, global_batch=Cc)
        out = ops.pairwise(coef, z, g, v)
        torch.cuda.synchronize()
        err = ""
        if R * Cc * d <= 1024 * 1024 * 8:
            want = ref(coef, z, g, v, Cc * 1e6)
            err = "  max rel err " + " ".join("%s %.1e" % (k, float(((out[k].double() - want[k]).abs() / (want[k].abs() + 1e-3)).max()))
                                               for k in ("lse_s", "lse_k", "rz", "cg", "cv"))
        n = 50
        for _ in range(5):
            ops.pairwise(coef, z, g, v, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.pairwise(coef, z, g, v, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / n
        print("rows %5d cols %5d d %2d var~%g: %8.1f us per call, %.3f T (i,j,k) elements/s%s" % (R, Cc, d, vs, us, R * Cc * d / us / 1e6, err), flush=True)


if __name__ == "__main__":
    main()
