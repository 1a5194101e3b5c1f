#!/usr/bin/env python3
"""Packed seeds on the shape-generic kernels: K independent models of BASELINE configs[0] (cleanIVAE d=5 M=40 H=100 xtanh,
batch 64, TC-decomposed loss) or configs[2] (cleanVAE, batch 1024) advanced together by ivae_elbo_adam_steps -- what
`python -m ivae_amd --config ivae.yaml --n-sims K` runs -- or of the plain-ELBO sweep shapes (d, d, 40, 50) lrelu, batch 64
(cfg1: d = 10, d20, d40: what `python -m ivae_amd.sweep` packs).  usage: time_packed.py [cfg0|cfg1|cfg2|d20|d40] [K ...]
IVAE_TUNE="tile_rows,waves,wgrad_rows" overrides the library's tiling choices (0 = keep)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from ivae_amd.trainer import init_packed_params

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg0"
Ks = [int(a) for a in sys.argv[2:]] or [1, 8, 32, 64, 128, 256]
dev = torch.device("cuda:0")
if os.environ.get("IVAE_TUNE"):
    tr, nw, wr = (int(v) for v in os.environ["IVAE_TUNE"].split(","))
    ops.TUNING.update(tile_rows=tr, waves=nw, wgrad_rows=wr)
plain = {"cfg1": 10, "d20": 20, "d40": 40}
if cfg in plain:
    d = plain[cfg]
    shape, B, ck = ops.ModelShape(d, d, 40, 50, 3, "lrelu", .1), 64, dict(a=1., b=1., c=1., d=1., pairwise=False, decoder_var=.01)
elif cfg == "cfg0":
    shape, B, ck = ops.ModelShape(5, 5, 40, 100, 3, "xtanh", .1), 64, dict(a=100., b=1., c=0., d=10., pairwise=True, decoder_var=.1)
else:
    shape, B, ck = ops.ModelShape(5, 5, 0, 100, 3, "xtanh", .1), 1024, dict(a=1000., b=1., c=15., d=1., pairwise=True, decoder_var=.1)
steps = 100
for K in Ks:
    g = torch.Generator(device=dev).manual_seed(0)
    P = init_packed_params(shape, K, 0, dev, xavier=False)
    m, v = torch.zeros_like(P), torch.zeros_like(P)
    xs = torch.randn(steps, K, B, shape.data_dim, generator=g, device=dev)
    us = None
    if shape.aux_dim:
        lab = torch.randint(0, shape.aux_dim, (steps, K, B), generator=g, device=dev)
        us = torch.zeros(steps, K, B, shape.aux_dim, device=dev).scatter_(3, lab.unsqueeze(-1), 1.)
    eps = torch.randn(steps, K, B, shape.latent_dim, generator=g, device=dev)
    coef = ops.make_coef(N=40000., global_batch=B, **ck)
    ws = ops.Workspace()
    z = torch.empty_like(eps)
    ops.elbo_adam_steps(shape, coef, P, m, v, xs[:10], None if us is None else us[:10], eps[:10], ws, 0, 1e-3, n_models=K, z_out=z[:10])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, _ = ops.elbo_adam_steps(shape, coef, P, m, v, xs, us, eps, ws, 10, 1e-3, n_models=K, z_out=z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{cfg} K={K:4d}: {1e6 * dt:8.1f} us/step  {K * B / dt / 1e6:8.2f} M samples/s  loss {float(loss[-1].mean()):.2f}", flush=True)
