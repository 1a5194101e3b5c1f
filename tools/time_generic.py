#!/usr/bin/env python3
"""Time one (elbo fwd+bwd [+pairwise] + Adam) step of the shape-generic path at the non-headline BASELINE configs.
usage: time_generic.py [cfg ...]   cfg in: cfg0 cfg2 cfg3 cfg1g cfgw cfg3p   (BASELINE.json configs[0], [2], [3]; [1] on generic kernels)
Prints ms/step, samples/s and the dense-equivalent fp32 FLOP fraction per config."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops

CFG = {
    # name: (shape, batch, coef kwargs, N)
    "cfg0": (ops.ModelShape(5, 5, 40, 100, 3, "xtanh", .1), 64, dict(a=100., b=1., c=0., d=10., pairwise=True, decoder_var=.1), 40000),
    "cfg1g": (ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1), 64, dict(pairwise=False), 40000),
    "cfg2": (ops.ModelShape(5, 5, 0, 100, 3, "xtanh", .1), 1024, dict(a=1000., b=1., c=15., d=1., pairwise=True, decoder_var=.1), 40000),
    "cfg3": (ops.ModelShape(20, 20, 200, 50, 3, "lrelu", .1), 4096, dict(pairwise=False), 1000000),
    "cfgw": (ops.ModelShape(10, 10, 40, 200, 3, "lrelu", .1), 256, dict(pairwise=False), 40000),   # IVAE_wrapper defaults (wrappers.py:14-16)
    "cfg3L": (ops.ModelShape(20, 20, 200, 50, 3, "lrelu", .1), 32768, dict(pairwise=False), 1000000),   # chip-filling batch
    "h100L": (ops.ModelShape(5, 5, 40, 100, 3, "xtanh", .1), 16384, dict(pairwise=False), 1000000),      # yaml shapes, chip-filling batch, plain
    "h200L": (ops.ModelShape(10, 10, 40, 200, 3, "lrelu", .1), 16384, dict(pairwise=False), 1000000),   # wrapper shapes, chip-filling batch
    "d20": (ops.ModelShape(20, 20, 40, 50, 3, "lrelu", .1), 64, dict(pairwise=False), 40000),          # sweep shapes (packs)
    "d40": (ops.ModelShape(40, 40, 40, 50, 3, "lrelu", .1), 64, dict(pairwise=False), 40000),
    "cfg3m8": (ops.ModelShape(20, 20, 8, 50, 3, "lrelu", .1), 4096, dict(pairwise=False), 1000000),      # configs[3] with 8 segments: what its one-hot u costs
    "cfg3p": (ops.ModelShape(20, 20, 200, 50, 3, "lrelu", .1), 4096, dict(a=1., b=1., c=2., d=1., pairwise=True), 1000000),
}


def flops_per_sample(shape, batch, pairwise):
    H, L = shape.hidden_dim, shape.n_layers
    tot = 0
    for m in shape.mlps:
        i, o = shape.mlp_io(m)
        dims = [i] + [H] * (L - 1) + [o]
        w = [dims[k] * dims[k + 1] for k in range(L)]
        tot += 2 * (2 * sum(w) + sum(w) - (0 if m == "f" else w[0]))
    if pairwise:
        tot += int(2.5 * 8 * batch * shape.latent_dim)
    return tot


def main():
    dev = torch.device("cuda:0")
    if os.environ.get("IVAE_TUNE"):          # "tile_rows,waves,wgrad_rows" (0 = the library's choice)
        tr, nw, wr = (int(v) for v in os.environ["IVAE_TUNE"].split(","))
        ops.TUNING.update(tile_rows=tr, waves=nw, wgrad_rows=wr)
    names = sys.argv[1:] or ["cfg0", "cfg1g", "cfg2", "cfg3", "cfg3p"]
    for name in names:
        shape, B, ck, N = CFG[name]
        g = torch.Generator(device=dev).manual_seed(0)
        P = shape.n_params()
        params = (torch.rand(P, generator=g, device=dev) - .5) * .2
        m = torch.zeros_like(params)
        v = torch.zeros_like(params)
        x = torch.randn(B, shape.data_dim, generator=g, device=dev)
        u = None
        if shape.aux_dim:
            u = torch.zeros(B, shape.aux_dim, device=dev)
            u[torch.arange(B, device=dev), torch.randint(0, shape.aux_dim, (B,), generator=g, device=dev)] = 1
        eps = torch.randn(B, shape.latent_dim, generator=g, device=dev)
        coef = ops.make_coef(N=N, global_batch=B, **ck)
        ws = ops.Workspace()

        def step(t):
            loss, z, grads = ops.elbo_fwd_bwd(shape, coef, params, x, u, eps, ws)
            ops.adam_step(params, m, v, grads, 1e-3, t)
            return loss

        for t in range(1, 6):
            loss = step(t)
        torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for t in range(6, 6 + n):
            loss = step(t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        fl = flops_per_sample(shape, B, ck.get("pairwise", False)) * B
        print(f"{name}: B={B} P={P}  {1e3 * dt:.3f} ms/step  {B / dt / 1e6:.3f} M samples/s  "
              f"{fl / dt / 1e12:.3f} TFLOP/s ({100 * fl / dt / 157.3e12:.2f}% of fp32 roof)  loss {float(loss[0]):.3f}", flush=True)


if __name__ == "__main__":
    main()
