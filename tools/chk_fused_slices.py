#!/usr/bin/env python3
"""One-off: ivae_train_steps on a pack longer than one launch holds (300 models: slices of 256 + 44) against the same
models run as separate calls -- in-kernel noise (the Philox stream id is the model's number inside the pack) and
host-supplied noise; a dataset shorter than a batch; row indices outside the dataset (clamped, never a fault)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from ivae_amd.trainer import init_packed_params

dev = torch.device("cuda:0")
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
coef = ops.make_coef(1., 1., 1., 1., 40000., .01, False, 64)
g = torch.Generator(device=dev).manual_seed(0)
nm, N, K = 300, 500, 6
P0 = init_packed_params(shape, nm, 0, dev)
X = torch.randn(nm, N, 10, generator=g, device=dev)
lab = torch.randint(0, 40, (nm, N), generator=g, device=dev)
U = torch.zeros(nm, N, 40, device=dev).scatter_(2, lab.unsqueeze(-1), 1.)
idx = torch.randint(0, N, (nm, K, 64), generator=g, device=dev, dtype=torch.int32)
eps = torch.randn(nm, K, 64, 10, generator=g, device=dev)
lrs = [1e-2 if m % 3 else 1e-3 for m in range(nm)]


def run(sl, eps_, seed=5):
    P, m, v = P0[sl].clone(), torch.zeros_like(P0[sl]), torch.zeros_like(P0[sl])
    loss, z = ops.train_steps(shape, coef, P, m, v, X[sl].contiguous(), U[sl].contiguous(), idx[sl].contiguous(), 3, lrs[sl],
                              eps=None if eps_ is None else eps_[sl].contiguous(), seed=seed, want_z=True)
    return P, m, v, loss, z


for name, e in (("host eps", eps), ("in-kernel noise", None)):
    whole = run(slice(0, nm), e)
    # a model's noise stream is keyed by its number in ITS call: separate calls reproduce the pack only for host eps, or
    # when the sub-call starts at the same model number; so compare in-kernel noise against the first 256 only
    parts = [run(slice(0, 256), e), run(slice(256, nm), e)]
    for i, what in enumerate(("params", "exp_avg", "exp_avg_sq", "loss", "z")):
        a = whole[i]
        if e is None:
            ok = torch.equal(a[:256], parts[0][i])
            print(f"{name}: {what}: first 256 models of the 300-pack == a 256-pack: {ok}")
        else:
            b = torch.cat([parts[0][i], parts[1][i]])
            print(f"{name}: {what}: 300-pack == 256-pack + 44-pack: {torch.equal(a, b)}")
# dataset shorter than a batch, indices outside it
Ns = 40
Xs, Us = X[:3, :Ns].contiguous(), U[:3, :Ns].contiguous()
bad_idx = torch.randint(-50, 2 * Ns, (3, 4, 64), generator=g, device=dev, dtype=torch.int32)
good_idx = bad_idx.clamp(0, Ns - 1)
out = []
for ix in (bad_idx, good_idx):
    P, m, v = P0[:3].clone(), torch.zeros_like(P0[:3]), torch.zeros_like(P0[:3])
    loss, _ = ops.train_steps(shape, coef, P, m, v, Xs, Us, ix.contiguous(), 0, 1e-2, eps=eps[:3, :4].contiguous())
    out.append((P, loss))
torch.cuda.synchronize()
print("indices outside a 40-row dataset behave as clamped ones:", torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]),
      "finite:", bool(torch.isfinite(out[0][1]).all()))
