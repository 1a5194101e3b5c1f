#!/usr/bin/env python3
"""Soak run of the fused kernel: 256 packed configs[1] models trained at a constant rate for many epochs, one epoch
(625 steps) per call; the first time a model's loss stops being finite its last losses are printed (a timed-out wave
hand-off poisons a model's losses with NaN from one step to the next; an optimisation that really diverges
shows it in the losses before) together with the state before that epoch.  The run is repeated with the same seed: the
same models at the same steps = deterministic arithmetic, not a race.  usage: soak_fused.py [epochs] [lr] [data: randn|gen]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ivae_amd import ops
from ivae_amd.trainer import FusedPackedTrainer

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
kind = sys.argv[3] if len(sys.argv) > 3 else "randn"
dev = torch.device("cuda:0")
M = 256
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
g = torch.Generator(device=dev).manual_seed(0)
if kind == "gen":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import reference_datasets
    x, u = reference_datasets(M, 40, 1000, 10, first_seed=1, device=dev)
else:
    x = torch.randn(M, 40000, 10, generator=g, device=dev)
    u = torch.zeros(M, 40000, 40, device=dev)
    u[:, torch.arange(40000), torch.arange(40000) % 40] = 1


def soak(tag):
    tr = FusedPackedTrainer(shape, M, 64, lr=lr, seed=0, device=dev)
    tr.set_data(x, u)
    seen, events = set(), []
    for e in range(epochs):
        before = tr.params.clone()
        loss = tr.run(625)                       # (M, 625)
        torch.cuda.synchronize()
        bad = (~torch.isfinite(loss[:, -1])).nonzero().flatten().tolist()
        for m in [m for m in bad if m not in seen]:
            row = loss[m]
            first = int((~torch.isfinite(row)).nonzero()[0])
            events.append((m, e * 625 + first))
            tail = row[max(0, first - 8):first + 1].tolist()
            pb = before[m]
            print(f"[{tag}] model {m}: first non-finite loss at step {e * 625 + first}; losses before it "
                  f"{' '.join('%.4g' % t for t in tail)}; params at the epoch's start: finite {bool(torch.isfinite(pb).all())}, "
                  f"max |p| {pb.abs().max().item():.3g}", flush=True)
        seen.update(bad)
        if e % 40 == 39:
            fin = loss[:, -1][torch.isfinite(loss[:, -1])]
            print(f"[{tag}] after {e + 1} epochs: {len(seen)} model(s) lost; finite losses {fin.min().item():.2f} .. {fin.max().item():.2f}", flush=True)
    return events


a = soak("run 1")
b = soak("run 2")
print("identical events in both runs:", a == b, len(a))
