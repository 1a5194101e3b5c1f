#!/usr/bin/env python3
"""Phase anatomy of k_train_steps from the diagnostic build (make prof): s_memtime stamps of
model 0 / wave 0 at every barrier.  Shares, not absolute run time, are what to read."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("IVAE_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ivae_amd",
                                          "libivae_amd_diag.so"))
import numpy as np
import torch
from ivae_amd import _lib, ops
from ivae_amd.trainer import FusedPackedTrainer

# (stamp slot written after the barrier that ENDS the phase, label); slot 0 = step start
PHASES = [(1, "P0 adam(last) + XU/eps -> LDS"), (8, "FWD wave-local chains (g|v, decoder|prior)"), (9, "BF3 dgrad+wgrad head"),
          (10, "BF2 dgrad->wgrad stream + adam"), (11, "BF1 dz + head deltas | wgrad + adam"), (13, "G3"), (14, "G2"), (15, "G1"),
          (16, "V3"), (17, "V2"), (18, "V1"), (19, "L3"), (20, "L2"), (21, "L1")]

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = 24
dev = torch.device("cuda:0")
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
tr = FusedPackedTrainer(shape, M, 64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, 4000, 10, generator=g, device=dev)
u = torch.zeros(M, 4000, 40, device=dev)
u[:, torch.arange(4000), torch.arange(4000) % 40] = 1
tr.set_data(x, u)
tr.run(8)
NST = 128
buf = torch.zeros(steps * NST, dtype=torch.int64, device=dev)
assert _lib.lib().ivae_debug_set_prof(C.c_void_p(buf.data_ptr())) == 0
tr.run(steps)
torch.cuda.synchronize()
allst = buf.cpu().numpy().reshape(steps, NST).astype(np.float64)
ids = [0] + [i for i, _ in PHASES]
t = allst[:, ids]
seg = np.diff(np.concatenate([t[:-1], t[1:, :1]], 1), axis=1)[4:]      # cycles per phase (the last one ends at the next step's start)
mean = seg.mean(0)
tot = mean.sum()
print(f"models={M}  cycles/step (wave 0 of model 0): {tot:.0f}")
for (_, l), c in zip(PHASES, mean):
    print(f"  {l:44s} {c:9.0f}  {100*c/tot:5.1f}%")
# forward-stage timeline of wave 0 (g chain, decoder; slots 32..) and wave 4 (logv chain, prior; slots 48..), cycles from
# the stage's first stamp of wave 0
raw = buf.cpu().numpy().reshape(steps, NST).astype(np.float64)[6:]
names0 = ["start", "g L0", "g L1", "g head", "wait v", "z", "dec L0", "dec L1", "dec head+dO"]
names4 = ["start", "v L0", "v L1", "v head", "prior L0", "prior L1 (+wait g)", "prior head", "wait z"]
for wv, base, names in (("wave0", 32, names0), ("wave4", 48, names4)):
    d = raw[:, base:base + len(names)] - raw[:, 32:33]
    m = d.mean(0)
    print(wv, " ".join(f"{n}@{int(v)}(+{int(v - (m[i - 1] if i else 0))})" for i, (n, v) in enumerate(zip(names, m))))
print("wave0 g L1: product took", int((raw[:, 32 + 9] - raw[:, 32 + 1]).mean()), "epilogue", int((raw[:, 32 + 2] - raw[:, 32 + 9]).mean()))
print("stage barrier released (STAMP 8) @", int((raw[:, 8] - raw[:, 32]).mean()))


# in-stream stamps of the G2 phase (dgrad -> wgrad stream), waves 0 and 4, cycles since the phase's barrier release (stamp 13)
for wv, base in (("wave0", 64), ("wave4", 96)):
    rel = lambda i: int((raw[:, base + i] - raw[:, 13]).mean())
    print(wv, "G2 stream: entry", rel(0), "first MFMA", rel(1), "4th dgrad group", rel(2), "dgrad MFMAs issued", rel(3),
          "5th wgrad group", rel(4), "wgrad MFMAs issued", rel(5), "stores done", rel(8), "adam done", rel(9),
          "| barrier14 @", int((raw[:, 14] - raw[:, 13]).mean()))
