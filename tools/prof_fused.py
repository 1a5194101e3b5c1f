#!/usr/bin/env python3
"""Phase anatomy of k_train_steps from the diagnostic build (make prof): s_memtime stamps of
model 0 / wave 0 at every barrier.  Shares, not absolute run time, are what to read."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IVAE_AMD_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ivae_amd",
                                          "libivae_amd_prof.so")
import numpy as np
import torch
from ivae_amd import _lib, ops
from ivae_amd.trainer import FusedPackedTrainer

LABELS = ["P0 adam(last) + XU/eps -> LDS", "F1 enc L1 (g,v)", "F2 enc L2 (g,v)", "F3 heads g,v + prior L1", "F4 prior L2 + z",
          "F5 prior head + dec L1", "D2 dec L2", "D3 dec head + loss", "BF3 dgrad+wgrad head", "BF2 dgrad+wgrad+adam",
          "BF1 dz | wgrad + adam", "E deltas + adam + restore", "G3", "G2", "G1", "V3", "V2", "V1", "L3", "L2", "L1"]
NSEG = len(LABELS)

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
buf = torch.zeros(steps * 64, dtype=torch.int64, device=dev)
assert _lib.lib().ivae_debug_set_prof(C.c_void_p(buf.data_ptr())) == 0
tr.run(steps)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(steps, 64)[:, :NSEG].astype(np.float64)
seg = np.diff(np.concatenate([t[:-1], t[1:, :1]], 1), axis=1)[4:]      # (steps-1, 34) cycles per segment
mean = seg.mean(0)
tot = mean.sum()
print(f"models={M}  cycles/step (wave 0 of model 0): {tot:.0f}")
for l, c in zip(LABELS, mean):
    print(f"  {l:28s} {c:9.0f}  {100*c/tot:5.1f}%")
# sub-phase stamps of the F2 phase: wave 0 (slots 32..) and wave 4 (slots 48..): 0 = before product g, 1 = after its
# MFMAs, 2 = after product v (+ epilogue g in its shadow), 3 = after epilogue v, 4 = after the barrier
raw = buf.cpu().numpy().reshape(steps, 64).astype(np.float64)[6:]
for wv, base in (("wave0", 32), ("wave4", 48)):
    d = raw[:, base:base + 5] - raw[:, base:base + 1]
    print(wv, "F2 sub-stamps (cycles from the first):", np.round(d.mean(0)).astype(int).tolist())
print("F2 phase start (STAMP 2) -> first sub-stamp, wave0:", int(np.round((raw[:, 32] - raw[:, 2]).mean())))
