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

LABELS = ["P0 gather", "F1 enc L1 (g,v)", "F2 enc L2", "F3 heads + L1 prior", "L2 prior", "L3 prior head", "Z reparam",
          "D1 dec L1", "D2 dec L2", "D3 dec head+loss", "DB3 compute", "DB3 adam", "DB2 compute", "DB2 adam",
          "DB1 compute", "DB1 adam+E+restore g", "GB3 compute", "GB3 adam", "GB2 compute+reload XU", "GB2 adam",
          "GB1 compute", "GB1 adam+restore v", "VB3 compute", "VB3 adam", "VB2 compute", "VB2 adam", "VB1 compute",
          "VB1 adam+restore l", "LB3 compute", "LB3 adam", "LB2 compute", "LB2 adam", "LB1 compute", "LB1 adam"]

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
t = buf.cpu().numpy().reshape(steps, 64)[:, :34].astype(np.float64)
seg = np.diff(np.concatenate([t[:-1], t[1:, :1]], 1), axis=1)[4:]      # (steps-1, 34) cycles per segment
mean = seg.mean(0)
tot = mean.sum()
print(f"models={M}  cycles/step (wave 0 of model 0): {tot:.0f}")
for l, c in zip(LABELS, mean):
    print(f"  {l:28s} {c:9.0f}  {100*c/tot:5.1f}%")
