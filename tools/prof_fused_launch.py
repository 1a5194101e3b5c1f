#!/usr/bin/env python3
"""Where the fixed cost of a k_train_steps launch sits (diagnostic build, make diag): the step-start stamps of model 0 / wave 0
over ONE 20-step launch -- is the first step longer than the others (cold instruction and data caches), and how much of the
launch lies before the first step's start and after the last step's end (state in / out)?  usage: prof_fused_launch.py"""
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

M, steps, NST = 256, 20, 128
dev = torch.device("cuda:0")
shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
tr = FusedPackedTrainer(shape, M, 64, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, 40000, 10, generator=g, device=dev)          # 625 steps per epoch: no launch below is cut at an epoch's end
u = torch.zeros(M, 40000, 40, device=dev)
u[:, torch.arange(40000), torch.arange(40000) % 40] = 1
tr.set_data(x, u)
for _ in range(6):
    tr.run(steps)
torch.cuda.synchronize()
buf = torch.zeros(steps * NST, dtype=torch.int64, device=dev)
assert _lib.lib().ivae_debug_set_prof(C.c_void_p(buf.data_ptr())) == 0
res = []
for rep in range(5):
    buf.zero_()
    tr.prepare_timing(2)
    torch.cuda.synchronize()
    tr._ev, tr._launch_steps = [], []
    tr.run(steps, timed=True)
    assert tr._launch_steps == [steps]
    kern_s, _ = tr.mean_kernel_seconds()
    st = buf.cpu().numpy().reshape(steps, NST).astype(np.float64)
    start = st[:, 0]                                   # stamp 0: step start (after the previous step's last barrier)
    per_step = np.diff(start)
    res.append((kern_s * 1e6, per_step, st[-1, 21] - start[-1]))
kern = np.mean([r[0] for r in res])
ps = np.mean([r[1] for r in res], 0)
last = np.mean([r[2] for r in res])
steady = ps[5:].mean()
print("launch (HIP events): %.1f us; steps 0..18 in stamp ticks: %s" % (kern, " ".join("%.0f" % v for v in ps)))
print("steady step %.0f ticks; step 0 %+.0f, step 1 %+.0f, step 2 %+.0f ticks against it; the last step up to its last barrier %.0f" % (
    steady, ps[0] - steady, ps[1] - steady, ps[2] - steady, last))
tick_us = 31.2 / steady if steady else 0
print("(with a steady step = 31.2 us: 1 tick = %.4f us) first-step excess %.1f us; 20 steps %.1f us; launch - 20 steady steps = %.1f us" % (
    tick_us, (ps[0] - steady) * tick_us, (ps.sum() + last) * tick_us, kern - 20 * 31.2))
