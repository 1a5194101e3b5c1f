#!/usr/bin/env python3
"""One-off: autograd usage patterns of the drop-in modules beyond `loss.backward()` once per step -- two elbo() calls
summed, a scaled loss, gradient accumulation without zero_grad, elbo() under no_grad, backward twice with retain_graph,
parameters re-homed by .to() / .cpu() round trips -- against the flat gradient of ops.elbo_fwd_bwd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ivae_amd
from ivae_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = ivae_amd.iVAE(4, 6, 5, n_layers=3, hidden_dim=20, activation="lrelu").to(dev)
B = 33
g = torch.Generator(device=dev).manual_seed(1)
xs = [torch.randn(B, 6, generator=g, device=dev) for _ in range(2)]
us = [torch.rand(B, 5, generator=g, device=dev) for _ in range(2)]
es = [torch.randn(B, 4, generator=g, device=dev) for _ in range(2)]


def flat_grad():
    return torch.cat([p.grad.reshape(-1) for p in m.parameters()])


def ref(i):
    co = ops.make_coef(1., 1., 1., 1., 1., float(m.decoder_var), False, B)
    loss, z, gr = ops.elbo_fwd_bwd(m._shape, co, m._flat_params(), xs[i], us[i], es[i], ops.Workspace())
    return loss[0], gr          # loss = -elbo


ok = True
def check(name, got, want, tol=1e-6):
    global ok
    e = float((got - want).abs().max() / want.abs().max())
    print(f"{name}: rel. error {e:.1e}")
    ok &= e <= tol

l0, g0 = ref(0); l1, g1 = ref(1)
m.zero_grad()
e0, _ = m.elbo(xs[0], us[0], eps=es[0]); e1, _ = m.elbo(xs[1], us[1], eps=es[1])
(e0 + e1).mul(-1).backward()
check("two calls summed", flat_grad(), g0 + g1)
check("elbo value", e0.detach(), -l0)
m.zero_grad()
e0, _ = m.elbo(xs[0], us[0], eps=es[0]); (e0 * -2.5).backward()
check("scaled loss", flat_grad(), 2.5 * g0)
e1, _ = m.elbo(xs[1], us[1], eps=es[1]); e1.mul(-1).backward()           # no zero_grad: accumulates
check("accumulation", flat_grad(), 2.5 * g0 + g1)
m.zero_grad()
e0, _ = m.elbo(xs[0], us[0], eps=es[0]); l = e0.mul(-1); l.backward(retain_graph=True); l.backward()
check("backward twice", flat_grad(), 2 * g0)
with torch.no_grad():
    en, zn = m.elbo(xs[0], us[0], eps=es[0])
print("no_grad: requires_grad", en.requires_grad); ok &= not en.requires_grad
check("no_grad value", en, -l0)
m = m.cpu().to(dev)
m.zero_grad()
e0, _ = m.elbo(xs[0], us[0], eps=es[0]); e0.mul(-1).backward()
check("after .cpu().to(dev)", flat_grad(), g0)
opt = torch.optim.SGD(m.parameters(), lr=.1)
before = m._flat_params().clone()
opt.step()
check("SGD step moves the arena", m._flat_params(), before - .1 * g0)
print("ALL OK" if ok else "FAILED")
