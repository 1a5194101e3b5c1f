"""PyTorch-CPU restatement of the iVAE training step (TEST INFRASTRUCTURE ONLY).

Same third-party arithmetic as the reference (ATen CPU kernels, autograd,
``torch.optim.Adam``) in the same order, written functionally over a parameter
dict so it needs no ``nn.Module``.  Used (a) as the autograd twin of
``np_ref``'s analytic backward, (b) as the ``cpu_baseline`` ("port") that
``bench.py`` times on the GPU box's host cores, where the reference's own
Python is not present.  Pinned by tests/test_oracle_golden.py.
"""
import math

import torch
import torch.nn.functional as F

from .np_ref import Coef, Spec  # noqa: F401  (re-exported for callers)


def _act(p, kind, slope):
    """models/nets.py:88-95,109-112."""
    if kind == "lrelu":
        return F.leaky_relu(p, negative_slope=slope)
    if kind == "xtanh":
        return p.tanh() + slope * p
    if kind == "sigmoid":
        return torch.sigmoid(p)
    if kind == "none":
        return p
    raise ValueError(kind)


def mlp(P, name, h, spec):
    """models/nets.py:114-121."""
    for li in range(spec.n_layers):
        h = F.linear(h, P[f"{name}.fc.{li}.weight"], P[f"{name}.fc.{li}.bias"])
        if li != spec.n_layers - 1:
            h = _act(h, spec.activation, spec.slope)
    return h


def _lognormal(x, mu, v, log2pi):
    """models/nets.py:52,310."""
    return -0.5 * (log2pi + v.log() + (x - mu).pow(2).div(v))


def forward(P, x, u, eps, spec):
    """models/nets.py:450-455 / 587-591 / 165-170 / 224-228 -> (f, g, v, z, l)."""
    if spec.aux_dim > 0:
        xu = torch.cat((x, u), 1)
        l = mlp(P, "logl", u, spec).exp()
    else:
        xu = x
        l = torch.ones(1, dtype=x.dtype)
    g = mlp(P, "g", xu, spec)
    v = mlp(P, "logv", xu, spec).exp()
    z = eps.mul(v.sqrt()).add(g)
    f = mlp(P, "f", z, spec)
    return f, g, v, z, l


def loss_fn(P, x, u, eps, spec, coef):
    """Quantity to minimise: family A's return value (nets.py:184,242) and the
    negation of family B's (nets.py:471-475,599).  Returns (loss, z, parts)."""
    f, g, v, z, l = forward(P, x, u, eps, spec)
    log2pi = math.log(2 * math.pi)
    B, dl = z.shape
    dv = torch.full((1,), spec.decoder_var, dtype=x.dtype)
    lpx = _lognormal(x, f, dv, log2pi).sum(-1)
    lqc = _lognormal(z, g, v, log2pi).sum(-1)
    lp = _lognormal(z, torch.zeros(1, dtype=x.dtype), l, log2pi).sum(-1)
    parts = dict(f=f, g=g, v=v, z=z, l=l, lpx=lpx, lqc=lqc, lp=lp)
    if coef.pairwise:
        T = _lognormal(z.view(B, 1, dl), g.view(1, B, dl), v.view(1, B, dl), log2pi)
        logMN = math.log(B * coef.N)
        lq = torch.logsumexp(T.sum(-1), dim=1) - logMN
        lqi = (torch.logsumexp(T, dim=1) - logMN).sum(-1)
        row = coef.a * lpx - coef.b * (lqc - lq) - coef.c * (lq - lqi) - coef.d * (lqi - lp)
        parts.update(lq=lq, lqi=lqi)
    else:
        row = coef.a * lpx - coef.b * lqc + coef.d * lp
    return -row.mean(), z, parts


def grads(P, x, u, eps, spec, coef):
    """loss.backward() of runners/ivae.py:82 / wrappers.py:69 -> (loss, z, {name: grad})."""
    Pg = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    loss, z, _ = loss_fn(Pg, x, u, eps, spec, coef)
    loss.backward()
    return loss.detach(), z.detach(), {k: v.grad for k, v in Pg.items()}


class Trainer:
    """zero_grad -> elbo -> backward -> Adam.step loop of runners/ivae.py:69-91 and
    wrappers.py:60-70 over a parameter dict; the timed CPU baseline."""

    def __init__(self, P, spec, coef, lr=1e-2):
        self.P = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        self.spec, self.coef = spec, coef
        self.opt = torch.optim.Adam(list(self.P.values()), lr=lr)

    def step(self, x, u, eps=None):
        self.opt.zero_grad()
        if eps is None:
            eps = torch.randn(x.shape[0], self.spec.latent_dim, dtype=x.dtype)
        loss, z, _ = loss_fn(self.P, x, u, eps, self.spec, self.coef)
        loss.backward()
        self.opt.step()
        return loss.detach(), z.detach()


def xavier_params(spec, seed=0, dtype=torch.float32):
    """Random parameters with the reference's init ranges (nets.py:10-12,432):
    xavier-uniform weights, U(+-1/sqrt(fan_in)) biases.  For tests/bench only
    (the product modules reuse nn.Linear's own init)."""
    gen = torch.Generator().manual_seed(seed)
    P = {}
    for m in spec.mlps:
        for li, (i, o) in enumerate(spec.layer_dims(m)):
            bound = math.sqrt(6.0 / (i + o))
            P[f"{m}.fc.{li}.weight"] = ((torch.rand(o, i, generator=gen, dtype=dtype) * 2 - 1) * bound)
            P[f"{m}.fc.{li}.bias"] = ((torch.rand(o, generator=gen, dtype=dtype) * 2 - 1) / math.sqrt(i))
    return P
