"""Differential check of the two independent implementations of a training step: the fused persistent kernel
(ivae_train_steps, host-supplied noise) against the shape-generic kernels (ivae_elbo_adam_steps) on the same rows,
noise, loss weights and Adam settings -- random fused shapes (d = 4..10, with / without u; u dense, not one-hot),
slopes (0 = plain ReLU .. 1), decoder variances, loss weights, step counts, starting step numbers with non-zero
moments, three packed models with their own learning rates.  Neither side is the oracle here (both are pinned against
it elsewhere); what this adds is coverage of settings no fixture has, at no oracle cost.

Measured over 1700 such cases (tools/fuzz_fused_vs_generic.py): first-step losses agree to <= 1e-6; ONE case (840) had the
two sides 2e-4 apart in one MLP's hidden-layer gradients -- a pre-activation of 1.7e-8 (typical size 0.27: inside fp32
rounding of the kink of the leaky ReLU), which the generic kernels' summation order put on the other side of zero than
fp64 and the fused kernel did: one row's derivative there is 1 on one side and the slope on the other
(tools/dbg_diff_case.py prints both sides against the fp64 oracle).  Such a case is not an error of either side and is
not among the 80 below."""
import numpy as np
import pytest
import torch

from ivae_amd import ops
from ivae_amd.trainer import init_packed_params

pytestmark = pytest.mark.gpu


def run_case(case):
    dev = torch.device("cuda:0")
    r = np.random.RandomState(case)
    d = int(r.randint(4, 11))
    M = int(r.choice([0, 40]))
    slope = float(r.choice([0., .1, .5, 1.]))
    dvar = float(r.choice([.01, .1, 1.]))
    a, b, dd = (float(t) for t in r.uniform(.3, 3., 3))
    K = int(r.choice([1, 2, 5, 9]))
    step0 = int(r.choice([0, 1, 17, 1000]))
    nm, N = 3, 300
    lrs = [float(t) for t in r.choice([1e-2, 1e-3, 3e-4], nm)]
    shape = ops.ModelShape(d, d, M, 50, 3, "lrelu", slope)
    assert ops.train_steps_supported(shape, 64)
    g = torch.Generator(device=dev).manual_seed(case)
    P0 = init_packed_params(shape, nm, case, dev)
    m0 = torch.rand(P0.shape, generator=g, device=dev) * 1e-3 if step0 else torch.zeros_like(P0)
    v0 = torch.rand(P0.shape, generator=g, device=dev) * 1e-4 if step0 else torch.zeros_like(P0)
    X = torch.randn(nm, N, d, generator=g, device=dev)
    U = None
    if M:
        lab = torch.randint(0, M, (nm, N), generator=g, device=dev)
        U = torch.zeros(nm, N, M, device=dev).scatter_(2, lab.unsqueeze(-1), 1.) * (1. + .1 * torch.randn(nm, N, 1, generator=g, device=dev))
    idx = torch.randint(0, N, (nm, K, 64), generator=g, device=dev, dtype=torch.int32)
    eps = torch.randn(nm, K, 64, d, generator=g, device=dev)
    coef = ops.make_coef(a, b, 1., dd, 40000., dvar, False, 64)
    Pf, mf, vf = P0.clone(), m0.clone(), v0.clone()
    lf, _ = ops.train_steps(shape, coef, Pf, mf, vf, X, U, idx, step0, lrs, eps=eps)
    Pg, mg, vg = P0.clone(), m0.clone(), v0.clone()
    li = idx.long()
    xs = torch.stack([torch.stack([X[m][li[m, s]] for m in range(nm)]) for s in range(K)])          # (K, nm, 64, d)
    us = None if U is None else torch.stack([torch.stack([U[m][li[m, s]] for m in range(nm)]) for s in range(K)])
    es = eps.permute(1, 0, 2, 3).contiguous()
    lg, _ = ops.elbo_adam_steps(shape, coef, Pg, mg, vg, xs, us, es, ops.Workspace(), step0, lrs, n_models=nm)
    torch.cuda.synchronize()
    lf, lg = lf.double().cpu().numpy(), lg.double().cpu().numpy().T                                 # (nm, K)
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))
    return dict(K=K, finite=bool(np.isfinite(lf).all()), desc=dict(d=d, M=M, slope=slope, dvar=dvar, K=K, step0=step0, lrs=lrs),
                loss1=np.abs(lf[:, 0] - lg[:, 0]).max() / max(np.abs(lg[:, 0]).max(), 1.),
                lossK=np.abs(lf - lg).max() / max(np.abs(lg).max(), 1.), p=rel(Pf, Pg), m=rel(mf, mg), v=rel(vf, vg))


@pytest.mark.parametrize("case", range(80))
def test_fused_kernel_agrees_with_generic_kernels(case):
    e = run_case(case)
    assert e["finite"], e
    assert e["loss1"] <= 1e-5, e                  # the first step's loss: the parity bar
    if e["K"] == 1:
        # one step: the moments are the gradient (times constants) and must meet the bar; the parameters themselves are
        # NOT compared at 1e-5 -- Adam's first update is lr * g / (|g| + 1e-8), which turns a gradient entry that is
        # rounding residue (1e-11 against 0) into a visible fraction of lr
        assert e["m"] <= 2e-5 and e["v"] <= 2e-5, e
        assert e["p"] <= 1e-3, e
    else:
        # later steps: Adam at these rates amplifies last-bit differences; the losses stay together
        assert e["lossK"] <= 1e-3, e
