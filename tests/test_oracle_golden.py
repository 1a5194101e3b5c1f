"""Pin the CPU oracle (oracle/np_ref.py, oracle/torch_ref.py) against golden
vectors produced by the reference itself (tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import np_ref, torch_ref
from tests.util import Case, case_names, relmax

CASES = case_names()

# fp32 tolerances: forward tensors/loss 1e-5 rel (north_star), grads per-tensor max-norm 1e-5 is the
# target for the HIP path; the fp32 numpy oracle itself differs from ATen by summation order, so it is
# checked at 5e-5 and the fp64 oracle (the tight ground truth) must sit within fp32 rounding of ATen.
FWD_TOL = 2e-5
GRAD_TOL_F32 = 1e-4
GRAD_TOL_F64 = 3e-5


@pytest.mark.parametrize("name", CASES)
def test_np_forward_matches_reference(name):
    c = Case(name)
    x, u, eps = c.inputs()
    for dtype, tol in ((np.float32, FWD_TOL), (np.float64, FWD_TOL)):
        fw = np_ref.elbo_forward(c.params, x, u, eps, c.spec, c.coef(), dtype)
        for k in ("g", "v", "l", "f", "z", "lpx", "lqc", "lp"):
            assert relmax(fw[k], c.z[k]) < tol, (k, dtype)
        assert abs(float(fw["loss"]) - c.loss()) <= 1e-5 * abs(c.loss()), dtype


@pytest.mark.parametrize("name", CASES)
def test_np_backward_matches_reference(name):
    c = Case(name)
    x, u, eps = c.inputs()
    ref = c.grads
    for dtype, tol in ((np.float64, GRAD_TOL_F64), (np.float32, GRAD_TOL_F32)):
        fw = np_ref.elbo_forward(c.params, x, u, eps, c.spec, c.coef(), dtype)
        G = np_ref.elbo_backward(c.params, x, u, fw, c.spec, c.coef(), dtype)
        assert set(G) == set(ref)
        for k in ref:
            assert relmax(G[k], ref[k]) < tol, (k, dtype, relmax(G[k], ref[k]))


@pytest.mark.parametrize("name", CASES)
def test_torch_ref_matches_reference(name):
    c = Case(name)
    x, u, eps = (torch.from_numpy(t) for t in c.inputs())
    P = {k: torch.from_numpy(v) for k, v in c.params.items()}
    loss, z, G = torch_ref.grads(P, x, u, eps, c.spec, c.coef())
    assert abs(float(loss) - c.loss()) <= 2e-6 * abs(c.loss())
    assert relmax(z.numpy(), c.z["z"]) < 1e-6
    for k, ref in c.grads.items():
        assert relmax(G[k].numpy(), ref) < 2e-6, k


@pytest.mark.parametrize("name", [n for n in CASES])
def test_adam_trajectory(name):
    """Four optimiser steps (lr 1e-2 x3, then 1e-3) with np_ref fp32 arithmetic
    reproduce torch.optim.Adam's parameters and moments."""
    c = Case(name)
    P = {k: v.astype(np.float32) for k, v in c.params.items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    for k in range(4):
        x, u, eps = c.inputs(k)
        co = c.coef(k)
        fw = np_ref.elbo_forward(P, x, u, eps, c.spec, co, np.float64)
        assert abs(float(fw["loss"]) - c.loss(k)) <= 5e-4 * abs(c.loss(k)), k
        G = np_ref.elbo_backward(P, x, u, fw, c.spec, co, np.float64)
        for n in P:
            P[n], M[n], V[n] = np_ref.adam_step(P[n], G[n].astype(np.float32), M[n], V[n], k + 1,
                                               c.meta["lrs"][k])
        if c.has_adam(k + 1):
            for n in P:
                # Adam's first steps move every weight by ~lr regardless of gradient scale, so
                # compare absolutely against lr
                assert np.abs(P[n] - c.z[f"adam{k+1}.p.{n}"]).max() < 2e-2 * c.meta["lrs"][k], (k, n)
                assert relmax(M[n], c.z[f"adam{k+1}.m.{n}"]) < 2e-3, (k, n)
                assert relmax(V[n], c.z[f"adam{k+1}.v.{n}"]) < 2e-3, (k, n)


def test_adam_single_step_exact():
    """One Adam step on the fixture's own gradient is bit-close to torch's."""
    c = Case("ivae_plain_cfg2")
    for n, p in c.params.items():
        g = c.grads[n]
        p1, m1, v1 = np_ref.adam_step(p, g, np.zeros_like(p), np.zeros_like(p), 1, 1e-2)
        assert np.abs(p1 - c.z[f"adam1.p.{n}"]).max() < 1e-7
        assert relmax(m1, c.z[f"adam1.m.{n}"]) < 1e-6
        assert relmax(v1, c.z[f"adam1.v.{n}"]) < 1e-6


def test_mcc_matches_reference(golden_dir):
    z = np.load(golden_dir + "/mcc.npz")
    for i in range(int(z["n_cases"])):
        cc = np_ref.corr_block(z[f"x{i}"], z[f"y{i}"])
        assert np.abs(cc - z[f"cc{i}"]).max() < 1e-12
        assert abs(np_ref.mcc(z[f"x{i}"], z[f"y{i}"]) - float(z[f"mcc{i}"])) < 1e-12
