"""Multi-process tests of ivae_amd/dist.py on CPU ranks (gloo, world_size 2): model/row partitioning, the
data-parallel gradient exchange for the plain ELBO and the all-gather / all-reduce exchange of the pairwise
(total-correlation) ELBO.  Kernels are replaced by the oracle-backed double in tests/oracle_backend.py, so what is
checked here is the exchange logic: N ranks must reproduce the single-rank full-batch result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ivae_amd.dist import DataParallelTrainer, gather_model_results, model_range, row_range
from ivae_amd.ops import ModelShape
from oracle import np_ref, torch_ref
from tests import oracle_backend

SHAPE = ModelShape(4, 3, 5, 8, 3, "lrelu", .1)
SPEC = np_ref.Spec(4, 3, 5, 8, 3, "lrelu", .1, .1)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(B, steps, seed=0):
    rng = np.random.RandomState(seed)
    x = rng.randn(steps, B, 4).astype(np.float32)
    u = np.zeros((steps, B, 5), np.float32)
    u[np.arange(steps)[:, None], np.arange(B)[None], rng.randint(0, 5, (steps, B))] = 1
    eps = rng.randn(steps, B, 3).astype(np.float32)
    P = torch_ref.xavier_params(SPEC, seed=1)
    flat = torch.cat([P[n].reshape(-1) for n in SPEC.param_names()]).clone()
    return x, u, eps, flat


def _train(rank, world, B, steps, coef, group=None):
    x, u, eps, flat = _problem(B, steps)
    tr = DataParallelTrainer(SHAPE, flat, decoder_var=.1, lr=1e-2, group=group, backend=oracle_backend)
    lo, hi = row_range(B, rank, world)
    losses = []
    t = torch.from_numpy
    for s in range(steps):
        loss, z = tr.step(t(x[s, lo:hi]), t(u[s, lo:hi]), t(eps[s, lo:hi]), B, **coef)
        losses.append(float(loss[0]))
    return np.array(losses), tr.params.numpy().copy(), tr.exp_avg_sq.numpy().copy()


def _worker(rank, world, port, B, steps, coef, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        losses, params, v = _train(rank, world, B, steps, coef)
        # packed-sweep result gathering: 5 models over 2 ranks -> ranges of 3 and 2
        lo, hi = model_range(5, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32).unsqueeze(1) * torch.ones(1, 3)
        allres = gather_model_results(local, 5)
        out.put((rank, losses, params, v, allres.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,coef", [
    (2, 12, dict(pairwise=False)),                                           # iVAE plain ELBO (nets.py:475)
    (2, 13, dict(a=100., b=1., c=0., d=10., N=4000., pairwise=True)),        # cleanIVAE weights of configs/ivae.yaml, ragged
    (2, 12, dict(a=100., b=1., c=0., d=10., N=4000., pairwise=True)),        # equal shards: one all-gather of (g, v), reduce-scatter of (cg, cv)
    (2, 13, dict(a=1000., b=1., c=15., d=1., N=4000., pairwise=True)),       # beta-TCVAE weights of configs/btcvae.yaml
    (4, 12, dict(a=1000., b=1., c=15., d=1., N=4000., pairwise=True)),       # four ranks, equal shards: the ranks' row blocks must land in rank order
    (3, 13, dict(a=100., b=1., c=.5, d=10., N=4000., pairwise=True)),        # three ranks, shards of 5 / 4 / 4 rows (padded path)
    (4, 14, dict(pairwise=False)),                                           # four ranks, plain ELBO, shards of 4 / 4 / 3 / 3 rows
    (8, 16, dict(a=100., b=1., c=0., d=10., N=4000., pairwise=True)),        # a whole node's eight ranks, two rows each
    (8, 19, dict(pairwise=False)),                                           # eight ranks, shards of 3 / 3 / 3 / 2 ... rows
])
def test_ranks_reproduce_single_rank(world, B, coef):
    steps = 3
    ref_losses, ref_params, ref_v = _train(0, 1, B, steps, coef)
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, steps, coef, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out.get() for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, p0, v0, g0), (_, l1, p1, v1, g1) = res[0], res[-1]
    # replicas stay bit-identical without any parameter broadcast
    for (_, l, p, v, g) in res[1:]:
        assert np.array_equal(p0, p) and np.array_equal(v0, v) and np.array_equal(l0, l)
    # and equal the single-rank full-batch run (fp32 interface, different summation split: 1e-5)
    assert np.allclose(l0, ref_losses, rtol=1e-5, atol=0)
    assert np.abs(p0 - ref_params).max() < 1e-5 * max(1., np.abs(ref_params).max())
    assert np.allclose(v0, ref_v, rtol=1e-4, atol=1e-12)
    want = np.repeat(np.arange(5, dtype=np.float32)[:, None], 3, 1)
    assert np.array_equal(g0, want) and np.array_equal(g1, want)


def test_partitions_cover_everything_once():
    for n, world in ((256, 8), (5, 2), (3, 8), (4096, 3)):
        spans = [model_range(n, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        model_range(4, 2, 2)


def test_shard_oracle_adds_up_to_full_batch():
    """The row-shard form of the oracle itself: shard losses / gradients (pairwise included) sum to the full batch."""
    B = 11
    x, u, eps, flat = _problem(B, 1)
    P = {k: v.numpy().astype(np.float64) for k, v in torch_ref.xavier_params(SPEC, seed=1).items()}
    coef = np_ref.Coef(a=100., b=1., c=.5, d=10., N=4000., pairwise=True)
    fw = np_ref.elbo_forward(P, x[0], u[0], eps[0], SPEC, coef, np.float64)
    G = np_ref.elbo_backward(P, x[0], u[0], fw, SPEC, coef, np.float64)
    parts = []
    for lo, hi in ((0, 4), (4, 11)):
        parts.append(np_ref.pairwise_terms(fw["z"][lo:hi], fw["g"], fw["v"], coef, B))
    cg, cv = sum(p["cg"] for p in parts), sum(p["cv"] for p in parts)
    loss, acc = 0., {k: 0. for k in G}
    for (lo, hi), pw in zip(((0, 4), (4, 11)), parts):
        f2 = np_ref.elbo_forward(P, x[0, lo:hi], u[0, lo:hi], eps[0, lo:hi], SPEC, coef, np.float64, global_batch=B,
                                 pair_cols=(fw["g"], fw["v"]))
        g2 = np_ref.elbo_backward(P, x[0, lo:hi], u[0, lo:hi], f2, SPEC, coef, np.float64, global_batch=B,
                                  pair=dict(rz=pw["rz"], cg=cg[lo:hi], cv=cv[lo:hi]))
        loss += f2["loss"]
        for k in G:
            acc[k] = acc[k] + g2[k]
    assert abs(loss - fw["loss"]) < 1e-10 * abs(fw["loss"])
    for k in G:
        assert np.abs(acc[k] - G[k]).max() < 1e-10 * max(1., np.abs(G[k]).max()), k
