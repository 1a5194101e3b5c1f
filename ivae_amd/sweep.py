"""Paper-style sweep (BASELINE configs[4]): many independent iVAE runs -- seeds x latent dimensions -- packed as
batched-model launches, a disjoint set of runs per GPU, no data-path collective; every run is scored by the Mean
Correlation Coefficient with the correlation matrix computed on the GPU and the Hungarian step on the host.

    python -m ivae_amd.sweep --dims 5,10,20,40 --seeds 64 --epochs 20                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ivae_amd.sweep --dims ... --seeds 64

Each run is what `main.py --config ivae.yaml`-style experiments do per seed (runners/ivae.py:31-123 with the
`iVAE` class of models/nets.py:391-486): its own dataset (data/data.py generator semantics, seed = run seed), its
own xavier-initialised model (seed = run seed), `epochs` passes of shuffled batches of 64 with Adam, then the
full-data evaluation on a *sampled* z (runners/ivae.py:118-123).  Runs of one latent dimension form one pack:
the fused persistent kernel where it covers the shape, the generic MFMA kernels otherwise.
"""
import argparse
from concurrent.futures import ThreadPoolExecutor
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist
from scipy.optimize import linear_sum_assignment

from . import ops
from .data import dataset_path, generate_data
from .dist import gather_model_results, model_range
from .trainer import FusedPackedTrainer, PackedTrainer, PlateauLR


def device_datasets(n_models, n_seg, n_per_seg, d, seeds, device):
    """Piecewise-stationary nonlinear-ICA data with the generator semantics of data/data.py:134-304 (gaussian
    sources modulated per segment by L ~ U(.5, 3), 3-layer lrelu mixing with column-normalised uniform matrices,
    without the condition-number rejection step), drawn on the device, one generator seed per run.
    Returns x (n, N, d), u (n, N, n_seg) one-hot, s (n, N, d)."""
    N = n_seg * n_per_seg
    xs, ss = [], []
    for sd in seeds:
        g = torch.Generator(device=device).manual_seed(int(sd))
        L = torch.rand(n_seg, 1, d, generator=g, device=device) * 2.5 + 0.5
        s = (torch.randn(n_seg, n_per_seg, d, generator=g, device=device) * L).reshape(N, d)
        A = torch.rand(d, d, generator=g, device=device) * 2 - 1
        A = A / A.pow(2).sum(0, keepdim=True).sqrt()
        x = s
        for layer in range(3):
            x = x @ A
            if layer < 2:
                x = torch.where(x > 0, x, 0.1 * x)
        xs.append(x)
        ss.append(s)
    lab = torch.arange(n_seg, device=device).repeat_interleave(n_per_seg)
    u = torch.zeros(N, n_seg, device=device)
    u[torch.arange(N, device=device), lab] = 1
    return torch.stack(xs).contiguous(), u.unsqueeze(0).expand(n_models, N, n_seg).contiguous(), torch.stack(ss).contiguous()


def _host_dataset(job):
    n_seg, n_per_seg, d, sd, cache = job
    path = None
    if cache:       # the reference's own cache (data/data.py:357-372: same file name, same keys): either code base reads it
        path = dataset_path(cache, n_per_seg, n_seg, d, d, 3, int(sd), 'gauss', 'lrelu')
        if os.path.exists(path):
            with np.load(path) as z:
                return tuple(np.ascontiguousarray(z[k], np.float32) for k in ('x', 'u', 's'))
    S, X, U, M, L = generate_data(n_per_seg, n_seg, d, n_layers=3, seed=int(sd), prior='gauss', activation='lrelu',
                                  repeat_linearity=True)
    if path is not None:
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        tmp = '{}.{}.tmp.npz'.format(path, os.getpid())
        np.savez_compressed(tmp, s=S, x=X, u=U, m=M, L=L)
        os.replace(tmp, path)            # (whole files only: another rank may be looking for the same dataset)
    return tuple(np.ascontiguousarray(a, np.float32) for a in (X, U, S))


def host_datasets_numpy(n_seg, n_per_seg, jobs, workers=None, cache=None):
    """The bit-exact host generator (ivae_amd.data.generate_data == data/data.py:200-304, repeat_linearity=True as
    data.py:369), one dataset per (d, seed) job -> {(d, seed): (X, U, S)}.  The datasets are independent (the generator
    seeds the legacy numpy stream per dataset), so they are made by a pool of forked workers -- only while this process
    has not initialised the GPU (a fork of a process that holds a HIP context is not safe); afterwards one by one.
    `cache`: a folder (with trailing separator, as the reference's `data_path`) of `.npz` files in the reference's cache format;
    datasets found there are loaded instead of generated, the others are generated and left there."""
    args = [(n_seg, n_per_seg, d, sd, cache) for d, sd in jobs]
    workers = min(workers or 16, len(args), os.cpu_count() or 1)
    if workers > 1 and not torch.cuda.is_initialized():
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            out = pool.map(_host_dataset, args, chunksize=1)
    else:
        out = [_host_dataset(a) for a in args]
    return dict(zip(jobs, out))


def host_datasets(n_seg, n_per_seg, d, seeds, device, made=None):
    """One dataset per run seed on the device; `made`: datasets host_datasets_numpy already holds."""
    made = made or host_datasets_numpy(n_seg, n_per_seg, [(d, sd) for sd in seeds])
    xs, us, ss = zip(*[made[(d, sd)] for sd in seeds])
    return tuple(torch.stack([torch.from_numpy(a) for a in t]).to(device) for t in (xs, us, ss))


def evaluate_pack(shape, params, x, u, s_true, rows_per_call=8192, seed=0):
    """Full-data MCC of every model of a pack: z sampled through ivae_forward in row blocks (all models per launch),
    correlation blocks by ivae_corr on the device, assignment on the host (metrics/mcc.py:326-335).
    Returns (n_models,) float64 numpy."""
    n, N, _ = x.shape
    dev = x.device
    g = torch.Generator(device=dev).manual_seed(seed)
    ws = ops.Workspace()
    z = torch.empty(n, N, shape.latent_dim, device=dev)
    for lo in range(0, N, rows_per_call):
        hi = min(N, lo + rows_per_call)
        eps = torch.randn(n, hi - lo, shape.latent_dim, generator=g, device=dev)
        co = ops.make_coef(N=float(N), global_batch=hi - lo)
        z[:, lo:hi] = ops.forward(shape, co, params, x[:, lo:hi].contiguous(),
                                  u[:, lo:hi].contiguous() if u is not None else None, eps, ws, n)
    cc = torch.stack([ops.corr(s_true[m], z[m]) for m in range(n)]).abs().cpu().numpy()    # one host copy per pack
    return np.array([c[linear_sum_assignment(-c)].mean() for c in cc])


def train_pack(tr, n_models, epochs, steps_per_epoch, lr, patience):
    """`epochs` passes over a pack's datasets.  After every epoch each run's own ReduceLROnPlateau(factor .1, patience)
    sees that run's mean epoch loss (runners/ivae.py:39,108-114; wrappers.py:45,82-83) and the pack's next epoch is
    launched with one learning rate per run (`model_lr` of ivae_train_steps / ivae_elbo_adam_steps); patience None = no
    scheduler (`no_scheduler: True`).  One host sync per epoch (the reference syncs every step).  -> final rates."""
    sched = None if patience is None else PlateauLR(n_models, lr, factor=0.1, patience=patience)
    lrs = [float(lr)] * n_models
    for _ in range(epochs):
        tr.lr = lrs if sched is not None else lr
        out = tr.run(steps_per_epoch)
        if sched is not None:
            losses = out if isinstance(tr, FusedPackedTrainer) else tr.last_losses         # (n_models, steps)
            lrs = sched.step(losses.double().mean(1).cpu().tolist())
    return lrs


def run_sweep(dims, n_seeds, epochs, n_seg=40, n_per_seg=1000, batch=64, lr=1e-2, hidden=50, data="host",
              device=None, rank=0, world=1, log=None, concurrent=True, patience=0, lr_out=None, datasets=None):
    """Runs this rank's share of the (dim, seed) grid; returns (local run ids, local MCCs, seconds spent training,
    training steps x models done).  concurrent: the packs of the different latent dimensions train at the same time, each
    on a HIP stream of its own -- a 64-model pack of the fused kernel occupies 64 of the 256 CUs and a 64-model pack of
    the generic kernels is latency-bound, so one pack at a time leaves most of the chip idle; the packs are independent,
    their results do not depend on the interleaving.  patience: of every run's ReduceLROnPlateau (0 as in
    runners/ivae.py:39; None: constant learning rate); lr_out: a dict that receives {run index: final learning rate}."""
    device = device or torch.device("cuda", 0)
    runs = [(d, s) for d in dims for s in range(n_seeds)]
    lo, hi = model_range(len(runs), rank, world)
    mine = runs[lo:hi]
    mcc_local = np.zeros(hi - lo)
    N = n_seg * n_per_seg
    nb = N // batch
    steps = epochs * nb
    packs = []
    made = None
    if data == "host":      # `datasets`: what host_datasets_numpy made for these runs before the GPU was initialised (main)
        made = datasets if datasets is not None else host_datasets_numpy(n_seg, n_per_seg, mine)
    for d in sorted({r[0] for r in mine}):
        idx = [i for i, r in enumerate(mine) if r[0] == d]
        seeds = [mine[i][1] for i in idx]
        shape = ops.ModelShape(d, d, n_seg, hidden, 3, "lrelu", .1)
        if data == "host":
            x, u, s_true = host_datasets(n_seg, n_per_seg, d, seeds, device, made)
        else:
            x, u, s_true = device_datasets(len(seeds), n_seg, n_per_seg, d, [1000 * d + s for s in seeds], device)
        fused = ops.train_steps_supported(shape, batch)
        cls = FusedPackedTrainer if fused else PackedTrainer
        tr = cls(shape, n_models=len(seeds), batch=batch, lr=lr, seed=1000 * d + seeds[0], device=device)
        tr.set_data(x, u)
        packs.append(dict(d=d, idx=idx, shape=shape, fused=fused, tr=tr, x=x, u=u, s=s_true, n=len(seeds)))
    t_train, work = 0.0, 0
    if concurrent and len(packs) > 1:
        # generic packs first: thousands of short launches each, enqueued while the earlier packs already run
        order = sorted(packs, key=lambda p: (p["fused"], -p["d"]))
        streams = [torch.cuda.Stream(device) for _ in order]
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        def train(p, st):      # a host thread per pack: a launch call blocks while its stream's queue is full, and the
            torch.cuda.set_device(device)       # library calls (a whole epoch each) run without the interpreter lock
            with torch.cuda.stream(st):
                # fused: ivae_train_steps launches; generic: one ivae_elbo_adam_steps call per epoch
                p["lrs"] = train_pack(p["tr"], p["n"], epochs, nb, lr, patience)
        with ThreadPoolExecutor(len(order)) as pool:
            for f in [pool.submit(train, p, st) for p, st in zip(order, streams)]:
                f.result()
        torch.cuda.synchronize(device)
        t_train = time.perf_counter() - t0
    else:
        for p in packs:
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            p["lrs"] = train_pack(p["tr"], p["n"], epochs, nb, lr, patience)
            torch.cuda.synchronize(device)
            t_train += time.perf_counter() - t0
    for p in packs:
        work += steps * p["n"]
        m = evaluate_pack(p["shape"], p["tr"].params, p["x"], p["u"], p["s"], seed=p["d"])
        mcc_local[p["idx"]] = m
        if lr_out is not None:
            lr_out.update({lo + i: v for i, v in zip(p["idx"], p["lrs"])})
        if log:
            log("rank %d: d=%d, %d runs, %s, %d steps: MCC mean %.3f (min %.3f, max %.3f)" % (
                rank, p["d"], p["n"], "fused" if p["fused"] else "generic", steps, m.mean(), m.min(), m.max()))
    return (lo, hi), mcc_local, t_train, work


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--dims", default="5,10,20,40")
    ap.add_argument("--seeds", type=int, default=64, help="runs per latent dimension")
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--nps", type=int, default=1000, help="samples per segment")
    ap.add_argument("--ns", type=int, default=40, help="segments")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--data", choices=["host", "device"], default="host",
                    help="host (default): the bit-exact numpy generator of data/data.py, as north_star asks (seconds per "
                         "dataset); device: a throughput-only stand-in drawn on the GPU WITHOUT the generator's "
                         "condition-number rejection -- its MCCs are not on reference inputs")
    ap.add_argument("--data-path", default=None,
                    help="folder for the host generator's datasets in the reference's cache format (tcl_*.npz, data/data.py:357-372): "
                         "generated once, loaded afterwards (a full-length sweep spends 7x its training time generating them)")
    ap.add_argument("--no-scheduler", action="store_true", help="constant learning rate (the reference's no_scheduler: True); "
                                                                "default: every run has its own ReduceLROnPlateau(factor .1, patience 0)")
    ap.add_argument("--serial", action="store_true", help="train one pack (latent dimension) at a time instead of all "
                                                          "of them concurrently on separate streams")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dims = [int(t) for t in args.dims.split(",")]
    t0 = time.perf_counter()
    made = None
    if args.data == "host":      # the host generator's worker pool forks: before this process initialises the GPU
        runs = [(d, s) for d in dims for s in range(args.seeds)]
        lo, hi = model_range(len(runs), rank, world)
        made = host_datasets_numpy(args.ns, args.nps, runs[lo:hi], workers=max(1, 16 // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))),
                                   cache=None if args.data_path is None else os.path.join(args.data_path, ""))
    t_data = time.perf_counter() - t0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    (lo, hi), mcc, t_train, work = run_sweep(dims, args.seeds, args.epochs, args.ns, args.nps, args.batch, args.lr,
                                             data=args.data, device=dev, rank=rank, world=world,
                                             log=lambda s: print(s, flush=True), concurrent=not args.serial,
                                             patience=None if args.no_scheduler else 0, datasets=made)
    n_runs = len(dims) * args.seeds
    allm = gather_model_results(torch.from_numpy(mcc).to(dev), n_runs).cpu().numpy()
    stats = torch.tensor([t_train, float(work)], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        t_train = float(tmax[0])
        work = float(stats[1])
    if rank == 0:
        per_dim = {}
        for i, d in enumerate(dims):
            m = allm[i * args.seeds:(i + 1) * args.seeds]
            per_dim[str(d)] = {"mcc_mean": round(float(m.mean()), 4), "mcc_std": round(float(m.std()), 4)}
        print(json.dumps({"sweep": "iVAE seeds x dims, packed runs", "n_runs": n_runs, "n_gpus": world,
                          "epochs": args.epochs, "batch": args.batch,
                          "lr_schedule": "constant" if args.no_scheduler else "ReduceLROnPlateau(.1, patience 0) per run",
                          "packs": "one at a time" if args.serial else "concurrent, one HIP stream per latent dimension",
                          "data": ("host numpy generator (bit-exact data/data.py)" if args.data == "host" else
                                   "device stand-in, NOT the reference generator (no condition-number rejection)"),
                          "train_samples_per_s": round(work * args.batch / t_train, 1),
                          "train_seconds_max_over_ranks": round(t_train, 2),
                          "host_data_seconds": round(t_data, 2),
                          "wall_seconds": round(time.perf_counter() - t0, 2), "mcc": per_dim}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
