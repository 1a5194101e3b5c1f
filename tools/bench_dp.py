#!/usr/bin/env python3
"""BASELINE configs[3]: large-batch iVAE (d=20, 200 segments x 5000 samples, batch 4096), data-parallel over the
ranks of one node (strong scaling: the global batch is fixed, each rank takes batch/N rows), one gradient all-reduce
(RCCL) per step.  Prints one JSON line on rank 0.

    python tools/bench_dp.py                                   # one GPU
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_dp.py --gpus N
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--pairwise", action="store_true", help="include the BxBxd total-correlation terms (all-gather path)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # start the ranks ourselves (a child process, before this one touches the GPU) and leave with its exit code
        import socket
        import subprocess
        have = torch.cuda.device_count()
        if args.gpus > have:
            sys.exit("bench_dp.py: --gpus %d asked for, %d GPU(s) visible" % (args.gpus, have))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                                  str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench_dp.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    from ivae_amd import ops
    from ivae_amd.dist import DataParallelTrainer, row_range
    from ivae_amd.trainer import init_packed_params

    shape = ops.ModelShape(20, 20, 200, 50, 3, "lrelu", .1)
    B, N = args.batch, 200 * 5000
    lo, hi = row_range(B, rank, world)
    g = torch.Generator(device=dev).manual_seed(1)           # same stream on every rank: identical global batches
    params = init_packed_params(shape, 1, 0, dev)[0].contiguous()
    tr = DataParallelTrainer(shape, params, decoder_var=.01, lr=1e-3)
    nb = 8                                                   # a few resident global batches, cycled
    xs = torch.randn(nb, B, 20, generator=g, device=dev)
    lab = torch.randint(0, 200, (nb, B), generator=g, device=dev)
    us = torch.zeros(nb, B, 200, device=dev).scatter_(2, lab.unsqueeze(-1), 1.)
    es = torch.randn(nb, B, 20, generator=g, device=dev)
    kw = dict(a=1., b=1., c=2., d=1., N=float(N), pairwise=True) if args.pairwise else dict(N=float(N))

    def run(n):
        for s in range(n):
            i = s % nb
            loss, _ = tr.step(xs[i, lo:hi], us[i, lo:hi], es[i, lo:hi], B, **kw)
        return loss

    run(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    loss = run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0          # this rank's time; the MAX over ranks below is the job's
    if world > 1:
        dist.barrier()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "iVAE train samples/sec (fwd+bwd+Adam), data-parallel", "value": round(args.steps * B / dt, 1),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong",
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: iVAE d=20, M=200, 3-layer lrelu MLPs hidden 50, global batch %d, "
                                   "Adam" % B, "pairwise": bool(args.pairwise),
                       "parallelism": "dp%d: row shards, one all-reduce of [grad | loss] per step" % world},
            "final_loss": round(float(loss[0]), 3)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
