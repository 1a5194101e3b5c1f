#!/usr/bin/env python3
"""Headline benchmark: iVAE train samples/sec (ELBO fwd + bwd + Adam) on MI355X.

Workload (BASELINE.json configs[1]): iVAE d=10, M=40 segments x 1000 samples, 3-layer
leaky-ReLU MLPs hidden 50, batch 64.  One GPU trains a packed set of independent models
(seed sweep, north_star "batched-model set per GPU"); every model does the full sequential
fwd+bwd+Adam step at batch 64 on its own shuffled synthetic dataset.  Ranks hold disjoint
model sets and never communicate on the data path (weak scaling).

    python bench.py --gpus N --steps K --warmup W     (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed) and
`cpu_baseline` (oracle/torch_ref port of the same step on the host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak


DATA_SEEDS = 4        # datasets drawn by the reference generator per rank; the packed models cycle through them


def reference_datasets(n_models, n_seg, n_per_seg, d, first_seed, device):
    """Inputs of SURVEY.md §8(d): `generate_data(n_per_seg, n_seg, d, d, n_layers=3, seed, prior='gauss',
    activation='lrelu', repeat_linearity=True)` -- ivae_amd.data's bit-exact restatement of data/data.py:200-304,367-369
    (sha256-pinned against the reference's arrays, tests/test_data_cpu.py) -- for DATA_SEEDS seeds (about a second of host
    numpy each), tiled over the packed models: model m trains on dataset m % DATA_SEEDS, with its own initialisation,
    shuffles and noise.  -> x (n_models, N, d), u (n_models, N, n_seg) one-hot, resident in HBM."""
    from ivae_amd.data import generate_data
    xs, us = [], []
    for k in range(DATA_SEEDS):
        _, X, U, _, _ = generate_data(n_per_seg, n_seg, d, d, n_layers=3, seed=first_seed + k, prior='gauss',
                                      activation='lrelu', batch_size=0, repeat_linearity=True)
        xs.append(torch.from_numpy(np.ascontiguousarray(X, np.float32)))
        us.append(torch.from_numpy(np.ascontiguousarray(U, np.float32)))
    pick = torch.arange(n_models) % DATA_SEEDS
    return torch.stack(xs)[pick].to(device).contiguous(), torch.stack(us)[pick].to(device).contiguous()


def flops_per_sample(shape):
    """2*(Wsum_fwd + Wsum_wgrad + Wsum_dgrad), dgrad excluding the first layers
    of logl/g/logv.  Config 2 -> 103 kFLOP."""
    H, L = shape.hidden_dim, shape.n_layers
    tot = 0
    for m in shape.mlps:
        i, o = shape.mlp_io(m)
        dims = [i] + [H] * (L - 1) + [o]
        w = [dims[k] * dims[k + 1] for k in range(L)]
        wsum = sum(w)
        dgrad = wsum - (0 if m == "f" else w[0])
        tot += 2 * (2 * wsum + dgrad)
    return tot


def measured_traffic(kernel, models, steps_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/hbm_traffic.json:
    FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc runs of this same command at two launch lengths).  A launch moves
    a fixed part (parameters and Adam moments in and out, once) plus a per-step part (the gathered batch rows), both
    proportional to the number of packed models, so the figure for this run's geometry is
    models/256 * (fixed + per_step * steps_per_launch)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None
    if t.get("kernel") != kernel or "fixed_kb_per_launch" not in t:
        return None
    return round(1e3 * models / t["models_per_gpu"] * (t["fixed_kb_per_launch"] + t["kb_per_step"] * steps_per_launch), 1)


def cpu_port_rate(dims, act, B, coef_kw, seconds, decoder_var=.01):
    """oracle/torch_ref.Trainer (== the reference's loop: elbo -> backward -> Adam.step, a torch-CPU port pinned by the
    reference fixtures) at one shape, timed on the host at 1 thread and at all allotted threads (the reference is SLOWER
    multi-threaded at batch 64, SURVEY.md §6, so both are reported and the better one is the baseline).
    -> {threads: samples/s}, host cores, threads allotted."""
    from oracle import np_ref, torch_ref
    dd, dl, M, H = dims
    spec = np_ref.Spec(data_dim=dd, latent_dim=dl, aux_dim=M, hidden_dim=H, n_layers=3, activation=act, decoder_var=decoder_var)
    coef = np_ref.Coef(**coef_kw)
    g = torch.Generator().manual_seed(0)
    n_b = 8
    x = torch.randn(n_b, B, dd, generator=g)
    u = None
    if M:
        u = torch.zeros(n_b * B, M)
        u[torch.arange(n_b * B), torch.randint(0, M, (n_b * B,), generator=g)] = 1
        u = u.view(n_b, B, M)
    host = len(os.sched_getaffinity(0))
    avail = int(os.environ.get("OMP_NUM_THREADS", "0")) or host
    results = {}
    for nt in sorted({1, min(avail, 16)}):
        torch.set_num_threads(nt)
        tr = torch_ref.Trainer(torch_ref.xavier_params(spec, 0), spec, coef, lr=1e-2)
        tr.step(x[0], None if u is None else u[0])                 # warm-up (allocations, thread pool)
        t0, steps = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds / 2 or steps < 2:
            tr.step(x[steps % n_b], None if u is None else u[steps % n_b])
            steps += 1
        results[nt] = steps * B / (time.perf_counter() - t0)
    return results, host, avail


def cpu_baseline(seconds=10.0):
    """The headline workload on the host cores: ONE model (the reference trains one model per process), batch 64."""
    results, host, avail = cpu_port_rate((10, 10, 40, 50), "lrelu", 64, dict(pairwise=False), seconds)
    best = max(results, key=results.get)
    return {"value": round(results[best], 1), "unit": "samples/s", "cores": best, "host_cores": host,
            "threads_allotted": avail, "by_threads": {str(k): round(v, 1) for k, v in results.items()}, "kind": "port",
            "sample": "cfg2 iVAE d=10 M=40 H=50 B=64 single model, oracle/torch_ref.Trainer (torch CPU elbo+backward+"
                      "Adam), ~%.0f s per thread setting; by threads: %s" % (
                          seconds / 2, {k: round(v) for k, v in results.items()})}


# ---- the other BASELINE configs on the shape-generic kernels (one model each; extra keys of the JSON line) ----------
OTHER_CONFIGS = [
    # key, workload, (data, latent, aux, hidden, act), batch, loss weights, N
    ("configs[0]", "configs/ivae.yaml: cleanIVAE d=5 M=40 H=100 xtanh, batch 64, TC-decomposed loss (pairwise B x B x d terms)",
     (5, 5, 40, 100, "xtanh"), 64, dict(a=100., b=1., c=0., d=10., pairwise=True, decoder_var=.1), 40000),
    ("configs[2]", "configs/btcvae.yaml: cleanVAE d=5 H=100 xtanh, batch 1024, beta-TCVAE weights (pairwise terms)",
     (5, 5, 0, 100, "xtanh"), 1024, dict(a=1000., b=1., c=15., d=1., pairwise=True, decoder_var=.1), 40000),
    ("configs[3]", "large-batch iVAE d=20 M=200 H=50 lrelu, batch 4096, plain ELBO (one GPU's view of the DP run)",
     (20, 20, 200, 50, "lrelu"), 4096, dict(pairwise=False), 1000000),
]


def other_config_lines(dev, steps=200, cpu=True):
    """fwd+bwd[+pairwise]+Adam through ivae_elbo_adam_steps (one library call for all steps), HIP events on the launch
    stream around the call; FLOPs per SURVEY.md §8(d): MLP part + 2.5 * 8 * B * d per sample for the pairwise terms."""
    from ivae_amd import ops
    out = []
    for key, workload, (dd, dl, M, H, act), B, ck, N in OTHER_CONFIGS:
        shape = ops.ModelShape(dd, dl, M, H, 3, act, .1)
        g = torch.Generator(device=dev).manual_seed(0)
        P = shape.n_params()
        params = (torch.rand(P, generator=g, device=dev) - .5) * .2
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        xs = torch.randn(steps, B, dd, generator=g, device=dev)
        us = None
        if M:
            us = torch.zeros(steps * B, M, device=dev)
            us[torch.arange(steps * B, device=dev), torch.randint(0, M, (steps * B,), generator=g, device=dev)] = 1
            us = us.view(steps, B, M)
        eps = torch.randn(steps, B, dl, generator=g, device=dev)
        coef = ops.make_coef(N=N, global_batch=B, **ck)
        ws = ops.Workspace()
        z = torch.empty_like(eps)
        ops.elbo_adam_steps(shape, coef, params, m, v, xs[:20], None if us is None else us[:20], eps[:20], ws, 0, 1e-3,
                            z_out=z[:20])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        loss, _ = ops.elbo_adam_steps(shape, coef, params, m, v, xs, us, eps, ws, 20, 1e-3, z_out=z)
        e1.record()
        torch.cuda.synchronize()
        dt = 1e-3 * e0.elapsed_time(e1) / steps
        fl = flops_per_sample(shape) + (int(2.5 * 8 * B * dl) if ck.get("pairwise") else 0)
        tf = fl * B / dt / 1e12
        assert bool(torch.isfinite(loss).all()), key
        line = {"config": key, "workload": workload, "impl": "shape-generic kernels, one model", "value": round(B / dt, 1),
                "unit": "samples/s", "ms_per_step": round(1e3 * dt, 5), "steps": steps,
                "roofline": {"bound": "mfma", "achieved": round(tf, 4), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / FP32_PEAK_TFLOPS, 5), "algorithmic_flops_per_sample": fl}}
        if ck.get("pairwise"):
            # The B x B x d terms are not a contraction.  Per (i, j, k) the four walks issue 14.5 packed-fp32 / plain vector
            # instructions and 3 v_exp_f32 per lane; with every SIMD busy the chip sustains one packed instruction per
            # 2.07 ns and one v_exp_f32 per 3.4 ns per SIMD (the clock drops to ~1.3 GHz under this load:
            # tools/ubench/valu_pk_rate.hip, profiles/r3_ubench_valu_pk_rate.txt):
            # 1024 SIMDs x 64 lanes / (14.5 * 2.07 + 3 * 3.4) ns = 1.63 T elements/s.
            el = B * B * dl
            line["pairwise_roofline"] = {"bound": "valu/exp", "unit": "T (i,j,k) elements/s", "peak": 1.63,
                                         "elements_per_step": el, "achieved_at_least": round(el / dt / 1e12, 4),
                                         "note": "instruction-issue roof of the pairwise kernels (power-limited rate, measured); "
                                                 "achieved_at_least divides by the WHOLE step's time, MLP kernels included; "
                                                 "the mfma line above prices the whole step by the MLP FLOPs only"}
        if cpu:
            res, host, avail = cpu_port_rate((dd, dl, M, H), act, B, {k: v for k, v in ck.items() if k != "decoder_var"} | {"N": float(N)},
                                             4.0, decoder_var=ck.get("decoder_var", .01))
            best = max(res, key=res.get)
            line["cpu_baseline"] = {"value": round(res[best], 1), "unit": "samples/s", "cores": best, "host_cores": host,
                                    "by_threads": {str(k): round(v, 1) for k, v in res.items()}, "kind": "port",
                                    "sample": "oracle/torch_ref.Trainer at this shape, ~2 s per thread setting"}
            line["x_cpu_baseline"] = round(B / dt / res[best], 1)
        out.append(line)
    return out


def run_dp(args, rank, world, dev):
    """--mode dp: BASELINE configs[3] -- ONE large-batch iVAE (d=20, 200 segments x 5000 samples, batch 4096), data-parallel
    over the ranks (north_star's second multi-GPU mode; strong scaling: the global batch is fixed, rank r takes
    row_range(batch, r, N) of it), per step ivae_elbo_fwd_bwd on the shard, ONE all-reduce(sum) of [gradient | loss] over
    RCCL, the identical fused Adam on every rank."""
    import torch.distributed as dist
    from ivae_amd import ops
    from ivae_amd.dist import DataParallelTrainer, row_range
    from ivae_amd.trainer import init_packed_params
    shape = ops.ModelShape(20, 20, 200, 50, 3, "lrelu", .1)
    B, N = 4096, 200 * 5000
    lo, hi = row_range(B, rank, world)
    g = torch.Generator(device=dev).manual_seed(1)           # same stream on every rank: identical global batches
    params = init_packed_params(shape, 1, 0, dev)[0].contiguous()
    tr = DataParallelTrainer(shape, params, decoder_var=.01, lr=1e-3)
    nb = 8                                                   # a few resident global batches, cycled
    xs = torch.randn(nb, B, 20, generator=g, device=dev)
    lab = torch.randint(0, 200, (nb, B), generator=g, device=dev)
    us = torch.zeros(nb, B, 200, device=dev).scatter_(2, lab.unsqueeze(-1), 1.)
    es = torch.randn(nb, B, 20, generator=g, device=dev)

    def run(n):
        for s_ in range(n):
            i = s_ % nb
            loss, _ = tr.step(xs[i, lo:hi], us[i, lo:hi], es[i, lo:hi], B, N=float(N))
        return loss

    run(max(args.warmup, 10))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    loss = run(args.steps)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    fps = flops_per_sample(shape)
    gpu_s = 1e-3 * e0.elapsed_time(e1) / args.steps
    tf = fps * B / gpu_s / 1e12
    out = {"metric": "iVAE train samples/sec (fwd+bwd+Adam), data-parallel large batch (BASELINE configs[3])",
           "value": round(args.steps * B / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
           "warmup": max(args.warmup, 10), "ms_per_step": round(1e3 * dt / args.steps, 5), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3]: iVAE d=20, 200 segments, 3-layer lrelu MLPs hidden 50, global batch "
                                  "4096, plain ELBO + Adam", "global_batch": B, "rows_per_rank": hi - lo,
                      "collective": ("none (one rank)" if world == 1 else
                                     "%s all-reduce(sum) of [gradient | loss] (%d floats) per step, world size %d" % (
                                         "RCCL" if BACKEND == "nccl" else BACKEND + " [REHEARSAL: ranks share GPUs, not a measurement]",
                                         params.numel() + 1, world)),
                      "backend": BACKEND if world > 1 else None, "impl": "shape-generic kernels (k_chain, k_wgrad, k_reduce+Adam)",
                      "parallelism": "dp%d" % world},
           "final_loss": round(float(loss[0]), 3),
           "roofline": {"bound": "mfma", "kernel": "one step's launches on rank 0 (HIP events around the K steps, collective included)",
                        "achieved": round(tf / world, 4), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s per GPU",
                        "frac": round(tf / world / FP32_PEAK_TFLOPS, 5), "traffic": None,
                        "algorithmic_flops_per_sample": fps}}
    if world == 1 and not args.no_cpu_baseline:
        res, host, avail = cpu_port_rate((20, 20, 200, 50), "lrelu", B, dict(pairwise=False), 8.0)
        best = max(res, key=res.get)
        out["cpu_baseline"] = {"value": round(res[best], 1), "unit": "samples/s", "cores": best, "host_cores": host,
                               "by_threads": {str(k): round(v, 1) for k, v in res.items()}, "kind": "port",
                               "sample": "oracle/torch_ref.Trainer at this shape and batch, ~4 s per thread setting"}
    print(json.dumps(out))


# "nccl" (= RCCL) always, except when tests rehearse the N > 1 path of this file on a one-GPU box: IVAE_BENCH_BACKEND=gloo
# lets the ranks share a GPU (RCCL refuses two ranks on one device); such a line says so in config.parallelism and is not a
# measurement of anything.
BACKEND = os.environ.get("IVAE_BENCH_BACKEND", "nccl")


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run as a CHILD
    process, before this process touches the GPU) and leave with its exit code."""
    import socket
    import subprocess
    have = torch.cuda.device_count()          # counts devices without initialising the GPU runtime
    if n > have and BACKEND == "nccl":
        sys.exit("bench.py: --gpus %d asked for, %d GPU(s) visible; refusing to report a smaller run as N=%d" % (n, have, n))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1250)
    ap.add_argument("--warmup", type=int, default=125)
    ap.add_argument("--models", type=int, default=256, help="independent models packed per GPU")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--impl", choices=["fused", "generic"], default="fused")
    ap.add_argument("--steps-per-launch", type=int, default=0, help="fused path: steps per k_train_steps launch (default 125)")
    ap.add_argument("--spinup-ms", type=float, default=300.0,
                    help="untimed device warm-up before the W warmup steps: training steps on a scratch copy of the "
                         "models, so that clocks and caches are at their steady state when the K timed steps start")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["pack", "dp"], default="pack",
                    help="pack (default): the headline -- independent packed models per GPU, no collective (weak scaling); "
                         "dp: BASELINE configs[3], one large-batch model data-parallel over the GPUs with an RCCL gradient "
                         "all-reduce per step (strong scaling)")

    ap.add_argument("--no-other-configs", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if BACKEND != "nccl":                     # rehearsal on a box with fewer GPUs than ranks: the ranks share the GPUs there are
        local_rank %= torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        if BACKEND == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(BACKEND)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.mode == "dp":
        run_dp(args, rank, world, dev)
        if world > 1:
            dist.destroy_process_group()
        return

    from ivae_amd import ops
    from ivae_amd.trainer import FusedPackedTrainer, PackedTrainer

    shape = ops.ModelShape(10, 10, 40, 50, 3, "lrelu", .1)
    M, B = args.models, args.batch
    x, u = reference_datasets(M, 40, 1000, 10, first_seed=1 + DATA_SEEDS * rank, device=dev)
    fused = args.impl == "fused"
    if fused:
        tr = FusedPackedTrainer(shape, n_models=M, batch=B, lr=1e-2, seed=rank, device=dev)
        if args.steps_per_launch > 0:
            tr.CHUNK = args.steps_per_launch
    else:
        tr = PackedTrainer(shape, n_models=M, batch=B, lr=1e-2, seed=rank, device=dev)
    tr.set_data(x, u)

    def advance(n, timed):
        if fused:
            tr.run(n, timed=timed)
        else:
            for _ in range(n):
                tr.step(timed=timed)

    if args.spinup_ms > 0:
        # The chip idles at a low clock and a K-step timed region can be shorter than its ramp: run the same kernels on
        # a scratch copy of the state first (not counted, state restored), then the W warmup steps, then time K steps.
        saved = [t.clone() for t in (tr.params, tr.exp_avg, tr.exp_avg_sq)]
        t_keep, t0 = tr.t, time.perf_counter()
        # whole launches only: a profiler's per-kernel mean over the process then averages launches of one length
        chunk = tr.CHUNK if fused else 125
        while time.perf_counter() - t0 < args.spinup_ms * 1e-3:
            advance(chunk * max(1, 125 // chunk), False)
            torch.cuda.synchronize()
        for dst, src in zip((tr.params, tr.exp_avg, tr.exp_avg_sq), saved):
            dst.copy_(src)
        tr.t = t_keep
    advance(args.warmup, False)
    if fused:      # the events of the timed launches exist before the timed region starts
        tr.prepare_timing((args.steps + tr.CHUNK - 1) // tr.CHUNK + 1)
        tr.prepare_indices(args.steps)         # (a run's first permutation: input preparation, like the data itself)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    advance(args.steps, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0          # this rank's K steps; the MAX over ranks below is the job's time
    if world > 1:
        dist.barrier()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = tr.losses().mean().item()
    assert np.isfinite(loss), "training diverged"

    if rank == 0:
        samples = args.steps * B * M * world
        fps = flops_per_sample(shape)
        if fused:
            kern_s, steps_per_launch = tr.mean_kernel_seconds()
            algo_flops = fps * B * M * steps_per_launch
        else:
            kern_s, steps_per_launch = tr.mean_kernel_seconds(), 1
            algo_flops = tr.dominant_flops_per_sample(fps) * B * M
        achieved = algo_flops / kern_s / 1e12
        out = {
            "metric": "iVAE train samples/sec (fwd+bwd+Adam) at d=10,M=40",
            "value": round(samples / dt, 1), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: iVAE d=10, 40 segments x 1000 samples, 3-layer lrelu MLPs "
                                   "hidden 50, batch 64, plain ELBO + Adam lr 1e-2",
                       "models_per_gpu": M, "batch": B, "impl": tr.impl, "rng": tr.rng,
                       "device_warmup_ms": args.spinup_ms,
                       "parallelism": "independent model sets per GPU, no collective" + (
                           "" if BACKEND == "nccl" else " [REHEARSAL over %s: ranks share GPUs, not a measurement]" % BACKEND)},
            "per_model": {"steps_per_s": round(args.steps / dt, 1), "samples_per_s": round(args.steps * B / dt, 1)},
            "final_mean_loss": round(loss, 3),
            "roofline": {"bound": "mfma", "kernel": tr.dominant_kernel, "achieved": round(achieved, 4),
                         "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 5),
                         "traffic": measured_traffic(tr.dominant_kernel, M, steps_per_launch) if fused else None,
                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, profiles/hbm_traffic.json)",
                         "avg_launch_us": round(kern_s * 1e6, 2),
                         "timing": ("HIP events on the launch stream, recorded inside the library directly before and after the "
                                    "kernel launch (ivae_debug_launch_events)" if fused else
                                    "HIP events on the launch stream around each step's launches"),
                         "steps_per_launch": steps_per_launch, "algorithmic_flops_per_launch": algo_flops},
        }
        if fused:
            # MFMA slots actually issued: 608.5 v_mfma_f32_16x16x4_f32 per wave and step (PMC SQ_INSTS_MFMA,
            # profiles/r2_pmc_instruction_mix_fused.txt; the tiling pads 50 -> 64 columns and 10 -> 16), 8 waves, 2048 FLOP each
            executed = 608.5 * 8 * 2048 * M * steps_per_launch
            out["roofline"]["executed_flops_per_launch"] = executed
            out["roofline"]["executed_frac"] = round(executed / kern_s / 1e12 / FP32_PEAK_TFLOPS, 5)
            out["roofline"]["executed_note"] = "matrix-pipe occupancy: padded MFMA FLOPs / peak; frac prices algorithmic FLOPs only"
            out["config"]["input_pipeline"] = ("FusedPackedTrainer.run, the code path training uses: a launch points into the "
                                               "epoch's device-side permutation table (no per-launch index slice); the "
                                               "permutation of each new epoch is drawn inside the timed region")
            out["config"]["datasets"] = "%d datasets from the bit-exact host generator (seeds %d..%d), tiled over the models" % (
                DATA_SEEDS, 1, DATA_SEEDS)
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_config_lines(dev, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["per_model"]["x_cpu_baseline"] = round(out["per_model"]["samples_per_s"] / out["cpu_baseline"]["value"], 2)
            out["per_model"]["note"] = ("one packed model against the CPU baseline's one model: the like-for-like ratio; "
                                        "`value` is %d such models" % M)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
