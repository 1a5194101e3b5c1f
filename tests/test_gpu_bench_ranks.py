"""bench.py's N > 1 path, rehearsed on the one GPU of a test box: the driver launches it as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` over RCCL, which needs N GPUs; with
IVAE_BENCH_BACKEND=gloo the ranks share cuda:0 and rendezvous over gloo instead, so everything else in that path runs as
it will on the node: rank / world from the environment, per-rank datasets and seeds, barrier -> K timed steps -> barrier,
MAX over ranks, whole-job samples, ONE JSON line from rank 0.  The figures of such a run mean nothing (two processes
time-slice one GPU) and the line says so."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(n, extra):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, IVAE_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 alone prints
    return json.loads(lines[0])


def test_two_ranks_pack_mode_prints_one_whole_job_line():
    out = _launch(2, ["--steps", "20", "--warmup", "5", "--spinup-ms", "0"])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["warmup"] == 5 and out["scaling"] == "weak"
    assert out["unit"] == "samples/s" and out["higher_is_better"] is True and out["vs_baseline"] is None
    assert "REHEARSAL" in out["config"]["parallelism"]
    # whole-job samples: both ranks' models / the slower rank's time
    M, B = out["config"]["models_per_gpu"], out["config"]["batch"]
    assert abs(out["value"] - 2 * M * B * 20 / (out["ms_per_step"] * 1e-3 * 20)) <= 1e-3 * out["value"]
    assert "cpu_baseline" not in out and "other_configs" not in out        # rank 0 at N = 1 only
    assert out["roofline"]["steps_per_launch"] == 20 and 0 < out["roofline"]["frac"] < 1
    assert out["final_mean_loss"] == out["final_mean_loss"]                # finite: training ran on both ranks


def test_two_ranks_data_parallel_mode():
    out = _launch(2, ["--mode", "dp", "--steps", "10", "--warmup", "2"])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["parallelism"] == "dp2"
    assert out["config"]["rows_per_rank"] == 2048 and out["config"]["global_batch"] == 4096
    assert out["final_loss"] == out["final_loss"]
