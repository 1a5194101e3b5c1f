#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the four PMC passes of tools/pmc_traffic.sh (gpurun_out/pmc/{FETCH,WRITE}_SIZE_{125,20}.txt):
two launch lengths give the per-launch fixed part (state in / out) and the per-step part (batch rows, loss words).
usage: pmc_fit.py <round-tag>"""
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"


def mean(ctr, steps):
    line = open(os.path.join(R, "gpurun_out", "pmc", f"{ctr}_{steps}.txt")).read().splitlines()[0]
    m = re.search(r"n=\s*(\d+) mean=\s*([\d.]+) mean_dur_us=\s*([\d.]+)", line)
    return int(m.group(1)), float(m.group(2)), float(m.group(3)), line


rows = {(c, s): mean(c, s) for c in ("FETCH_SIZE", "WRITE_SIZE") for s in (125, 20)}
tot = {s: rows[("FETCH_SIZE", s)][1] + rows[("WRITE_SIZE", s)][1] for s in (125, 20)}
per_step = (tot[125] - tot[20]) / (125 - 20)
fixed = tot[125] - 125 * per_step
out = {
    "_comment": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (tools/pmc_traffic.sh) of python3 bench.py --steps 2S "
                "--warmup S --steps-per-launch S at S = 125 and S = 20; per-launch means of k_train_steps<FCfg<10,10,40,50>>, "
                "rocprofv3 unit = KB (raw summary: profiles/%s_pmc_hbm_traffic.txt; fit: tools/pmc_fit.py).  Two points give the "
                "fixed part (state in/out once per launch) and the per-step part (gathered batch rows + one loss word per model); "
                "both scale with models_per_gpu.  The guide's gfx950 x2 correction of FETCH_SIZE is calibrated for 16-B-per-lane "
                "streaming reads; this kernel's reads are 4-B row gathers and 4-B state loads (uncalibrated), so raw values are "
                "reported.  Algorithmic: 3342 KB per step (256 models x 64 rows x 204 B), 61.3 MB of state in + 61.3 MB out per "
                "launch.  bench.py's untimed spin-up advances in whole launches, so every dispatch of a pass has the pass's launch "
                "length (except one short launch per 625-step epoch at S = 20) and the mean durations below are per-launch times." % tag,
    "kernel": "k_train_steps", "models_per_gpu": 256,
    "fixed_kb_per_launch": round(fixed, 1), "kb_per_step": round(per_step, 1),
    "fetch_kb": {str(s): rows[("FETCH_SIZE", s)][1] for s in (125, 20)},
    "write_kb": {str(s): rows[("WRITE_SIZE", s)][1] for s in (125, 20)},
    "mean_launch_us": {str(s): rows[("FETCH_SIZE", s)][2] for s in (125, 20)},
    "round": tag,
}
json.dump(out, open(os.path.join(R, "profiles", "hbm_traffic.json"), "w"), indent=2)
with open(os.path.join(R, "profiles", f"{tag}_pmc_hbm_traffic.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.sh) of bench.py at 125 and 20 steps per launch; unit KB\n")
    for s in (125, 20):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f.write(rows[(c, s)][3] + "\n")
print(json.dumps({k: v for k, v in out.items() if k != "_comment"}))
