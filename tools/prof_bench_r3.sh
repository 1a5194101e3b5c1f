#!/bin/bash
# Round-3 evidence: rocprofv3 --kernel-trace --stats of the default bench command and of the driver-style 20-step run.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r3
rm -rf $R/gpurun_out/r3/bench
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3/bench -o run -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $R/gpurun_out/r3/bench_prof.log 2>&1
python3 $R/tools/kernel_stats.py $(find $R/gpurun_out/r3/bench -name '*.db' | head -1) $R/gpurun_out/r3/r3_fused_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-configs (1250 steps, 125 per launch, 256 models)" "bench line of the same run: $(grep '^{"metric' $R/gpurun_out/r3/bench_prof.log | cut -c1-1200)"
head -8 $R/gpurun_out/r3/r3_fused_bench_kernel_stats.csv
rm -rf $R/gpurun_out/r3/bench
