#!/bin/bash
# rocprofv3 --kernel-trace --stats of one model's step on the shape-generic kernels at BASELINE configs[0], [2], [3]
# (tools/time_generic.py) -> gpurun_out/ks_cfg{0,2,3}.csv; copy the ones to keep into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in cfg0 cfg2 cfg3; do
  rm -rf $R/gpurun_out/prof_$c
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o run -- python3 $R/tools/time_generic.py $c > $R/gpurun_out/prof_$c.log 2>&1
  python3 $R/tools/kernel_stats.py $(find $R/gpurun_out/prof_$c -name '*.db' | head -1) $R/gpurun_out/ks_$c.csv
  rm -rf $R/gpurun_out/prof_$c
  tail -1 $R/gpurun_out/prof_$c.log
  head -12 $R/gpurun_out/ks_$c.csv
done
