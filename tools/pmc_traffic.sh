#!/bin/bash
# HBM traffic of k_train_steps at two launch lengths (separate --pmc passes, as MI355X_MICROARCH.md prescribes):
#   tools/pmc_traffic.sh  ->  gpurun_out/pmc/{fetch,write}_{125,20}.txt ; tools/pmc_fit.py turns them into profiles/hbm_traffic.json
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
for steps in 125 20; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/pmc/${ctr}_${steps}
    rm -rf $d
    timeout -s KILL 120 rocprofv3 --pmc $ctr -d $d -o run -- python3 $R/bench.py --steps $((steps * 2)) --warmup $steps --steps-per-launch $steps --no-cpu-baseline --no-other-configs > $d.log 2>&1
    python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_train_steps > $R/gpurun_out/pmc/${ctr}_${steps}.txt
    rm -rf $d
    cat $R/gpurun_out/pmc/${ctr}_${steps}.txt
  done
done
