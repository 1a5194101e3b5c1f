#!/bin/bash
# Where the waves of k_train_steps spend their cycles (one --pmc pass, SQ counters only) -> gpurun_out/pmc_waits_fused.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/pmc_waits_f
rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $d -o run -- python3 $R/bench.py --steps 250 --warmup 125 --steps-per-launch 125 --no-cpu-baseline --no-other-configs > $d.log 2>&1
python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_train_steps > $R/gpurun_out/pmc_waits_fused.txt
rm -rf $d
cat $R/gpurun_out/pmc_waits_fused.txt
