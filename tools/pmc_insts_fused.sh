#!/bin/bash
# Instruction mix of k_train_steps (one --pmc pass, SQ counters only) -> gpurun_out/pmc_insts_fused.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/pmc_insts_f
rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH -d $d -o run -- python3 $R/bench.py --steps 250 --warmup 125 --steps-per-launch 125 --no-cpu-baseline --no-other-configs > $d.log 2>&1
python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_train_steps > $R/gpurun_out/pmc_insts_fused.txt
rm -rf $d
cat $R/gpurun_out/pmc_insts_fused.txt
