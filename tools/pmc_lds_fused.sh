#!/bin/bash
# LDS activity of k_train_steps (one --pmc pass, SQ counters only) -> gpurun_out/pmc_lds_fused.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/pmc_lds_f
rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_BUSY_CU_CYCLES -d $d -o run -- python3 $R/bench.py --steps 250 --warmup 125 --steps-per-launch 125 --no-cpu-baseline --no-other-configs > $d.log 2>&1
python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_train_steps > $R/gpurun_out/pmc_lds_fused.txt
rm -rf $d
cat $R/gpurun_out/pmc_lds_fused.txt
