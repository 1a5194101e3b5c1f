#!/bin/bash
# Instruction mix of the generic kernels (one --pmc pass, SQ counters only): tools/pmc_insts.sh <cfg> -> gpurun_out/pmc_insts_<cfg>.txt
set -e
cfg=${1:-cfg3}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/pmc_insts
rm -rf $d
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA -d $d -o run -- python3 $R/tools/time_generic.py $cfg > $d.log 2>&1
python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_ > $R/gpurun_out/pmc_insts_$cfg.txt
rm -rf $d
cat $R/gpurun_out/pmc_insts_$cfg.txt
