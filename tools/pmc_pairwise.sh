#!/bin/bash
# Wave states and instruction mix of the sliced pairwise kernels (two --pmc passes, SQ counters only) at rows = cols = 4096,
# d = 20 -> gpurun_out/pmc_pairwise.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_pairwise.txt
: > $out
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_INST_LDS"; do
  d=$R/gpurun_out/pmc_pw
  rm -rf $d
  timeout -s KILL 120 rocprofv3 --pmc $set -d $d -o run -- python3 $R/tools/time_pairwise.py 4096,20 > $d.log 2>&1
  python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_pair >> $out
  rm -rf $d
done
cat $out
