#!/bin/bash
# Where the waves of the generic kernels spend their cycles in a 256-model pack (one --pmc pass, SQ counters only):
#   tools/pmc_waits_packed.sh -> gpurun_out/pmc_waits_packed.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/pmc_waits
rm -rf $d
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $d -o run -- python3 $R/tools/time_packed.py cfg1 256 > $d.log 2>&1
python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_ > $R/gpurun_out/pmc_waits_packed.txt
rm -rf $d
grep -E "^k_|^void k_" $R/gpurun_out/pmc_waits_packed.txt
