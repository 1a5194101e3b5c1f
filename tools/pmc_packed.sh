#!/bin/bash
# HBM traffic of the generic kernels in a 256-model pack of the configs[1] shape (separate --pmc passes):
#   tools/pmc_packed.sh -> gpurun_out/pmc_packed.txt  (per kernel: mean FETCH_SIZE / WRITE_SIZE per launch, raw counter units)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_packed.txt
: > $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  d=$R/gpurun_out/pmc_packed_$ctr
  rm -rf $d
  timeout -s KILL 150 rocprofv3 --pmc $ctr -d $d -o run -- python3 $R/tools/time_packed.py cfg1 256 > $d.log 2>&1
  python3 $R/tools/pmc_summary.py $(find $d -name '*.db' | head -1) k_ >> $out
  rm -rf $d
done
cat $out
