#!/bin/bash
# prof_cmd.sh <name> <python args...>: rocprofv3 --kernel-trace --stats of `python3 <args>` -> gpurun_out/ks_<name>.csv (+ log)
set -e
name=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
d=$R/gpurun_out/prof_$name
rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $d -o run -- python3 "$@" > $R/gpurun_out/prof_$name.log 2>&1
python3 $R/tools/kernel_stats.py $(find $d -name '*.db' | head -1) $R/gpurun_out/ks_$name.csv "rocprofv3 --kernel-trace --stats -- python3 $*" "$(grep '^{"metric' $R/gpurun_out/prof_$name.log | cut -c1-1500)"
rm -rf $d
head -6 $R/gpurun_out/ks_$name.csv | cut -c1-400
