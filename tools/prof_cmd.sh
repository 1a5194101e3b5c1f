#!/bin/bash
# prof_cmd.sh <tag> <python script + args...>: rocprofv3 kernel stats of a command -> gpurun_out/ks_<tag>.csv
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_$tag
( cd $R && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o run -- python3 "$@" > $R/gpurun_out/prof_$tag.log 2>&1 )
python3 $R/tools/kernel_stats.py $(find $R/gpurun_out/prof_$tag -name '*.db' | head -1) $R/gpurun_out/ks_$tag.csv
rm -rf $R/gpurun_out/prof_$tag
