#!/bin/bash
# prof_one.sh <tag> <cfg> [IVAE_TUNE]: rocprofv3 kernel stats of tools/time_generic.py <cfg> -> gpurun_out/ks_<tag>.csv
set -e
tag=$1; cfg=$2; export IVAE_TUNE=${3:-}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o run -- python3 $R/tools/time_generic.py $cfg > $R/gpurun_out/prof_$tag.log 2>&1
python3 $R/tools/kernel_stats.py $(find $R/gpurun_out/prof_$tag -name '*.db' | head -1) $R/gpurun_out/ks_$tag.csv
rm -rf $R/gpurun_out/prof_$tag
echo "== $tag $cfg tune=$IVAE_TUNE: $(tail -1 $R/gpurun_out/prof_$tag.log)"
grep -E "k_fwd|k_dec|k_bwd|k_wgrad|k_reduce|k_row" $R/gpurun_out/ks_$tag.csv | cut -d, -f1,2,4 | cut -c1-40,100-
