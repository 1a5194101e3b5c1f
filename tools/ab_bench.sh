#!/bin/bash
# ab_bench.sh <variant.so> [rounds]: bench.py (headline workload only) alternating the shipped library and a variant in one
# GPU call, so that box-to-box variation drops out -> us per step of each run.  AB_ARGS: extra bench.py arguments
# (e.g. AB_ARGS='--steps 20 --warmup 5', the driver's line)
R=${GRAFT_REPO_ROOT:-/root/repo}
n=${2:-3}
for i in $(seq $n); do
  for lib in "" "$1"; do
    if [ -n "$lib" ]; then export IVAE_AMD_LIB=$R/$lib; else unset IVAE_AMD_LIB; fi
    timeout -k 10 120 python3 $R/bench.py --no-cpu-baseline --no-other-configs $AB_ARGS 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print('${lib:-shipped}', d['ms_per_step'] * 1e3, 'us/step, kernel', d['roofline']['avg_launch_us'], 'us/launch, frac', d['roofline']['frac'])" || exit 1
  done
done
