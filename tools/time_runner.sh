#!/bin/bash
# Wall time of the main.py-style flow (python -m ivae_amd) at the reference's yaml configs, and where one run's host time goes.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for args in "--config ivae.yaml" "--config btcvae.yaml" "--config ivae.yaml --n-sims 8"; do
  t0=$(date +%s.%N)
  timeout -k 10 300 python3 -m ivae_amd $args --run /tmp/run_$$ 2>&1 | tail -2
  t1=$(date +%s.%N)
  echo "python -m ivae_amd $args: $(python3 -c "print(round($t1 - $t0, 2))") s wall"
done
timeout -k 10 300 python3 - <<'PY'
import cProfile, io, pstats, runpy, sys
sys.argv = ["ivae_amd", "--config", "ivae.yaml", "--run", "/tmp/run_prof"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_module("ivae_amd", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue()[:5000])
PY
