#!/bin/bash
# cmp_packed.sh <lib name ...>: packed sweep shapes at tile_rows 32 (default) and 16 for library variants
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do
  if [ "$n" = tree ]; then unset IVAE_AMD_LIB; else export IVAE_AMD_LIB=$R/var/libivae_$n.so; fi
  for t in 0,0,0 16,0,0; do
    echo "== $n tune $t"
    IVAE_TUNE=$t timeout -k 10 100 python3 $R/tools/time_packed.py cfg1 256 2>/dev/null || exit 1
    IVAE_TUNE=$t timeout -k 10 100 python3 $R/tools/time_packed.py d40 256 2>/dev/null || exit 1
    IVAE_TUNE=$t timeout -k 10 100 python3 $R/tools/time_packed.py cfg0 64 256 2>/dev/null || exit 1
  done
done
