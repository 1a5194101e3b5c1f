#!/bin/bash
# cmp_libs.sh <lib name ...>: the generic-path timing set for the tree's library ("tree") and build/libivae_<name>.so variants
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do
  if [ "$n" = tree ]; then unset IVAE_AMD_LIB; else export IVAE_AMD_LIB=$R/var/libivae_$n.so; fi
  echo "== $n"
  timeout -k 10 150 python3 $R/tools/time_generic.py cfg0 cfg1g cfg2 cfg3 cfgw cfg3L 2>/dev/null | cut -c1-60 || exit 1
  timeout -k 10 100 python3 $R/tools/time_packed.py cfg1 256 2>/dev/null || exit 1
  timeout -k 10 100 python3 $R/tools/time_packed.py d40 256 2>/dev/null || exit 1
  timeout -k 10 100 python3 $R/tools/time_packed.py cfg0 64 2>/dev/null || exit 1
done
