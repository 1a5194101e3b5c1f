#!/bin/bash
# cmp_pack.sh <lib name ...>: packed plain-ELBO sweep shapes (d = 10, 20, 40; 256 models) for the tree's library ("tree")
# and var/libivae_<name>.so variants, in one GPU call
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do
  if [ "$n" = tree ]; then unset IVAE_AMD_LIB; else export IVAE_AMD_LIB=$R/var/libivae_$n.so; fi
  echo "== $n"
  for c in cfg1 d20 d40; do
    timeout -k 10 100 python3 $R/tools/time_packed.py $c 256 2>/dev/null || exit 1
  done
done
