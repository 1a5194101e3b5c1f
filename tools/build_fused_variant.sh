#!/bin/bash
# build_fused_variant.sh <name> <extra -D flags...>: recompiles only fused_train.hip (config-2 shape) and links it with the
# objects `make` left behind -> var/libivae_<name>.so (travels to the GPU box, never committed).  Timing experiments only.
name=$1; shift
set -e
mkdir -p var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function '-DFT_SHAPES(X)=X(10,10,40,50)' "$@" \
    -c ivae_amd/csrc/fused_train.hip -o var/fused_$name.o
objs=$(ls ivae_amd/csrc/*.o | grep -v fused_train.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs var/fused_$name.o -o var/libivae_$name.so
rm -f var/fused_$name.o
