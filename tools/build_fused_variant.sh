#!/bin/bash
# build_fused_variant.sh <name> <extra -D flags...>: like build_variant.sh, but recompiles only fused_train.hip (config-2 shape)
# and links it with the objects `make` left behind.  Timing experiments only (never shipped).
name=$1; shift
set -e
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function '-DFT_SHAPES(X)=X(10,10,40,50)' "$@" \
    -c ivae_amd/csrc/fused_train.hip -o build/fused_$name.o
objs=$(ls ivae_amd/csrc/*.o | grep -v fused_train.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs build/fused_$name.o -o build/libivae_$name.so
