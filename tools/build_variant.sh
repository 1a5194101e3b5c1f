#!/bin/bash
# build_variant.sh <name> <extra -D flags...>: library variant under build/ for timing experiments (never shipped)
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -shared ivae_amd/csrc/*.hip -o build/libivae_$name.so
