#!/bin/bash
# build_variant.sh <name> [-Dflags...]: var/libivae_<name>.so = the library with elbo_mfma.hip and capi.hip (which builds
# the layouts) compiled with extra flags; the other objects are the tree's own (run `make` first).
# Use with IVAE_AMD_LIB=/root/repo/var/libivae_<name>.so.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p var
for f in elbo_mfma capi; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c ivae_amd/csrc/$f.hip -o var/${f}_$name.o &
done
wait
others=$(ls ivae_amd/csrc/*.o | grep -v -e elbo_mfma.o -e capi.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 var/elbo_mfma_$name.o var/capi_$name.o $others -o var/libivae_$name.so
rm -f var/elbo_mfma_$name.o var/capi_$name.o
