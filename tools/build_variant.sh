#!/bin/bash
# tools/build_variant.sh name [-DMACRO ...]: dieselfluid_amd/lib/libdslsph_<name>.so with extra macros (travels to the GPU box, git-ignored)
name=$1; shift
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 "$@" \
  -o dieselfluid_amd/lib/libdslsph_$name.so dieselfluid_amd/csrc/dslsph.hip
