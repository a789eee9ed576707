#!/bin/bash
# tools/build_variant.sh <git-rev|WORK> <name>: the library as of <git-rev> (WORK: the working tree), built to
# variants/libdslsph_<name>.so (git-ignored, travels with gpurun).  A/B runs of two builds in ONE gpurun call -- boxes differ
# by several per cent from call to call --: DSL_LIB=variants/libdslsph_<name>.so python bench.py ...
# The variant must speak the working tree's ABI (the Python binding binds every symbol of include/dslsph.h).
set -e
rev=$1; name=$2; shift 2  # (further arguments: extra compiler flags, e.g. -DDSL_WALK_MIN_GROUPS=256)
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/variants"
if [ "$rev" = WORK ]; then src="$root"; else
  src=$(mktemp -d); git -C "$root" archive "$rev" dieselfluid_amd/csrc include | tar -x -C "$src"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 \
  "$@" -o "$root/variants/libdslsph_$name.so" "$src/dieselfluid_amd/csrc/dslsph.hip"
[ "$rev" = WORK ] || rm -rf "$src"
ls -la "$root/variants/libdslsph_$name.so"
