#!/bin/bash
# r03 call R: probe: what would two targets per lane cost the density sweep (a second test on every record read)?
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
B="--no-cpu-baseline --exact-steps 0 --developed-steps 0 --steps 20 --warmup 5"
for v in base probepair; do
  lib=dieselfluid_amd/lib/libdslsph_$v.so
  [ "$v" = base ] && lib=dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B > $out/r_$v.json 2> $out/r_$v.err; echo "$v rc=$?"
  python tools/benchline.py $out/r_$v.json
done
