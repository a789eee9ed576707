#!/bin/bash
# A/B of prebuilt library variants on the GPU box: tools/gpu_variants.sh name1 name2 ...
# (each dieselfluid_amd/lib/libdslsph_<name>.so, built in the container with tools/build_variant.sh; "base" = the product build)
out=gpurun_out/r2
mkdir -p $out
for v in "$@"; do
  lib=dieselfluid_amd/lib/libdslsph_$v.so
  [ "$v" = base ] && lib=dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --developed-steps 0 > $out/var_$v.json 2> $out/var_$v.err || { echo "$v FAILED"; tail -3 $out/var_$v.err; exit 1; }
  python - "$v" <<'PY'
import json,sys
v=sys.argv[1]
j=json.loads(open(f'gpurun_out/r2/var_{v}.json').read().strip().splitlines()[-1])
print(v, j['value'], j['ms_per_step'], j['kernels_ms'])
PY
  grep "dsl diag" $out/var_$v.err
done
exit 0
