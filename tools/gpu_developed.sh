#!/bin/bash
# developed-flow A/B: tools/gpu_developed.sh <variant> ...  (the default bench incl. its 10000-step segment)
out=gpurun_out/r2; mkdir -p $out
for v in "$@"; do
  lib=dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline > $out/dev_$v.json 2> $out/dev_$v.err || { echo "$v FAILED"; tail -3 $out/dev_$v.err; exit 1; }
  python -c "
import json; j=json.loads(open('$out/dev_$v.json').read().strip().splitlines()[-1]); print('$v', j['value'], j['ms_per_step'], 'developed', j['developed']['value'], j['developed']['ms_per_step'], j['developed']['kernels_ms'])"
done
