#!/bin/bash
out=gpurun_out/r2; mkdir -p $out
for v in "$@"; do
  lib=dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py --method pcisph --n3 160 --extra-terms --steps 20 --warmup 5 --no-cpu-baseline > $out/pcixs_$v.json 2> $out/pcixs_$v.err || { echo "$v FAILED"; tail -3 $out/pcixs_$v.err; exit 1; }
  python -c "
import json; j=json.loads(open('$out/pcixs_$v.json').read().strip().splitlines()[-1]); print('pci+xs $v', j['value'], j['ms_per_step'], j['kernels_ms'])"
done
