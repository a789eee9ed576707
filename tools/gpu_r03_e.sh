#!/bin/bash
# r03 call E: why did the developed run blow up in call D?  base twice (with / without the exact engine), single-buffer
# density, the old one-word-ahead loop, the pre-r03 kernels
set -o pipefail
out=gpurun_out/r3
mkdir -p $out
B="--no-cpu-baseline --steps 20 --warmup 5"
run() {  # tag lib extra...
  tag=$1; lib=$2; shift 2
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B "$@" > $out/e_$tag.json 2> $out/e_$tag.err; echo "$tag rc=$?"
  python - $out/e_$tag.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); dv=d['developed']
print(sys.argv[1], d['value'], '| developed', dv['value'], 'max_cell', dv['max_cell_count'], 'max_vel', dv['max_vel'], dv['kernels_ms'])
PY
}
run base_noexact dieselfluid_amd/lib/libdslsph.so --exact-steps 0
run base_exact dieselfluid_amd/lib/libdslsph.so
run sb_noexact dieselfluid_amd/lib/libdslsph_sb.so --exact-steps 0
run oldahead_noexact dieselfluid_amd/lib/libdslsph_oldahead.so --exact-steps 0
run old_noexact dieselfluid_amd/lib/libdslsph_old.so --exact-steps 0
timeout -k 10 600 python -m pytest tests/test_gpu_developed.py tests/test_gpu_parity.py -q -x > $out/pytest_e.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_e.log
