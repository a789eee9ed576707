#!/bin/bash
set -o pipefail
out=gpurun_out/r2
mkdir -p $out
tools/mfma_rate > $out/mfma_rate.log 2>&1; cat $out/mfma_rate.log
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -DDSL_DIAG_STAMPS -o /tmp/libdsl_diag.so dieselfluid_amd/csrc/dslsph.hip 2> $out/diag_build.log
DSL_DENSITY_KERNEL=valu DSL_LIB=/tmp/libdsl_diag.so timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_diag.json 2> $out/bench_diag.err; echo "diag rc=$?"
grep "dsl diag" $out/bench_diag.err
for k in valu quad; do
DSL_DENSITY_KERNEL=$k timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_$k.json 2> $out/bench_$k.err
python -c "
import json,sys
j=json.loads(open('gpurun_out/r2/bench_$k.json').read().strip().splitlines()[-1]); print('$k', j['value'], j['kernels_ms'])"
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_b.log
