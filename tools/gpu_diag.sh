#!/bin/bash
# bench + per-phase stamps + parity suite on the GPU box; logs under gpurun_out/r2/
set -o pipefail
out=gpurun_out/r2
mkdir -p $out
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python -c "
import json,sys
j=json.loads(open('gpurun_out/r2/bench.json').read().strip().splitlines()[-1]); print('bench', j['value'], j['kernels_ms'], j['roofline']['pass_frac_68B'])"
if [ "$1" != "nodiag" ]; then
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -DDSL_DIAG_STAMPS -o /tmp/libdsl_diag.so dieselfluid_amd/csrc/dslsph.hip 2> $out/diag_build.log
DSL_LIB=/tmp/libdsl_diag.so timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench_diag.json 2> $out/bench_diag.err; echo "diag rc=$?"
grep "dsl diag" $out/bench_diag.err
fi
if [ "$2" != "notest" ]; then
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
fi
