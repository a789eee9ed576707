#!/bin/bash
# r03 call A: new slab / boundary tests, baseline bench line (with `exact` and `developed`), per-phase stamps
set -o pipefail
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_slab.py tests/test_gpu_boundary.py tests/test_abi.py -x -q > $out/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_a.log
timeout -k 10 400 python bench.py > $out/bench_base.json 2> $out/bench_base.err; echo "bench rc=$?"
python tools/benchline.py $out/bench_base.json
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -DDSL_DIAG_STAMPS -o /tmp/libdsl_diag.so dieselfluid_amd/csrc/dslsph.hip 2> $out/diag_build.log
DSL_LIB=/tmp/libdsl_diag.so timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --developed-steps 0 --exact-steps 0 > $out/bench_diag.json 2> $out/bench_diag.err; echo "diag rc=$?"
grep "dsl diag" $out/bench_diag.err
