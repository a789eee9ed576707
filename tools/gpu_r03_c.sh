#!/bin/bash
# r03 call C: double-buffered density / PCISPH density: parity suite, A/B against the previous build, PCISPH, stamps
set -o pipefail
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $out/pytest_c.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest_c.log
B="--no-cpu-baseline --steps 20 --warmup 5"
for v in prev base; do
  lib=dieselfluid_amd/lib/libdslsph_$v.so
  [ "$v" = base ] && lib=dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B > $out/c_$v.json 2> $out/c_$v.err; echo "$v rc=$?"
  python tools/benchline.py $out/c_$v.json
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B --method pcisph --n3 160 > $out/c_pci_$v.json 2> $out/c_pci_$v.err; echo "pci $v rc=$?"
  python tools/benchline.py $out/c_pci_$v.json
done
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 -DDSL_DIAG_STAMPS -o /tmp/libdsl_diag.so dieselfluid_amd/csrc/dslsph.hip 2> $out/diag_build.log
DSL_LIB=/tmp/libdsl_diag.so timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --developed-steps 0 --exact-steps 0 > $out/bench_diag_c.json 2> $out/bench_diag_c.err; echo "diag rc=$?"
grep "dsl diag" $out/bench_diag_c.err
