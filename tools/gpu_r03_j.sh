#!/bin/bash
# r03 call J: (1) soak: which builds blow up in the developed run?  (2) two-word walk: parity + A/B
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
: > $out/soak.jsonl
soak() { tag=$1; shift; env "$@" timeout -k 10 150 python tools/soak_developed.py 252 10500 $tag 2>> $out/soak.err | grep '^{' >> $out/soak.jsonl; tail -1 $out/soak.jsonl | cut -c1-260; }
soak keys0_a DSL_CELL_KEYS=0
soak keys0_b DSL_CELL_KEYS=0
soak keys1_a DSL_CELL_KEYS=1
soak sb_keys0_a DSL_CELL_KEYS=0 DSL_LIB=$PWD/dieselfluid_amd/lib/libdslsph_sb.so
soak sb_keys0_b DSL_CELL_KEYS=0 DSL_LIB=$PWD/dieselfluid_amd/lib/libdslsph_sb.so
soak keys0_c DSL_CELL_KEYS=0
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_developed.py tests/test_gpu_edge_cases.py -q -x > $out/pytest_j.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_j.log
timeout -k 10 300 python bench.py --no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5 > $out/j_base.json 2> $out/j_base.err; echo "bench rc=$?"
python tools/benchline.py $out/j_base.json
