#!/bin/bash
# r03 call P: two-run window in the developed-flow force walk: parity, A/B, soak
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_developed.py tests/test_gpu_edge_cases.py -q -x > $out/pytest_p.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_p.log
B="--no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5"
for v in nowin base; do
  lib=dieselfluid_amd/lib/libdslsph_$v.so
  [ "$v" = base ] && lib=dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B > $out/p_$v.json 2> $out/p_$v.err; echo "$v rc=$?"
  python tools/benchline.py $out/p_$v.json
done
for rep in 1 2; do
  timeout -k 10 150 python tools/soak_developed.py 252 10500 win_$rep 2>> $out/soak_p.err | grep '^{' >> $out/soak_p.jsonl; tail -1 $out/soak_p.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['tag'], j['steps'], j['bad_at'], j['last'][-1])"
done
