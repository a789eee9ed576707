#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_golden.py tests/test_gpu_lsh.py -q -x > $out/pytest_x.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_x.log
B="--no-cpu-baseline --exact-steps 0 --developed-steps 0"
timeout -k 10 200 python bench.py $B --steps 40 --warmup 10 > $out/x_16m.json 2> $out/x_16m.err; python tools/benchline.py $out/x_16m.json
timeout -k 10 200 python bench.py $B --n3 100 --steps 200 --warmup 20 > $out/x_1m.json 2> $out/x_1m.err; python tools/benchline.py $out/x_1m.json
timeout -k 10 200 python tools/slab_periodic_bench.py --native --nccl --no-timing --steps 200 --warmup 20 2>> $out/x_slab.err | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('slab', j['ms_per_step'])"
