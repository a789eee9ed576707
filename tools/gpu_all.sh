#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r2/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r2/pytest_all.log
for mode in "--native" "--native --no-overlap" "" "--nccl"; do
timeout -k 10 300 python tools/slab_periodic_bench.py $mode --no-timing --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$mode', j['driver'], 'overlap', j['overlap'], 'ms/step', j['ms_per_step'], 'host', j['host_enqueue_ms_per_step'])"
done
