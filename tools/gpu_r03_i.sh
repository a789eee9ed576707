#!/bin/bash
# r03 call I: in-cell ordering from per-cell key rows (one scatter pass): full parity suite, A/B with the developed flow,
# slab rank
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $out/pytest_i.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_i.log
B="--no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5"
for ck in 0 1; do
  DSL_CELL_KEYS=$ck timeout -k 10 300 python bench.py $B > $out/i_keys$ck.json 2> $out/i_keys$ck.err; echo "keys$ck rc=$?"
  python tools/benchline.py $out/i_keys$ck.json
done
: > $out/slab_runs_i.jsonl
for ck in 0 1; do
  DSL_CELL_KEYS=$ck timeout -k 10 200 python tools/slab_periodic_bench.py --native --nccl --no-timing --steps 200 --warmup 20 2>> $out/slab_runs_i.err | grep '^{' | sed "s/^{/{\"cell_keys\": $ck, /" >> $out/slab_runs_i.jsonl
done
python - <<'PY'
import json, os
for l in open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3/slab_runs_i.jsonl'):
    j = json.loads(l); print('cell_keys', j['cell_keys'], j['driver'], 'overlap', j['overlap'], j['ms_per_step'])
PY
