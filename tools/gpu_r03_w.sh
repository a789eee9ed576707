#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/pytest_w.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_w.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rep in 1 2 3; do
  timeout -k 10 150 python tools/soak_developed.py 252 10500 final_pair_$rep 2>> $out/soak_w.err | grep '^{' >> $out/soak_w.jsonl; tail -1 $out/soak_w.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['tag'], j['steps'], j['bad_at'], j['last'][-1])"
done
timeout -k 10 400 python tools/slab_soak.py 126 4000 1000 > $out/slab_soak_w.jsonl 2>> $out/soak_w.err; tail -1 $out/slab_soak_w.jsonl | cut -c1-220
