#!/bin/bash
# r03 call L: after the explicit LDS drain in front of every barrier: soak x8, parity suite, bench
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
: > $out/soak_l.jsonl
for rep in 1 2 3 4 5 6 7 8; do
  timeout -k 10 150 python tools/soak_developed.py 252 10500 fixed_$rep 2>> $out/soak_l.err | grep '^{' >> $out/soak_l.jsonl; tail -1 $out/soak_l.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['tag'], j['steps'], j['bad_at'], j['last'][-1])"
done
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $out/pytest_l.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_l.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $out/l_base.json 2> $out/l_base.err; echo "bench rc=$?"
python tools/benchline.py $out/l_base.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/l_base.json').read().strip().splitlines()[-1]); print('developed', d['developed']['max_vel'], d['developed']['max_cell_count'], 'exact', d['exact']['value'])
PY
timeout -k 10 300 python bench.py --no-cpu-baseline --method pcisph --n3 160 --steps 20 --warmup 5 > $out/l_pci.json 2> $out/l_pci.err; echo "pci rc=$?"
python tools/benchline.py $out/l_pci.json
