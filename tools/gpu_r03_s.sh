#!/bin/bash
# r03 call S: density with two targets per lane (k_density_pair): full parity suite, A/B, soak
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $out/pytest_s.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_s.log
B="--no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5"
for dp in 0 1; do
  DSL_DENSITY_PAIR=$dp timeout -k 10 300 python bench.py $B > $out/s_pair$dp.json 2> $out/s_pair$dp.err; echo "pair$dp rc=$?"
  python tools/benchline.py $out/s_pair$dp.json
  DSL_DENSITY_PAIR=$dp timeout -k 10 300 python bench.py $B --method pcisph --n3 160 --developed-steps 0 > $out/s_pci_pair$dp.json 2> $out/s_pci_pair$dp.err; echo "pci pair$dp rc=$?"
  python tools/benchline.py $out/s_pci_pair$dp.json
done
for rep in 1 2; do
  timeout -k 10 150 python tools/soak_developed.py 252 10500 pair_$rep 2>> $out/soak_s.err | grep '^{' >> $out/soak_s.jsonl; tail -1 $out/soak_s.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['tag'], j['steps'], j['bad_at'], j['last'][-1])"
done
