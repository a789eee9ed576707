#!/bin/bash
# per-kernel durations of a developed flow: tools/gpu_dev_trace.sh [n3=126] [steps=6000]; steps 200-400 and the last 200
# steps of a rocprofv3 kernel trace of tools/long_run.py
out=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $out
N3=${1:-126}; STEPS=${2:-6000}
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/devtrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/devtrace -- python3 $GRAFT_REPO_ROOT/tools/long_run.py $N3 $STEPS $STEPS > /dev/null 2> $out/devtrace.err
python3 - <<'PY' | tee $out/devtrace_${N3}.txt
import csv, glob, collections, os
f = glob.glob('/tmp/devtrace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_cell_rank' in r['Kernel_Name']]
first, last = idx[200], idx[400]
a, b = idx[-201], idx[-1]
def summ(lo, hi, tag):
    d = collections.defaultdict(float)
    for r in rows[lo:hi]:
        d[r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
    n = sum(1 for r in rows[lo:hi] if 'k_cell_rank' in r['Kernel_Name'])
    print(tag, 'steps', n)
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]):
        print(f"  {v / n:8.1f} us/step  {k}")
summ(first, last, 'lattice (steps 200-400)')
summ(a, b, 'developed (last 200 steps)')
PY
