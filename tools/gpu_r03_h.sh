#!/bin/bash
# r03 call H: one-launch band pack + fused count bump: slab tests, one emulated middle rank of 16M / 8 (native driver,
# bands through RCCL to the own rank) with the old and the new pack, kernel timeline of one step
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_slab.py -q -x > $out/pytest_h.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_h.log
: > $out/slab_runs.jsonl
for pk in 0 1; do
 for mode in "--native --nccl --no-timing" "--native --nccl --no-timing --no-overlap"; do
  DSL_PACK_ONEPASS=$pk timeout -k 10 200 python tools/slab_periodic_bench.py $mode --steps 200 --warmup 20 2>> $out/slab_runs.err | grep '^{' | sed "s/^{/{\"pack_onepass\": $pk, /" >> $out/slab_runs.jsonl
 done
done
python - <<'PY'
import json, os
for l in open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3/slab_runs.jsonl'):
    j = json.loads(l); print('pack_onepass', j['pack_onepass'], j['driver'], 'overlap', j['overlap'], j['ms_per_step'], 'host', j['host_enqueue_ms_per_step'], 'live', j['live_with_ghosts'], 'owned', j['owned'])
PY
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_slab -- python3 $GRAFT_REPO_ROOT/tools/slab_periodic_bench.py --native --nccl --no-timing --steps 100 --warmup 20 > /dev/null 2> $out/prof_slab.err
cp $(find /tmp/prof_slab -name "*kernel_stats.csv" | head -1) $out/slab_native_kernel_stats.csv
python3 - <<'PY'
import csv, glob, os
f = glob.glob('/tmp/prof_slab/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_cell_rank' in r['Kernel_Name']]
a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
t0 = int(rows[a]['Start_Timestamp'])
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3/slab_native_timeline.txt', 'w')
for r in rows[a:b + 1]:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    name = r['Kernel_Name'].split('(')[0][-60:]
    out.write(f"{s/1000:9.1f} {e/1000:9.1f} {(e-s)/1000:8.1f} us  q{r.get('Queue_Id','?')} {name}\n")
out.close()
PY
cat $out/slab_native_timeline.txt
