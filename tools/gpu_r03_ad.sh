#!/bin/bash
# r03 call AD: binned PCISPH step in a drifted state (4M, steps 100 / 400 / 1000): per-kernel times
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
tag=${1:-ad}
for s in 100 400 1000; do
  timeout -k 10 300 python tools/pci_drifted_state.py save 160 $s /tmp/pci$s.npz > /dev/null || exit 1
  echo "state $s"; timeout -k 10 200 python tools/pci_drifted_state.py run /tmp/pci$s.npz 20 1 | tee $out/${tag}_run$s.json
done
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o $tag -- python $GRAFT_REPO_ROOT/tools/pci_drifted_state.py run /tmp/pci400.npz 20 1 > $out/${tag}_prof.log 2>&1; echo "rocprof rc=$?"
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); cp $f $out/${tag}_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
