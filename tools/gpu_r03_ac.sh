#!/bin/bash
# r03 call AC: per-kernel trace of the binned PCISPH step in a drifted state (4M, step 400)
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 300 python tools/pci_drifted_state.py save 160 400 /tmp/pci400.npz || exit 1
timeout -k 10 200 python tools/pci_drifted_state.py run /tmp/pci400.npz 20 1 | tee $out/ac_run.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ac -o ac -- python $GRAFT_REPO_ROOT/tools/pci_drifted_state.py run /tmp/pci400.npz 20 1 > $out/ac_prof.log 2>&1; echo "rocprof rc=$?"
f=$(find /tmp/prof_ac -name "*kernel_stats.csv" | head -1); cp $f $out/ac_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {r['Percentage']}")
PY
