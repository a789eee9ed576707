#!/bin/bash
# r03 call O: what the developed flow's extra kernel time is made of: VALU instructions, lane utilisation and clocks of the
# density / force launches on the lattice and 10000 steps in (state saved by one process, profiled in another)
set -o pipefail
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r3
mkdir -p $out
python tools/dev_state.py save /tmp/dev.npz 252 10000 > $out/dev_save.log 2>&1; echo "save rc=$?"; tail -1 $out/dev_save.log
python tools/dev_state.py save /tmp/lat.npz 252 20 >> $out/dev_save.log 2>&1; echo "save lattice rc=$?"
export TMPDIR=/tmp
cd /tmp
for tag in lat dev; do
  rm -rf /tmp/prof_$tag
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAVES -d /tmp/prof_$tag -- python3 $R/tools/dev_state.py run /tmp/$tag.npz 252 30 > $out/pmc_$tag.log 2>&1; echo "rocprof $tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
out = open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r3/dev_pmc_summary.txt', 'w')
for tag in ('lat', 'dev'):
    f = glob.glob(f'/tmp/prof_{tag}/**/*counter_collection.csv', recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '')
        key = (int(r['Dispatch_Id']), name)
        per.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
    by = collections.OrderedDict()
    for (d, name), v in per.items():
        if v.get('SQ_INSTS_VALU', 0) > 1e6: by.setdefault(name, []).append(v)
    for name, rows in by.items():
        sel = rows[5:]
        if not sel: continue
        avg = lambda c: sum(v.get(c, 0) for v in sel) / len(sel)
        iv, tc, ga, il, act = avg('SQ_INSTS_VALU'), avg('SQ_THREAD_CYCLES_VALU'), avg('GRBM_GUI_ACTIVE'), avg('SQ_INSTS_LDS'), avg('SQ_ACTIVE_INST_VALU')
        line = (f"{tag} {name[:64]:64s} n={len(sel):3d} VALU insts {iv:.4g} lanes/inst {tc / max(act * 4, 1):.1f} "
                f"clocks {ga / 8:.4g} clk/inst/SIMD {ga / 8 * 1024 / max(iv, 1):.2f} LDS insts {il:.4g}")
        print(line); out.write(line + "\n")
out.close()
PY
