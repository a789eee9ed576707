#!/bin/bash
# kernel stats of the short bench command for one library variant: tools/gpu_kstats.sh <variant>
out=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $out
lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph_$1.so; [ "$1" = base ] && lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph.so
export TMPDIR=/tmp DSL_LIB=$lib; cd /tmp; rm -rf /tmp/ks_$1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$1 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --developed-steps 0 --steps 10 --warmup 3 > /dev/null 2> $out/ks_$1.err
f=$(find /tmp/ks_$1 -name "*kernel_stats.csv" | head -1); cp $f $out/ks_$1.csv
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{float(r['AverageNs'])/1000:9.1f} us x{r['Calls']:>4}  {r['Name'].split('(')[0][-70:]}")
PY
