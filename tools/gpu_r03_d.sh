#!/bin/bash
# r03 call D: force-kernel mask-word prefetch depth A/B; pipelined MFMA microbenchmark with the co-execution counters
set -o pipefail
out=gpurun_out/r3
mkdir -p $out
B="--no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5"
for v in base pf3 pf8; do
  lib=dieselfluid_amd/lib/libdslsph_$v.so
  [ "$v" = base ] && lib=dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$PWD/$lib timeout -k 10 300 python bench.py $B > $out/d_$v.json 2> $out/d_$v.err; echo "$v rc=$?"
  python tools/benchline.py $out/d_$v.json
done
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_pipe.hip -o /tmp/mfma_pipe 2> $out/mfma_pipe_build.log; echo "build rc=$?"
timeout -k 10 120 /tmp/mfma_pipe > $out/mfma_pipe.log 2>&1; echo "mfma_pipe rc=$?"; cat $out/mfma_pipe.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/mfma_prof -o mfma -- /tmp/mfma_pipe > $GRAFT_REPO_ROOT/$out/mfma_prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find /tmp/mfma_prof -name "*counter_collection.csv" | head -3
f=$(find /tmp/mfma_prof -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp $f $out/mfma_pipe_counters.csv
python - <<'PY'
import csv, collections, glob, os
f='gpurun_out/r3/mfma_pipe_counters.csv'
if os.path.exists(f):
    rows=list(csv.DictReader(open(f)))
    agg=collections.OrderedDict()
    for r in rows:
        key=(r.get('Dispatch_Id'), r.get('Kernel_Name'))
        agg.setdefault(key, {})[r['Counter_Name']]=float(r['Counter_Value'])
    for (d,k),v in agg.items():
        print(d, k[:24], {n:int(x) for n,x in v.items()})
PY
