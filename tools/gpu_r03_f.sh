#!/bin/bash
# r03 call F: EXACT with shared exact divisions + PF=8: full parity suite, bench line, MFMA microbench counters
set -o pipefail
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -x -m gpu > $out/pytest_f.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest_f.log
timeout -k 10 400 python bench.py --no-cpu-baseline > $out/f_base.json 2> $out/f_base.err; echo "bench rc=$?"
python tools/benchline.py $out/f_base.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/f_base.json').read().strip().splitlines()[-1]); print('exact', d['exact']); print('developed', d['developed']['max_vel'], d['developed']['max_cell_count'])
PY
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_pipe.hip -o /tmp/mfma_pipe 2> /dev/null
mkdir -p $out/mfma_prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/mfma_prof -- /tmp/mfma_pipe > /dev/null 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find $out/mfma_prof -type f | head -10
