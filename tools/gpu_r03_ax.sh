#!/bin/bash
# r03 call AX: un-binned PCISPH iteration with the first targets' predictor state requested one tile ahead: tests, A/B
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_pci_drift.py tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_developed.py -x -q -m gpu -k "pci" > $out/pytest_ax.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_ax.log | tail -3
[ $rc -eq 0 ] || exit 1
for v in base prev base prev base prev; do
  lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$lib timeout -k 10 300 python bench.py --method pcisph --n3 160 --no-cpu-baseline --drift-steps 0 > $out/ax_$v.json 2> $out/ax_$v.err || { echo "$v FAILED"; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("$out/ax_$v.json") if l.startswith("{")][-1])
print("$v pcisph", j["value"], j["ms_per_step"], "pci_density", j["kernels_ms"]["pci_density"])
PY
done
