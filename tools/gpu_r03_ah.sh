#!/bin/bash
# r03 call AH: PCISPH tests after the asynchronous drift look + escaped-query flag; PCISPH 4M bench line twice (auto / never)
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_pci_drift.py tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_developed.py -x -q -m gpu -k "pci" > $out/pytest_ah.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_ah.log | tail -5
[ $rc -eq 0 ] || exit 1
for mode in 0 -1 0 -1; do
DSL_PCI_BINNED=$mode timeout -k 10 300 python bench.py --method pcisph --n3 160 --no-cpu-baseline --drift-steps 0 > $out/ah_pcisph_4m_$mode.json 2> $out/ah_pcisph_4m.err; echo "pcisph 4m mode $mode rc=$?"
python - <<PY
import json
j=json.loads([l for l in open("$out/ah_pcisph_4m_$mode.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["kernels_ms"]["pci_density"], j["kernels_ms"]["cell_rank"])
PY
done
