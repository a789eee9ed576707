#!/bin/bash
# r03 call AG: full GPU suite on the build with binned PCISPH queries; PCISPH bench lines with the drifted measurement; long run
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_ag.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_ag.log | tail -5
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --method pcisph --n3 160 --no-cpu-baseline > $out/ag_pcisph_4m.json 2> $out/ag_pcisph_4m.err; echo "pcisph 4m rc=$?"
python - <<PY
import json
j=json.loads([l for l in open("$out/ag_pcisph_4m.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["kernels_ms"] if "kernels_ms" in j else "", "\n drifted:", j["drifted"])
PY
timeout -k 10 400 python bench.py --method pcisph --n3 400 --extra-terms --no-cpu-baseline --drift-steps 200 > $out/ag_pcisph_64m.json 2> $out/ag_pcisph_64m.err; echo "pcisph 64m rc=$?"
python - <<PY
import json
j=json.loads([l for l in open("$out/ag_pcisph_64m.json") if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], "\n drifted:", j["drifted"])
PY
timeout -k 10 400 python tools/pci_long_run.py 160 1500 100 > $out/ag_pci_long.jsonl 2> $out/ag_pci_long.err; echo "long rc=$?"
cut -c1-170 $out/ag_pci_long.jsonl
timeout -k 10 300 python bench.py --no-cpu-baseline --developed-steps 0 > $out/ag_bench.json 2> $out/ag_bench.err; echo "bench rc=$?"; python tools/benchline.py $out/ag_bench.json
