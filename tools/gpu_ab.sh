#!/bin/bash
# A/B of the density kernels + parity suite on the GPU box; logs under gpurun_out/r2/
set -o pipefail
out=gpurun_out/r2
mkdir -p $out
tools/mfma_layout > $out/mfma_layout.log 2>&1; echo "mfma_layout rc=$?" | tee -a $out/summary.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_a.log 2>&1; echo "pytest rc=$?" | tee -a $out/summary.log
tail -5 $out/pytest_a.log
DSL_DENSITY_KERNEL=valu timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_valu.json 2> $out/bench_valu.err; echo "bench valu rc=$?" | tee -a $out/summary.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_quad.json 2> $out/bench_quad.err; echo "bench quad rc=$?" | tee -a $out/summary.log
python - <<'PY'
import json
for k in ("valu","quad"):
    try:
        j=json.loads(open(f"gpurun_out/r2/bench_{k}.json").read().strip().splitlines()[-1])
        print(k, j["value"], j["ms_per_step"], j["kernels_ms"], j["roofline"]["pass_frac_68B"])
    except Exception as e:
        print(k, "failed", e)
PY
