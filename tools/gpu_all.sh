#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r2/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r2/pytest_all.log
timeout -k 10 300 python bench.py --math exact --steps 5 --warmup 2 --no-cpu-baseline --developed-steps 0 > gpurun_out/r2/wcsph_16m_exact_bench.json 2>/dev/null; python -c "
import json; j=json.loads(open('gpurun_out/r2/wcsph_16m_exact_bench.json').read().strip().splitlines()[-1]); print('16M exact', j['value'], j['ms_per_step'], j['kernels_ms'])"
