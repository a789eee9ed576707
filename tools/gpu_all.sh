#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2
python tools/slab_native_debug.py 2>&1 | grep "^steps"
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r2/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2/pytest_all.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err
python -c "
import json
j=json.loads(open('gpurun_out/r2/bench.json').read().strip().splitlines()[-1]); print('bench', j['value'], j['kernels_ms'], j['roofline']['pass_frac_68B'])"
