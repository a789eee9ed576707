#!/bin/bash
# headline + developed-flow bench (kernel breakdown of both)
mkdir -p gpurun_out/r2
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_full.json 2> gpurun_out/r2/bench_full.err
python -c "
import json; j=json.loads(open('gpurun_out/r2/bench_full.json').read().strip().splitlines()[-1]); print(j['value'], j['kernels_ms']); print(j['developed'])"
