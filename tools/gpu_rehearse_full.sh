#!/bin/bash
# bench.py's N>1 path at the full 16M size on the one-GPU box: N ranks on cuda:0, gloo transport (Python protocol):
# checks slab planes, capacities and message sizes of the configurations the driver's SCALE run uses
mkdir -p gpurun_out/r2
for n in 2 4; do
  DSL_BENCH_BACKEND=gloo DSL_BENCH_DEVICE=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2953$n bench.py --gpus $n --steps 8 --warmup 9 --no-cpu-baseline > gpurun_out/r2/bench_n${n}_full_gloo.json 2> gpurun_out/r2/bench_n${n}_full_gloo.err; echo "n=$n rc=$?"
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
try:
    j = json.loads(open(f'gpurun_out/r2/bench_n{n}_full_gloo.json').read().strip().splitlines()[-1])
    print(n, j['value'], j['ms_per_step'], 'overflow', j['slab_overflow'], 'missed', j['slab_band_missed'], 'live rank0', j['n_live_rank0'], j['config']['parallelism'])
except Exception as e:
    print(n, 'no json', e)
PY
done
