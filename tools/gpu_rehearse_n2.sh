#!/bin/bash
# bench.py's N>1 path rehearsed on the one-GPU box: 2 ranks on cuda:0, gloo transport (the Python protocol)
mkdir -p gpurun_out/r2
DSL_BENCH_BACKEND=gloo DSL_BENCH_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --n3 126 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench_n2_gloo.json 2> gpurun_out/r2/bench_n2_gloo.err; echo "n2 rc=$?"; tail -c 1500 gpurun_out/r2/bench_n2_gloo.json
