#!/bin/bash
# per-rank cost of the 8-way slab split emulated on one GPU: python driver (copy / RCCL) and the library's own driver
out=gpurun_out/r2
mkdir -p $out
: > $out/slab_runs.jsonl
for mode in "" "--nccl" "--native --nccl" "--native --nccl --no-overlap"; do
  timeout -k 10 200 python tools/slab_periodic_bench.py $mode --steps 100 --warmup 20 >> $out/slab_runs.jsonl 2>> $out/slab_runs.err || { echo "slab bench ($mode) failed"; tail -5 $out/slab_runs.err; }
done
cat $out/slab_runs.jsonl
