#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2; mkdir -p $out
for g in 1 0; do
DSL_SLAB_GRAPHS=$g DSL_SLAB_GRAPH_DEBUG=1 timeout -k 10 200 python tools/slab_periodic_bench.py --native --nccl --no-timing --steps 200 --warmup 20 2> $out/slab3.err | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('graphs=$g', j['driver'], j['overlap'], j['ms_per_step'], 'host', j['host_enqueue_ms_per_step'])"
grep "dsl\]" $out/slab3.err
done
