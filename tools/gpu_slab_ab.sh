#!/bin/bash
for v in prev base prev base; do
  lib=$PWD/dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=$PWD/dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$lib timeout -k 10 200 python tools/slab_periodic_bench.py --native --nccl --no-timing --steps 200 --warmup 20 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('slab split $v', j['ms_per_step'])"
done
