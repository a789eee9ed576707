#!/bin/bash
# r03 call Y: rehearsal of bench.py --gpus N on one card (gloo): python protocol and the library driver over the host-staged transport
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
for drv in python native; do
  for n in 2 4; do
    DSL_BENCH_BACKEND=gloo DSL_BENCH_SLAB_DRIVER=$drv timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --n3 126 --steps 10 --warmup 3 --no-cpu-baseline > $out/y_${drv}_n$n.json 2> $out/y_${drv}_n$n.err; echo "$drv n=$n rc=$?"
    grep '^{' $out/y_${drv}_n$n.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['slab_driver'], j['slab_overflow'], j['slab_band_missed'], j['max_vel'])"
  done
done
DSL_BENCH_BACKEND=gloo DSL_BENCH_SLAB_DRIVER=native timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29619 bench.py --gpus 2 --n3 80 --steps 10 --warmup 3 --no-cpu-baseline --method pcisph --extra-terms > $out/y_pci_n2.json 2> $out/y_pci_n2.err; echo "pcisph native n=2 rc=$?"
grep '^{' $out/y_pci_n2.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['slab_driver'], j['slab_overflow'])"
