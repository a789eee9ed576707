#!/bin/bash
# r03 call AO: the three-runs-per-loop walk where the pass-sharing instantiation runs on a LATTICE: PCISPH set-up sweep, slab rank
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
for v in base notriple base notriple; do
  lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$lib timeout -k 10 300 python bench.py --method pcisph --n3 160 --no-cpu-baseline > $out/ao_pci_$v.json 2> $out/ao_pci_$v.err || { echo "$v FAILED"; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("$out/ao_pci_$v.json") if l.startswith("{")][-1])
print("$v pcisph", j["value"], j["ms_per_step"], "viscous", j["kernels_ms"]["viscous"], "| drifted", j["drifted"]["value"], j["drifted"]["kernels_ms"]["viscous"])
PY
  DSL_LIB=$lib timeout -k 10 200 python tools/slab_periodic_bench.py --native --nccl --no-timing --steps 200 --warmup 20 2>/dev/null | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v slab', j['ms_per_step'])"
done
