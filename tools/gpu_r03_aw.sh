#!/bin/bash
# r03 call AW: tile-list box shapes (DSL_TILE_BOX) on the 16M bench: which one the XCDs' L2s like best
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out; : > $out/aw_boxes.txt
for box in ${BOXES:-8,4,4 4,4,8 4,8,4 8,8,2 16,4,2 2,8,8 4,4,4 8,8,4 16,8,1 8,4,4}; do
  DSL_TILE_BOX=$box timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --developed-steps 0 --exact-steps 0 > $out/aw_box.json 2> $out/aw_box.err || { echo "$box FAILED"; continue; }
  python - "$box" <<PY | tee -a $out/aw_boxes.txt
import json,sys
j=json.loads([l for l in open("$out/aw_box.json") if l.startswith("{")][-1])
print(sys.argv[1], j['value'], j['ms_per_step'], j['kernels_ms']['density'], j['kernels_ms']['force_integrate'])
PY
done
