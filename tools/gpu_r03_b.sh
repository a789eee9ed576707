#!/bin/bash
# r03 call B: EXACT slab diff diagnostic, full GPU suite on the packed-run-table build, tile list order A/B
set -o pipefail
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 300 python tools/slab_exact_diff.py 2 16 10 > $out/slab_exact_diff.log 2>&1; echo "diff rc=$?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $out/slab_exact_diff.log | tail -30
B="--no-cpu-baseline --exact-steps 0 --steps 20 --warmup 5"
for box in 0 8,4,4 4,4,8 16,4,2 4,8,4; do
  DSL_TILE_BOX=$box timeout -k 10 200 python bench.py $B --developed-steps 0 > $out/box_$box.json 2> $out/box_$box.err; echo "box $box rc=$?"
  python tools/benchline.py $out/box_$box.json
done
for box in 0 8,4,4; do
  DSL_TILE_BOX=$box timeout -k 10 300 python bench.py $B > $out/boxdev_$box.json 2> $out/boxdev_$box.err; echo "boxdev $box rc=$?"
  python tools/benchline.py $out/boxdev_$box.json
done
timeout -k 10 900 python -m pytest tests -q -m gpu --deselect "tests/test_gpu_slab.py::test_hip_slabs_match_single_engine" --deselect "tests/test_gpu_slab.py::test_pcisph_slabs_match_single_engine" > $out/pytest_b.log 2>&1; echo "pytest rc=$?"; tail -8 $out/pytest_b.log
