#!/bin/bash
# r03 call AK: full GPU suite on the build with query rows; smoke; default bench
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/pytest_ak.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_ak.log | tail -5
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/ak_bench.json 2> $out/ak_bench.err; echo "bench rc=$?"; python tools/benchline.py $out/ak_bench.json
