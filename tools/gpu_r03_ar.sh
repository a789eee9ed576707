#!/bin/bash
# r03 call AR: three runs per walk in the EXACT density kernel: EXACT bench, parity tests
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
for v in base base; do
  timeout -k 10 300 python bench.py --math exact --steps 5 --warmup 2 --no-cpu-baseline --developed-steps 0 > $out/ar_$v.json 2> $out/ar_$v.err || { echo "$v FAILED"; exit 1; }
  echo -n "$v "; python tools/benchline.py $out/ar_$v.json
done
timeout -k 10 900 python -m pytest tests/test_gpu_developed.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_host.py tests/test_gpu_slab.py -x -q -m gpu > $out/pytest_ar.log 2>&1; echo "pytest rc=$?"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_ar.log | tail -3
