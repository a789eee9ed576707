#!/bin/bash
# r03 call AQ: three runs per loop in the EXACT force walk: A/B on the EXACT bench, then the parity tests (bit for bit)
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
for v in base noexact3 base noexact3; do
  lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$lib timeout -k 10 300 python bench.py --math exact --steps 5 --warmup 2 --no-cpu-baseline --developed-steps 0 > $out/aq_$v.json 2> $out/aq_$v.err || { echo "$v FAILED"; exit 1; }
  echo -n "$v "; python tools/benchline.py $out/aq_$v.json
done
timeout -k 10 900 python -m pytest tests/test_gpu_developed.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_host.py -x -q -m gpu > $out/pytest_aq.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_aq.log
