#!/bin/bash
# r03 call AN: three runs per loop in the developed-flow force walk: A/B on the bench's developed measurement, then the FAST parity tests
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
for v in base notriple base notriple; do
  lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --exact-steps 0 > $out/an_$v.json 2> $out/an_$v.err || { echo "$v FAILED"; tail -3 $out/an_$v.err; exit 1; }
  echo -n "$v "; python tools/benchline.py $out/an_$v.json
done
timeout -k 10 900 python -m pytest tests/test_gpu_developed.py tests/test_gpu_parity.py tests/test_gpu_boundary.py -x -q -m gpu > $out/pytest_an.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_an.log
