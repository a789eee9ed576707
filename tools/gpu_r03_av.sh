#!/bin/bash
# r03 call AV: the masks' "valid" word removed (derived from run lengths): parity tests, A/B against the previous build
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_developed.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_host.py tests/test_gpu_slab.py -x -q -m gpu > $out/pytest_av.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_av.log | tail -3
[ $rc -eq 0 ] || exit 1
bash tools/gpu_variants.sh base prev base prev
for v in base prev; do
  lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/dieselfluid_amd/lib/libdslsph.so
  DSL_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline > $out/av_$v.json 2> $out/av_$v.err; echo -n "$v "; python tools/benchline.py $out/av_$v.json
done
