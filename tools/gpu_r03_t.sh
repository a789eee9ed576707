#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 300 python tools/pair_diag.py 1 > $out/pair_diag1.log 2>&1; echo "rc=$?"; grep -v "amdgpu.ids" $out/pair_diag1.log | tail -40
