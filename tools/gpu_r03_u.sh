#!/bin/bash
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_developed.py -q -m gpu > $out/pytest_u.log 2>&1; echo "pytest rc=$?"; tail -6 $out/pytest_u.log
