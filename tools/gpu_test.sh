#!/bin/bash
# run (a subset of) the GPU tests on the box: tools/gpu_test.sh [pytest args]
set -o pipefail
mkdir -p gpurun_out/r2
timeout -k 10 1100 python -m pytest "$@" > gpurun_out/r2/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/r2/pytest_sel.log
