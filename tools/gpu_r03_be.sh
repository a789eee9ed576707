#!/bin/bash
# r03 call BE: binned PCISPH without its tiny memset nodes: parity tests, drifted-state timing, determinism
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_pci_drift.py tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_developed.py -x -q -m gpu -k "pci" > $out/pytest_be.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_be.log | tail -3
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/pci_drifted_state.py save 160 400 /tmp/pci400.npz > /dev/null
for i in 1 2 3; do timeout -k 10 200 python tools/pci_drifted_state.py run /tmp/pci400.npz 40 1 | cut -c1-60; done
for i in 1 2; do timeout -k 10 300 python tools/pci_soak.py 160 400 2>/dev/null | grep '^{'; done
