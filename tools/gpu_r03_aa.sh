#!/bin/bash
# r03 call AA: binned DensityF queries -- parity tests, then the long PCISPH run again
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_pci_drift.py tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q -m gpu -k "pci" -s > $out/pytest_aa.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $out/pytest_aa.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/pci_long_run.py 160 1500 50 > $out/aa_pci_long.jsonl 2> $out/aa_pci_long.err; echo "rc=$?"
cut -c1-200 $out/aa_pci_long.jsonl
