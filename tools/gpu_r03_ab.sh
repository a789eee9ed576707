#!/bin/bash
# r03 call AB: query-tiled DensityF -- parity tests, the long PCISPH run with and without the LDS-tiled query sweep
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_pci_drift.py tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q -m gpu -k "pci" > $out/pytest_ab.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $out/pytest_ab.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/pci_long_run.py 160 1500 100 > $out/ab_pci_long.jsonl 2> $out/ab_pci_long.err; echo "rc=$?"
cut -c1-300 $out/ab_pci_long.jsonl
DSL_PCI_QTILED=0 timeout -k 10 400 python tools/pci_long_run.py 160 300 100 > $out/ab_pci_long_untiled.jsonl 2> $out/ab_pci_long_untiled.err; echo "rc=$?"
cut -c1-200 $out/ab_pci_long_untiled.jsonl
