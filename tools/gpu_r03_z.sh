#!/bin/bash
# r03 call Z: PCISPH 4M over 1500 steps -- where the never-resynchronised predictor goes and what it costs
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 500 python tools/pci_long_run.py 160 1500 50 > $out/z_pci_long.jsonl 2> $out/z_pci_long.err; echo "rc=$?"
cat $out/z_pci_long.jsonl | cut -c1-330
