#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 500 python tools/pci_switch_point.py 160 96 | tee $out/af_switch.jsonl
