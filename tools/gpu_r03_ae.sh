#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
for s in 100 400 1000; do
  timeout -k 10 300 python tools/pci_drifted_state.py save 160 $s /tmp/pci$s.npz > /dev/null || exit 1
  echo "state $s"; timeout -k 10 300 python tools/pci_query_spread.py /tmp/pci$s.npz | tee -a $out/ae_spread.txt
done
