#!/bin/bash
# r03 call AT: soaks of the final build -- developed WCSPH flow (three-runs-per-loop walk) x3, binned PCISPH x3: repeated runs must end in the same bits
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
: > $out/at_soak.jsonl
for i in 1 2 3; do
  timeout -k 10 300 python tools/soak_developed.py 252 10500 final$i 2>/dev/null | grep '^{' | tee -a $out/at_soak.jsonl | cut -c1-220
done
: > $out/at_pci_soak.jsonl
for i in 1 2 3; do
  timeout -k 10 300 python tools/pci_soak.py 160 800 2>/dev/null | grep '^{' | tee -a $out/at_pci_soak.jsonl
done
