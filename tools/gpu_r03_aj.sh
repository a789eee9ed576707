#!/bin/bash
# r03 call AJ: query rows (no scan / scatter pass in the binned PCISPH iteration): parity tests, A/B on drifted states
out=$GRAFT_REPO_ROOT/gpurun_out/r3; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_pci_drift.py tests/test_gpu_parity.py tests/test_gpu_slab.py tests/test_gpu_developed.py -x -q -m gpu -k "pci" > $out/pytest_aj.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep -v "Gloo\|socket\|amdgpu.ids" $out/pytest_aj.log | tail -4
[ $rc -eq 0 ] || exit 1
rm -f $out/aj_ab.jsonl
for s in 100 400 1000; do
  timeout -k 10 300 python tools/pci_drifted_state.py save 160 $s /tmp/pci$s.npz > /dev/null || exit 1
  for rows in 1 0 1 0; do
    echo -n "state $s rows $rows: "; DSL_PCI_QROWS=$rows timeout -k 10 200 python tools/pci_drifted_state.py run /tmp/pci$s.npz 20 1 | tee -a $out/aj_ab.jsonl | cut -c1-60,170-260
  done
done
