#!/bin/bash
# r03 call G: one-launch prefix scan: parity suite, A/B at 16M / 1M / PCISPH 4M
set -o pipefail
out=gpurun_out/r3
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_developed.py tests/test_gpu_golden.py tests/test_gpu_lsh.py -q -x > $out/pytest_g.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_g.log
B="--no-cpu-baseline --exact-steps 0 --developed-steps 0"
for s in 0 1; do
  DSL_SCAN_ONEPASS=$s timeout -k 10 200 python bench.py $B --steps 40 --warmup 10 > $out/g_16m_scan$s.json 2> $out/g_16m_scan$s.err; echo "16M scan$s rc=$?"; python tools/benchline.py $out/g_16m_scan$s.json
  DSL_SCAN_ONEPASS=$s timeout -k 10 200 python bench.py $B --n3 100 --steps 200 --warmup 20 > $out/g_1m_scan$s.json 2> $out/g_1m_scan$s.err; echo "1M scan$s rc=$?"; python tools/benchline.py $out/g_1m_scan$s.json
  DSL_SCAN_ONEPASS=$s timeout -k 10 200 python bench.py $B --method pcisph --n3 160 --steps 40 --warmup 10 > $out/g_pci_scan$s.json 2> $out/g_pci_scan$s.err; echo "pci scan$s rc=$?"; python tools/benchline.py $out/g_pci_scan$s.json
done
timeout -k 10 600 python -m pytest tests/test_gpu_slab.py tests/test_gpu_boundary.py tests/test_gpu_host.py -q -x > $out/pytest_g2.log 2>&1; echo "pytest2 rc=$?"; tail -4 $out/pytest_g2.log
