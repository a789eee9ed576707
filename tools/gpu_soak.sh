#!/bin/bash
# long runs of the final build: slab soak (conservation, status words) and the melting lattice (step time, finiteness)
out=gpurun_out/r2
mkdir -p $out
timeout -k 10 500 python tools/slab_soak.py 126 6000 1000 2>/dev/null | grep '^{' > $out/slab_soak.jsonl; echo "soak rc=$?"; tail -2 $out/slab_soak.jsonl
timeout -k 10 500 python tools/long_run.py 126 8000 1000 2>/dev/null | grep '^{' > $out/long_run_2m.jsonl; echo "long rc=$?"; tail -3 $out/long_run_2m.jsonl
