#!/bin/bash
set -o pipefail
out=gpurun_out/r2
mkdir -p $out
timeout -k 10 400 python bench.py --method pcisph --n3 160 --steps 20 --warmup 5 --no-cpu-baseline > $out/pcisph_4m_bench.json 2> $out/pcisph_4m.err; tail -c 900 $out/pcisph_4m_bench.json; echo
timeout -k 10 600 python bench.py --method pcisph --n3 400 --extra-terms --steps 10 --warmup 3 --no-cpu-baseline > $out/pcisph_64m_bench.json 2> $out/pcisph_64m.err; tail -c 900 $out/pcisph_64m_bench.json; echo
timeout -k 10 300 python bench.py --n3 100 --steps 50 --warmup 10 --no-cpu-baseline --developed-steps 0 > $out/wcsph_1m_bench.json 2>/dev/null; python -c "
import json; j=json.loads(open('$out/wcsph_1m_bench.json').read().strip().splitlines()[-1]); print('1M', j['value'], j['ms_per_step'])"
timeout -k 10 300 python bench.py --math exact --steps 5 --warmup 2 --no-cpu-baseline --developed-steps 0 > $out/wcsph_16m_exact_bench.json 2>/dev/null; python -c "
import json; j=json.loads(open('$out/wcsph_16m_exact_bench.json').read().strip().splitlines()[-1]); print('16M exact', j['value'], j['ms_per_step'], j['kernels_ms'])"
