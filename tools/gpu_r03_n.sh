#!/bin/bash
# r03 call N: robustness: the new capped-grid determinism test, slab soak (split step, 6000 steps), PCISPH 4M x 200
# steps twice (must agree), 64M WCSPH 600 steps finite, final two soak repeats of the default build
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r3
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_developed.py tests/test_gpu_parity.py -q -x > $out/pytest_n.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_n.log
timeout -k 10 400 python tools/slab_soak.py 126 6000 1000 > $out/slab_soak.jsonl 2> $out/slab_soak.err; echo "slab soak rc=$?"; tail -2 $out/slab_soak.jsonl
python - > $out/pcisph_repeat.log 2>&1 <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from dieselfluid_amd import SPHEngine, scenes
res = []
for rep in range(2):
    p, pos = scenes.dambreak_scene(160, math_mode=1)
    p.pci_max_iters = 4; p.eos_w = p.eos_w / 4; p.delta = 1.0e-7
    e = SPHEngine(p, device=0); e.upload("positions", pos); e.reset_forces(); e.pcisph_begin()
    for _ in range(10): e.pcisph_step(20)
    st = e.stats()
    x = e.download("positions"); res.append(x)
    print("rep", rep, "max_vel", st.max_vel, "iters", st.pci_iters, "err", st.pci_max_error, "finite", bool(np.isfinite(x).all()), flush=True)
    e.close()
print("pcisph 4M x 200 steps: two runs bit-identical:", bool(np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32))))
PY
cat $out/pcisph_repeat.log | tail -3
for rep in 1 2; do
  timeout -k 10 150 python tools/soak_developed.py 252 10500 final_$rep 2>> $out/soak_n.err | grep '^{' >> $out/soak_n.jsonl; tail -1 $out/soak_n.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['tag'], j['steps'], j['bad_at'], j['last'][-1])"
done
timeout -k 10 200 python tools/soak_developed.py 400 600 n400 2>> $out/soak_n.err | tail -1 | cut -c1-200
