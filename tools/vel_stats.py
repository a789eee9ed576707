#!/usr/bin/env python3
"""How far do particles move per step?  Statistics of |v| dt / h over all particles of the 16M bench scene (or --n3) at
a few points of the run: what a skin (DSL_OPT_SKIN) of s h / 2 buys under a global max-displacement bound, and what
it would buy under a bound on the bulk."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dieselfluid_amd import SPHEngine, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n3", type=int, default=252)
ap.add_argument("--points", type=str, default="25,500,2000,5000,10000")
a = ap.parse_args()
p, pos = scenes.dambreak_scene(a.n3)
eng = SPHEngine(p, device=0)
eng.upload("positions", pos); eng.reset_forces(); del pos
done = 0
for pt in [int(x) for x in a.points.split(",")]:
    while done < pt:
        k = min(500, pt - done); eng.wcsph_step(k); done += k
    v = eng.download("velocities").astype(np.float64)
    d = np.sqrt((v * v).sum(axis=1)) * p.dt / p.h
    del v
    qs = [50, 90, 99, 99.9, 99.99, 99.999, 100]
    out = {"step": done, "disp_over_h_per_step": {str(q): float(np.percentile(d, q)) for q in qs},
           "n_above_0.025h": int((d > 0.025).sum()), "n_above_0.05h": int((d > 0.05).sum()), "n": int(d.size)}
    print(json.dumps(out), flush=True)
