"""Developed-flow state of the bench scene, saved and re-run (tools only; GPU box).
  python tools/dev_state.py save /tmp/dev.npz [n3] [steps]   # step the dam-break `steps` times, save x, v
  python tools/dev_state.py run  /tmp/dev.npz [n3] [steps]   # load it, step `steps` more (what a profiler should see)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dieselfluid_amd import SPHEngine, scenes

mode, path = sys.argv[1], sys.argv[2]
n3 = int(sys.argv[3]) if len(sys.argv) > 3 else 252
steps = int(sys.argv[4]) if len(sys.argv) > 4 else (10000 if mode == "save" else 40)
p, pos = scenes.dambreak_scene(n3, math_mode=1)
eng = SPHEngine(p, device=0)
if mode == "save":
    eng.upload("positions", pos)
    eng.reset_forces()
    done = 0
    while done < steps:
        eng.wcsph_step(min(1000, steps - done))
        done += min(1000, steps - done)
        print("saved-run step", done, "max_vel", eng.stats().max_vel, flush=True)
    np.savez(path, pos=eng.download("positions"), vel=eng.download("velocities"))
else:
    z = np.load(path)
    eng.upload("positions", z["pos"])
    eng.upload("velocities", z["vel"])
    eng.reset_forces()
    eng.wcsph_step(5)
    eng.sync()
    t0 = time.perf_counter()
    eng.wcsph_step(steps)
    eng.sync()
    print("developed ms/step", round((time.perf_counter() - t0) / steps * 1e3, 4), "max_cell", eng.stats().max_cell_count, flush=True)
eng.close()
