import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from oracle import pyoracle as po
from dieselfluid_amd import SPHEngine, scenes
n3 = 12
for steps in (1, 2):
    p, pos = scenes.dambreak_scene(n3, math_mode=1)
    h = p.h
    t = np.arange(0.25 * h, 1.0, 0.5 * h, dtype=np.float32)
    u, v = np.meshgrid(t, t, indexing="ij")
    plate = np.stack([u.reshape(-1), np.full(u.size, -0.4 * h, np.float32), v.reshape(-1)], axis=1).astype(np.float32)
    p.capacity = n3 ** 3 + plate.shape[0]
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos); eng.upload("forces", frc); eng.add_boundary_particles(plate)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc); ora.add_boundary(plate)
    eng.wcsph_step(steps); ora.wcsph_step(steps)
    n = n3 ** 3
    gx, ox = eng.download("positions")[:n], ora.positions()
    gb, ob = ~np.isfinite(gx).all(axis=1), ~np.isfinite(ox).all(axis=1)
    print(f"steps {steps}: gpu bad {gb.sum()} oracle bad {ob.sum()} both {np.sum(gb & ob)} gpu-only {np.sum(gb & ~ob)} oracle-only {np.sum(~gb & ob)}")
    for k in np.nonzero(gb != ob)[0][:8]:
        print("   particle", k, "start", pos[k], "gpu", gx[k], "ora", ox[k], "rho gpu", eng.download("densities")[k], "ora", ora.densities()[k])
    eng.close()
