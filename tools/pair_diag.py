"""Diagnostic (GPU box): densities of a developed 20^3 snapshot from k_density_pair, from k_density_tiled and from the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from oracle import pyoracle as po
from dieselfluid_amd import SPHEngine, scenes

def run(pair, x, v, p, steps):
    e = SPHEngine(p, device=0)
    e.set_option("density_pair", 1 if pair else 0)
    e.upload("positions", x); e.upload("velocities", v)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (x.shape[0], 1)); e.upload("forces", frc)
    e.density_all()
    rho = e.download("densities")
    e.wcsph_step(steps)
    out = (rho, e.download("positions"), e.download("velocities"), e.download("densities"))
    e.close()
    return out

p, pos = scenes.dambreak_scene(20, math_mode=1)
e = SPHEngine(p, device=0); e.set_option("density_pair", int(sys.argv[1]) if len(sys.argv) > 1 else 0)  # which kernel makes the snapshot
e.upload("positions", pos); e.reset_forces(); e.wcsph_step(2500)
x, v = e.download("positions"), e.download("velocities"); e.close()
frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (x.shape[0], 1))
ora = po.OracleSPH.from_state(helpers.oracle_params(p), x, vel=v, force=frc)
ora.density_all() if hasattr(ora, "density_all") else None
for steps in (1, 2, 3, 4, 6, 8, 10):
    a = run(True, x, v, p, steps); b = run(False, x, v, p, steps)
    o = po.OracleSPH.from_state(helpers.oracle_params(p), x, vel=v, force=frc); o.wcsph_step(steps)
    nd = int(np.count_nonzero(a[0].view(np.uint32) != b[0].view(np.uint32)))
    print("densities before the step: pair vs tiled differ in", nd, "of", a[0].size, "max rel", float(np.max(np.abs(a[0] - b[0]) / np.maximum(b[0], 1e-30))))
    for name, r in (("pair", a), ("tiled", b)):
        print(name, "vs oracle after", steps, "step: x", helpers.rel_err(r[1], o.positions()), "v abs", float(np.abs(r[2].astype(np.float64) - o.velocities()).max()),
              "rho", helpers.rel_err(r[3], o.densities()), "bound", helpers.fast_velocity_tolerance(p, steps))
    for name, r in (("pair", a), ("tiled", b)):
        d = np.abs(r[1].astype(np.float64) - o.positions()).max(axis=1)
        bad = np.nonzero(d > 2e-6)[0]
        print("   ", name, "particles with |dx| > 2e-6:", bad.size, [(int(i), float(d[i]), [float(t) for t in x[i]]) for i in bad[np.argsort(-d[bad])][:4]])
