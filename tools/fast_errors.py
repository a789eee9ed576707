#!/usr/bin/env python3
"""FAST-mode error against the oracle, printed next to the error model used for the test tolerances:
a density known to eps_rho moves the Tait pressure by gamma*eps_rho*B and the pressure acceleration by about
gamma*eps_rho*c_s^2/h, i.e. a velocity by that times dt per step."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from dieselfluid_amd import SPHEngine, scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def run(name, p, x, v, steps):
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (x.shape[0], 1))
    eng = SPHEngine(p, device=0)
    eng.upload("positions", x)
    if v is not None:
        eng.upload("velocities", v)
    eng.upload("forces", frc)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), x, vel=v, force=frc)
    eng.wcsph_step(steps); ora.wcsph_step(steps)
    gx, gv, gr = eng.download("positions"), eng.download("velocities"), eng.download("densities")
    ev = np.abs(gv.astype(np.float64) - ora.velocities()).max()
    cs2 = float(p.eos_w) / float(p.mass)
    model = float(p.eos_gamma) * 2e-6 * cs2 / float(p.h) * float(p.dt) * steps
    print(f"{name:28s} steps {steps:3d}: x rel {helpers.rel_err(gx, ora.positions()):.2e}  rho rel {helpers.rel_err(gr, ora.densities()):.2e}  "
          f"|dv| abs {ev:.2e} (= {ev / np.abs(ora.velocities()).max():.2e} of max|v| {np.abs(ora.velocities()).max():.3f}); "
          f"model gamma*2e-6*c_s^2/h*dt*steps = {model:.2e}")
    eng.close()


p, pos = scenes.dambreak_scene(16, math_mode=1)
run("lattice n3=16", p, pos, None, 1)
run("lattice n3=16", p, pos, None, 10)
p, pos = scenes.dambreak_scene(20, math_mode=1)
eng = SPHEngine(p, device=0)
eng.upload("positions", pos)
eng.reset_forces()
eng.wcsph_step(2500)
x, v = eng.download("positions"), eng.download("velocities")
eng.close()
run("melted n3=20", p, x, v, 1)
run("melted n3=20", p, x, v, 10)
run("melted n3=20", p, x, v, 40)
