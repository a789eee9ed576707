#!/usr/bin/env python3
"""GPU A/B of the two FAST density kernels (k_density_quad vs k_density_tiled, selected per engine by
DSL_DENSITY_KERNEL at dsl_create): densities and one WCSPH step must agree to summation-order rounding."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from dieselfluid_amd import SPHEngine, scenes  # noqa: E402


def engine(p, kind):
    os.environ["DSL_DENSITY_KERNEL"] = kind
    e = SPHEngine(p, device=0)
    os.environ.pop("DSL_DENSITY_KERNEL", None)
    return e


def compare(name, p, pos, vel=None, steps=2):
    out = {}
    for kind in ("valu", "quad"):
        e = engine(p, kind)
        e.upload("positions", pos)
        if vel is not None:
            e.upload("velocities", vel)
        e.reset_forces()
        res = []
        for _ in range(steps):
            e.density_all()
            res.append(e.download("densities").copy())
            e.force_pass()
            res.append(e.download("positions").copy())
        out[kind] = res
        e.close()
    ok = True
    for i, (a, b) in enumerate(zip(out["valu"], out["quad"])):
        # (not bit for bit between two engines: the in-cell slot order comes from atomics)
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        same = bool(np.nanmax(d) <= 3e-6 * np.nanmax(np.abs(a)))
        ok &= same
        if not same:
            print(f"  {name}: item {i} ({'rho' if i % 2 == 0 else 'pos'}) differs: max {np.nanmax(d):.3e} at {np.nanargmax(d)}"
                  f" ({int((d > 0).sum())} entries), scale {np.nanmax(np.abs(a)):.3e}")
    print(f"{name}: {'identical' if ok else 'DIFFERENT'}")
    return ok


def main():
    ok = True
    p, pos = scenes.dambreak_scene(12, math_mode=1)
    ok &= compare("lattice n3=12", p, pos)
    p, pos = scenes.dambreak_scene(20, math_mode=1)
    ok &= compare("lattice n3=20", p, pos)
    for n in (7, 65, 300, 2000):
        p, _ = scenes.dambreak_scene(12, math_mode=1, positions=False)
        p.n_particles = n
        p.dt = p.dt * 0.02
        rng = np.random.default_rng(n)
        pos = (0.3 + 0.2 * rng.random((n, 3))).astype(np.float32)
        ok &= compare(f"clump n={n}", p, pos)
    # a melted state: run the lattice for a while first
    p, pos = scenes.dambreak_scene(24, math_mode=1)
    e = SPHEngine(p, device=0)
    e.upload("positions", pos)
    e.reset_forces()
    e.wcsph_step(1500)
    mpos, mvel = e.download("positions"), e.download("velocities")
    print("melted: max_cell_count", e.stats().max_cell_count)
    e.close()
    ok &= compare("melted n3=24", p, mpos, mvel, steps=3)
    return 0 if ok else 1


if __name__ == "__main__":
    raise SystemExit(main())
