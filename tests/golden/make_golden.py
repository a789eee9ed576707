#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/dsl_oracle.c) in this
container.  The reference itself cannot run here (Go, no toolchain), so these vectors
pin the oracle against regressions and give the GPU tests oracle-independent expected
values; they are NOT outputs of the reference ("parity unpinned").

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers  # noqa: E402
from dieselfluid_amd import scenes  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def fixture_init_lsh():
    """(1) sph.Init on the n3 = 8 and 16 lattices, lsh_ref sampler with fixed hash
    vectors: densities and forces after Init, pcidelta of the default system."""
    hv = po.default_hash_vectors(7)
    out = {"hash_vectors": hv}
    for n3 in (8, 16):
        s = po.OracleSPH.init(po.params_reference(n3), hash_vectors=hv, pci=True)
        out[f"n{n3}_densities"] = s.densities()
        out[f"n{n3}_forces"] = s.forces()
        out[f"n{n3}_delta"] = np.float32(s.delta)
        out[f"n{n3}_samples0"] = s.get_samples(0)
    save("init_lsh_ref.npz", **out)


def fixture_reference_grid():
    """(2a) reference constants, grid neighbours, 12^3 jittered lattice: the individual
    passes and one PCISPH step with 5 and 4 max iterations."""
    n3 = 12
    p, _ = scenes.reference_scene(n3)
    pos = helpers.jittered_lattice(n3, 0.2)
    vel = helpers.seeded_velocities(n3 ** 3, 0.1)
    q = helpers.oracle_params(p)
    out = {"positions": pos, "velocities": vel}
    s = po.OracleSPH.from_state(q, pos, vel=vel)
    s.density_all()
    out["densities"] = s.densities()
    s.viscous_all()
    out["viscous_force"] = s.forces()
    s2 = po.OracleSPH.from_state(q, pos, vel=vel)
    s2.density_all()
    s2.gradient_pressure_force()
    out["pressure_force"] = s2.forces()
    s2.pressure_all()
    out["pressures"] = s2.pressures()
    pos2 = helpers.jittered_lattice(n3, 0.1)
    vel2 = helpers.seeded_velocities(n3 ** 3, 0.05)
    out["pci_positions0"], out["pci_velocities0"] = pos2, vel2
    for iters in (5, 4):
        q.pci_max_iters = iters
        s3 = po.OracleSPH.from_state(q, pos2, vel=vel2)
        s3.delta = 1.0e-4
        s3.pcisph_begin()
        s3.pcisph_step(1)
        out[f"pci{iters}_positions"] = s3.positions()
        out[f"pci{iters}_velocities"] = s3.velocities()
        out[f"pci{iters}_error"] = np.float32(s3.pci_error)
        out[f"pci{iters}_iters"] = np.int32(s3.pci_iters)
    save("reference_grid_n12.npz", **out)


def fixture_dambreak():
    """(2b) build-defined dam-break (pressure + viscosity + walls), 12^3, after 1 and 10
    WCSPH steps."""
    n3 = 12
    p, pos = scenes.dambreak_scene(n3)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    s = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    out = {"positions0": pos}
    s.wcsph_step(1)
    out["x1"], out["v1"], out["rho1"] = s.positions(), s.velocities(), s.densities()
    s.wcsph_step(9)
    out["x10"], out["v10"], out["rho10"] = s.positions(), s.velocities(), s.densities()
    save("dambreak_n12.npz", **out)


if __name__ == "__main__":
    fixture_init_lsh()
    fixture_reference_grid()
    fixture_dambreak()
