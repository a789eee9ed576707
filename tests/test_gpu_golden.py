"""GPU vs the committed golden fixtures (tests/golden/*.npz): the expected values come
from files, not from an oracle call, so this parity check is independent of oracle/
being present on the GPU box."""
import os

import numpy as np
import pytest

import helpers


def _agree(a, b, tol, floor=0.0):
    """tol == 0: bit for bit; else max |a - b| < tol * max |b|"""
    if tol == 0:
        return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    return helpers.rel_err(a, b, floor=floor) < tol

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("math_mode,tol", [(0, 0), (1, 5e-5)])
def test_reference_grid_passes_against_fixture(math_mode, tol):
    from dieselfluid_amd import SPHEngine, scenes
    z = np.load(os.path.join(G, "reference_grid_n12.npz"))
    p, _ = scenes.reference_scene(12)
    p.math_mode = math_mode
    eng = SPHEngine(p)
    eng.upload("positions", z["positions"])
    eng.upload("velocities", z["velocities"])
    eng.density_all()
    assert _agree(eng.download("densities"), z["densities"], tol)
    eng.viscous_all()
    assert _agree(eng.download("forces"), z["viscous_force"], tol)
    eng2 = SPHEngine(p)
    eng2.upload("positions", z["positions"])
    eng2.upload("velocities", z["velocities"])
    eng2.density_all()
    eng2.gradient_pressure_force()
    assert _agree(eng2.download("forces"), z["pressure_force"], tol)
    eng2.pressure_all()
    assert _agree(eng2.download("pressures"), z["pressures"], 8 * tol)
    for iters in (5, 4):
        p.pci_max_iters, p.delta = iters, 1.0e-4
        e3 = SPHEngine(p)
        e3.upload("positions", z["pci_positions0"])
        e3.upload("velocities", z["pci_velocities0"])
        e3.pcisph_begin()
        e3.pcisph_step(1)
        assert _agree(e3.download("positions"), z[f"pci{iters}_positions"], 4 * tol)
        assert _agree(e3.download("velocities"), z[f"pci{iters}_velocities"], 4 * tol)
        st = e3.stats()
        assert st.pci_iters == int(z[f"pci{iters}_iters"])
        e3.close()


@pytest.mark.parametrize("math_mode,tol_x,tol_v", [(0, 0, 0), (1, 1e-5, 1e-3)])
def test_dambreak_against_fixture(math_mode, tol_x, tol_v):
    from dieselfluid_amd import SPHEngine, scenes
    z = np.load(os.path.join(G, "dambreak_n12.npz"))
    p, pos = scenes.dambreak_scene(12, math_mode=math_mode)
    eng = SPHEngine(p)
    eng.upload("positions", pos)
    eng.upload("forces", np.tile(np.array(p.force_reset[:], dtype=np.float32), (12 ** 3, 1)))
    eng.wcsph_step(1)
    assert _agree(eng.download("positions"), z["x1"], tol_x)
    assert _agree(eng.download("densities"), z["rho1"], 10 * tol_x)
    eng.wcsph_step(9)
    assert _agree(eng.download("positions"), z["x10"], tol_x)
    assert _agree(eng.download("velocities"), z["v10"], tol_v, floor=1e-2)
