"""The library's fall-back forms, selected through dsl_set_option (include/dslsph.h, DSL_OPT_*): each
is product code some configuration or failure path reaches, so each is held to the same parity bar as the default --
DSL_MATH_EXACT bit for bit against the oracle, DSL_MATH_FAST to the stated tolerances."""
import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
EXACT, FAST = 0, 1


@pytest.mark.parametrize("variant", ["density_pair=0", "cell_keys=0", "tile_box=0", f"tile_box={2 | 2 << 8 | 2 << 16}",
                                     "persistent_blocks=8", "skin=0.1"])
@pytest.mark.parametrize("math_mode,tol_x", [(EXACT, 0), (FAST, 2e-6)])
def test_wcsph_dambreak_under_a_library_switch(variant, math_mode, tol_x):
    """10 steps of the 16^3 dam-break (tests/test_gpu_parity.py::test_wcsph_dambreak_10_steps) with one option set: the
    lane-per-target FAST density kernel, in-cell ordering in two passes, the tile list in linear order / in small
    boxes, eight workgroups walking all the tiles, the skin step (which DSL_MATH_EXACT ignores)."""
    from dieselfluid_amd import SPHEngine, scenes
    k, v = variant.split("=")
    n3 = 16
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    eng = SPHEngine(p, device=0)
    eng.set_option(k, float(v))
    eng.upload("positions", pos)
    eng.upload("forces", frc)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    eng.wcsph_step(10); ora.wcsph_step(10)
    gx, gv, gr = eng.download("positions"), eng.download("velocities"), eng.download("densities")
    eng.close()
    if math_mode == EXACT:
        assert np.array_equal(gx.view(np.uint32), ora.positions().view(np.uint32))
        assert np.array_equal(gv.view(np.uint32), ora.velocities().view(np.uint32))
        assert np.array_equal(gr.view(np.uint32), ora.densities().view(np.uint32))
    else:
        assert helpers.rel_err(gx, ora.positions()) < tol_x
        assert np.abs(gv.astype(np.float64) - ora.velocities()).max() < helpers.fast_velocity_tolerance(p, 10)
        assert helpers.rel_err(gr, ora.densities()) < 10 * tol_x
