"""The skin step (DSL_OPT_SKIN, kernels_skin.hpp): dsl_wcsph_step with neighbour lists that live for several steps.
The sums are the reference's sums over { |x_i - x_j| < h } (sph_field.go:155-200,251-269) -- a listed pair beyond h
contributes exactly 0 -- so the skin step is held to the FAST tolerances of tests/test_gpu_parity.py against the oracle,
over windows that contain both re-use steps and rebuilds."""
import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
FAST = 1


def _scene(n3):
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=FAST)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    return p, pos, frc


def _engine(p, pos, skin, vel=None):
    from dieselfluid_amd import SPHEngine
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    if vel is not None:
        eng.upload("velocities", vel)
    eng.reset_forces()
    if skin:
        eng.set_option("skin", skin)
    return eng


def _check(eng, ora, p, steps, tol_x=2e-6):
    gx, gv, gr = eng.download("positions"), eng.download("velocities"), eng.download("densities")
    assert helpers.rel_err(gx, ora.positions()) < tol_x
    assert np.abs(gv.astype(np.float64) - ora.velocities()).max() < helpers.fast_velocity_tolerance(p, steps)
    assert helpers.rel_err(gr, ora.densities()) < 10 * tol_x


@pytest.mark.parametrize("skin", [0.05, 0.1, 0.2])
def test_skin_steps_match_the_oracle_over_reuse_and_rebuild(skin):
    """10 steps of the 16^3 dam-break, the tolerances of test_wcsph_dambreak_10_steps; the particles get a seeded
    velocity large enough that the window holds at least one rebuild besides the first, and re-use steps in between."""
    n3 = 16
    p, pos, frc = _scene(n3)
    # |v| dt up to ~0.012 h per step: the bound s h / 2 is reached after 2 (s = 0.05) to 8 (s = 0.2) steps
    cs = float(np.sqrt(p.eos_w / p.mass))
    vel = helpers.seeded_velocities(n3 ** 3, scale=0.03 * cs)
    eng = _engine(p, pos, skin, vel)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel, force=frc)
    eng.wcsph_step(10); ora.wcsph_step(10)
    steps, rebuilds = eng.get_option("skin_steps"), eng.get_option("skin_rebuilds")
    assert steps == 10 and 2 <= rebuilds < 10, (steps, rebuilds)
    assert eng.get_option("skin_list_overflow") == 0
    _check(eng, ora, p, 10)
    eng.close()


def test_skin_equals_plain_steps_to_tolerance_and_is_reproducible():
    """30 steps from rest: skin and plain engines agree to the FAST tolerance, two skin runs agree bit for bit (the
    rebuild decision is device state, a function of the simulation alone), and most steps re-use their lists."""
    n3 = 20
    p, pos, _ = _scene(n3)
    runs = []
    for skin in (0.0, 0.1, 0.1):
        eng = _engine(p, pos, skin)
        eng.wcsph_step(30)
        runs.append((eng.download("positions"), eng.download("velocities"), eng.download("densities"),
                     eng.get_option("skin_steps"), eng.get_option("skin_rebuilds")))
        eng.close()
    plain, a, b = runs
    assert plain[3] == 0 and a[3] == 30 and 1 <= a[4] <= 10, (plain[3:], a[3:])
    for k in range(3):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
    assert helpers.rel_err(a[0], plain[0]) < 2e-6
    assert np.abs(a[1].astype(np.float64) - plain[1]).max() < 2 * helpers.fast_velocity_tolerance(p, 30)
    assert helpers.rel_err(a[2], plain[2]) < 2e-5


def test_lists_built_at_predicted_positions_live_longer():
    """DSL_OPT_SKIN_PREDICT: a rebuild sorts, sweeps and lists at the REFERENCE positions x + tau v and displacement is
    measured against those, so the budget s h / 2 covers the way from -tau v to +tau v.  A block drifting at 0.01 h per
    step (s = 0.1: 5 steps per build when built where the particles are; 30 steps: the library does not look at its
    give-up flag before step 32) must need fewer rebuilds with the prediction
    than without, report its tau, and match the oracle -- which has no lists at all -- either way; two predicted runs
    agree bit for bit (tau is device state, a function of the simulation alone)."""
    n3 = 16
    p, pos, frc = _scene(n3)
    vel = np.zeros_like(pos)
    vel[:, 0] = 0.01 * p.h / p.dt
    steps = 30
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel, force=frc)
    ora.wcsph_step(steps)
    rebuilds, runs = {}, []
    for predict in (0.0, 0.8, 0.8):
        eng = _engine(p, pos, 0.1, vel)
        eng.set_option("skin_predict", predict)
        assert eng.get_option("skin_predict") == pytest.approx(predict)
        eng.wcsph_step(steps)
        tau = eng.get_option("skin_tau_steps")
        assert eng.get_option("skin_steps") == steps and eng.get_option("skin_list_overflow") == 0
        assert eng.get_option("skin_suspensions") == 0
        rebuilds[predict] = eng.get_option("skin_rebuilds")
        assert (tau == 0.0) if predict == 0.0 else (2.0 < tau <= 16.0), tau  # 0.8 x 0.05 h / (0.01 h per step) = 4 steps
        _check(eng, ora, p, steps)
        runs.append((eng.download("positions"), eng.download("velocities")))
        eng.close()
    assert 5 <= rebuilds[0.0] <= 10 and rebuilds[0.8] <= rebuilds[0.0] - 2, rebuilds
    assert np.array_equal(runs[1][0].view(np.uint32), runs[2][0].view(np.uint32))
    assert np.array_equal(runs[1][1].view(np.uint32), runs[2][1].view(np.uint32))
    with pytest.raises(Exception):
        _engine(p, pos, 0.1).set_option("skin_predict", 1.5)


def test_both_list_builders_write_the_same_lists():
    """DSL_OPT_LIST_BUILD: the lock-step builder (every lane one field per trip, from a queue of its non-empty mask words)
    and the first form (one bit loop per mask word) produce the same entries in the same order with the same padding, so
    two runs that differ only in the builder end in the same bits -- over 25 steps with several rebuilds, from rest and
    with seeded velocities (lists of very different lengths inside a wave)."""
    n3 = 24
    p, pos, _ = _scene(n3)
    cs = float(np.sqrt(p.eos_w / p.mass))
    for vel in (None, helpers.seeded_velocities(n3 ** 3, scale=0.02 * cs)):
        out = []
        for lockstep in (1, 0):
            eng = _engine(p, pos, 0.1, vel)
            eng.set_option("list_build", lockstep)
            assert eng.get_option("list_build") == lockstep
            eng.wcsph_step(25)
            assert eng.get_option("skin_steps") == 25 and eng.get_option("skin_rebuilds") >= 2
            assert eng.get_option("skin_list_overflow") == 0
            out.append((eng.download("positions"), eng.download("velocities"), eng.download("densities"),
                        eng.get_option("skin_rebuilds"), eng.get_option("skin_fields_own"), eng.get_option("skin_fields_padded")))
            eng.close()
        a, b = out
        assert a[3:] == b[3:], (a[3:], b[3:])
        for k in range(3):
            assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


def test_a_fast_particle_forces_rebuilds():
    """One particle shot through the block at 0.3 h per step: the displacement bound is outrun every step, every step
    rebuilds, and the results still match the oracle (which has no lists at all)."""
    n3 = 16
    p, pos, frc = _scene(n3)
    vel = np.zeros_like(pos)
    vel[7] = (0.3 * p.h / p.dt, 0.0, 0.0)
    eng = _engine(p, pos, 0.1, vel)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel, force=frc)
    eng.wcsph_step(6); ora.wcsph_step(6)
    assert eng.get_option("skin_rebuilds") == 6
    gx, gr = eng.download("positions"), eng.download("densities")
    assert helpers.rel_err(gx, ora.positions()) < 2e-6
    assert helpers.rel_err(gr, ora.densities()) < 2e-5
    eng.close()


def test_the_library_suspends_a_skin_the_flow_outruns():
    """A particle that crosses s h / 2 every step makes every step a rebuild; after 32 skin steps (the library looks at
    step counts fixed in advance) the skin is suspended and the plain step takes over -- same results, to the FAST
    tolerance, as an engine that never had a skin."""
    n3 = 12
    p, pos, frc = _scene(n3)
    vel = np.zeros_like(pos)
    vel[5] = (0.0, 0.0, 0.2 * p.h / p.dt)
    a = _engine(p, pos, 0.1, vel)
    b = _engine(p, pos, 0.0, vel)
    a.wcsph_step(40); b.wcsph_step(40)
    assert a.get_option("skin_suspensions") == 1 and a.get_option("skin_steps") == 32 and a.get_option("skin_rebuilds") >= 16  # (looked at every 32 steps)
    assert b.get_option("skin_steps") == 0
    assert helpers.rel_err(a.download("positions"), b.download("positions")) < 2e-6
    assert helpers.rel_err(a.download("densities"), b.download("densities")) < 2e-5
    a.close(); b.close()


def test_a_skin_whose_cells_crowd_the_lds_image_is_suspended():
    """s = 0.2 on the jittered lattice: 4 x 4 x 3 cells of 2.4 dx stage 14 x 14 x 12 = 2352 records and more, more than a
    tile's LDS image holds -- such tiles get no lists and fall back to the global-memory sweep (correct, several times
    slower).  The library notices at its first look (step 2) and goes back to the plain step; results to the FAST
    tolerance of an engine that never had a skin."""
    n3 = 40
    p, pos, frc = _scene(n3)
    a = _engine(p, pos, 0.2)
    b = _engine(p, pos, 0.0)
    a.wcsph_step(6); b.wcsph_step(6)
    assert a.get_option("skin_suspensions") == 1 and a.get_option("skin_steps") == 2
    assert helpers.rel_err(a.download("positions"), b.download("positions")) < 2e-6
    assert helpers.rel_err(a.download("densities"), b.download("densities")) < 2e-5
    a.close(); b.close()


def test_skin_default_follows_the_size_and_the_math_mode():
    """DSL_OPT_SKIN defaults to 0.07 for DSL_MATH_FAST handles of 200,000 particles and more (where a step outweighs
    the gated launches: +24 % at 262k, +21 % at 1M), to 0 below that and in DSL_MATH_EXACT."""
    from dieselfluid_amd import SPHEngine, scenes
    for n3, mode, want in ((16, 1, 0.0), (64, 1, 0.07), (128, 1, 0.07), (128, 0, 0.0)):
        p, _ = scenes.dambreak_scene(n3, math_mode=mode, positions=False)
        eng = SPHEngine(p, device=0)
        assert abs(eng.get_option("skin") - want) < 1e-7, (n3, mode)
        eng.close()


def test_other_entry_points_between_skin_steps():
    """Downloads, an upload and the per-pass API in the middle of a skin run settle the device-side state (which
    slot -> particle map is current, the grid's cell size) and the run goes on: the result equals a run that was never
    interrupted to the FAST tolerance, and host order survives every rebuild."""
    n3 = 16
    p, pos, frc = _scene(n3)
    a = _engine(p, pos, 0.1)
    b = _engine(p, pos, 0.1)
    a.wcsph_step(12)
    for _ in range(4):
        b.wcsph_step(3)
        x = b.download("positions")     # settles
        v = b.download("velocities")
        b.density_all()                 # plain cells again: the per-pass API on the same handle
        b.upload("velocities", v)       # round trip through host order
    xa, xb = a.download("positions"), b.download("positions")
    assert helpers.rel_err(xb, xa) < 2e-6
    assert np.abs(b.download("velocities").astype(np.float64) - a.download("velocities")).max() < 2 * helpers.fast_velocity_tolerance(p, 12)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    ora.wcsph_step(12)
    _check(a, ora, p, 12)
    a.close(); b.close()


def test_skin_is_ignored_where_it_does_not_apply():
    """DSL_MATH_EXACT keeps rebuilding every step (its in-cell order is the oracle's): bit for bit with the option set."""
    from dieselfluid_amd import SPHEngine, scenes
    n3 = 12
    p, pos = scenes.dambreak_scene(n3, math_mode=0)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    eng.upload("forces", frc)
    eng.set_option("skin", 0.1)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    eng.wcsph_step(5); ora.wcsph_step(5)
    assert eng.get_option("skin_steps") == 0
    assert np.array_equal(eng.download("positions").view(np.uint32), ora.positions().view(np.uint32))
    with pytest.raises(Exception):
        eng.set_option("skin", 0.5)
    eng.close()
