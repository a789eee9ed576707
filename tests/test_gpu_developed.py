"""GPU parity where the product actually runs: a DEVELOPED flow (the lattice has melted, cell occupancy
spreads, runs exceed 32 candidates, tiles exceed one pass of targets) against the oracle, and the
full-size PCISPH configurations of BASELINE.json (configs[2]: 4.1M particles x 4 iterations; configs[4]'s
single-GPU workload: 64M particles + XSPH + cohesion) through size-independent properties."""
import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
EXACT, FAST = 0, 1
BEGIN, ITERATE, CHECK, END = 0, 1, 2, 3


def _engine(p):
    from dieselfluid_amd import SPHEngine
    return SPHEngine(p, device=0)


@pytest.fixture(scope="module")
def melted():
    """dam-break block of 20^3 particles advanced on the GPU until the column has collapsed.  Advanced in EXACT mode: that
    trajectory is the oracle's, bit for bit, so the snapshot does not change whenever a FAST kernel rounds differently
    (it used to be a FAST run, and every such change handed the tests below another flow to be calibrated on)"""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(20, math_mode=EXACT)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.reset_forces()
    eng.wcsph_step(2500)
    x, v = eng.download("positions"), eng.download("velocities")
    st = eng.stats()
    eng.nn()
    occ = np.diff(eng.download_cell_start())
    eng.close()
    assert np.isfinite(x).all() and np.isfinite(v).all()
    return x, v, st.max_cell_count, occ


def test_the_snapshot_is_a_developed_flow(melted):
    x, v, max_cell, occ = melted
    full = occ[occ > 0]
    # the lattice start has exactly 8 per cell; here the occupancy spreads and runs of 3 cells pass 32
    assert max_cell >= 14 and full.std() > 1.5
    assert np.abs(v).max() > 0.5  # the front is moving (m/s; gravity scale sqrt(2 g L) = 4.4)
    assert x[:, 0].max() > 1.5    # ... and has left the initial block [0, 1]^3


@pytest.mark.parametrize("math_mode,steps,tol_x,tol_rho", [
    (EXACT, 1, 0, 0), (EXACT, 10, 0, 0),
    # FAST (fma, v_rcp, v_rsq, EOS series): positions 5x / densities 3x what tools/fast_errors.py measures;
    # velocities: the absolute bound of helpers.fast_velocity_tolerance
    (FAST, 1, 3e-7, 2e-5), (FAST, 10, 6e-7, 2e-5)])
def test_steps_from_a_developed_state_match_the_oracle(melted, math_mode, steps, tol_x, tol_rho):
    """1 and 10 WCSPH steps (pressure + viscosity + walls) from the melted snapshot: EXACT bit for bit
    (cells ordered by id = the oracle's DSLO_ORDER_CELL), FAST to the stated tolerances."""
    from dieselfluid_amd import scenes
    x, v, _, _ = melted
    p, _ = scenes.dambreak_scene(20, math_mode=math_mode, positions=False)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (x.shape[0], 1))
    eng = _engine(p)
    eng.upload("positions", x)
    eng.upload("velocities", v)
    eng.upload("forces", frc)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), x, vel=v, force=frc)
    eng.wcsph_step(steps); ora.wcsph_step(steps)
    gx, gv, gr = eng.download("positions"), eng.download("velocities"), eng.download("densities")
    if tol_x == 0:
        assert np.array_equal(gx.view(np.uint32), ora.positions().view(np.uint32))
        assert np.array_equal(gv.view(np.uint32), ora.velocities().view(np.uint32))
        assert np.array_equal(gr.view(np.uint32), ora.densities().view(np.uint32))
    else:
        # Particles about to touch a wall are left out of the FAST comparison: the wall is a clamp + reflect, i.e. an EVENT,
        # and a particle that crosses the plane one step earlier in one arithmetic than in the other loses its normal
        # velocity a step apart (seen with particles 1e-5 from the x = 0 plane: 0.15 m/s and 1e-4 in x after 10 steps,
        # identically for every FAST density kernel).  "About to": off the plane now, but within four times its own travel
        # of it; particles resting ON a wall (most of a collapsed column lies on the floor) stay in.
        ox = ora.positions()
        speed = np.linalg.norm(v, axis=1)
        reach = 4.0 * speed * float(p.dt) * steps + 1.0e-6
        near = np.zeros(x.shape[0], dtype=bool)
        for a in range(3):
            lo, hi = x[:, a] - p.box_min[a], p.box_max[a] - x[:, a]
            near |= ((lo > 0) & (lo < reach)) | ((hi > 0) & (hi < reach))
        keep = ~near
        assert keep.sum() > 0.75 * x.shape[0]
        assert helpers.rel_err(gx[keep], ox[keep]) < tol_x
        # (densities of a developed flow agree to 3e-6 .. 5e-6, velocities to 0.2 .. 0.6 of this bound: tools/pair_diag.py)
        assert np.abs(gv[keep].astype(np.float64) - ora.velocities()[keep]).max() < helpers.fast_velocity_tolerance(p, steps, eps_rho=6.0e-6)
        assert helpers.rel_err(gr[keep], ora.densities()[keep]) < tol_rho
    eng.close()


def _pci_params(p, iters, extra):
    p.pci_max_iters = iters
    p.eos_w = p.eos_w / iters  # the gradient is added once per iteration (pcisph_darwin.go:93)
    p.delta = 1.0e-7
    p.pci_max_error = -1.0     # never converged (the error word is >= 0): every iteration runs
    if extra:
        p.xsph_eps = 0.25
        p.st_kappa = 25.0 * p.h * p.h


def _predicted_density_f64(p, x, xp, probe, near):
    """SPHField.DensityF (sph_field.go:137-152) in float64: W0 + sum_j m F(|x*_i - x_j|) over the CURRENT
    positions of every particle within h of the predicted position, the particle itself included."""
    h, m = float(p.h), float(p.mass)
    A = 315.0 / (64.0 * 3.141592653589 * h ** 3)
    xn = x[near].astype(np.float64)
    out = np.empty(probe.shape[0])
    for k, g in enumerate(probe):
        d2 = ((xn - xp[g].astype(np.float64)) ** 2).sum(axis=1)
        msk = d2 < h * h
        out[k] = A + m * A * ((1.0 - d2[msk] / (h * h)) ** 2).sum()
    return out


@pytest.mark.parametrize("n3,extra,steps", [(160, False, 2), (400, True, 1)])
def test_full_size_pcisph_properties(n3, extra, steps):
    """BASELINE configs[2] (n3 = 160: 4,096,000 particles, 4 correction iterations) and configs[4]'s
    workload on one GPU (n3 = 400: 64,000,000 particles, 4 iterations, XSPH + cohesion), FAST math.
    Size-independent checks: the loop runs exactly max_iters iterations with the tolerance below 0; after the
    first iteration of a step the pressure accumulator of every particle in a probe box is
    (rho* - rho0) delta with rho* a float64 brute-force DensityF at the downloaded predicted positions;
    the state stays finite and inside the wall box; max|v| equals the downloaded maximum."""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=FAST)
    _pci_params(p, 4, extra)
    n = n3 ** 3
    eng = _engine(p)
    eng.upload("positions", pos)
    del pos
    eng.reset_forces()
    eng.pcisph_begin()
    if n3 == 400:
        # the binned iteration's per-cell query rows cost 512 B per GRID CELL -- 33 GB for this scene's
        # box.  Side arrays that scale with the box are only allocated within a budget (dsl_params.reserved[0]; default the
        # larger of 4 GiB and 64 B per particle): here the queries take the sorted-array form and the cells' key rows the
        # two-pass ordering, and the whole handle stays where the particles put it
        eng.pcisph_set_binning(1)
    eng.pcisph_step(steps)
    if n3 == 400:
        assert eng.get_option("pci_qrows") == 0 and eng.get_option("cell_keys") == 0
        assert eng.get_option("device_bytes") < 24e9, eng.get_option("device_bytes")  # (was 19 + 8.6 + 33 GB)
    st = eng.stats()
    assert st.pci_iters == 4 and st.steps == steps
    assert np.isfinite(st.pci_max_error) and st.pci_max_error > 0
    # one more step, in phases: BEGIN, first ITERATE, then look
    eng.pcisph_phase(BEGIN)
    eng.pcisph_phase(ITERATE)
    x = eng.download("positions")
    xp = eng.download("pci_positions")
    press = eng.download("pressures")
    assert x.shape == (n, 3) and np.isfinite(x).all() and np.isfinite(xp).all() and np.isfinite(press).all()
    h = np.float32(p.h)
    lo = np.array([0.955, 0.50, 0.45], np.float32)  # against the free face x = 1 of the block
    inside = np.nonzero(np.all((x >= lo) & (x < lo + 5 * h), axis=1))[0]
    near = np.nonzero(np.all((x >= lo - 2 * h) & (x < lo + 7 * h), axis=1))[0]
    assert 300 < inside.shape[0] < 5000
    # the predictor state is never re-synchronised (pcisph_darwin.go:28-41) but after two steps it is
    # still within a fraction of h of the particle
    assert np.abs(xp[inside] - x[inside]).max() < 0.5 * h
    rho_star = _predicted_density_f64(p, x, xp, inside, near)
    want = (rho_star - float(p.ref_density)) * float(p.delta)
    # rho* is dominated by the W0 it starts from (sph_field.go:139: no mass factor; W0 ~ 600 rho0 here), so
    # the accumulator is known to a float32 rounding of its own size plus the density sum's FAST tolerance
    assert np.abs(press[inside] - want).max() < 4e-7 * np.abs(want).max() + 3e-5 * float(p.ref_density) * float(p.delta)
    eng.pcisph_phase(CHECK)
    for _ in range(3):
        eng.pcisph_phase(ITERATE)
        eng.pcisph_phase(CHECK)
    eng.pcisph_phase(END)
    st = eng.stats()
    assert st.pci_iters == 4 and st.steps == steps + 1
    x1, v1 = eng.download("positions"), eng.download("velocities")
    assert np.isfinite(x1).all() and np.isfinite(v1).all()
    for a in range(3):
        assert x1[:, a].min() >= p.box_min[a] and x1[:, a].max() <= p.box_max[a]
    vmax = np.sqrt((v1.astype(np.float64) ** 2).sum(axis=1)).max()
    assert st.max_vel >= vmax * (1 - 1e-6)  # maxVel is a running maximum (fluid.go:186-191)
    assert np.array_equal(eng.download("pressures"), np.zeros(n, dtype=np.float32))  # Update: Press = 0 (fluid.go:192)
    eng.close()


@pytest.mark.parametrize("method", ["wcsph", "pcisph"])
def test_results_do_not_depend_on_how_tiles_are_dealt_to_workgroups(method):
    """The persistent tile kernels hand a workgroup one tile after another (double-buffered LDS image, rotating tile
    tables: kernels_tiled.hpp).  A scene of a few hundred tiles gives every workgroup ONE tile, so that hand-over is
    never exercised by the other tests -- round 3 shipped a barrier without its LDS drain there and only a 10000-step
    16M soak noticed.  Here the grid is capped at 8 workgroups (DSL_OPT_PERSISTENT_BLOCKS), 60+ tiles each, and the run
    must give the same BITS as the uncapped engine: FAST arithmetic does not depend on which workgroup sweeps a tile."""
    from dieselfluid_amd import SPHEngine, scenes
    n3 = 64
    p, pos = scenes.dambreak_scene(n3, math_mode=1)
    if method == "pcisph":
        p.pci_max_iters = 4
        p.eos_w = p.eos_w / 4
        p.delta = 1.0e-7
    res = []
    # (the third engine walks the tile list the static way -- every workgroup a share dealt in advance -- where the first
    # two draw their tiles from the per-XCD counters of DSL_OPT_TILE_QUEUE: 2 = at any size)
    for cap, queue in ((0, 2), (8, 2), (0, 0)):
        eng = SPHEngine(p, device=0)
        eng.set_option("persistent_blocks", cap)
        eng.set_option("tile_queue", queue)
        eng.upload("positions", pos)
        eng.reset_forces()
        if method == "pcisph":
            eng.pcisph_begin()
            for _ in range(6):
                eng.pcisph_step(10)
        else:
            for _ in range(10):
                eng.wcsph_step(60)
        res.append((eng.download("positions"), eng.download("velocities"), eng.download("densities")))
        assert np.isfinite(res[-1][0]).all()
        eng.close()
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
