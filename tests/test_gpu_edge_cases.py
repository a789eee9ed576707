"""GPU edge cases: tiny and ragged inputs, coincident particles (collisions), particles
outside the grid box, crowded cells that invalidate the 32-bit neighbour masks or overflow the
LDS tile, and a full-size (16M) run checked through size-independent properties."""
import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
EXACT, FAST = 0, 1


def _engine(p):
    from dieselfluid_amd import SPHEngine
    return SPHEngine(p, device=0)


def _dambreak_params(n, math_mode, n3_like=12):
    """dam-break parameter set (h = 2dx of an n3_like block) for an arbitrary particle count"""
    from dieselfluid_amd import scenes
    p, _ = scenes.dambreak_scene(n3_like, math_mode=math_mode, positions=False)
    p.n_particles = n
    return p


def _compare_steps(p, pos, vel, steps, tol_x, tol_v, mode=po.NEIGH_GRID, before=None):
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (pos.shape[0], 1))
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.upload("forces", frc)
    if before is not None:
        before(eng)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p, mode=mode), pos, vel=vel, force=frc)
    eng.wcsph_step(steps); ora.wcsph_step(steps)
    gx, gv, ox, ov = eng.download("positions"), eng.download("velocities"), ora.positions(), ora.velocities()
    assert np.array_equal(np.isnan(gx), np.isnan(ox))
    assert helpers.rel_err(np.nan_to_num(gx), np.nan_to_num(ox)) < tol_x
    assert helpers.rel_err(np.nan_to_num(gv), np.nan_to_num(ov), floor=1e-2) < tol_v
    return eng, ora


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
@pytest.mark.parametrize("n", [1, 2, 7, 65])
def test_tiny_and_ragged_particle_counts(n, math_mode):
    """n = 1 (no neighbour at all: rho = 0), n not a multiple of the wave or block size."""
    p = _dambreak_params(n, math_mode)
    rng = np.random.default_rng(n)
    pos = (0.3 + 0.2 * rng.random((n, 3))).astype(np.float32)  # one clump, a few h wide
    vel = helpers.seeded_velocities(n, 0.1, seed=n)
    # an isolated particle has rho = 0 -> P/rho^2 = 0/0: the reference produces NaN there too;
    # the pressure term is switched off for the single-particle case to compare finite numbers
    if n == 1:
        p.wcsph_pressure_force = 0
    p.dt = p.dt * 0.02  # the random clump is far denser than rest density: keep the motion small
    _compare_steps(p, pos, vel, 3, 1e-5, 1e-3, mode=po.NEIGH_ALL)


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
def test_coincident_particles(math_mode):
    """collisions: pairs of particles at exactly the same position (r = 0): Norm() of the zero
    vector is zero (vector.go:322-331), so the pair exerts no pressure force but does count in
    the density and viscosity sums."""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(8, math_mode=math_mode)
    pos = pos.copy()
    pos[1] = pos[0]
    pos[100] = pos[37]
    pos[101] = pos[37]
    vel = helpers.seeded_velocities(pos.shape[0], 0.2)
    eng, ora = _compare_steps(p, pos, vel, 2, 1e-5, 2e-3)
    assert np.isfinite(eng.download("positions")).all()


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
def test_particles_outside_the_grid_box(math_mode):
    """cell coordinates are clamped; clamping is monotone, so the 27-cell sweep stays complete
    for particles that have left the box.  The brute-force oracle rule is the judge."""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(8, math_mode=math_mode)
    p.walls = 0
    pos = pos.copy()
    pos[:64] -= np.float32(0.9)           # a clump far below/left of the grid box
    pos[64:128, 0] += np.float32(9.0)     # and one beyond the far x end
    vel = helpers.seeded_velocities(pos.shape[0], 0.1)
    _compare_steps(p, pos, vel, 2, 1e-5, 2e-3, mode=po.NEIGH_ALL)


def test_crowded_run_invalidates_masks_but_tile_still_fits():
    """FAST: a cluster of 60 extra particles in one cell makes the x-runs through it longer
    than the 32 mask bits; those particles must fall back to the full sweep inside the tiled
    kernel while everything else keeps walking masks."""
    from dieselfluid_amd import scenes
    n3 = 10
    p, pos = scenes.dambreak_scene(n3, math_mode=FAST)
    rng = np.random.default_rng(3)
    h = p.h
    centre = np.array([4.5 * h, 3.5 * h, 2.5 * h], dtype=np.float32)  # middle of one grid cell
    extra = (centre + (rng.random((60, 3)).astype(np.float32) - 0.5) * np.float32(0.8 * h)).astype(np.float32)
    pos = np.concatenate([pos, extra]).astype(np.float32)
    p.n_particles = pos.shape[0]
    vel = helpers.seeded_velocities(pos.shape[0], 0.05)
    p.dt = p.dt * 0.05  # the cluster is violently over-pressured: keep the step small
    def crowded(eng):
        eng.nn()
        assert eng.stats().max_cell_count >= 60  # x-runs through this cell exceed the 32 mask bits

    _compare_steps(p, pos, vel, 2, 1e-5, 5e-3, before=crowded)


def test_tile_overflow_falls_back():
    """FAST with h = 5 dx: ~125 particles per cell, every 4x4x4 tile exceeds the LDS budget and
    the kernels must take the global-memory sweep."""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(12, math_mode=FAST, h_over_dx=5.0)
    vel = helpers.seeded_velocities(pos.shape[0], 0.05)
    def crowded(eng):
        eng.nn()
        assert eng.stats().max_cell_count > 36  # 6x6x6 cells x that many records cannot fit the LDS tile

    _compare_steps(p, pos, vel, 2, 2e-5, 5e-3, before=crowded)


def _probe_sets(x, lo, hi, h):
    inside = np.nonzero(np.all((x >= lo) & (x < hi), axis=1))[0]
    near = np.nonzero(np.all((x >= lo - 2 * h) & (x < hi + 2 * h), axis=1))[0]
    return inside, near


@pytest.mark.parametrize("n3", [100, 252])
def test_full_size_16m_properties(n3):
    """BASELINE full sizes -- configs[1]'s 1,000,000 particles (n3 = 100) and the bench's 16,003,008 (n3 = 252) --, FAST math: checked through
    size-independent properties -- the slot map stays a permutation, cell_start is a valid prefix
    table of the sorted cells, and for every particle inside a probe box the density AND the result of
    the fused force+integrate kernel (new position and velocity) equal a brute-force float64
    evaluation of the reference formulas on the downloaded state (helpers.brute_force_step_f64)."""
    from dieselfluid_amd import SPHEngine, scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=FAST)
    n = n3 ** 3
    eng = SPHEngine(p)
    eng.upload("positions", pos)
    del pos
    eng.reset_forces()
    eng.wcsph_step(3)
    eng.nn()
    eng.density_all()
    ids = eng.download_ids()
    assert ids.shape == (n,)
    seen = np.zeros(n, dtype=np.uint8)
    seen[ids] = 1
    assert seen.all(), "slot -> particle map must be a permutation"
    cs = eng.download_cell_start()
    assert cs[0] == 0 and cs[-1] == n and np.all(np.diff(cs) >= 0)
    cell_of_slot = np.repeat(np.arange(cs.size - 1), np.diff(cs))
    assert np.all(np.diff(ids)[cell_of_slot[1:] == cell_of_slot[:-1]] > 0), "cells must be ordered by particle id"
    del cell_of_slot, seen
    st = eng.stats()
    assert np.diff(cs).max() == st.max_cell_count
    x = eng.download("positions")
    v = eng.download("velocities")
    rho = eng.download("densities")
    assert np.isfinite(x).all() and np.isfinite(rho).all()
    h = np.float32(p.h)
    # probe box at the free surface corner of the block (the part of the fluid that is moving)
    lo = np.array([0.90, 0.90, 0.45], np.float32)
    inside, near = _probe_sets(x, lo, lo + 5 * h, h)
    assert 300 < inside.shape[0] < 5000
    want_rho, want_x, want_v = helpers.brute_force_step_f64(p, x, v, inside, near)
    assert helpers.rel_err(rho[inside], want_rho) < 2e-5
    eng.force_pass()  # the fused pressure + viscosity + integrate + walls kernel on exactly this state
    x1, v1 = eng.download("positions"), eng.download("velocities")
    assert np.isfinite(x1).all() and np.isfinite(v1).all()
    # (a) against the float64 brute force: displacement and velocity change of this one step.  The
    # pressure sum is ill-conditioned in float32 (pair terms ~100x the net force, EOS exponent 7.16 on a
    # density known to 1e-6), so float32 and float64 differ by a few 1e-4 of the largest change in the box.
    dx_want, dx_got = want_x - x[inside].astype(np.float64), x1[inside].astype(np.float64) - x[inside].astype(np.float64)
    assert np.abs(dx_got - dx_want).max() < 1e-3 * np.abs(dx_want).max() + 2e-7 * np.abs(want_x).max()
    dv_want, dv_got = want_v - v[inside].astype(np.float64), v1[inside].astype(np.float64) - v[inside].astype(np.float64)
    assert np.abs(dv_got - dv_want).max() < 1e-3 * np.abs(dv_want).max()
    # (b) against the float32 oracle: the particles within 2h of the box, taken out of the 16M state
    # in ascending id, ARE the complete neighbourhood of every probe particle and of each of its
    # neighbours, so one oracle step of that subset gives the probe particles what a 16M-particle oracle
    # run would -- at the FAST tolerances of the small-scale parity tests.
    order = near  # (np.nonzero is ascending and host order is id order)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (order.shape[0], 1))
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), x[order], vel=v[order], force=frc)
    ora.wcsph_step(1)
    sel = np.searchsorted(order, inside)
    ox, ov = ora.positions()[sel], ora.velocities()[sel]
    assert np.abs(x1[inside].astype(np.float64) - ox).max() < 1e-6 * np.abs(ox).max()
    dv_ora = ov.astype(np.float64) - v[inside].astype(np.float64)
    assert np.abs(dv_got - dv_ora).max() < helpers.fast_velocity_tolerance(p, 1)  # (measured: 0.13 of the model bound)
    # Update resets every force and pressure after a step (fluid.go:192-193)
    assert st.steps == 3 and eng.stats().steps == 3
    # (c) the SKIN step (DSL_OPT_SKIN: neighbour lists against h (1 + s), kernels_skin.hpp) at full size.  One step from
    # the state just checked -- list build + walk -- against the float64 brute force of that state, at (a)'s tolerances;
    eng.set_option("skin", 0.1)
    xa, va = x1, v1
    inside_k, near_k = _probe_sets(xa, lo, lo + 5 * h, h)
    _, wx, wv = helpers.brute_force_step_f64(p, xa, va, inside_k, near_k)
    s0, r0 = eng.get_option("skin_steps"), eng.get_option("skin_rebuilds")
    eng.wcsph_step(1)
    assert (eng.get_option("skin_steps"), eng.get_option("skin_rebuilds")) == (s0 + 1, r0 + 1)
    xb, vb = eng.download("positions"), eng.download("velocities")
    dxw, dxg = wx - xa[inside_k].astype(np.float64), xb[inside_k].astype(np.float64) - xa[inside_k].astype(np.float64)
    assert np.abs(dxg - dxw).max() < 1e-3 * np.abs(dxw).max() + 2e-7 * np.abs(wx).max()
    dvw, dvg = wv - va[inside_k].astype(np.float64), vb[inside_k].astype(np.float64) - va[inside_k].astype(np.float64)
    assert np.abs(dvg - dvw).max() < 1e-3 * np.abs(dvw).max()
    # then four more -- one build, three steps that walk the same lists -- against a twin that sorts and sweeps every
    # step, at the FAST tolerances of tests/test_gpu_parity.py
    twin = SPHEngine(p)
    twin.set_option("skin", 0.0)
    twin.upload("positions", xb)
    twin.upload("velocities", vb)
    twin.reset_forces()
    twin.wcsph_step(4)
    eng.wcsph_step(4)
    assert (eng.get_option("skin_steps"), eng.get_option("skin_rebuilds")) == (s0 + 5, r0 + 2)
    assert twin.get_option("skin_steps") == 0
    xs, vs, xt, vt = eng.download("positions"), eng.download("velocities"), twin.download("positions"), twin.download("velocities")
    twin.close()
    # (d) DSL_OPT_GRID_OVERSUB (from 8M particles on the plain step's tile kernels are launched with 8 x the workgroups the
    # chip holds): which workgroup sweeps a tile must not matter -- a second twin with persistent workgroups, same bits
    if n >= 8000000:
        twin1 = SPHEngine(p)
        twin1.set_option("skin", 0.0)
        twin1.set_option("grid_oversub", 1)
        twin1.upload("positions", xb)
        twin1.upload("velocities", vb)
        twin1.reset_forces()
        twin1.wcsph_step(4)
        x1t, v1t = twin1.download("positions"), twin1.download("velocities")
        twin1.close()
        assert np.array_equal(x1t.view(np.uint32), xt.view(np.uint32)) and np.array_equal(v1t.view(np.uint32), vt.view(np.uint32))
    assert helpers.rel_err(xs, xt) < 2e-6
    assert np.abs(vs.astype(np.float64) - vt).max() < 2 * helpers.fast_velocity_tolerance(p, 4)
    assert eng.stats().steps == 8


def test_shared_short_passes_match_the_oracle():
    """Tiles whose last pass is short (here: every tile of a 10^3 block is part empty, and a dense
    clump pushes one tile beyond 512 targets) are swept by the instantiation that gives 2-16 lanes
    to a target.  The engine switches to it once the tile statistics of an earlier build have reached
    the host, i.e. after a synchronising call; compare every chunk of steps with the oracle."""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(10, math_mode=FAST)
    rng = np.random.default_rng(5)
    h = p.h
    centre = np.array([2.5 * h, 2.5 * h, 2.5 * h], dtype=np.float32)
    extra = (centre + (rng.random((600, 3)).astype(np.float32) - 0.5) * 3.9 * h).astype(np.float32)  # ~1 tile
    pos = np.concatenate([pos, extra]).astype(np.float32)
    p.n_particles = pos.shape[0]
    p.dt = p.dt * 0.02
    vel = helpers.seeded_velocities(pos.shape[0], 0.1, seed=3)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (pos.shape[0], 1))
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.upload("forces", frc)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel, force=frc)
    for chunk in (1, 2, 2):
        eng.wcsph_step(chunk)
        ora.wcsph_step(chunk)
        gx, gv = eng.download("positions"), eng.download("velocities")  # synchronises: statistics arrive
        assert helpers.rel_err(gx, ora.positions()) < 1e-5
        assert helpers.rel_err(gv, ora.velocities(), floor=1e-2) < 5e-3
    rho = eng.download("densities")
    assert np.all(np.isfinite(rho)) and rho.max() > 1.5 * p.ref_density  # the clump really is crowded


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
def test_set_params_refreshes_derived_state(math_mode):
    """dsl_set_params with a new EOS constant re-derives P/rho^2 from the cached densities, and a new mass
    marks the densities stale: the gradient pass after it equals that of an engine created with the new values."""
    from dieselfluid_amd import scenes
    n3 = 10
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    vel = helpers.seeded_velocities(n3 ** 3, 0.05, seed=5)

    def gradient_forces(eng):
        eng.gradient_pressure_force()
        return eng.download("forces")

    for field, factor in (("eos_gamma", 0.5), ("mass", 1.25)):
        q, _ = scenes.dambreak_scene(n3, math_mode=math_mode, positions=False)
        setattr(q, field, getattr(q, field) * factor)
        late, fresh = _engine(p), _engine(q)
        for eng in (late, fresh):
            eng.upload("positions", pos)
            eng.upload("velocities", vel)
            eng.nn()
            eng.density_all()
        late.set_params(q)
        if field == "mass":  # densities scale with the mass: they have to be taken again
            late.density_all()
        fa, fb = gradient_forces(late), gradient_forces(fresh)
        assert np.array_equal(fa, fb), field
        late.close(); fresh.close()


def test_a_tile_of_more_than_65535_particles():
    """a tile's table packs its in-row cell boundaries into 16 bits (TileMeta::trow).  A tile that holds more
    particles than that -- here: 42^3 = 74,088 particles whose grid box is ONE tile, everything beyond it clamped into
    its outermost cells -- is beyond the LDS budget anyway and takes the global-memory sweep; its targets are then
    walked by the 32-bit target prefix alone.  Densities against a float64 KD-tree evaluation of sph_field.go:155-172."""
    from scipy.spatial import cKDTree
    from dieselfluid_amd import SPHEngine, scenes
    n3 = 42
    p, pos = scenes.dambreak_scene(n3, math_mode=FAST)
    for a in range(3):  # a 4 x 4 x 4-cell grid in the block's corner: 1 tile
        p.grid_min[a] = 0.0
        p.grid_max[a] = 4.0 * p.h
    eng = SPHEngine(p)
    eng.upload("positions", pos)
    eng.density_all()
    assert eng.stats().grid_cells == 64
    rho = eng.download("densities")
    eng.close()
    h, m = float(p.h), float(p.mass)
    A = 315.0 / (64.0 * 3.141592653589 * h ** 3)
    x = pos.astype(np.float64)
    tree = cKDTree(x)
    pairs = tree.query_pairs(h, output_type="ndarray")
    d2 = ((x[pairs[:, 0]] - x[pairs[:, 1]]) ** 2).sum(axis=1)
    w = m * A * (1.0 - d2 / (h * h)) ** 2
    want = np.bincount(pairs[:, 0], w, n3 ** 3) + np.bincount(pairs[:, 1], w, n3 ** 3)
    assert helpers.rel_err(rho, want) < 2e-5


@pytest.mark.parametrize("math_mode", [0, 1])
def test_a_box_eight_times_taller_and_no_budget_for_side_arrays(math_mode):
    """side arrays that cost memory per GRID CELL (the cells' key rows, the PCISPH query rows) must not
    decide how tall a domain may be.  The 16^3 dam-break in a box -- and a grid -- 8 x taller, with a 1 MiB budget for each
    such array (dsl_params.reserved[0]): neither is allocated, the sort orders its cells in two passes, the binned PCISPH
    iteration keeps its queries in the sorted array -- and the results are the oracle's (EXACT: bit for bit)."""
    from dieselfluid_amd import SPHEngine, scenes
    n3 = 16
    res = {}
    for budget in (0, 1):
        p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
        p.box_max[1] *= 8.0
        p.grid_max[1] = p.box_max[1] + p.h
        p.reserved[0] = budget
        p.pci_max_iters = 3
        p.eos_w = p.eos_w / 3
        p.delta = 1.0e-7
        frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
        eng = SPHEngine(p)
        assert eng.get_option("cell_keys") == (0 if budget else 1)
        eng.upload("positions", pos)
        eng.upload("forces", frc)
        eng.wcsph_step(6)
        eng.pcisph_begin()
        eng.pcisph_set_binning(1)
        eng.pcisph_step(3)
        if math_mode == 1:  # (the rows belong to the FAST sweep; EXACT keeps the sorted array whatever the budget)
            assert eng.get_option("pci_qrows") == (0 if budget else 1)
        res[budget] = (eng.download("positions"), eng.download("velocities"), eng.get_option("device_bytes"))
        if budget:
            ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
            ora.wcsph_step(6)
            ora.pcisph_begin()
            ora.pcisph_step(3)
            if math_mode == 0:
                assert np.array_equal(res[1][0].view(np.uint32), ora.positions().view(np.uint32))
                assert np.array_equal(res[1][1].view(np.uint32), ora.velocities().view(np.uint32))
            else:
                assert helpers.rel_err(res[1][0], ora.positions()) < 2e-6
        eng.close()
    assert res[1][2] < res[0][2]  # the side arrays are really not there
    if math_mode == 0:
        assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))


def test_steps_captured_into_a_host_graph_replay_to_the_same_bits():
    """A host may capture the engine's launches into a graph of its own (dsl_set_stream + stream capture) and replay it.
    Two plain steps are the unit that can be replayed (the position / velocity sets and the slot maps ping-pong per
    step on the host side).  The tile queue (DSL_OPT_TILE_QUEUE) alternates between two counter blocks per LAUNCH CALL,
    which a replay does not repeat -- so a capturing stream gets the static walk -- and two replays of the captured pair
    must equal four steps taken directly, bit for bit (FAST results do not depend on how tiles are dealt)."""
    import torch
    from dieselfluid_amd import SPHEngine, scenes
    n3 = 32
    p, pos = scenes.dambreak_scene(n3, math_mode=FAST)
    out = []
    for captured in (False, True):
        eng = SPHEngine(p, device=0)
        eng.set_option("skin", 0.0)
        eng.set_option("tile_queue", 2)
        eng.upload("positions", pos)
        eng.reset_forces()
        eng.wcsph_step(2)  # (allocations and first-use paths out of the way)
        if captured:
            s = torch.cuda.Stream()
            eng.sync()
            eng.set_stream(s.cuda_stream)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                eng.wcsph_step(2)
            g.replay()
            g.replay()
            torch.cuda.synchronize()
            eng.use_own_stream()
        else:
            eng.wcsph_step(4)
        out.append((eng.download("positions"), eng.download("velocities")))
        eng.close()
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
