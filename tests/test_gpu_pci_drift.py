"""PCISPH over more than a handful of steps.  The reference seeds its predictor state once (pcisph_darwin.go:28-41) and
advances it in every correction iteration (:57-73) without ever copying the particles back into it, so the points
DensityF is evaluated at (sph_field.go:137-152) drift away from the particles they belong to -- cells, then tiles, then
the whole box.  The library follows them: once 0.2 % have left their particle's tile it sorts the QUERY points into grid
cells of their own before every DensityF sweep (kernels_sph.hpp: k_pci_predict_bin ... k_pci_density_binned).  Nothing a
host can observe may change with that switch in DSL_MATH_EXACT, and in DSL_MATH_FAST a run must not depend on how its
steps were grouped into calls (the switch is looked at every 4 steps, a function of the step count alone)."""
import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
EXACT, FAST = 0, 1


def _scene(n3, math_mode):
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    p.pci_max_iters = 4
    p.eos_w = p.eos_w / 4   # the gradient is added once per iteration (pcisph_darwin.go:93)
    p.delta = 1.0e-7
    p.pci_max_error = -1.0  # never converged: every iteration runs
    return p, pos


def _tile_leavers(p, x, xp):
    """fraction of predicted positions whose 4x4x4-cell tile is not their particle's (the library's criterion)"""
    g0 = np.array(p.grid_min[:], dtype=np.float32)
    def tile(a):
        c = np.floor((a - g0) * np.float32(1.0 / p.h))
        return np.floor(np.clip(c, 0, None) / 4)
    return float(np.any(tile(x) != tile(xp), axis=1).mean())


def test_exact_run_through_the_switch_is_the_oracles_bit_for_bit():
    """40 EXACT steps of a 16^3 dam-break block: the predictor drifts, the library starts binning the queries on the way
    (asserted), and positions, velocities, predictor state, iteration count and error stay the oracle's bits."""
    from dieselfluid_amd import SPHEngine
    p, pos = _scene(16, EXACT)
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    eng.reset_forces()
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (pos.shape[0], 1))
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    ora.delta = p.delta
    eng.pcisph_begin(); ora.pcisph_begin()
    assert eng.pcisph_binning() == (0, False)
    seen = []
    for chunk in range(10):
        eng.pcisph_step(4); ora.pcisph_step(4)
        st = eng.stats()
        seen.append(eng.pcisph_binning()[1])
        assert st.pci_iters == ora.pci_iters
        assert np.float32(st.pci_max_error) == np.float32(ora.pci_error)
        for name, want in (("positions", ora.positions()), ("velocities", ora.velocities()),
                           ("pci_positions", ora.pci_positions()), ("pci_velocities", ora.pci_velocities())):
            assert np.array_equal(eng.download(name).view(np.uint32), want.view(np.uint32)), (name, chunk)
    frac = _tile_leavers(p, eng.download("positions"), eng.download("pci_positions"))
    print(f"binning active after each 4 steps: {seen}; {frac:.3f} of the predicted positions are in another tile")
    assert frac > 0.05 and seen[-1] and not seen[0]
    eng.close()


@pytest.mark.parametrize("mode", [0, 1, -1])
def test_fast_run_does_not_depend_on_call_grouping(mode):
    """40 FAST steps as 40 calls of one step, 5 calls of 8 and the phase-by-phase form: the same bits, whatever the
    binning mode (0 = the library decides, every 4 steps)."""
    from dieselfluid_amd import SPHEngine
    p, pos = _scene(16, FAST)
    res = []
    for grouping in ("1", "8", "phases"):
        eng = SPHEngine(p, device=0)
        eng.pcisph_set_binning(mode)
        eng.upload("positions", pos)
        eng.reset_forces()
        eng.pcisph_begin()
        if grouping == "phases":
            for _ in range(40):
                eng.pcisph_phase(0)
                for _ in range(4):
                    eng.pcisph_phase(1)
                    eng.pcisph_phase(2)
                eng.pcisph_phase(3)
        else:
            k = int(grouping)
            for _ in range(40 // k):
                eng.pcisph_step(k)
        res.append((eng.download("positions"), eng.download("velocities"), eng.download("pci_positions"), eng.download("pressures"),
                    eng.pcisph_binning()[1]))
        assert np.isfinite(res[-1][0]).all()
        eng.close()
    for r in res[1:]:
        for a, b in zip(res[0][:4], r[:4]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert r[4] == res[0][4]
    assert res[0][4] == (mode >= 0)  # by step 36 more than 0.2 % of the queries have left their tile


@pytest.mark.parametrize("variant", ["", "pci_qincr=0", "pci_qrows=0", "pci_qpair=0", "pci_qtiled=0"])
def test_binned_density_matches_a_float64_brute_force_after_the_drift(variant):
    """FAST, 64^3 particles, 60 steps (median drift ~ 1/3 h, every fifth query in another tile): the pressure accumulator
    after EACH of the next step's four correction iterations is the sum of (rho* - rho0) delta with rho* a float64
    brute-force DensityF at the predicted positions downloaded after that iteration -- for queries INSIDE the fluid, at its
    surface and outside it.  From the second iteration on the default form keeps the query rows and moves only the queries
    that have changed cells (tombstones, late entries: k_pci_predict_bin<.., INCR>).
    The variants are the forms the library falls back to: rows filled afresh in every iteration, the sorted query array
    instead of per-cell rows (also what an allocation failure of the rows selects), one query per lane, the global-memory
    sweep."""
    from dieselfluid_amd import SPHEngine
    from scipy.spatial import cKDTree
    p, pos = _scene(64, FAST)
    eng = SPHEngine(p, device=0)
    if variant:  # (library options, include/dslsph.h: DSL_OPT_PCI_*)
        k, v = variant.split("=")
        eng.set_option(k, float(v))
    eng.upload("positions", pos)
    eng.reset_forces()
    eng.pcisph_begin()
    eng.pcisph_step(60)
    assert eng.pcisph_binning() == (0, True)
    eng.pcisph_phase(0)
    x = eng.download("positions")
    assert np.isfinite(x).all()
    h, m = float(p.h), float(p.mass)
    A = 315.0 / (64.0 * 3.141592653589 * h ** 3)
    tree = cKDTree(x.astype(np.float64))
    probe = np.random.default_rng(5).choice(x.shape[0], 4000, replace=False)
    want = np.zeros(probe.shape[0])
    cells_before = None
    for it in range(4):
        eng.pcisph_phase(1)
        xp, press = eng.download("pci_positions"), eng.download("pressures")
        assert np.isfinite(xp).all() and np.isfinite(press).all()
        cells = np.floor((xp.astype(np.float64) - np.array(p.grid_min[:])) / h).astype(np.int64)
        if cells_before is not None:
            moved = int((cells != cells_before).any(axis=1).sum())
            assert moved > 100  # (queries that change cells between two iterations: what the kept rows have to follow)
        cells_before = cells
        rho = np.empty(probe.shape[0])
        nn = np.empty(probe.shape[0], dtype=np.int64)
        for k, g in enumerate(probe):
            q = xp[g].astype(np.float64)
            idx = tree.query_ball_point(q, h)
            d2 = ((x[idx].astype(np.float64) - q) ** 2).sum(axis=1)
            d2 = d2[d2 < h * h]
            nn[k] = d2.shape[0]
            rho[k] = A + m * A * ((1.0 - d2 / (h * h)) ** 2).sum()
        want += (rho - float(p.ref_density)) * float(p.delta)
        err = np.abs(press[probe] - want).max()
        print(f"iteration {it}: neighbour counts of the probed queries: min {nn.min()} median {int(np.median(nn))} max {nn.max()}; max error {err:.3e}")
        assert nn.min() < 8 and nn.max() > 30  # queries outside / at the surface / inside
        assert err < (it + 1) * (4e-7 * np.abs(want).max() + 3e-5 * float(p.ref_density) * float(p.delta))
        eng.pcisph_phase(2)
    eng.pcisph_phase(3)
    assert eng.stats().pci_iters == 4
    eng.close()


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
def test_particles_outside_the_grid_keep_their_neighbours(math_mode):
    """The binned form finishes a query further than h outside the grid's bounds on the spot (no neighbour there) -- but
    only while every PARTICLE lies inside them.  Here the grid stops a quarter of the way up the fluid block: the
    particles above are clamped into its top cells by the cell rule (in the oracle as well), and the queries up there,
    whole cells outside the grid, must still find them.  Those top cells hold ~44 particles and as many queries: more
    than a cell's row of 32 query slots (the spill list) and more than a tile's LDS image (the global-memory sweep)."""
    from dieselfluid_amd import SPHEngine
    p, pos = _scene(12, math_mode)
    p.grid_max[1] = p.grid_min[1] + 0.25 * (float(pos[:, 1].max()) - p.grid_min[1])
    assert (pos[:, 1] > p.grid_max[1] + 2 * p.h).sum() > 100
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (pos.shape[0], 1))
    res = []
    for binning in (1, -1):
        eng = SPHEngine(p, device=0)
        eng.pcisph_set_binning(binning)
        eng.upload("positions", pos)
        eng.reset_forces()
        eng.pcisph_begin()
        eng.pcisph_step(3)
        res.append((eng.download("positions"), eng.download("velocities"), eng.download("pci_positions"), eng.stats().pci_max_error))
        eng.close()
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    ora.delta = p.delta
    ora.pcisph_begin()
    ora.pcisph_step(3)
    for got in res:
        if math_mode == EXACT:
            assert np.array_equal(got[0].view(np.uint32), ora.positions().view(np.uint32))
            assert np.array_equal(got[1].view(np.uint32), ora.velocities().view(np.uint32))
            assert np.float32(got[3]) == np.float32(ora.pci_error)
        else:
            assert helpers.rel_err(got[0], ora.positions()) < 1e-4
            assert abs(got[3] - ora.pci_error) <= 2e-3 * abs(ora.pci_error)
