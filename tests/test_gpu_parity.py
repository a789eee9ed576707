"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle on identical
seeded inputs.  EXACT math mode is held to BIT-FOR-BIT agreement wherever the arithmetic is
float32 + sqrt + divide (the counting sort orders every cell by particle id, so the device sums
in the oracle's DSLO_ORDER_CELL order); where a double pow is involved (Tait EOS) the two libms may
differ in the last place and a tolerance is written next to the check.  FAST mode is held to the
float32 tolerances written next to each check."""
import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

EXACT, FAST = 0, 1


def _engine(p):
    from dieselfluid_amd import SPHEngine
    return SPHEngine(p, device=0)


def _reference_system(n3, math_mode, amp=0.2, vel_scale=0.1):
    """Reference constants (h=1, m=1, rho0=N/8, mu=1.3059, dt=0.01) on the jittered
    [-1,1)^3 lattice of SURVEY 8c fixture (2), grid neighbours."""
    from dieselfluid_amd import scenes
    p, _ = scenes.reference_scene(n3)
    p.math_mode = math_mode
    pos = helpers.jittered_lattice(n3, amp)
    vel = helpers.seeded_velocities(n3 ** 3, vel_scale)
    return p, pos, vel


def _agree(a, b, tol, floor=0.0):
    """tol == 0: bit for bit; else max |a - b| < tol * max |b|"""
    if tol == 0:
        return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    return helpers.rel_err(a, b, floor=floor) < tol


@pytest.mark.parametrize("math_mode,tol", [(EXACT, 0), (FAST, 2e-5)])
def test_density_pass(math_mode, tol):
    """D: SPHField.Density (sph_field.go:155-172).  EXACT: bit for bit against the oracle's
    DSLO_ORDER_CELL sums (cells in z, y, x order, ascending particle index inside a cell)."""
    p, pos, vel = _reference_system(12, math_mode)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.density_all()
    rho = eng.download("densities")
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos)
    ora.density_all()
    assert _agree(rho, ora.densities(), tol)


@pytest.mark.parametrize("h_over_dx", [2.0, 5.0])
def test_cells_are_ordered_by_particle_id(h_over_dx):
    """The counting sort leaves every cell ascending in particle id (dsl_params.sort_unordered = 0).  h = 2 dx: ~8
    particles per cell, ordered in the scatter pass itself from the cells' key rows; h = 5 dx: ~125 per cell, more
    than a key row holds (kCellKeys = 32): the flag-gated two-pass fallback."""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(16, math_mode=FAST, h_over_dx=h_over_dx)
    if h_over_dx > 2.0:
        p.dt = p.dt * 0.2
    rng = np.random.default_rng(5)
    perm = rng.permutation(pos.shape[0])  # ids unrelated to the position in the lattice
    eng = _engine(p)
    eng.upload("positions", pos[perm])
    eng.reset_forces()
    eng.wcsph_step(3)
    eng.nn()
    ids, cs = eng.download_ids(), eng.download_cell_start()
    assert np.array_equal(np.sort(ids), np.arange(pos.shape[0]))  # a permutation: nobody lost, nobody twice
    assert (np.diff(cs).max() > 32) == (h_over_dx > 2.0)
    cell_of_slot = np.repeat(np.arange(cs.size - 1), np.diff(cs))
    same_cell = cell_of_slot[1:] == cell_of_slot[:-1]
    assert np.all(np.diff(ids)[same_cell] > 0)
    eng.close()


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
def test_two_runs_agree_bit_for_bit(math_mode):
    """Reproducibility: the same scene stepped by two engines gives identical bits (the in-cell
    order no longer depends on which wave's atomic landed first)."""
    from dieselfluid_amd import scenes
    p, pos = scenes.dambreak_scene(16, math_mode=math_mode)
    res = []
    for _ in range(2):
        eng = _engine(p)
        eng.upload("positions", pos)
        eng.reset_forces()
        eng.wcsph_step(25)
        res.append((eng.download("positions"), eng.download("velocities"), eng.download("densities")))
        eng.close()
    for a, b in zip(*res):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_density_bit_exact_in_device_order():
    """With the oracle fed the particles in the device's sorted slot order (so both sum
    each cell in the same order) EXACT mode must agree bit for bit."""
    p, pos, vel = _reference_system(12, EXACT)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.density_all()
    spos = eng.download("positions", sorted_order=True)
    srho = eng.download("densities", sorted_order=True)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), spos)
    ora.density_all()
    assert np.array_equal(srho.view(np.uint32), ora.densities().view(np.uint32))


def test_sort_is_a_permutation_and_sorted():
    """N: counting sort -- ids are a permutation, cell ids ascend, cell_start matches."""
    p, pos, vel = _reference_system(12, EXACT)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.nn()
    ids = eng.download_ids()
    assert np.array_equal(np.sort(ids), np.arange(eng.n))
    spos = eng.download("positions", sorted_order=True)
    assert np.array_equal(spos, pos[ids])
    st = eng.stats()
    dims = np.array(st.grid_dims[:])
    inv = np.float32(1.0) / np.float32(p.h)
    gmin = np.array(p.grid_min[:], dtype=np.float32)
    c = np.floor((spos - gmin) * inv).astype(np.int64)
    c = np.clip(c, 0, dims - 1)
    cell = (c[:, 2] * dims[1] + c[:, 1]) * dims[0] + c[:, 0]
    assert np.all(np.diff(cell) >= 0)
    cs = eng.download_cell_start()
    assert cs[0] == 0 and cs[-1] == eng.n
    counts = np.bincount(cell, minlength=st.grid_cells)
    assert np.array_equal(np.diff(cs), counts)
    assert st.max_cell_count == counts.max()


@pytest.mark.parametrize("math_mode,tol", [(EXACT, 0), (FAST, 5e-5)])
def test_reference_pass_sequence(math_mode, tol):
    """The passes of sph.Init and one PCISPH-style force build-up, one C-ABI call per
    reference method: DensityAll, ExternalAll, ViscousAll, GradientPressureForce,
    PressureAll, Update (fluid.go:127-197)."""
    p, pos, vel = _reference_system(12, math_mode)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel)
    g = np.array([0, -9.81, 0], dtype=np.float32)
    eng.density_all(); ora.density_all()
    eng.external_all(g); ora.external_all(g)
    eng.viscous_all(); ora.viscous_all()
    f1 = eng.download("forces")
    assert _agree(f1, ora.forces(), tol)
    eng.gradient_pressure_force(); ora.gradient_pressure_force()
    f2 = eng.download("forces")
    assert _agree(f2, ora.forces(), tol)
    eng.pressure_all(); ora.pressure_all()
    # (rho/rho0)^gamma - 1 amplifies the summation-order noise of rho by ~gamma/(ratio^gamma - 1)
    assert _agree(eng.download("pressures"), ora.pressures(), 8 * tol)
    eng.update(); ora.update()
    assert _agree(eng.download("positions"), ora.positions(), tol)
    assert _agree(eng.download("velocities"), ora.velocities(), tol)
    # Update resets force and pressure (fluid.go:192-193)
    assert np.array_equal(eng.download("forces"), ora.forces())
    assert np.array_equal(eng.download("pressures"), ora.pressures())
    st = eng.stats()
    assert abs(st.max_vel - ora.max_vel) <= tol * ora.max_vel
    assert abs(st.max_f - ora.max_f) <= tol * ora.max_f


def test_wcsph_free_fall_known_answer():
    """W: the reference WCSPH loop has no pressure force and double gravity: after steps
    1,2,3 v_y = -0.1962,-0.3924,-0.5886 and y(-1) = -1.001962,-1.005886,-1.0117719
    (SURVEY 8c).  Bit-exact in EXACT mode."""
    from dieselfluid_amd import scenes
    p, pos = scenes.reference_scene(8)
    p.math_mode = EXACT
    eng = _engine(p)
    eng.upload("positions", pos)
    frc = np.tile(np.array([0, -9.81, 0], dtype=np.float32), (eng.n, 1))
    eng.upload("forces", frc)  # state after sph.Init: gravity (viscous term is 0 at rest)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    want_v = [np.float32(-0.1962), np.float32(-0.3924), np.float32(-0.5886)]
    want_y = [np.float32(-1.001962), np.float32(-1.005886), np.float32(-1.0117719)]
    for k in range(3):
        eng.wcsph_step(1); ora.wcsph_step(1)
        v = eng.download("velocities"); x = eng.download("positions")
        assert np.array_equal(v, ora.velocities())
        assert np.array_equal(x, ora.positions())
        assert v[0, 1] == want_v[k] and x[0, 1] == want_y[k]


@pytest.mark.parametrize("math_mode,tol_x", [(EXACT, 0), (FAST, 2e-6)])
def test_wcsph_dambreak_10_steps(math_mode, tol_x):
    """Build-defined dam-break (pressure + viscosity + walls) through the fused
    force+integrate kernel, 10 steps, against the oracle's pass-by-pass loop."""
    from dieselfluid_amd import scenes
    n3 = 16
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("forces", frc)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    eng.wcsph_step(10); ora.wcsph_step(10)
    assert _agree(eng.download("positions"), ora.positions(), tol_x)
    if math_mode == EXACT:
        assert _agree(eng.download("velocities"), ora.velocities(), 0)
    else:  # absolute, from the error model of helpers.fast_velocity_tolerance (measured: 4e-5 m/s)
        assert np.abs(eng.download("velocities").astype(np.float64) - ora.velocities()).max() < helpers.fast_velocity_tolerance(p, 10)
    assert _agree(eng.download("densities"), ora.densities(), 10 * tol_x)


@pytest.mark.parametrize("binning", [0, 1])
@pytest.mark.parametrize("math_mode,tol", [(EXACT, 0), (FAST, 2e-4)])
def test_pcisph_steps(math_mode, tol, binning):
    """PC: PciMethod.Run (pcisph_darwin.go:43-101), 2 steps with 5 and 4 max iterations.
    binning = 1: DensityF's query points sorted into cells of their own (dsl_pcisph_set_binning), same bar."""
    p, pos, vel = _reference_system(12, math_mode, amp=0.1, vel_scale=0.05)
    for iters in (5, 4):
        p.pci_max_iters = iters
        p.delta = 1.0e-4
        eng = _engine(p)
        eng.pcisph_set_binning(binning)
        eng.upload("positions", pos)
        eng.upload("velocities", vel)
        ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel)
        ora.delta = p.delta
        eng.pcisph_begin(); ora.pcisph_begin()
        for step in range(2):
            eng.pcisph_step(1); ora.pcisph_step(1)
            st = eng.stats()
            assert st.pci_iters == ora.pci_iters
            assert abs(st.pci_max_error - ora.pci_error) <= tol * max(abs(ora.pci_error), 1e-3)
            assert _agree(eng.download("positions"), ora.positions(), tol)
            assert _agree(eng.download("velocities"), ora.velocities(), tol)
            assert _agree(eng.download("pci_positions"), ora.pci_positions(), tol)
            assert _agree(eng.download("pci_velocities"), ora.pci_velocities(), tol)
        eng.close()


@pytest.mark.parametrize("binning", [0, 1])
@pytest.mark.parametrize("math_mode,tol", [(EXACT, 0), (FAST, 1e-4)])
def test_pcisph_dambreak_scene(math_mode, tol, binning):
    """PCISPH on the dam-break block (h = 2dx, ~8 particles per cell): in FAST mode this is
    the LDS-tiled path (tiled viscosity sweep, cached gradient term, tiled DensityF), in
    EXACT mode the pass-by-pass kernels; both against the oracle's pcisph_darwin.go loop."""
    from dieselfluid_amd import scenes
    n3 = 12
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    p.pci_max_iters = 4
    p.delta = 2.0e-7
    vel = helpers.seeded_velocities(n3 ** 3, 0.2)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    eng = _engine(p)
    eng.pcisph_set_binning(binning)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.upload("forces", frc)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel, force=frc)
    ora.delta = p.delta
    eng.pcisph_begin(); ora.pcisph_begin()
    assert eng.pcisph_binning() == (binning, bool(binning))
    for step in range(3):
        eng.pcisph_step(1); ora.pcisph_step(1)
        st = eng.stats()
        assert st.pci_iters == ora.pci_iters
        assert abs(st.pci_max_error - ora.pci_error) <= 20 * tol * max(abs(ora.pci_error), 1e-3)
        assert _agree(eng.download("positions"), ora.positions(), tol)
        assert _agree(eng.download("velocities"), ora.velocities(), 20 * tol, floor=1e-2)
        assert _agree(eng.download("pci_positions"), ora.pci_positions(), tol)


@pytest.mark.parametrize("math_mode,tol", [(EXACT, 0), (FAST, 5e-5)])
def test_field_operators(math_mode, tol):
    """Div, Curl, Laplacian, Interpolate (sph_field.go:124-135,203-294)."""
    p, pos, vel = _reference_system(12, math_mode)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.density_all()
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel)
    ora.density_all()
    assert _agree(eng.field_div("velocities"), ora.field_div("velocity"), tol)
    assert _agree(eng.field_curl("velocities"), ora.field_curl("velocity"), tol)
    # (differences of nearly equal densities: the least forgiving of these sums)
    assert _agree(eng.field_laplacian("densities"), ora.field_laplacian("density"), 2 * tol)
    assert _agree(eng.field_laplacian("pressures"), ora.field_laplacian("pressure"), 8 * tol)
    q = (pos[::7] * np.float32(0.93) + np.float32(0.01)).astype(np.float32)
    assert _agree(eng.field_interpolate(q, "densities"), ora.field_interpolate(q, "density"), tol)
    assert _agree(eng.field_interpolate(q, "pressures"), ora.field_interpolate(q, "pressure"), 8 * tol)
    # a tensor field other than velocity: forces after ExternalAll + ViscousAll
    g = np.array([0, -9.81, 0], dtype=np.float32)
    eng.external_all(g); ora.external_all(g)
    eng.viscous_all(); ora.viscous_all()
    assert _agree(eng.field_div("forces"), ora.field_div("force"), 2 * tol)


@pytest.mark.parametrize("math_mode,tol_x,tol_v", [(EXACT, 0, 0), (FAST, 1e-5, 1e-3)])
@pytest.mark.parametrize("method", ["wcsph", "pcisph"])
def test_xsph_and_surface_tension_terms(method, math_mode, tol_x, tol_v):
    """BASELINE configs[4]'s extra terms (build-defined, oracle first): XSPH advection and the
    cohesion force, in the fused WCSPH step and in the PCISPH step."""
    from dieselfluid_amd import scenes
    n3 = 12
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    p.xsph_eps = 0.25
    p.st_kappa = 40.0
    p.pci_max_iters = 3
    p.delta = 2.0e-7
    vel = helpers.seeded_velocities(n3 ** 3, 0.5)
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.upload("forces", frc)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel, force=frc)
    ora.delta = p.delta
    # the terms must matter in this set-up, otherwise the test proves nothing
    p0, _ = scenes.dambreak_scene(n3, math_mode=math_mode)
    p0.pci_max_iters, p0.delta = 3, 2.0e-7
    ref0 = po.OracleSPH.from_state(helpers.oracle_params(p0), pos, vel=vel, force=frc)
    ref0.delta = p0.delta
    if method == "wcsph":
        eng.wcsph_step(5); ora.wcsph_step(5); ref0.wcsph_step(5)
    else:
        eng.pcisph_begin(); ora.pcisph_begin(); ref0.pcisph_begin()
        eng.pcisph_step(3); ora.pcisph_step(3); ref0.pcisph_step(3)
    assert helpers.rel_err(ref0.positions(), ora.positions()) > 5e-5
    assert _agree(eng.download("positions"), ora.positions(), tol_x)
    assert _agree(eng.download("velocities"), ora.velocities(), tol_v, floor=1e-2)


def test_upload_download_roundtrip_after_sort():
    """P: buffers keep the reference's host order across the device's re-sorting."""
    p, pos, vel = _reference_system(8, EXACT)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.nn()
    assert np.array_equal(eng.download("positions"), pos)
    assert np.array_equal(eng.download("velocities"), vel)
    vel2 = (vel * np.float32(2)).astype(np.float32)
    eng.upload("velocities", vel2)  # upload while the device order is permuted
    assert np.array_equal(eng.download("velocities"), vel2)


def test_render_handoff_decimated_and_device_pointers():
    """positions for the renderer without the full per-step read-back."""
    import ctypes as C
    import torch
    p, pos, vel = _reference_system(8, EXACT)
    eng = _engine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.nn()
    for stride in (1, 3, 8):
        assert np.array_equal(eng.download_decimated("positions", stride), pos[::stride])
        assert np.array_equal(eng.download_decimated("velocities", stride), vel[::stride])
    (xp, yp, zp), idp, n = eng.device_pointers("positions")
    assert n == eng.n and xp and yp and zp and idp
    # read the device arrays through HIP without going through the library
    hip = C.CDLL("libamdhip64.so")
    x = np.empty(n, dtype=np.float32)
    ids = np.empty(n, dtype=np.int32)
    assert hip.hipMemcpy(x.ctypes.data_as(C.c_void_p), C.c_void_p(xp), C.c_size_t(4 * n), 2) == 0
    assert hip.hipMemcpy(ids.ctypes.data_as(C.c_void_p), C.c_void_p(idp), C.c_size_t(4 * n), 2) == 0
    assert np.array_equal(x, pos[ids, 0])


def test_error_paths_do_not_abort():
    from dieselfluid_amd import SPHEngine, DslError, scenes
    p, pos = scenes.reference_scene(4)
    p.n_boundary = -3
    with pytest.raises(DslError):
        SPHEngine(p)
    p.n_boundary = 0
    p.capacity = p.n_particles - 1
    with pytest.raises(DslError):
        SPHEngine(p)
    p.capacity = 0
    eng = SPHEngine(p)
    with pytest.raises(DslError):
        eng.add_boundary_particles(pos[:2])  # no room: capacity == n_particles
    with pytest.raises(DslError):
        eng.upload("positions", pos[:-1])
