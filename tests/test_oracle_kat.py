"""Pins the CPU oracle: the exact vector checks the reference's own tests hold
(math/math_test.go:13-110) and the hand-derived known-answer values of SURVEY.md 8c.
The reference has no SPH golden vectors and cannot be built here (no Go toolchain), so
these are the only external anchors: parity is "unpinned by the reference"."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po

f32 = np.float32


def _v(*a):
    return po.f32(a)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ---- math/math_test.go ---------------------------------------------------------------
def test_vector_mag_matches_reference_test():
    """math_test.go:80 -- Mag({2,2,2}) == float32(math.Sqrt(12))"""
    L = po.lib()
    a = _v(2, 2, 2)
    assert L.dslo_vec_mag(_p(a), 3) == f32(np.sqrt(np.float64(12.0)))


def test_vector_dot_and_cross_match_reference_test():
    """math_test.go:40-50 Dot({1,2,3},{1,1,1}) == 6; :75 Cross({-2,-2,-2},{1,2,1}) == {2,0,-2}"""
    L = po.lib()
    assert L.dslo_vec_dot3(_p(_v(1, 2, 3)), _p(_v(1, 1, 1))) == f32(6.0)
    out = np.zeros(3, dtype=f32)
    L.dslo_vec_cross3(_p(_v(-2, -2, -2)), _p(_v(1, 2, 1)), _p(out))
    assert np.array_equal(out, _v(2, 0, -2))


def test_vector_add_scale_match_reference_test():
    """math_test.go:13-24 Add({1,1,1},{1,1,1}) == {2,2,2}; :66 Scale({2,2,2}, 2) == {4,4,4};
    :69 Add({2,2,2},{2,2,2}) == {4,4,4}"""
    L = po.lib()
    out = np.zeros(3, dtype=f32)
    assert L.dslo_vec_add(_p(_v(1, 1, 1)), 3, _p(_v(1, 1, 1)), 3, _p(out)) == 3
    assert np.array_equal(out, _v(2, 2, 2))
    assert L.dslo_vec_scale(_p(_v(2, 2, 2)), 3, f32(2.0), _p(out)) == 3
    assert np.array_equal(out, _v(4, 4, 4))
    assert L.dslo_vec_add(_p(_v(2, 2, 2)), 3, _p(_v(2, 2, 2)), 3, _p(out)) == 3
    assert np.array_equal(out, _v(4, 4, 4))


def test_vector_add_of_an_empty_vector_is_the_zero_vector():
    """vector.go:167-173 (and math_test.go:58-64: Vec{} is not {0,0,0}): the rule behind the collapsed
    lattice of sph.Init(1.0, Vec{}, ...)"""
    L = po.lib()
    out = np.ones(3, dtype=f32)
    empty = np.zeros(1, dtype=f32)
    assert L.dslo_vec_add(_p(empty), 0, _p(_v(1, 2, 3)), 3, _p(out)) == 3
    assert np.array_equal(out, _v(0, 0, 0))
    out[:] = 1
    assert L.dslo_vec_sub(_p(_v(1, 2, 3)), 3, _p(empty), 0, _p(out)) == 3
    assert np.array_equal(out, _v(0, 0, 0))


def test_vector_proj_and_refl_match_reference_test():
    """math_test.go:86-98 Proj({2,2,0},{0,2,0}) == {0,2,0}; :100-106 Refl({1,-1,0},{0,1,0}) == {1,1,0}"""
    L = po.lib()
    out = np.zeros(3, dtype=f32)
    L.dslo_vec_proj3(_p(_v(2, 2, 0)), _p(_v(0, 2, 0)), _p(out))
    assert np.array_equal(out, _v(0, 2, 0))
    L.dslo_vec_refl3(_p(_v(1, -1, 0)), _p(_v(0, 1, 0)), _p(out))
    assert np.array_equal(out, _v(1, 1, 0))


def test_vector_norm_of_zero_is_zero():
    """vector.go:322-331"""
    L = po.lib()
    out = np.ones(3, dtype=f32)
    L.dslo_vec_norm3(_p(_v(0, 0, 0)), _p(out))
    assert np.array_equal(out, _v(0, 0, 0))


# ---- K: kernel/std_kernel.go --------------------------------------------------------------
def test_kernel_constants_h1():
    k = po.lib().dslo_build_kernel(1.0)
    assert f32(k.A) == f32(1.5666814) and f32(k.B) == f32(-14.323944) and f32(k.C) == f32(28.647888)


def test_kernel_values():
    L = po.lib()
    k = L.dslo_build_kernel(1.0)
    assert L.dslo_kernel_F(k, 0.5) == f32(0.88125825)
    assert L.dslo_kernel_O1D(k, 0.5) == f32(-3.580986)
    assert L.dslo_kernel_O2D(k, 0.5) == f32(14.323944)
    assert L.dslo_kernel_F(k, 1.0) == 0.0          # x >= h
    assert L.dslo_kernel_O1D(k, 1.0) == 0.0
    assert L.dslo_kernel_O2D(k, 1.0) == 0.0        # not cut ('>') but C*(1-1) = 0
    assert L.dslo_kernel_O2D(k, np.nextafter(f32(1.0), f32(2.0))) == 0.0
    assert L.dslo_kernel_F(k, 0.0) == f32(k.A)     # W0


# ---- E: model/model.go:92-101 ------------------------------------------------------------
def test_tait_eos_values():
    L = po.lib()
    assert L.dslo_tait_eos(400.0, 512.0, 3.5) == f32(3.5)         # clamped at d0 -> p0
    assert L.dslo_tait_eos(512.0, 512.0, 0.0) == 0.0
    assert L.dslo_tait_eos(1024.0, 512.0, 0.0) == f32(42.643494)
    assert L.dslo_tait_eos(512.0, 87.0, 0.0) == f32(97485.29)


# ---- I / S0 ---------------------------------------------------------------------------------
def test_init_lattice_and_constants_n16():
    prm = po.params_reference(16)
    assert prm.ref_density == 512.0 and prm.mass == 1.0 and prm.dt == f32(0.01) and prm.h == 1.0
    pos = po.lattice_positions(16)
    assert pos.shape == (4096, 3)
    # particle id k + 16(16 i + j) sits at (-1 + 0.125 i, -1 + 0.125 j, -1 + 0.125 k)
    for (i, j, k) in [(0, 0, 0), (3, 5, 7), (15, 15, 15), (8, 0, 1)]:
        pid = k + 16 * (16 * i + j)
        assert np.array_equal(pos[pid], _v(-1 + 0.125 * i, -1 + 0.125 * j, -1 + 0.125 * k))
    assert po.lib().dslo_lsh_size(4096, 255) == 24


def test_zero_length_origin_collapses_lattice():
    """sph_test.go:10 passes vector.Vec{}: V.Add returns (0,0,0) on a length mismatch
    (vector.go:171-173) so every particle sits at the origin."""
    pos = po.lattice_positions(8, origin=())
    assert np.all(pos == 0.0)


def test_interior_lattice_point_has_2102_neighbours_within_h():
    r = np.arange(-8, 9)
    x, y, z = np.meshgrid(r, r, r, indexing="ij")
    d2 = (x * x + y * y + z * z) * (0.125 ** 2)
    assert int(np.sum((d2 < 1.0) & (d2 > 0))) == 2102


# ---- W: free fall -----------------------------------------------------------------------------
def test_wcsph_free_fall_sequence():
    """Update resets F to gravity and the driver adds gravity again (2g, fluid.go:193 +
    wcsph.go:19); no pressure force in the loop."""
    s = po.OracleSPH.init(po.params_reference(16))
    want_v = [f32(-0.1962), f32(-0.3924), f32(-0.5886)]
    want_y = [f32(-1.001962), f32(-1.005886), f32(-1.0117719)]
    for k in range(3):
        s.wcsph_step()
        v, x = s.velocities(), s.positions()
        assert np.all(v[:, 1] == want_v[k]) and np.all(v[:, 0] == 0) and np.all(v[:, 2] == 0)
        assert x[0, 1] == want_y[k]


# ---- T: CacheIncr ------------------------------------------------------------------------------
def test_cache_incr_rebuilds_every_fourth_call():
    s = po.OracleSPH.init(po.params_reference(8))
    pattern = [s.cache_incr()[1] for _ in range(12)]
    assert pattern == [False, False, False, True] * 3


# ---- N: lsh -------------------------------------------------------------------------------------
def test_lsh_hash_rules():
    hv = np.full((8, 3), 0.25, dtype=f32)
    s = po.OracleSPH.init(po.params_reference(8), hash_vectors=hv)
    assert s.lsh_hash(_v(-1, -1, -1)) == 0            # all dots <= 0
    assert s.lsh_hash(_v(0, 0, 0)) == 0               # sgn(0) = 0
    assert s.lsh_hash(_v(1, 1, 1)) == 255 % 255       # code 255 aliases bucket 0


def test_get_samples_cycles_short_bucket():
    """lsh.go:142-156: a bucket with fewer than 100 entries is re-read from its start."""
    hv = po.default_hash_vectors()
    prm = po.params_reference(4)
    pos = po.lattice_positions(4)
    s = po.OracleSPH.from_state(prm, pos, hash_vectors=hv)
    smp = s.get_samples(0)
    assert smp.shape == (100,)
    b = s.lsh_hash(pos[0])
    members = [i for i in range(64) if s.lsh_hash(pos[i]) == b]
    assert 0 in members
    want = (members * (100 // len(members) + 1))[:100]
    assert list(smp) == want
    flat = s.lsh_data_1d()
    assert flat.shape == (255 * s.lsh_size,)


# ---- PD ------------------------------------------------------------------------------------------
def test_pcidelta_formula():
    """delta = -1/(beta*denom), beta = dt^2 m^2 2/rho0^2 (fluid.go:265-277), recomputed in
    float64 from the same 8^3 lattice walk."""
    prm = po.params_reference(16)
    s = po.OracleSPH.init(prm, pci=True)
    pos = po.lattice_positions(8).astype(np.float64)
    k = po.lib().dslo_build_kernel(1.0)
    idx, tr = [], 0
    for i in range(512):
        if i % 2 == 0:
            idx.append(256 + tr)
        else:
            idx.append(256 - tr)
            tr += 1
    assert idx[0] == 256 and idx[1] == 256 and 0 not in idx
    d1, d2 = np.zeros(3), 0.0
    for x in idx:
        p = pos[x]
        r = np.linalg.norm(p)
        if r * r < 1.0:
            g = (p / r if r > 0 else np.zeros(3)) * (-(k.B * (1 - r) ** 2))
            d1 += g
            d2 += g @ g
    denom = -(d1 @ d1) - d2
    beta = (0.01 ** 2) * 1.0 * (2.0 / 512.0 ** 2)
    want = -1.0 / (beta * denom)
    assert abs(s.delta - want) / abs(want) < 1e-5
    assert s.delta > 0


# ---- P: Get quirk -------------------------------------------------------------------------------
def test_particle_array_get_index_equal_n_is_empty():
    """particle_array.go:98,107: index == n_particles returns the zero particle."""
    L = po.lib()

    class PA(C.Structure):
        _fields_ = [("positions", C.POINTER(C.c_float)), ("velocities", C.POINTER(C.c_float)),
                    ("densities", C.POINTER(C.c_float)), ("forces", C.POINTER(C.c_float)),
                    ("pressures", C.POINTER(C.c_float)), ("n", C.c_int), ("nb", C.c_int),
                    ("mass", C.c_float), ("rd", C.c_float)]
    pa = PA()
    L.dslo_particles_init.argtypes = [C.POINTER(PA), C.c_int, C.c_int, C.c_float, C.c_float]
    L.dslo_particles_get.argtypes = [C.POINTER(PA), C.c_int]
    L.dslo_particles_get.restype = po.Particle
    L.dslo_particles_free.argtypes = [C.POINTER(PA)]
    assert L.dslo_particles_init(C.byref(pa), 4, 3, 2.0, 0.5) == 0
    assert pa.rd == 1.0  # ReferenceDensity = density*mass (particle_array.go:26)
    for i in range(7 * 3):
        pa.positions[i] = 1.0 + i
    q = L.dslo_particles_get(C.byref(pa), 4)          # == n_particles -> zero particle
    assert list(q.position) == [0, 0, 0]
    q = L.dslo_particles_get(C.byref(pa), 5)          # boundary: position only
    assert list(q.position) == [16.0, 17.0, 18.0] and q.density == 0
    q = L.dslo_particles_get(C.byref(pa), 7)          # >= Total
    assert list(q.position) == [0, 0, 0]
    L.dslo_particles_free(C.byref(pa))


# ---- grid-mode self consistency ----------------------------------------------------------------
@pytest.mark.parametrize("n3", [8, 12])
def test_grid_mode_equals_brute_force(n3):
    """The cell-grid candidate rule must find exactly the particles the brute-force rule
    finds; with both visiting candidates in ascending index order the float32 results are
    bit-identical."""
    import helpers
    pos = helpers.jittered_lattice(n3, 0.3)
    vel = helpers.seeded_velocities(n3 ** 3)
    out = []
    for mode in (po.NEIGH_GRID, po.NEIGH_ALL):
        prm = po.params_reference(n3)
        prm.neigh_mode, prm.neigh_order = mode, po.ORDER_ASCENDING
        s = po.OracleSPH.from_state(prm, pos, vel=vel)
        s.density_all()
        s.viscous_all()
        s.gradient_pressure_force()
        out.append((s.densities(), s.forces()))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))


# ---- two particles, closed form -------------------------------------------------------------------
def test_two_particles_closed_form():
    """Two particles 0.5 h apart, every pass worked out by hand from the reference's formulas
    (std_kernel.go:33-76, model.go:92-101, sph_field.go:155-200,251-269, fluid.go:127-197) in float64:
    Density, LaplacianForce x mu (one neighbour: the running-sum x m quirk is a plain product), Gradient with the
    pressures of FLUID_DENSITY = 87 (model.go:41) and Update.  Independent of the oracle's code paths: numbers only."""
    PI = 3.141592653589
    A, B, Ck = 315.0 / (64.0 * PI), -45.0 / PI, 90.0 / PI  # h = 1
    prm = po.params_reference(4)
    prm.neigh_mode = po.NEIGH_ALL
    pos = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.0]], dtype=np.float32)
    vel = np.array([[0.0, 0.0, 0.0], [0.2, 0.0, 0.0]], dtype=np.float32)
    s = po.OracleSPH.from_state(prm, pos, vel=vel)
    # D: rho = m F(0.5) = A (1 - 0.25)^2
    s.density_all()
    rho = A * 0.5625
    assert np.allclose(s.densities(), [rho, rho], rtol=2e-7)
    # V: F_0 = mu * ((v_1 - v_0) / rho_1) * O2D(0.5) * m, O2D = C (1 - 0.5); F_1 = -F_0
    s.viscous_all()
    fv = float(prm.mu) * (0.2 / rho) * (Ck * 0.5)
    f = s.forces()
    assert np.allclose(f[:, 0], [fv, -fv], rtol=1e-6) and np.all(f[:, 1:] == 0)
    # G with rho = 100 > 87: P = (2.15/7.16) ((100/87)^7.16 - 1); grad_0 = dir * (-O1D) with dir = +x,
    # O1D = B (1 - 0.5)^2; F = 2 P / rho^2; result = grad F rho m, ADDED to the force (fluid.go:168-169)
    s.set_forces(np.zeros((2, 3), dtype=np.float32))
    s.set_densities(np.array([100.0, 100.0], dtype=np.float32))
    s.gradient_pressure_force()
    P = (2.15 / 7.16) * ((100.0 / 87.0) ** 7.16 - 1.0)
    g0 = (-(B * 0.25)) * (2.0 * P / 100.0 ** 2) * 100.0
    f = s.forces()
    assert np.allclose(f[:, 0], [g0, -g0], rtol=2e-6) and np.all(f[:, 1:] == 0)
    assert g0 > 0  # the reference's sign: particle 0 is pushed TOWARDS its neighbour
    # U: v += F/m dt, x += v dt, force reset (0, -9.81 m, 0), pressure 0 (dt = 0.01)
    s.update()
    v0 = g0 * 0.01
    assert np.allclose(s.velocities()[:, 0], [v0, 0.2 - v0], rtol=2e-6)
    assert np.allclose(s.positions()[:, 0], [v0 * 0.01, 0.5 + (0.2 - v0) * 0.01], rtol=2e-6, atol=1e-9)
    assert np.allclose(s.forces(), [[0.0, -9.81, 0.0]] * 2) and np.all(s.pressures() == 0)


def test_three_particles_one_pcisph_step_worked_by_hand():
    """Three collinear particles, one PCISPH step restated from the reference's formulas alone (float64, this function):
    DensityAll (sph_field.go:155-172), then the correction loop of pcisph_darwin.go:52-98 -- predict (:57-73), DensityF at
    the predicted position, which STARTS at W0 and meets the particle itself as well (sph_field.go:137-152: rho* = 2 W0 +
    rho), the pressure accumulator (density_error * delta, :76-92), GradientPressureForce with P/rho^2 of the CURRENT
    densities, ADDED to the force (fluid.go:164-172, sph_field.go:175-200: the force on i points TOWARDS j) -- the SIGNED
    max error of :95-98 (all errors negative: 0 <= 1 %, the loop ends after its first iteration; errors of +30 %: it runs
    all of them), then Update (fluid.go:175-197).  Velocities start at zero, so ViscousAll adds nothing."""
    PI = 3.141592653589
    A, B = 315.0 / (64.0 * PI), -45.0 / PI  # h = 1
    x0 = np.array([[0.0, 0.0, 0.0], [0.4, 0.0, 0.0], [0.9, 0.0, 0.0]])
    dt, m, d0 = 0.01, 1.0, 1.0

    def F(r):
        return np.where(r < 1.0, A * (1.0 - r * r) ** 2, 0.0)

    def by_hand(ref_density, iters_max, delta):
        x, v = x0.copy(), np.zeros_like(x0)
        r = np.abs(x[:, None, 0] - x[None, :, 0])
        rho = np.array([sum(m * F(r[i, j]) for j in range(3) if j != i) for i in range(3)])
        force = np.zeros_like(x0)
        ppos, pvel = x.copy(), v.copy()
        press = np.zeros(3)
        P = (2.15 / 7.16) * ((np.maximum(rho, d0) / d0) ** 7.16 - 1.0)
        iters, err = 0, 0.0
        for it in range(iters_max):
            iters = it + 1
            err = 0.0
            pvel = pvel + (force / m) * dt
            ppos = ppos + pvel * dt
            for i in range(3):
                calc = A + sum(m * F(np.linalg.norm(ppos[i] - x[j])) for j in range(3))  # W0 + every sample, itself included
                e = calc - ref_density
                press[i] += e * delta
                err = max(err, e / ref_density)
            for i in range(3):
                g = np.zeros(3)
                for j in range(3):
                    if j == i:
                        continue
                    d = x[j] - x[i]
                    rr = np.linalg.norm(d)
                    if rr < 1.0:
                        g += (d / rr) * (-(B * (1.0 - rr) ** 2)) * (P[i] / rho[i] ** 2 + P[j] / rho[j] ** 2)
                force[i] += g * rho[i] * m
            if err <= 0.01:
                break
        v = v + (force / m) * dt
        x = x + v * dt
        return x, v, iters, err, rho

    for ref_density, iters_max, want_iters in ((8.0, 5, 1), (2.0, 3, 3)):
        prm = po.params_reference(4)
        prm.neigh_mode = po.NEIGH_ALL
        prm.eos_d0_grad = d0  # (FLUID_DENSITY = 87 would clamp every pressure of this tiny system to 0)
        prm.d0_override = ref_density
        prm.pci_max_iters = iters_max
        s = po.OracleSPH.from_state(prm, x0.astype(np.float32), vel=np.zeros((3, 3), dtype=np.float32),
                                    force=np.zeros((3, 3), dtype=np.float32))
        s.delta = 0.5
        s.pcisph_begin()
        s.pcisph_step(1)
        x, v, iters, err, rho = by_hand(ref_density, iters_max, 0.5)
        assert iters == want_iters and s.pci_iters == iters
        assert abs(s.pci_error - err) <= 2e-6 * max(abs(err), 1.0)
        assert err == 0.0 if want_iters == 1 else err > 0.3
        assert np.allclose(s.velocities(), v, rtol=5e-6, atol=1e-9) and np.allclose(s.positions(), x, rtol=5e-6, atol=1e-9)
        assert v[0, 0] > 0 and v[2, 0] < 0  # the reference's sign: the outer particles are pulled inwards
        assert np.allclose(s.forces(), [[0.0, -9.81, 0.0]] * 3) and np.all(s.pressures() == 0)
