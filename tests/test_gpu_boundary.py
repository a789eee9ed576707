"""(f2) Boundary particles / colliders on the device against the oracle: ParticleArray.AddBoundaryParticles
(model/particle_array.go:123-128), Get()'s position-only view of them (:94-117, including Get(N()) = the zero
particle), SPHField.BoundaryParticles (model/field/sph_field.go:75-85), Mesh.GenerateBoundaryParticles
(geom/mesh/mesh.go:60-76), the `pIndex < Total()` guards of Density/DensityF (sph_field.go:143,163) and their
absence in Gradient/LaplacianForce (:183,259: density 0 -> NaN / Inf, reproduced)."""
import numpy as np
import pytest

import helpers
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
EXACT, FAST = 0, 1


def _box_vertices(half, step):
    """vertex list of a tessellated axis-aligned box [-half, half]^3 (what a collider mesh would hold)"""
    t = np.arange(-half, half + 1e-6, step, dtype=np.float32)
    u, v = np.meshgrid(t, t, indexing="ij")
    faces = []
    for axis in range(3):
        for side in (-half, half):
            f = np.empty((u.size, 3), dtype=np.float32)
            f[:, axis] = side
            f[:, (axis + 1) % 3] = u.reshape(-1)
            f[:, (axis + 2) % 3] = v.reshape(-1)
            faces.append(f)
    return np.concatenate(faces).astype(np.float32)


def _system(math_mode, n3=8):
    from dieselfluid_amd import SPHEngine, scenes
    p, pos = scenes.reference_scene(n3)
    p.math_mode = math_mode
    rng = np.random.default_rng(11)
    pos = (pos + (rng.random(pos.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.05)).astype(np.float32)
    vel = helpers.seeded_velocities(n3 ** 3, 0.1, seed=7)
    bverts = _box_vertices(1.25, 0.25)
    bpos = po.mesh_boundary_particles(bverts)  # mesh.go:60-76 (the last vertex's particle stays at the origin)
    assert np.array_equal(bpos[:-1], bverts[:-1]) and np.array_equal(bpos[-1], np.zeros(3, np.float32))
    p.capacity = n3 ** 3 + bpos.shape[0]
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    eng.upload("velocities", vel)
    eng.add_boundary_particles(bpos)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=vel)
    assert ora.add_boundary(bpos) == n3 ** 3 + bpos.shape[0]
    return p, eng, ora, pos, bpos


def test_boundary_buffers_keep_the_reference_layout():
    p, eng, ora, pos, bpos = _system(EXACT)
    n, nb = pos.shape[0], bpos.shape[0]
    assert eng.n == n + nb and eng.n_fluid == n and eng.params.n_boundary == nb
    x = eng.download("positions")  # Total() particles, fluid first; the slice keeps what was uploaded
    assert x.shape == (n + nb, 3) and np.array_equal(x[:n], pos) and np.array_equal(x[n:], bpos)
    assert eng.download("velocities").shape == (n, 3) and eng.download("densities").shape == (n,)
    eng.nn()  # the sort moves boundary particles like any other; host order survives
    assert np.array_equal(eng.download("positions"), np.concatenate([pos, bpos]))
    with pytest.raises(Exception):
        eng.upload("positions", pos)  # N instead of Total() particles
    eng.close()


@pytest.mark.parametrize("math_mode,tol", [(EXACT, 0), (FAST, 2e-5)])
def test_reference_wcsph_loop_with_a_collider_box(math_mode, tol):
    """wcsph.go:14-26 (density, gravity, EOS, Update) with boundary particles around the fluid: they add to the
    densities (sph_field.go:155-172 counts every sample < Total()), nothing moves them.  EXACT: bit for bit."""
    p, eng, ora, pos, bpos = _system(math_mode)
    n = pos.shape[0]
    frc = np.tile(np.array([0, -9.81, 0], dtype=np.float32), (n, 1))
    eng.upload("forces", frc)
    ora2 = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=eng.download("velocities"), force=frc)
    ora2.add_boundary(bpos)
    free = po.OracleSPH.from_state(helpers.oracle_params(p), pos, vel=eng.download("velocities"), force=frc)
    eng.wcsph_step(3); ora2.wcsph_step(3); free.wcsph_step(3)
    rho, rho_o = eng.download("densities"), ora2.densities()
    assert np.isfinite(rho_o).all() and (rho_o - free.densities()).max() > 1.0  # the boundary really adds density
    if tol == 0:
        assert np.array_equal(rho.view(np.uint32), rho_o.view(np.uint32))
        assert np.array_equal(eng.download("velocities").view(np.uint32), ora2.velocities().view(np.uint32))
        assert np.array_equal(eng.download("positions")[:n].view(np.uint32), ora2.positions().view(np.uint32))
    else:
        assert helpers.rel_err(rho, rho_o) < tol
        assert helpers.rel_err(eng.download("positions")[:n], ora2.positions()) < 1e-6
    assert np.array_equal(eng.download("positions")[n:], bpos)  # Update loops over N() particles only
    eng.close()


def test_the_first_boundary_particle_is_read_at_the_origin():
    """particle_array.go:94-117: Get(index) with index == n_particles falls through to the zero Particle.  One
    boundary particle far away from the fluid therefore raises the density of the fluid around the ORIGIN."""
    from dieselfluid_amd import SPHEngine, scenes
    p, pos = scenes.reference_scene(8)
    p.capacity = 8 ** 3 + 2
    far = np.array([[3.5, 3.5, 3.5], [3.0, 3.5, 3.5]], dtype=np.float32)
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    eng.density_all()
    rho_free = eng.download("densities")
    eng.add_boundary_particles(far)
    eng.density_all()
    rho = eng.download("densities")
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos)
    ora.add_boundary(far)
    ora.density_all()
    assert np.array_equal(rho.view(np.uint32), ora.densities().view(np.uint32))
    r = np.sqrt((pos.astype(np.float64) ** 2).sum(axis=1))
    changed = rho != rho_free
    assert changed.any() and np.all(r[changed] < 1.0) and np.all(changed[(r < 0.99) & (r > 0)])
    assert np.array_equal(eng.download("positions")[-2:], far)  # the slice itself is untouched
    eng.close()


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
def test_gradient_and_viscosity_divide_by_the_boundary_density_zero(math_mode):
    """sph_field.go:183,259: no `< Total()` guard in Gradient / LaplacianForce: a boundary neighbour
    contributes P_j / 0^2 = NaN resp. (v_j - v_i) / 0 = Inf or NaN.  EXACT reproduces the oracle's values bit
    for bit, non-finite ones included; FAST is non-finite exactly where the oracle is and agrees elsewhere."""
    p, eng, ora, pos, bpos = _system(math_mode)
    eng.density_all(); ora.density_all()
    eng.viscous_all(); ora.viscous_all()
    fv, fv_o = eng.download("forces"), ora.forces()
    assert (~np.isfinite(fv_o)).any()  # (h = 1 and the boundary particle at the origin: here every particle has one)
    eng2, ora2 = _system(math_mode)[1:3]
    eng2.density_all(); ora2.density_all()
    eng2.gradient_pressure_force(); ora2.gradient_pressure_force()
    fg, fg_o = eng2.download("forces"), ora2.forces()
    assert np.isnan(fg_o).any()
    for got, want in ((fv, fv_o), (fg, fg_o)):
        if math_mode == EXACT:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        else:
            ok = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), ok)
            if ok.any():
                assert np.abs(got[ok] - want[ok]).max() < 5e-5 * np.abs(want[ok]).max()
    eng.close(); eng2.close()


@pytest.mark.parametrize("math_mode", [EXACT, FAST])
def test_fused_step_with_boundary_particles(math_mode):
    """the build's dam-break step (pressure + viscosity + walls, tiled kernels in FAST mode) with a collider
    plate under the block: particles next to it go non-finite exactly as the oracle's pass-by-pass loop says
    (pressure term over a boundary density of 0), the others agree with it; boundary particles stay put."""
    from dieselfluid_amd import SPHEngine, scenes
    n3 = 12
    p, pos = scenes.dambreak_scene(n3, math_mode=math_mode)
    h = p.h
    t = np.arange(0.25 * h, 1.0, 0.5 * h, dtype=np.float32)
    u, v = np.meshgrid(t, t, indexing="ij")
    plate = np.stack([u.reshape(-1), np.full(u.size, -0.4 * h, np.float32), v.reshape(-1)], axis=1).astype(np.float32)
    p.capacity = n3 ** 3 + plate.shape[0]
    frc = np.tile(np.array(p.force_reset[:], dtype=np.float32), (n3 ** 3, 1))
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    eng.upload("forces", frc)
    eng.add_boundary_particles(plate)
    ora = po.OracleSPH.from_state(helpers.oracle_params(p), pos, force=frc)
    ora.add_boundary(plate)
    eng.wcsph_step(2); ora.wcsph_step(2)
    gx, ox = eng.download("positions"), ora.positions()
    n = n3 ** 3
    assert np.array_equal(gx[n:], plate)
    bad = ~np.isfinite(ox).all(axis=1)
    assert bad.any() and (~bad).any()
    assert np.array_equal(~np.isfinite(gx[:n]).all(axis=1), bad)
    if math_mode == EXACT:
        assert np.array_equal(gx[:n][~bad].view(np.uint32), ox[~bad].view(np.uint32))
    else:
        assert helpers.rel_err(gx[:n][~bad], ox[~bad]) < 1e-5
    eng.close()


def test_stale_densities_stay_on_their_particles():
    """new boundary particles (or a new mass) leave the densities stale, not scrambled -- the reference keeps
    the old value on the same particle until the next DensityAll (particle_array.go:94-117 reads whatever
    `densities[i]` holds).  A neighbour build in between moves the particles; their densities must move with them.
    The appended slots read as Get() reads a boundary particle."""
    from dieselfluid_amd import SPHEngine, scenes
    n3 = 8
    p, pos = scenes.reference_scene(n3)
    p.math_mode = EXACT
    rng = np.random.default_rng(5)
    pos = (pos + (rng.random(pos.shape, dtype=np.float32) - np.float32(0.5)) * np.float32(0.2)).astype(np.float32)
    bpos = _box_vertices(1.25, 0.25)
    p.capacity = n3 ** 3 + bpos.shape[0]
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos[::-1].copy())  # (an order the sort has to change)
    eng.density_all()
    rho0 = eng.download("densities")
    assert rho0.min() > 0 and np.unique(rho0).size > 100
    eng.add_boundary_particles(bpos)          # densities are stale now ...
    eng.nn()                                  # ... and the sort re-orders every array
    assert np.array_equal(eng.download("densities"), rho0)
    q = eng.params
    q.mass = 2.0 * q.mass
    eng.set_params(q)                         # stale again (rho scales with the mass), still on their particles
    eng.nn()
    assert np.array_equal(eng.download("densities"), rho0)
    eng.density_all()
    assert not np.array_equal(eng.download("densities"), rho0)
    eng.close()


def test_set_params_wants_the_current_boundary_count():
    """dsl_set_params with the creation-time block after dsl_add_boundary_particles used to leave
    dsl_get_params reporting n_boundary = 0; now it is refused, and the refreshed block is accepted."""
    from dieselfluid_amd import SPHEngine, scenes
    from dieselfluid_amd._lib import DslError
    n3 = 8
    p, pos = scenes.reference_scene(n3)
    bpos = _box_vertices(1.25, 0.5)
    p.capacity = n3 ** 3 + bpos.shape[0]
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    eng.add_boundary_particles(bpos)
    with pytest.raises(DslError):
        eng.set_params(p)                     # p.n_boundary is still 0
    q = eng.params
    assert q.n_boundary == bpos.shape[0]
    q.mu = 2.0 * q.mu
    eng.set_params(q)
    assert eng.params.n_boundary == bpos.shape[0] and eng.n_fluid == n3 ** 3
    eng.close()
