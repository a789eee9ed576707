"""Shared test helpers: map the product's dsl_params onto the oracle's parameter block
(the two structs are defined independently on purpose) and build seeded inputs."""
import numpy as np

from oracle import pyoracle as po


def oracle_params(p, mode=po.NEIGH_GRID, order=po.ORDER_CELL, n3=0):
    """dieselfluid_amd.Params -> oracle Params (field by field)."""
    q = po.params_reference(4)
    q.n3 = n3
    q.neigh_mode, q.neigh_order = mode, order
    # the engine's ref_density is the final D0(); NewParticleArray would multiply by mass again
    # (particle_array.go:26), so hand it over through SetReferenceDensity (particle_array.go:35-37)
    q.d0_override = p.ref_density
    for name in ("h", "mass", "ref_density", "mu", "dt", "eos_w", "eos_gamma", "eos_d0_grad", "pressure_sign",
                 "visc_running_mass", "wcsph_pressure_force", "wcsph_viscosity", "pci_max_iters", "pci_max_error",
                 "walls", "restitution", "xsph_eps", "st_kappa"):
        setattr(q, name, getattr(p, name))
    for name in ("force_reset", "external", "box_min", "box_max", "grid_min", "grid_max"):
        for a in range(3):
            getattr(q, name)[a] = getattr(p, name)[a]
    return q


def jittered_lattice(n3, amp=0.2, seed=1234, origin=(0.0, 0.0, 0.0)):
    """sph.Init lattice plus a seeded uniform jitter of +-amp*step (SURVEY 8c fixture 2)."""
    pos = po.lattice_positions(n3, origin)
    rng = np.random.default_rng(seed)
    step = np.float32(2.0 / n3)
    jit = (rng.random(pos.shape, dtype=np.float32) * np.float32(2) - np.float32(1)) * np.float32(amp) * step
    return (pos + jit).astype(np.float32)


def seeded_velocities(n, scale=0.1, seed=99):
    rng = np.random.default_rng(seed)
    return ((rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.float32(2 * scale)).astype(np.float32)


def rel_err(a, b, floor=0.0):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = max(np.max(np.abs(b)), floor, 1e-30)
    return float(np.max(np.abs(a - b)) / scale)


def brute_force_step_f64(p, x, v, probe, near):
    """One WCSPH step of the build's fused force+integrate kernel, in float64, by brute force, for
    the particles `probe` (index array) of the state (x, v); `near` lists every particle within 2h of
    a probe particle.  The formulas are the reference's (std_kernel.go:33-76, model.go:92-101,
    sph_field.go:155-200,251-269 without the running-mass product, fluid.go:175-197) plus the
    build-defined wall box.  Returns (rho_probe, x_new, v_new).  Independent of the oracle and of the
    engine: numpy only."""
    h, m = float(p.h), float(p.mass)
    PI = 3.141592653589
    A, B, Ck = 315.0 / (64.0 * PI * h ** 3), -45.0 / (PI * h ** 4), 90.0 / (PI * h ** 5)
    xn, vn = x[near].astype(np.float64), v[near].astype(np.float64)
    # densities of every particle within h of a probe particle
    xp = x[probe].astype(np.float64)
    close = np.zeros(xn.shape[0], dtype=bool)
    for k in range(xp.shape[0]):
        close |= ((xn - xp[k]) ** 2).sum(axis=1) < h * h
    rho_n = np.full(xn.shape[0], np.nan)
    for k in np.nonzero(close)[0]:
        d2 = ((xn - xn[k]) ** 2).sum(axis=1)
        msk = (d2 < h * h) & (d2 > 0)
        rho_n[k] = m * A * ((1.0 - d2[msk] / (h * h)) ** 2).sum()

    def pterm(rho):
        r = np.maximum(rho, float(p.eos_d0_grad)) / float(p.eos_d0_grad)
        return (float(p.eos_w) / float(p.eos_gamma)) * (r ** float(p.eos_gamma) - 1.0) / (rho * rho)

    pos_of = {int(g): k for k, g in enumerate(near)}
    rho_p = np.empty(xp.shape[0])
    x_new, v_new = np.empty_like(xp), np.empty_like(xp)
    dt = float(p.dt)
    for k, g in enumerate(probe):
        i = pos_of[int(g)]
        d = xn - xn[i]
        d2 = (d * d).sum(axis=1)
        msk = (d2 < h * h) & (d2 > 0)
        r = np.sqrt(d2[msk])
        rho_p[k] = rho_n[i]
        F = np.array([float(p.force_reset[a]) for a in range(3)])
        if p.wcsph_pressure_force:
            grad = (d[msk] / r[:, None]) * (-(B * (1.0 - r / h) ** 2))[:, None]  # Grad = dir * (-O1D)
            G = ((pterm(rho_n[i]) + pterm(rho_n[msk]))[:, None] * grad).sum(axis=0)
            F = F + float(p.pressure_sign) * rho_n[i] * m * G
        if p.wcsph_viscosity:
            assert not p.visc_running_mass
            V = (m * (vn[msk] - vn[i]) / rho_n[msk][:, None] * (Ck * (1.0 - r / h))[:, None]).sum(axis=0)
            F = F + float(p.mu) * V
        F = F + np.array([float(p.external[a]) for a in range(3)])
        vv = vn[i] + (F / m) * dt
        xx = xn[i] + vv * dt
        if p.walls:
            for a in range(3):
                if xx[a] < float(p.box_min[a]):
                    xx[a] = float(p.box_min[a])
                    if vv[a] < 0:
                        vv[a] = -vv[a] * float(p.restitution)
                if xx[a] > float(p.box_max[a]):
                    xx[a] = float(p.box_max[a])
                    if vv[a] > 0:
                        vv[a] = -vv[a] * float(p.restitution)
        x_new[k], v_new[k] = xx, vv
    return rho_p, x_new, v_new


def fast_velocity_tolerance(p, steps, eps_rho=2.0e-6):
    """Absolute velocity tolerance (m/s) of DSL_MATH_FAST against the oracle after `steps` steps of a
    scene with the pressure force on.  Error model: FAST densities agree with the oracle to eps_rho ~ 2e-6
    (fma + expanded r^2 instead of separately rounded operations); the Tait EOS turns that into
    gamma * eps_rho * B of pressure, the pressure gradient into about gamma * eps_rho * c_s^2 / h of
    acceleration (c_s^2 = eos_w / mass in the build's scenes), one step into that times dt of velocity.
    Measured (tools/fast_errors.py, MI355X): 0.08-0.2 of this bound after 1 step, 0.03-0.07 after 10
    (errors of successive steps do not add coherently); after ~40 steps of a developed flow the two
    trajectories part for good (a neighbour crossing r = h one step apart), which is why parity is
    checked over at most 10 steps.  Half the bound is asserted.  `eps_rho`: the density agreement the model starts
    from -- 2e-6 on a jittered lattice; a developed flow (cells of 4 to 26 particles, tile-relative coordinates up to 3.5
    cells) measures up to 5e-6 (tools/pair_diag.py), and its tests pass 6e-6."""
    cs2 = float(p.eos_w) / float(p.mass)
    return 0.5 * float(p.eos_gamma) * eps_rho * cs2 / float(p.h) * float(p.dt) * steps
