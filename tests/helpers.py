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
