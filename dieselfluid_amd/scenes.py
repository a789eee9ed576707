"""Initial conditions and parameter sets (host side, numpy only).

* reference_scene: sph.Init's lattice on [-1,1)^3 (geom/grid/point-grid.go:33-63,
  model/field/sph_field.go:87-108) with the reference constants.
* dambreak_scene: the build's synthetic dam-break -- no reference
  counterpart; the oracle carries an independent C restatement of the same generator and
  tests/test_scenes.py checks the two agree bit for bit.
"""
from __future__ import annotations

import numpy as np

from ._lib import Params
from .engine import reference_params

f32 = np.float32


def lattice_positions(n3: int, origin=(0.0, 0.0, 0.0)) -> np.ndarray:
    """pos[id = k + n3*(i*n3 + j)] = min + step*(i,j,k); float32 operation order of
    BuildKernGrid / GridPosition (point-grid.go:26-28,39-40,60-63)."""
    origin = np.asarray(origin, dtype=f32)
    if origin.shape != (3,):  # V.Add length mismatch -> (0,0,0) (vector.go:171-173)
        minb = np.zeros(3, dtype=f32)
    else:
        minb = (f32(-1.0) * np.ones(3, dtype=f32) + origin).astype(f32)
    inv = f32(1.0) / f32(n3)
    step = (inv * (minb * f32(-2.0)).astype(f32)).astype(f32)
    i, j, k = np.meshgrid(np.arange(n3), np.arange(n3), np.arange(n3), indexing="ij")
    ijk = np.stack([i, j, k], axis=-1).reshape(-1, 3).astype(f32)  # id order = i-major, k fastest
    return (minb[None, :] + (step[None, :] * ijk).astype(f32)).astype(f32)


def reference_scene(n3: int, grid_half_extent: float = 4.0):
    """(params, positions) for sph.Init(1.0, {0,0,0}, nil, n3, pci) in grid mode."""
    p = reference_params(n3)
    for a in range(3):
        p.grid_min[a] = -grid_half_extent
        p.grid_max[a] = grid_half_extent
    return p, lattice_positions(n3)


def _splitmix64(seed: int, counters: np.ndarray) -> np.ndarray:
    """counter-based splitmix64: mix(seed + (c+1)*0x9E3779B97F4A7C15), vectorised."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (counters.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def dambreak_positions(n3: int, dx: float, jitter: float = 0.05, seed: int = 1234) -> np.ndarray:
    """Fluid block of n3^3 particles at ((i,j,k)+0.5)*dx with +-jitter*dx uniform jitter
    from splitmix64(seed, 3*id + axis), id = k + n3*(i*n3 + j)."""
    n = n3 ** 3
    r = _splitmix64(seed, np.arange(3 * n, dtype=np.uint64))
    u = ((r >> np.uint64(40)).astype(f32) * f32(1.0 / 16777216.0)).astype(f32)
    jit = ((u * f32(2.0) - f32(1.0)).astype(f32) * f32(f32(jitter) * f32(dx))).astype(f32)
    i, j, k = np.meshgrid(np.arange(n3), np.arange(n3), np.arange(n3), indexing="ij")
    ijk = np.stack([i, j, k], axis=-1).reshape(-1, 3).astype(f32)
    base = ((ijk + f32(0.5)).astype(f32) * f32(dx)).astype(f32)
    return (base + jit.reshape(-1, 3)).astype(f32)


def dambreak_slab_ids(n3: int, axis: int, l0: int, l1: int) -> np.ndarray:
    """ids (k + n3*(i*n3 + j)) of the lattice layers l0 <= index < l1 along `axis`
    (0: i/x, 1: j/y, 2: k/z), ascending."""
    rng = [np.arange(n3)] * 3
    rng = list(rng)
    rng[axis] = np.arange(l0, l1)
    i, j, k = np.meshgrid(rng[0], rng[1], rng[2], indexing="ij")
    ids = (k + n3 * (i * n3 + j)).reshape(-1)
    return np.sort(ids).astype(np.int32)


def dambreak_positions_ids(n3: int, dx: float, ids, jitter: float = 0.05, seed: int = 1234) -> np.ndarray:
    """dambreak_positions restricted to the given particle ids (same values bit for bit)."""
    ids = np.asarray(ids, dtype=np.int64)
    k = ids % n3
    j = (ids // n3) % n3
    i = ids // (n3 * n3)
    ctr = (3 * ids[:, None] + np.arange(3)[None, :]).reshape(-1).astype(np.uint64)
    r = _splitmix64(seed, ctr)
    u = ((r >> np.uint64(40)).astype(f32) * f32(1.0 / 16777216.0)).astype(f32)
    jit = ((u * f32(2.0) - f32(1.0)).astype(f32) * f32(f32(jitter) * f32(dx))).astype(f32)
    ijk = np.stack([i, j, k], axis=-1).astype(f32)
    base = ((ijk + f32(0.5)).astype(f32) * f32(dx)).astype(f32)
    return (base + jit.reshape(-1, 3)).astype(f32)


def lattice_rest_density(h_over_dx: float, mass: float, dx: float) -> float:
    """Density the reference kernel (kernel/std_kernel.go:33-39) gives an interior particle
    of a perfect cubic lattice, self term excluded as in SPHField.Density."""
    h = h_over_dx * dx
    A = 315.0 / (64.0 * 3.141592653589 * h ** 3)
    r = int(np.ceil(h_over_dx)) + 1
    o = np.arange(-r, r + 1)
    ox, oy, oz = np.meshgrid(o, o, o, indexing="ij")
    d2 = (ox * ox + oy * oy + oz * oz).astype(np.float64) * dx * dx
    mask = (d2 > 0) & (d2 < h * h)
    q = 1.0 - d2[mask] / (h * h)
    return float(mass * A * np.sum(q * q))


def dambreak_scene(n3: int, *, h_over_dx: float = 2.0, jitter: float = 0.05, seed: int = 1234, fluid_edge: float = 1.0,
                   rho_phys: float = 1000.0, cfl: float = 0.25, visc: float = 0.02, math_mode: int = 1,
                   positions: bool = True):
    """(params, positions) of the synthetic dam-break: fluid block [0,L]^3 in the corner of a
    4L x 2L x L box, h = h_over_dx*dx, gravity -y, wall box = clamp + reflect.  All force
    terms use the reference's formulas (kernel, Tait EOS, Gradient, LaplacianForce, Update);
    the switches that are unphysical for m != 1 are set to their physical values
    (pressure repels, viscosity without the running-mass product, single gravity)."""
    L = float(fluid_edge)
    dx = L / n3
    h = h_over_dx * dx
    mass = rho_phys * dx ** 3
    rho0 = lattice_rest_density(h_over_dx, mass, dx)
    g = 9.81
    c_s = 10.0 * np.sqrt(2.0 * g * L)
    dt = cfl * h / c_s
    p = reference_params(4)  # reference constants as the starting point
    p.n_particles = n3 ** 3
    p.lsh_bucket_size = int(np.float32(p.n_particles // 255) * np.float32(1.5))
    p.dt = dt
    p.mass = mass
    p.h = h
    p.ref_density = rho0
    p.eos_d0_grad = rho0
    # reference Gradient carries rho_i*m instead of m_i*m_j (sph_field.go:199): absorb the
    # extra rho/m into the stiffness so that the sound speed is c_s
    p.eos_w = c_s * c_s * mass
    p.mu = visc * mass * dx * c_s
    p.pressure_sign = -1.0
    p.visc_running_mass = 0
    p.force_reset[0], p.force_reset[1], p.force_reset[2] = 0.0, -g * mass, 0.0
    p.external[0] = p.external[1] = p.external[2] = 0.0
    p.wcsph_pressure_force = 1
    p.wcsph_viscosity = 1
    p.walls = 1
    box = (4.0 * L, 2.0 * L, 1.0 * L)
    for a in range(3):
        p.box_min[a] = 0.0
        p.box_max[a] = box[a]
        p.grid_min[a] = -h
        p.grid_max[a] = box[a] + h
    p.restitution = 0.0
    p.math_mode = math_mode
    pos = dambreak_positions(n3, dx, jitter, seed) if positions else None
    return p, pos
