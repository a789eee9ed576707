"""ctypes harness around oracle/libdsloracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg, never from the product package
``dieselfluid_amd``.  See oracle/dsl_oracle.h for the parity status
("parity unpinned by the reference").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdsloracle.so")

NEIGH_LSH_REF, NEIGH_GRID, NEIGH_ALL = 0, 1, 2
ORDER_CELL, ORDER_ASCENDING = 0, 1
SAMPLES = 100


def build(force: bool = False) -> str:
    """Compile the oracle with oracle/Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "dsl_oracle.c")
    hdr = os.path.join(_HERE, "dsl_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or (os.path.exists(src) and os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libdsloracle.so"])
    return _LIB_PATH


class Kernel(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("A", "B", "C", "H1", "H_", "H2", "H3", "H4", "H5")]


class Particle(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("velocity", C.c_float * 3),
        ("force", C.c_float * 3),
        ("density", C.c_float),
        ("press", C.c_float),
    ]


class Params(C.Structure):
    _fields_ = [
        ("n3", C.c_int),
        ("neigh_mode", C.c_int),
        ("neigh_order", C.c_int),
        ("h", C.c_float),
        ("mass", C.c_float),
        ("ref_density", C.c_float),
        ("mu", C.c_float),
        ("dt", C.c_float),
        ("eos_w", C.c_float),
        ("eos_gamma", C.c_float),
        ("eos_d0_grad", C.c_float),
        ("pressure_sign", C.c_float),
        ("visc_running_mass", C.c_int),
        ("force_reset", C.c_float * 3),
        ("external", C.c_float * 3),
        ("wcsph_pressure_force", C.c_int),
        ("wcsph_viscosity", C.c_int),
        ("pci_max_iters", C.c_int),
        ("pci_max_error", C.c_float),
        ("walls", C.c_int),
        ("box_min", C.c_float * 3),
        ("box_max", C.c_float * 3),
        ("restitution", C.c_float),
        ("grid_min", C.c_float * 3),
        ("grid_max", C.c_float * 3),
        ("d0_override", C.c_float),
        ("xsph_eps", C.c_float),
        ("st_kappa", C.c_float),
    ]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    sig = {
        "dslo_build_kernel": (Kernel, [C.c_float]),
        "dslo_kernel_F": (C.c_float, [C.POINTER(Kernel), C.c_float]),
        "dslo_kernel_O1D": (C.c_float, [C.POINTER(Kernel), C.c_float]),
        "dslo_kernel_O2D": (C.c_float, [C.POINTER(Kernel), C.c_float]),
        "dslo_kernel_grad": (None, [C.POINTER(Kernel), C.c_float, fp, fp]),
        "dslo_tait_eos": (C.c_float, [C.c_float, C.c_float, C.c_float]),
        "dslo_tait_eos_ex": (C.c_float, [C.c_float] * 5),
        "dslo_vec_mag": (C.c_float, [fp, C.c_int]),
        "dslo_vec_dist3": (C.c_float, [fp, fp]),
        "dslo_vec_dot3": (C.c_float, [fp, fp]),
        "dslo_vec_norm3": (None, [fp, fp]),
        "dslo_vec_cross3": (None, [fp, fp, fp]),
        "dslo_vec_add": (C.c_int, [fp, C.c_int, fp, C.c_int, fp]),
        "dslo_vec_scale": (C.c_int, [fp, C.c_int, C.c_float, fp]),
        "dslo_vec_sub": (C.c_int, [fp, C.c_int, fp, C.c_int, fp]),
        "dslo_vec_proj3": (None, [fp, fp, fp]),
        "dslo_vec_refl3": (None, [fp, fp, fp]),
        "dslo_lsh_size": (C.c_int, [C.c_int, C.c_int]),
        "dslo_lattice_positions": (None, [C.c_int, fp, C.c_int, fp]),
        "dslo_params_reference": (Params, [C.c_int]),
        "dslo_params_sizeof": (C.c_size_t, []),
        "dslo_sph_init": (vp, [C.POINTER(Params), fp, C.c_int, fp, C.c_int, C.c_int]),
        "dslo_sph_from_state": (vp, [C.POINTER(Params), C.c_int, fp, fp, fp, fp, C.c_int]),
        "dslo_sph_free": (None, [vp]),
        "dslo_sph_add_boundary": (C.c_int, [vp, fp, C.c_int]),
        "dslo_mesh_boundary_particles": (None, [fp, C.c_int, fp]),
        "dslo_total": (C.c_int, [vp]),
        "dslo_cfl": (C.c_float, [vp]),
        "dslo_density_all": (None, [vp]),
        "dslo_pressure_all": (None, [vp]),
        "dslo_viscous_all": (None, [vp]),
        "dslo_external_all": (None, [vp, fp]),
        "dslo_gradient_pressure_force": (None, [vp]),
        "dslo_update": (None, [vp]),
        "dslo_cache_incr": (C.c_float, [vp, C.POINTER(C.c_int)]),
        "dslo_pcidelta": (C.c_float, [vp]),
        "dslo_density_f": (C.c_float, [vp, fp]),
        "dslo_sampler_update": (None, [vp]),
        "dslo_get_samples": (C.c_int, [vp, C.c_int, C.POINTER(C.POINTER(C.c_int))]),
        "dslo_get_samples_from_position": (C.c_int, [vp, fp, C.POINTER(C.POINTER(C.c_int))]),
        "dslo_lsh_get_data_1d": (None, [vp, C.POINTER(C.c_int)]),
        "dslo_lsh_hash_sys": None,
        "dslo_wcsph_step": (None, [vp]),
        "dslo_pcisph_begin": (None, [vp]),
        "dslo_pcisph_step": (None, [vp]),
        "dslo_xorshift64s": (C.c_uint64, [C.POINTER(C.c_uint64)]),
        "dslo_dambreak_positions": (None, [C.c_int, C.c_float, C.c_float, C.c_uint64, fp]),
        "dslo_n": (C.c_int, [vp]),
        "dslo_positions": (fp, [vp]),
        "dslo_velocities": (fp, [vp]),
        "dslo_forces": (fp, [vp]),
        "dslo_densities": (fp, [vp]),
        "dslo_pressures": (fp, [vp]),
        "dslo_pci_positions": (fp, [vp]),
        "dslo_pci_velocities": (fp, [vp]),
        "dslo_get_delta": (C.c_float, [vp]),
        "dslo_set_delta": (None, [vp, C.c_float]),
        "dslo_get_time": (C.c_float, [vp]),
        "dslo_get_max_vel": (C.c_float, [vp]),
        "dslo_get_max_f": (C.c_float, [vp]),
        "dslo_get_pci_error": (C.c_float, [vp]),
        "dslo_get_pci_iters": (C.c_int, [vp]),
        "dslo_get_lsh_size": (C.c_int, [vp]),
        "dslo_get_ref_density": (C.c_float, [vp]),
        "dslo_get_kernel": (None, [vp, C.POINTER(Kernel)]),
        "dslo_field_gradient": (None, [vp, C.c_int, fp]),
        "dslo_field_laplacian_force": (None, [vp, C.c_int, fp]),
        "dslo_lsh_hash_pos": (C.c_int, [vp, fp]),
        "dslo_field_div": (C.c_float, [vp, C.c_int, C.c_int]),
        "dslo_field_curl": (None, [vp, C.c_int, C.c_int, fp]),
        "dslo_field_laplacian": (C.c_float, [vp, C.c_int, C.c_int]),
        "dslo_field_interpolate": (C.c_float, [vp, fp, C.c_int]),
    }
    for name, s in sig.items():
        if s is None:
            continue
        fn = getattr(L, name)
        fn.restype, fn.argtypes = s
    assert L.dslo_params_sizeof() == C.sizeof(Params), "Params layout drifted from dsl_oracle.h"
    _lib = L
    return L


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def params_reference(n3: int) -> Params:
    """The reference's constants (model/sph/fluid.go:41-88 etc.)."""
    return lib().dslo_params_reference(int(n3))


def set_vec(field, v):
    for i in range(3):
        field[i] = float(v[i])


def mesh_boundary_particles(vertices) -> np.ndarray:
    """Mesh.GenerateBoundaryParticles (geom/mesh/mesh.go:60-76): one particle per vertex, the last one left at
    the origin by the reference's bound check"""
    v = f32(vertices).reshape(-1, 3)
    out = np.zeros_like(v)
    lib().dslo_mesh_boundary_particles(_fp(v), v.shape[0], _fp(out))
    return out


def default_hash_vectors(seed: int = 7) -> np.ndarray:
    """8 fixed projection vectors in (-0.5, 0.5)^3.  The reference draws them from Go's
    math/rand seeded with the wall-clock second (sampler/lsh/lsh.go:32-40), which cannot
    be reproduced here; they are therefore an explicit fixture."""
    st = C.c_uint64(seed * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF or 1)
    out = np.empty((8, 3), dtype=np.float32)
    L = lib()
    for i in range(8):
        for a in range(3):
            r = L.dslo_xorshift64s(C.byref(st))
            out[i, a] = np.float32(np.float32((r >> 40) / 16777216.0) - np.float32(0.5))
    return out


def lattice_positions(n3: int, origin=(0.0, 0.0, 0.0)) -> np.ndarray:
    """geom/grid/point-grid.go:33-63 + sph_field.go:87-108; origin=() reproduces the
    zero-length-origin collapse of sph_test.go:10."""
    pos = np.zeros((n3 ** 3, 3), dtype=np.float32)
    o = f32(list(origin) + [0.0] * (3 - len(origin)))
    lib().dslo_lattice_positions(n3, _fp(o), len(origin), _fp(pos))
    return pos


def dambreak_positions(n3: int, dx: float, jitter: float = 0.05, seed: int = 1234) -> np.ndarray:
    pos = np.zeros((n3 ** 3, 3), dtype=np.float32)
    lib().dslo_dambreak_positions(n3, C.c_float(dx), C.c_float(jitter), C.c_uint64(seed), _fp(pos))
    return pos


class OracleSPH:
    """Handle on one oracle SPH system (mirrors model/sph.SPH)."""

    def __init__(self, handle, prm: Params):
        self._h = C.c_void_p(handle)
        self.prm = prm
        self._L = lib()

    # -- construction ---------------------------------------------------------------
    @classmethod
    def init(cls, prm: Params, origin=(0.0, 0.0, 0.0), hash_vectors=None, pci=False) -> "OracleSPH":
        """sph.Init (model/sph/fluid.go:41-88)."""
        L = lib()
        o = f32(list(origin) + [0.0] * (3 - len(origin)))
        hv = f32(hash_vectors if hash_vectors is not None else default_hash_vectors())
        h = L.dslo_sph_init(C.byref(prm), _fp(o), len(origin), _fp(hv), hv.shape[0], int(pci))
        return cls(h, prm)

    @classmethod
    def from_state(cls, prm: Params, pos, vel=None, force=None, hash_vectors=None) -> "OracleSPH":
        L = lib()
        pos = f32(pos).reshape(-1, 3)
        n = pos.shape[0]
        vel_p = _fp(f32(vel).reshape(-1, 3)) if vel is not None else None
        frc_p = _fp(f32(force).reshape(-1, 3)) if force is not None else None
        hv = f32(hash_vectors if hash_vectors is not None else default_hash_vectors())
        h = L.dslo_sph_from_state(C.byref(prm), n, _fp(pos), vel_p, frc_p, _fp(hv), hv.shape[0])
        return cls(h, prm)

    def close(self):
        if self._h:
            self._L.dslo_sph_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state views (copies) -------------------------------------------------------
    @property
    def n(self) -> int:
        return self._L.dslo_n(self._h)

    def _view(self, fn, cols):
        n = self.n
        ptr = fn(self._h)
        arr = np.ctypeslib.as_array(ptr, shape=(n * cols,))
        return arr.reshape(n, cols) if cols > 1 else arr

    def positions(self):
        return self._view(self._L.dslo_positions, 3).copy()

    # -- boundary particles (particle_array.go:123-128, mesh.go:60-76) -------------------------
    def add_boundary(self, positions):
        """ParticleArray.AddBoundaryParticles + a sampler rebuild; returns Total()"""
        a = f32(positions).reshape(-1, 3)
        return int(self._L.dslo_sph_add_boundary(self._h, _fp(a), a.shape[0]))

    @property
    def total(self) -> int:
        return int(self._L.dslo_total(self._h))

    def all_positions(self):
        """positions of the Total() = N + Nb particles (the reference's positions slice)"""
        ptr = self._L.dslo_positions(self._h)
        return np.ctypeslib.as_array(ptr, shape=(self.total * 3,)).reshape(-1, 3).copy()

    def velocities(self):
        return self._view(self._L.dslo_velocities, 3).copy()

    def forces(self):
        return self._view(self._L.dslo_forces, 3).copy()

    def densities(self):
        return self._view(self._L.dslo_densities, 1).copy()

    def pressures(self):
        return self._view(self._L.dslo_pressures, 1).copy()

    def pci_positions(self):
        return self._view(self._L.dslo_pci_positions, 3).copy()

    def pci_velocities(self):
        return self._view(self._L.dslo_pci_velocities, 3).copy()

    def set_velocities(self, v):
        self._view(self._L.dslo_velocities, 3)[:] = f32(v).reshape(-1, 3)

    def set_forces(self, v):
        self._view(self._L.dslo_forces, 3)[:] = f32(v).reshape(-1, 3)

    def set_densities(self, v):
        self._view(self._L.dslo_densities, 1)[:] = f32(v).reshape(-1)

    def set_positions(self, v):
        self._view(self._L.dslo_positions, 3)[:] = f32(v).reshape(-1, 3)

    # -- scalars --------------------------------------------------------------------
    delta = property(lambda s: s._L.dslo_get_delta(s._h), lambda s, v: s._L.dslo_set_delta(s._h, C.c_float(v)))
    time = property(lambda s: s._L.dslo_get_time(s._h))
    max_vel = property(lambda s: s._L.dslo_get_max_vel(s._h))
    max_f = property(lambda s: s._L.dslo_get_max_f(s._h))
    pci_error = property(lambda s: s._L.dslo_get_pci_error(s._h))
    pci_iters = property(lambda s: s._L.dslo_get_pci_iters(s._h))
    lsh_size = property(lambda s: s._L.dslo_get_lsh_size(s._h))
    ref_density = property(lambda s: s._L.dslo_get_ref_density(s._h))

    def kernel(self) -> Kernel:
        k = Kernel()
        self._L.dslo_get_kernel(self._h, C.byref(k))
        return k

    # -- passes (model/sph/fluid.go:111-277) ---------------------------------------
    def cfl(self):
        return self._L.dslo_cfl(self._h)

    def density_all(self):
        self._L.dslo_density_all(self._h)

    def pressure_all(self):
        self._L.dslo_pressure_all(self._h)

    def viscous_all(self):
        self._L.dslo_viscous_all(self._h)

    def external_all(self, f):
        self._L.dslo_external_all(self._h, _fp(f32(f)))

    def gradient_pressure_force(self):
        self._L.dslo_gradient_pressure_force(self._h)

    def update(self):
        self._L.dslo_update(self._h)

    def cache_incr(self):
        r = C.c_int(0)
        c = self._L.dslo_cache_incr(self._h, C.byref(r))
        return c, bool(r.value)

    def pcidelta(self):
        return self._L.dslo_pcidelta(self._h)

    def density_f(self, pos):
        return self._L.dslo_density_f(self._h, _fp(f32(pos)))

    def sampler_update(self):
        self._L.dslo_sampler_update(self._h)

    def get_samples(self, i):
        p = C.POINTER(C.c_int)()
        n = self._L.dslo_get_samples(self._h, int(i), C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def get_samples_from_position(self, pos):
        p = C.POINTER(C.c_int)()
        n = self._L.dslo_get_samples_from_position(self._h, _fp(f32(pos)), C.byref(p))
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def lsh_hash(self, pos):
        return self._L.dslo_lsh_hash_pos(self._h, _fp(f32(pos)))

    def lsh_data_1d(self):
        out = np.zeros(255 * self.lsh_size, dtype=np.int32)
        self._L.dslo_lsh_get_data_1d(self._h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def field_gradient(self, i):
        out = np.zeros(3, dtype=np.float32)
        self._L.dslo_field_gradient(self._h, int(i), _fp(out))
        return out

    def field_laplacian_force(self, i):
        out = np.zeros(3, dtype=np.float32)
        self._L.dslo_field_laplacian_force(self._h, int(i), _fp(out))
        return out

    # -- field operators no solver calls (sph_field.go:124-135,203-294) -----------------
    SCALAR = {"density": 0, "pressure": 1}
    TENSOR = {"velocity": 0, "force": 1}

    def field_div(self, tensor="velocity"):
        f = self.TENSOR[tensor]
        return np.array([self._L.dslo_field_div(self._h, i, f) for i in range(self.n)], dtype=np.float32)

    def field_curl(self, tensor="velocity"):
        f = self.TENSOR[tensor]
        out = np.zeros((self.n, 3), dtype=np.float32)
        tmp = np.zeros(3, dtype=np.float32)
        for i in range(self.n):
            self._L.dslo_field_curl(self._h, i, f, _fp(tmp))
            out[i] = tmp
        return out

    def field_laplacian(self, scalar="density"):
        f = self.SCALAR[scalar]
        return np.array([self._L.dslo_field_laplacian(self._h, i, f) for i in range(self.n)], dtype=np.float32)

    def field_interpolate(self, positions, scalar="density"):
        f = self.SCALAR[scalar]
        pos = f32(positions).reshape(-1, 3)
        return np.array([self._L.dslo_field_interpolate(self._h, _fp(pos[k]), f) for k in range(pos.shape[0])],
                        dtype=np.float32)

    # -- drivers --------------------------------------------------------------------
    def wcsph_step(self, nsteps=1):
        for _ in range(nsteps):
            self._L.dslo_wcsph_step(self._h)

    def pcisph_begin(self):
        self._L.dslo_pcisph_begin(self._h)

    def pcisph_step(self, nsteps=1):
        for _ in range(nsteps):
            self._L.dslo_pcisph_step(self._h)
