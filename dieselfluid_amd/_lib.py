"""ctypes loader for libdslsph.so (include/dslsph.h).  Fails loudly: there is no
fallback path of any kind when the HIP library is missing or fails to load."""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_BUILT = os.path.join(_PKG, "lib", "libdslsph.so")  # what build_library() makes, always
# DSL_LIB: LOAD another build of the same library (A/B variants, diagnostic builds such as -DDSL_DIAG_STAMPS).
# It never redirects the build: a default-flags rebuild over a variant would make an A/B run measure the same
# code twice without noticing.
_LIB = os.environ.get("DSL_LIB") or _BUILT
_SRC_DIR = os.path.join(_PKG, "csrc")
_HDR = os.path.join(_ROOT, "include", "dslsph.h")

# -fno-slp-vectorize: packed FP32 (v_pk_*) issues slower than two scalar ops once operands
# are distinct registers (tools/valu_rate.hip), so keep the compiler from forming them.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-std=c++17", "-Wall"]


class DslError(RuntimeError):
    pass


class Params(C.Structure):
    """dsl_params (include/dslsph.h)."""
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("n_particles", C.c_int32), ("n_boundary", C.c_int32),
        ("lsh_buckets", C.c_int32), ("lsh_bucket_size", C.c_int32),
        ("dt", C.c_float), ("mass", C.c_float), ("delta", C.c_float), ("max_vel", C.c_float), ("h", C.c_float),
        ("ref_density", C.c_float), ("mu", C.c_float),
        ("eos_w", C.c_float), ("eos_gamma", C.c_float), ("eos_d0_grad", C.c_float),
        ("pressure_sign", C.c_float), ("visc_running_mass", C.c_int32),
        ("force_reset", C.c_float * 3), ("external", C.c_float * 3),
        ("wcsph_pressure_force", C.c_int32), ("wcsph_viscosity", C.c_int32),
        ("pci_max_iters", C.c_int32), ("pci_max_error", C.c_float),
        ("walls", C.c_int32), ("box_min", C.c_float * 3), ("box_max", C.c_float * 3), ("restitution", C.c_float),
        ("grid_min", C.c_float * 3), ("grid_max", C.c_float * 3),
        ("neigh_mode", C.c_int32), ("math_mode", C.c_int32), ("capacity", C.c_int32),
        ("xsph_eps", C.c_float), ("st_kappa", C.c_float),
        ("sort_unordered", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


class Stats(C.Structure):
    """dsl_stats (include/dslsph.h)."""
    _fields_ = [
        ("max_vel", C.c_float), ("max_f", C.c_float), ("pci_max_error", C.c_float), ("pci_iters", C.c_int32),
        ("steps", C.c_int64), ("grid_dims", C.c_int32 * 3), ("grid_cells", C.c_int32), ("max_cell_count", C.c_int32),
    ]


# dsl_transport (include/dslsph.h): the host's own transport behind dsl_comm_create_custom
TR_GROUP = C.CFUNCTYPE(C.c_int, C.c_void_p)
TR_XFER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
TR_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("group_start", TR_GROUP), ("group_end", TR_GROUP), ("send", TR_XFER),
                ("recv", TR_XFER), ("all_reduce_max_u32", TR_REDUCE)]


# every symbol include/dslsph.h declares: name -> (restype, argtypes)
_vp, _fp, _ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
EXPORTS = {
    "dsl_params_reference": (C.c_int, [C.POINTER(Params), C.c_int]),
    "dsl_create": (C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(_vp)]),
    "dsl_destroy": (C.c_int, [_vp]),
    "dsl_set_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "dsl_get_params": (C.c_int, [_vp, C.POINTER(Params)]),
    "dsl_upload": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "dsl_download": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "dsl_add_boundary_particles": (C.c_int, [_vp, _fp, C.c_size_t]),
    "dsl_download_decimated": (C.c_int, [_vp, C.c_int, C.c_int, _fp, C.c_size_t]),
    "dsl_device_pointers": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int)]),
    "dsl_set_hash_vectors": (C.c_int, [_vp, _fp, C.c_int]),
    "dsl_lsh_download_table": (C.c_int, [_vp, _ip, C.c_size_t]),
    "dsl_build_neighbours": (C.c_int, [_vp]),
    "dsl_density_pass": (C.c_int, [_vp]),
    "dsl_pressure_pass": (C.c_int, [_vp]),
    "dsl_viscous_pass": (C.c_int, [_vp]),
    "dsl_external_pass": (C.c_int, [_vp, _fp]),
    "dsl_gradient_pressure_pass": (C.c_int, [_vp]),
    "dsl_update_pass": (C.c_int, [_vp]),
    "dsl_force_pass": (C.c_int, [_vp]),
    "dsl_field_divergence": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "dsl_field_curl": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "dsl_field_laplacian": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "dsl_field_interpolate": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t, _fp]),
    "dsl_wcsph_step": (C.c_int, [_vp, C.c_int]),
    "dsl_pcisph_begin": (C.c_int, [_vp]),
    "dsl_pcisph_step": (C.c_int, [_vp, C.c_int]),
    "dsl_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "dsl_sync": (C.c_int, [_vp]),
    "dsl_set_stream": (C.c_int, [_vp, _vp]),
    "dsl_use_own_stream": (C.c_int, [_vp]),
    "dsl_timing_enable": (C.c_int, [_vp, C.c_int]),
    "dsl_timing_reset": (C.c_int, [_vp]),
    "dsl_timing_get": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dsl_download_sorted": (C.c_int, [_vp, C.c_int, _fp, C.c_size_t]),
    "dsl_download_ids": (C.c_int, [_vp, _ip, C.c_size_t]),
    "dsl_download_cell_start": (C.c_int, [_vp, _ip, C.c_size_t]),
    "dsl_slab_config": (C.c_int, [_vp, C.c_int, C.c_float, C.c_float]),
    "dsl_slab_message_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "dsl_slab_record_floats": (C.c_int, [_vp]),
    "dsl_slab_message_floats_for": (C.c_size_t, [_vp, C.c_int, C.c_int]),
    "dsl_pcisph_phase": (C.c_int, [_vp, C.c_int]),
    "dsl_pcisph_error_word": (C.c_int, [_vp, _vp, C.c_int]),
    "dsl_pcisph_set_binning": (C.c_int, [_vp, C.c_int]),
    "dsl_pcisph_get_binning": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dsl_slab_split": (C.c_int, [_vp, C.c_float, C.c_float]),
    "dsl_slab_pack": (C.c_int, [_vp, C.c_float, C.c_float, _vp, _vp, C.c_int, C.c_int]),
    "dsl_slab_pack_band": (C.c_int, [_vp, C.c_float, _vp, _vp, C.c_int, C.c_int, _vp]),
    "dsl_force_pass_split": (C.c_int, [_vp, C.c_int]),
    "dsl_slab_append": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "dsl_slab_append2": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int]),
    "dsl_slab_status": (C.c_int, [_vp, C.POINTER(C.c_int32), C.c_int]),
    "dsl_slab_overflow": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "dsl_get_count": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dsl_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "dsl_comm_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.POINTER(_vp)]),
    "dsl_comm_create_all": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "dsl_comm_create_custom": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "dsl_comm_destroy": (C.c_int, [_vp]),
    "dsl_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "dsl_comm_last_error": (C.c_char_p, []),
    "dsl_create_multi": (C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(C.c_int), C.POINTER(_vp), C.POINTER(_vp)]),
    "dsl_slab_attach": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]),
    "dsl_slab_detach": (C.c_int, [_vp]),
    "dsl_slab_image_shift": (C.c_int, [_vp, C.c_float, C.c_float]),
    "dsl_slab_exchange": (C.c_int, [_vp]),
    "dsl_slab_replan": (C.c_int, [_vp]),
    "dsl_slab_wcsph_step": (C.c_int, [_vp, C.c_int]),
    "dsl_slab_pcisph_step": (C.c_int, [_vp, C.c_int]),
    "dsl_set_option": (C.c_int, [_vp, C.c_int, C.c_double]),
    "dsl_get_option": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    "dsl_set_ids": (C.c_int, [_vp, _ip, C.c_size_t]),
    "dsl_reset_forces": (C.c_int, [_vp]),
    "dsl_last_error": (C.c_char_p, [_vp]),
    "dsl_version": (C.c_char_p, []),
}


def library_path() -> str:
    return _LIB


def _sources():
    return sorted(os.path.join(_SRC_DIR, f) for f in os.listdir(_SRC_DIR) if f.endswith((".hip", ".hpp")))


def build_library(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> dieselfluid_amd/lib/libdslsph.so (in-tree)."""
    deps = _sources() + [_HDR]
    if not force and os.path.exists(_BUILT) and all(os.path.getmtime(_BUILT) >= os.path.getmtime(d) for d in deps):
        return _BUILT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise DslError("hipcc not found: cannot build libdslsph.so (no fallback exists)")
    os.makedirs(os.path.dirname(_BUILT), exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", _BUILT, os.path.join(_SRC_DIR, "dslsph.hip")]
    subprocess.check_call(cmd)
    return _BUILT


_lib = None


def load_library() -> C.CDLL:
    """Load libdslsph.so and bind every symbol of include/dslsph.h; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        raise DslError(f"{_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
                       "The engine has no CPU or PyTorch fallback.")
    try:
        L = C.CDLL(_LIB)
    except OSError as e:  # pragma: no cover
        raise DslError(f"cannot load {_LIB}: {e}") from e
    for name, (res, args) in EXPORTS.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise DslError(f"{_LIB} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L
