"""SPHEngine: thin object wrapper over the C ABI (include/dslsph.h).

Method names follow model/sph.SPH (model/sph/fluid.go:111-215) and the solver drivers
(solver/wcsph/wcsph.go, solver/pcisph/pcisph_darwin.go) so the parity tests read like
the reference's call sequences.  Every method is one C-ABI call; no arithmetic happens
in Python.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import DslError, Params, Stats, load_library

BUF = {
    "positions": 0, "velocities": 1, "forces": 2, "densities": 3, "pressures": 4,
    "pci_positions": 5, "pci_velocities": 6,
}
_COMPS = {0: 3, 1: 3, 2: 3, 3: 1, 4: 1, 5: 3, 6: 3}
KERNEL_IDS = {
    "cell_rank": 0, "scan": 1, "scatter": 2, "density": 3, "force_integrate": 4, "pressure": 5, "viscous": 6,
    "gradient": 7, "external": 8, "update": 9, "pci_predict": 10, "pci_density": 11, "tile_list": 12,
    "neigh_lists": 13,
}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def reference_params(n3: int) -> Params:
    """sph.Init's constants (dsl_params_reference)."""
    L = load_library()
    p = Params()
    rc = L.dsl_params_reference(C.byref(p), int(n3))
    if rc:
        raise DslError(L.dsl_last_error(None).decode())
    return p


class Comm:
    """dsl_comm: the RCCL communicator libdslsph.so owns (include/dslsph.h).  Rank 0 makes the unique id,
    the host hands its 128 bytes to every rank (here: any callable `bcast(bytes_or_None) -> bytes`)."""

    def __init__(self, nranks: int, rank: int, device: int, bcast=None):
        self._L = load_library()
        self.ptr = C.c_void_p()
        self.nranks, self.rank = int(nranks), int(rank)
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            rc = self._L.dsl_comm_unique_id(ident)
            if rc:
                raise DslError(f"dsl_comm_unique_id failed ({rc}): {self._L.dsl_comm_last_error().decode()}")
        if nranks > 1:
            if bcast is None:
                raise DslError("Comm: nranks > 1 needs a broadcast function for the unique id")
            raw = bcast(bytes(ident) if rank == 0 else None)
            ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        rc = self._L.dsl_comm_create(self.nranks, self.rank, ident, int(device), C.byref(self.ptr))
        if rc:
            raise DslError(f"dsl_comm_create failed ({rc}): {self._L.dsl_comm_last_error().decode()}")

    def count(self) -> int:
        """ranks the communicator really spans (ncclCommCount)"""
        n = C.c_int(0)
        rc = self._L.dsl_comm_count(self.ptr, C.byref(n))
        if rc:
            raise DslError(f"dsl_comm_count failed ({rc}): {self._L.dsl_comm_last_error().decode()}")
        return int(n.value)

    def close(self):
        if getattr(self, "ptr", None):
            self._L.dsl_comm_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostStagedComm(Comm):
    """dsl_comm over the host's OWN transport (dsl_comm_create_custom, include/dslsph.h): the library makes
    the same call sequence as it makes to RCCL (group_start / send, recv per neighbour / group_end;
    all_reduce_max for the re-plan words and the PCISPH iteration error), and this table moves the messages
    through host memory with torch.distributed (gloo).  It exists for the places RCCL cannot go -- several
    ranks on ONE device (the 2- and 3-rank GPU tests of the library's step drivers, gloo rehearsals of
    bench.py --gpus N) -- and doubles as the worked example of a host-supplied transport.  Blocking: every
    operation synchronises the stream it is ordered on."""

    def __init__(self, nranks: int, rank: int, device: int, group=None):
        import numpy as np
        import torch
        import torch.distributed as dist
        from ._lib import TR_GROUP, TR_REDUCE, TR_XFER, Transport
        self._L = load_library()
        self.ptr = C.c_void_p()
        self.nranks, self.rank = int(nranks), int(rank)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        hip.hipSetDevice.argtypes = [C.c_int]
        H2D, D2H = 1, 2
        state = {"ops": None, "calls": []}
        self.calls = state["calls"]  # (tests) the sequence of transport calls the library made

        def run(ops):
            hip.hipSetDevice(int(device))
            for st in {o[4] for o in ops}:
                if hip.hipStreamSynchronize(st):
                    return 1
            works, recvs, keep = [], [], []
            for kind, buf, nbytes, peer, _st in ops:
                t = torch.empty(nbytes, dtype=torch.uint8)
                keep.append(t)
                if kind == "send":
                    if hip.hipMemcpy(t.data_ptr(), buf, nbytes, D2H):
                        return 1
                    works.append(dist.P2POp(dist.isend, t, peer, group=group))
                else:
                    works.append(dist.P2POp(dist.irecv, t, peer, group=group))
                    recvs.append((t, buf, nbytes))
            for w in dist.batch_isend_irecv(works):
                w.wait()
            for t, buf, nbytes in recvs:
                if hip.hipMemcpy(buf, t.data_ptr(), nbytes, H2D):
                    return 1
            return 0

        def guarded(fn):
            def call(*a):
                try:
                    return int(fn(*a))
                except Exception as e:  # an exception must not unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    self.error = e
                    return 1
            return call

        def group_start(_ctx):
            state["calls"].append("group_start")
            state["ops"] = []
            return 0

        def group_end(_ctx):
            state["calls"].append("group_end")
            ops, state["ops"] = state["ops"], None
            return run(ops) if ops else 0

        def xfer(kind):
            def f(_ctx, buf, nbytes, peer, stream):
                state["calls"].append((kind, int(peer), int(nbytes)))
                op = (kind, buf, int(nbytes), int(peer), stream)
                if state["ops"] is None:
                    return run([op])
                state["ops"].append(op)
                return 0
            return f

        def all_reduce(_ctx, buf, count, stream):
            state["calls"].append(("all_reduce_max", int(count)))
            hip.hipSetDevice(int(device))
            if hip.hipStreamSynchronize(stream):
                return 1
            a = np.zeros(int(count), dtype=np.uint32)
            if hip.hipMemcpy(a.ctypes.data, buf, a.nbytes, D2H):
                return 1
            t = torch.from_numpy(a.astype(np.int64))
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            a = t.numpy().astype(np.uint32)
            return 1 if hip.hipMemcpy(buf, a.ctypes.data, a.nbytes, H2D) else 0

        # (the CFUNCTYPE objects must outlive the communicator: the library keeps the raw pointers)
        self._cb = (TR_GROUP(guarded(group_start)), TR_GROUP(guarded(group_end)), TR_XFER(guarded(xfer("send"))),
                    TR_XFER(guarded(xfer("recv"))), TR_REDUCE(guarded(all_reduce)))
        self._table = Transport(None, *self._cb)
        rc = self._L.dsl_comm_create_custom(self.nranks, self.rank, int(device), C.cast(C.byref(self._table), C.c_void_p),
                                            C.byref(self.ptr))
        if rc:
            raise DslError(f"dsl_comm_create_custom failed ({rc}): {self._L.dsl_comm_last_error().decode()}")


class SPHEngine:
    def __init__(self, params: Params, device: int = 0):
        self._L = load_library()
        self._h = C.c_void_p()
        rc = self._L.dsl_create(C.byref(params), int(device), C.byref(self._h))
        if rc:
            raise DslError(f"dsl_create failed ({rc}): {self._L.dsl_last_error(None).decode()}")
        self.device = device
        self.capacity = int(params.capacity) if params.capacity else int(params.n_particles)

    # -- plumbing -----------------------------------------------------------------
    def _ck(self, rc):
        if rc:
            raise DslError(f"libdslsph error {rc}: {self._L.dsl_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.dsl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def n(self) -> int:
        """live particle count (changes only in slab mode)"""
        n = C.c_int(0)
        self._ck(self._L.dsl_get_count(self._h, C.byref(n), None))
        return n.value

    def n_owned(self) -> int:
        n, o = C.c_int(0), C.c_int(0)
        self._ck(self._L.dsl_get_count(self._h, C.byref(n), C.byref(o)))
        return o.value

    @property
    def params(self) -> Params:
        p = Params()
        self._ck(self._L.dsl_get_params(self._h, C.byref(p)))
        return p

    def set_params(self, p: Params):
        self._ck(self._L.dsl_set_params(self._h, C.byref(p)))

    def set_stream(self, stream_ptr):
        """run on the given hipStream_t (0 / None = HIP's default stream)"""
        self._ck(self._L.dsl_set_stream(self._h, C.c_void_p(stream_ptr or None)))

    def use_own_stream(self):
        self._ck(self._L.dsl_use_own_stream(self._h))

    # -- buffers (ComputeGPU.PassFloatBuffer / ReadFloatBuffer) -------------------
    def upload(self, name: str, arr):
        b = BUF[name]
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1)
        self._ck(self._L.dsl_upload(self._h, b, _fp(a), a.size))

    @property
    def n_fluid(self) -> int:
        """N(): particles that are not boundary particles (slab mode: the live count)"""
        nb = int(self.params.n_boundary)
        return self.n - nb if nb > 0 else self.n

    def add_boundary_particles(self, positions):
        """ParticleArray.AddBoundaryParticles (model/particle_array.go:123-128)"""
        a = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1)
        self._ck(self._L.dsl_add_boundary_particles(self._h, _fp(a), a.size))

    def download(self, name: str, sorted_order: bool = False) -> np.ndarray:
        b = BUF[name]
        n = self.n if name == "positions" else self.n_fluid
        out = np.empty(n * _COMPS[b], dtype=np.float32)
        fn = self._L.dsl_download_sorted if sorted_order else self._L.dsl_download
        self._ck(fn(self._h, b, _fp(out), out.size))
        return out.reshape(n, 3) if _COMPS[b] == 3 else out

    def download_decimated(self, name: str, stride: int) -> np.ndarray:
        """every stride-th particle of positions/velocities (render hand-off)"""
        n = self.n
        out = np.empty(3 * ((n + stride - 1) // stride), dtype=np.float32)
        self._ck(self._L.dsl_download_decimated(self._h, BUF[name], int(stride), _fp(out), out.size))
        return out.reshape(-1, 3)

    def device_pointers(self, name: str):
        """(x_ptr, y_ptr, z_ptr), ids_ptr, n of the live SoA device arrays (slot order)"""
        xyz = (C.c_void_p * 3)()
        ids = C.c_void_p()
        n = C.c_int(0)
        self._ck(self._L.dsl_device_pointers(self._h, BUF[name], xyz, C.byref(ids), C.byref(n)))
        return (xyz[0], xyz[1], xyz[2]), ids.value, n.value

    def set_ids(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self._ck(self._L.dsl_set_ids(self._h, a.ctypes.data_as(C.POINTER(C.c_int32)), a.size))

    def reset_forces(self):
        self._ck(self._L.dsl_reset_forces(self._h))

    # -- multi-GPU slabs (device record buffers are raw pointers, e.g. tensor.data_ptr()) --
    def slab_config(self, axis: int, lo: float, hi: float):
        self._ck(self._L.dsl_slab_config(self._h, int(axis), C.c_float(lo), C.c_float(hi)))

    def slab_message_floats(self, cap_full: int, cap_xonly: int) -> int:
        return int(self._L.dsl_slab_message_floats_for(self._h, int(cap_full), int(cap_xonly)))

    def slab_record_floats(self) -> int:
        return int(self._L.dsl_slab_record_floats(self._h))

    def pcisph_phase(self, phase: int):
        """0 begin step, 1 iterate, 2 check, 3 end step (include/dslsph.h)"""
        self._ck(self._L.dsl_pcisph_phase(self._h, int(phase)))

    def pcisph_set_binning(self, mode: int):
        """-1 never, 0 automatic, 1 always: sort DensityF's query points into grid cells of their own (include/dslsph.h)"""
        self._ck(self._L.dsl_pcisph_set_binning(self._h, int(mode)))

    def pcisph_binning(self):
        """(mode, active): active = the next correction iteration sorts its queries"""
        m, a = C.c_int(0), C.c_int(0)
        self._ck(self._L.dsl_pcisph_get_binning(self._h, C.byref(m), C.byref(a), None))
        return m.value, bool(a.value)

    def pcisph_query_escaped(self) -> bool:
        """slab mode: has a DensityF query point left what the ghost band covers (include/dslsph.h)?  Blocking."""
        e = C.c_int(0)
        self._ck(self._L.dsl_pcisph_get_binning(self._h, None, None, C.byref(e)))
        return bool(e.value)

    def pcisph_error_word(self, dev_word: int, store: bool):
        self._ck(self._L.dsl_pcisph_error_word(self._h, C.c_void_p(dev_word), 1 if store else 0))

    def slab_split(self, width: float, margin: float):
        self._ck(self._L.dsl_slab_split(self._h, C.c_float(width), C.c_float(margin)))

    def slab_pack(self, width_full: float, width: float, dev_lo: int, dev_hi: int, cap_full: int, cap_xonly: int):
        self._ck(self._L.dsl_slab_pack(self._h, C.c_float(width_full), C.c_float(width), C.c_void_p(dev_lo or None),
                                       C.c_void_p(dev_hi or None), int(cap_full), int(cap_xonly)))

    def slab_pack_band(self, width_full: float, dev_lo: int, dev_hi: int, cap_full: int, cap_xonly: int,
                       stream: int = 0):
        """between force_pass_split(BAND) and (INNER): packs the integrated band on `stream`"""
        self._ck(self._L.dsl_slab_pack_band(self._h, C.c_float(width_full), C.c_void_p(dev_lo or None),
                                            C.c_void_p(dev_hi or None), int(cap_full), int(cap_xonly),
                                            C.c_void_p(stream or None)))

    def force_pass_split(self, phase: int):
        self._ck(self._L.dsl_force_pass_split(self._h, int(phase)))

    def slab_append(self, dev_msg: int, cap_full: int, cap_xonly: int):
        self._ck(self._L.dsl_slab_append(self._h, C.c_void_p(dev_msg), int(cap_full), int(cap_xonly)))

    def slab_append2(self, dev_msg_a: int, dev_msg_b: int, cap_full: int, cap_xonly: int):
        self._ck(self._L.dsl_slab_append2(self._h, C.c_void_p(dev_msg_a or None), C.c_void_p(dev_msg_b or None),
                                          int(cap_full), int(cap_xonly)))

    def slab_status(self, reset_high_water: bool = False):
        """(overflow, band_missed, high-water full, high-water position-only); blocking"""
        st = (C.c_int32 * 4)()
        self._ck(self._L.dsl_slab_status(self._h, st, 1 if reset_high_water else 0))
        return tuple(int(v) for v in st)

    def slab_overflow(self) -> int:
        v = C.c_int(0)
        self._ck(self._L.dsl_slab_overflow(self._h, C.byref(v)))
        return v.value

    # -- the exchange behind the C ABI (RCCL inside libdslsph.so) ---------------------------------
    def slab_attach(self, comm, lo_rank: int, hi_rank: int, width_full: float, width: float, cap_full: int,
                    cap_xonly: int, overlap: bool):
        """comm: a Comm (or None for a lone slab); neighbour ranks, -1 at a domain end"""
        self._comm = comm  # keep it alive as long as the link
        self._ck(self._L.dsl_slab_attach(self._h, comm.ptr if comm is not None else None, int(lo_rank), int(hi_rank),
                                         C.c_float(width_full), C.c_float(width), int(cap_full), int(cap_xonly),
                                         1 if overlap else 0))

    def slab_detach(self):
        self._ck(self._L.dsl_slab_detach(self._h))

    def slab_image_shift(self, from_lo: float, from_hi: float):
        self._ck(self._L.dsl_slab_image_shift(self._h, C.c_float(from_lo), C.c_float(from_hi)))

    def slab_exchange(self):
        self._ck(self._L.dsl_slab_exchange(self._h))

    def slab_replan(self):
        self._ck(self._L.dsl_slab_replan(self._h))

    def slab_wcsph_step(self, nsteps: int = 1):
        self._ck(self._L.dsl_slab_wcsph_step(self._h, int(nsteps)))

    def slab_pcisph_step(self, nsteps: int = 1):
        self._ck(self._L.dsl_slab_pcisph_step(self._h, int(nsteps)))

    def download_ids(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.int32)
        self._ck(self._L.dsl_download_ids(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size))
        return out

    def download_cell_start(self) -> np.ndarray:
        cells = self.stats().grid_cells
        out = np.empty(cells + 1, dtype=np.int32)
        self._ck(self._L.dsl_download_cell_start(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size))
        return out

    # -- lsh_ref parity mode (sampler/lsh/lsh.go) ---------------------------------------
    def set_hash_vectors(self, vectors):
        v = np.ascontiguousarray(vectors, dtype=np.float32).reshape(-1, 3)
        self._ck(self._L.dsl_set_hash_vectors(self._h, _fp(v), v.shape[0]))

    def lsh_table(self) -> np.ndarray:
        p = self.params
        out = np.empty(p.lsh_buckets * p.lsh_bucket_size, dtype=np.int32)
        self._ck(self._L.dsl_lsh_download_table(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size))
        return out

    # -- model/sph.SPH passes -----------------------------------------------------
    def nn(self):
        self._ck(self._L.dsl_build_neighbours(self._h))

    def density_all(self):
        self._ck(self._L.dsl_density_pass(self._h))

    def pressure_all(self):
        self._ck(self._L.dsl_pressure_pass(self._h))

    def viscous_all(self):
        self._ck(self._L.dsl_viscous_pass(self._h))

    def external_all(self, f):
        a = np.ascontiguousarray(f, dtype=np.float32)
        self._ck(self._L.dsl_external_pass(self._h, _fp(a)))

    def gradient_pressure_force(self):
        self._ck(self._L.dsl_gradient_pressure_pass(self._h))

    def update(self):
        self._ck(self._L.dsl_update_pass(self._h))

    def force_pass(self):
        self._ck(self._L.dsl_force_pass(self._h))

    # -- SPHField operators no solver calls (sph_field.go:124-135,203-294) ---------
    def field_div(self, tensor: str = "velocities") -> np.ndarray:
        out = np.empty(self.n_fluid, dtype=np.float32)
        self._ck(self._L.dsl_field_divergence(self._h, BUF[tensor], _fp(out), out.size))
        return out

    def field_curl(self, tensor: str = "velocities") -> np.ndarray:
        out = np.empty(self.n_fluid * 3, dtype=np.float32)
        self._ck(self._L.dsl_field_curl(self._h, BUF[tensor], _fp(out), out.size))
        return out.reshape(-1, 3)

    def field_laplacian(self, scalar: str = "densities") -> np.ndarray:
        out = np.empty(self.n_fluid, dtype=np.float32)
        self._ck(self._L.dsl_field_laplacian(self._h, BUF[scalar], _fp(out), out.size))
        return out

    def field_interpolate(self, positions, scalar: str = "densities") -> np.ndarray:
        q = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
        out = np.empty(q.shape[0], dtype=np.float32)
        self._ck(self._L.dsl_field_interpolate(self._h, BUF[scalar], _fp(q), q.shape[0], _fp(out)))
        return out

    # -- solver drivers -----------------------------------------------------------
    def wcsph_step(self, nsteps: int = 1):
        self._ck(self._L.dsl_wcsph_step(self._h, int(nsteps)))

    def pcisph_begin(self):
        self._ck(self._L.dsl_pcisph_begin(self._h))

    def pcisph_step(self, nsteps: int = 1):
        self._ck(self._L.dsl_pcisph_step(self._h, int(nsteps)))

    def sync(self):
        self._ck(self._L.dsl_sync(self._h))

    def stats(self) -> Stats:
        s = Stats()
        self._ck(self._L.dsl_get_stats(self._h, C.byref(s)))
        return s

    # -- timing ---------------------------------------------------------------------
    def timing_enable(self, on=True):
        """True / 1: every kernel; 2: only the step's dominant kernels; False / 0: off"""
        self._ck(self._L.dsl_timing_enable(self._h, int(on)))

    def timing_reset(self):
        self._ck(self._L.dsl_timing_reset(self._h))

    # library options (include/dslsph.h: DSL_OPT_*)
    OPTIONS = {"skin": 1, "skin_steps": 2, "skin_rebuilds": 3, "skin_list_overflow": 4, "skin_suspensions": 5, "device_bytes": 6, "skin_fields_own": 7, "skin_fields_padded": 8, "skin_predict": 9, "skin_tau_steps": 10,
               "density_pair": 16, "cell_keys": 17, "tile_box": 18, "persistent_blocks": 19, "pci_qtiled": 20,
               "pci_qpair": 21, "pci_qrows": 22, "pci_qincr": 23, "list_build": 24, "grid_oversub": 25, "tile_queue": 26}

    def set_option(self, name: str, value: float):
        self._ck(self._L.dsl_set_option(self._h, self.OPTIONS[name], float(value)))

    def get_option(self, name: str) -> float:
        v = C.c_double(0.0)
        self._ck(self._L.dsl_get_option(self._h, self.OPTIONS[name], C.byref(v)))
        return float(v.value)

    def timing(self, kernel: str):
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._ck(self._L.dsl_timing_get(self._h, KERNEL_IDS[kernel], C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value
