"""Spatial-slab decomposition of the particle domain, one process per GPU.

The reference has no multi-process path; this is the build's own
host logic above the C ABI (dsl_slab_* in include/dslsph.h).  torch.distributed is the
transport only (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

Per step and per neighbour ONE message: every owned particle within 2h of the slab plane.
Those within h (and the migrants beyond the plane) travel as full records (x, v, global id:
28 bytes; 52 with the PCISPH predictor state), the outer half of the band as positions + id
(16 bytes: the id keeps every cell in the single-domain order on every rank).  The receiver keeps full records inside its own [lo,hi) as newly owned and the
rest as ghosts.  A 2h band lets the receiver recompute the ghosts' densities itself, so the
force pass needs no second exchange: ghosts within h of the plane see their full
neighbourhood, and only those contribute to owned particles.

The driver is engine-agnostic: anything implementing pack/append/nn/density_all/
force_pass (HipSlabEngine here; an oracle-backed stand-in lives in tests/) can be driven,
which is how the N>1 logic is tested on CPU with gloo.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist

from . import scenes
from ._lib import DslError
from .engine import SPHEngine


class SlabOverflow(DslError):
    """a band message, the particle capacity or the split margin was outgrown on some rank: particles or
    ghosts were lost, the run is no longer valid"""

RECORD = 7    # full record: x,y,z,vx,vy,vz,id-bits
RECORD_X = 4  # position-only record: x,y,z,id-bits
SPLIT_BAND, SPLIT_INNER = 1, 2


RECORD_PCI = 13  # full record once the PCISPH predictor state travels along


def message_floats(cap_full: int, cap_x: int, record: int = RECORD) -> int:
    """header record + full records + position-only records (include/dslsph.h)"""
    return (cap_full + 1) * record + cap_x * RECORD_X


def message_counts(msg: torch.Tensor):
    """(full, position-only) record counts stored as int32 bits in the header"""
    hdr = msg[:2].cpu().contiguous().view(torch.int32)
    return int(hdr[0]), int(hdr[1])


class HipSlabEngine:
    """SPHEngine + device message buffers (torch tensors used as plain device memory)."""

    def __init__(self, params, device: int, cap_full: int, cap_x: int, max_scale: float = 2.0):
        self.eng = SPHEngine(params, device=device)
        self.dev = torch.device("cuda", device)
        self.max_full, self.max_x = int(max_scale * cap_full), int(max_scale * cap_x)
        self.cap_full, self.cap_x = int(cap_full), int(cap_x)
        self._send = None  # allocated at the first pack: the record size depends on dsl_pcisph_begin
        # kernels and NCCL ops are ordered through torch's current stream
        self.eng.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        self.supports_split = params.math_mode == 1
        self.appends_pairs = True
        self.comm_stream = torch.cuda.Stream(self.dev) if self.supports_split else None

    def set_caps(self, cap_full: int, cap_x: int):
        self.cap_full, self.cap_x = min(int(cap_full), self.max_full), min(int(cap_x), self.max_x)

    def message_floats(self) -> int:
        return message_floats(self.cap_full, self.cap_x, self.eng.slab_record_floats())

    def max_message_floats(self) -> int:
        return message_floats(self.max_full, self.max_x, self.eng.slab_record_floats())

    def _buffers(self):
        if self._send is None or self._send[0].numel() < self.max_message_floats():
            self._send = [torch.zeros(self.max_message_floats(), dtype=torch.float32, device=self.dev)
                          for _ in range(2)]
        return self._send

    def _views(self, want_lo, want_hi):
        n = self.message_floats()
        return (self._send[0][:n] if want_lo else None, self._send[1][:n] if want_hi else None)

    # -- protocol -------------------------------------------------------------------
    def pack(self, width_full: float, width: float, want_lo: bool, want_hi: bool):
        """Asynchronous: both band messages are filled on the device, counts included."""
        send = self._buffers()
        self.eng.slab_pack(width_full, width, send[0].data_ptr() if want_lo else 0,
                           send[1].data_ptr() if want_hi else 0, self.cap_full, self.cap_x)
        return self._views(want_lo, want_hi)

    def split(self, width: float, margin: float):
        self.eng.slab_split(width, margin)

    def force_band(self):
        self.eng.force_pass_split(SPLIT_BAND)

    def pack_band(self, width_full: float, want_lo: bool, want_hi: bool):
        """On the engine's own stream, between the two force launches: three small kernels that take
        25 us there.  On a side stream, under the interior launch, they were starved (two tile
        workgroups fill a CU's registers) and ate the whole window that is meant for the transfer."""
        send = self._buffers()
        self.eng.slab_pack_band(width_full, send[0].data_ptr() if want_lo else 0,
                                send[1].data_ptr() if want_hi else 0, self.cap_full, self.cap_x, 0)
        return self._views(want_lo, want_hi)

    def force_inner(self):
        self.eng.force_pass_split(SPLIT_INNER)

    def append(self, msg: torch.Tensor, msg2: torch.Tensor = None):
        """one or both neighbours' messages, one launch"""
        ms = [None if m is None else (m if m.device == self.dev else m.to(self.dev)).contiguous() for m in (msg, msg2)]
        # the engine runs on torch's current stream, so the caching allocator's stream-ordered
        # reuse keeps the messages' memory valid until the append kernel has run
        self.eng.slab_append2(ms[0].data_ptr() if ms[0] is not None else 0,
                              ms[1].data_ptr() if ms[1] is not None else 0, self.cap_full, self.cap_x)

    def status(self, reset_high_water: bool = False):
        return self.eng.slab_status(reset_high_water)

    def nn(self):
        self.eng.nn()

    def density_all(self):
        self.eng.density_all()

    def force_pass(self):
        self.eng.force_pass()

    def owned_state(self, axis, lo, hi):
        """Particles this rank is responsible for: finite positions inside [lo,hi).  After a
        step the old ghosts carry NaN, the freshly received ones and the particles that have
        just left (already handed over by the end-of-step exchange) lie outside [lo,hi)."""
        ids = self.eng.download_ids()
        pos = self.eng.download("positions", sorted_order=True)
        vel = self.eng.download("velocities", sorted_order=True)
        with np.errstate(invalid="ignore"):
            own = np.isfinite(pos).all(axis=1) & (pos[:, axis] >= np.float32(lo)) & (pos[:, axis] < np.float32(hi))
        return ids[own], pos[own], vel[own]

    @property
    def n(self):
        return self.eng.n


class SlabDriver:
    """Runs the WCSPH step on one slab and exchanges the 2h band with the two neighbours.

    The exchange for step t+1 happens at the END of step t.  With an engine that supports the
    split force pass the bands are packed as soon as the cell layers near the planes are
    integrated, and the RCCL send/recv runs on a side stream while the other layers are."""

    REPLAN_EVERY = 8  # steps between message-size re-plans (one host sync + one tiny all-reduce)

    def __init__(self, engine, rank: int, world: int, axis: int, planes, width: float, width_full: float,
                 group=None, overlap=None, margin: float = 0.0):
        assert len(planes) == world + 1
        self.engine, self.rank, self.world, self.axis, self.group = engine, rank, world, axis, group
        self.width, self.width_full = width, width_full
        self.lo = -math.inf if rank == 0 else float(planes[rank])
        self.hi = math.inf if rank == world - 1 else float(planes[rank + 1])
        self.backend = dist.get_backend(group) if world > 1 else "none"
        self.comm_dev = torch.device("cpu") if self.backend != "nccl" else getattr(engine, "dev", torch.device("cuda"))
        self.overlap = bool(getattr(engine, "supports_split", False) and world > 1) if overlap is None else overlap
        self.margin = margin
        self.steps = 0
        self._recv = None
        self._ghosts_in = False  # the ghosts for the next step are already appended

    # -- halo + migration exchange ----------------------------------------------------
    def _neighbours(self):
        return (self.rank - 1 if self.rank > 0 else None, self.rank + 1 if self.rank < self.world - 1 else None)

    def _post(self, send):
        """starts the transfer of one fixed-size message per neighbour and direction; the record
        counts travel in the message headers, so nothing here waits for the GPU"""
        nbs = self._neighbours()
        n = self.engine.message_floats()
        if self._recv is None or self._recv[0].numel() < n:
            full = (self.engine.max_message_floats() if hasattr(self.engine, "max_message_floats") else
                    message_floats(getattr(self.engine, "max_full", self.engine.cap_full),
                                   getattr(self.engine, "max_x", self.engine.cap_x)))
            self._recv = [torch.zeros(max(full, n), dtype=torch.float32, device=self.comm_dev) for _ in range(2)]
        sbuf = [None if s is None else (s if s.device == self.comm_dev else s.to(self.comm_dev)) for s in send]
        ops = []
        for k in range(2):
            if nbs[k] is not None:
                ops.append(dist.P2POp(dist.isend, sbuf[k], nbs[k], group=self.group))
                ops.append(dist.P2POp(dist.irecv, self._recv[k][:n], nbs[k], group=self.group))
        return (dist.batch_isend_irecv(ops) if ops else []), n, sbuf

    def _finish(self, posted):
        """with NCCL, Work.wait() only orders the current stream behind the transfer"""
        works, n, _keep = posted
        for w in works:
            w.wait()
        nbs = self._neighbours()
        got = [self._recv[k][:n] for k in range(2) if nbs[k] is not None]
        if len(got) == 2 and getattr(self.engine, "appends_pairs", False):
            self.engine.append(got[0], got[1])
        else:
            for m in got:
                self.engine.append(m)
        self._ghosts_in = True

    def exchange(self):
        if self.world == 1:
            return
        lo_nb, hi_nb = self._neighbours()
        send = self.engine.pack(self.width_full, self.width, lo_nb is not None, hi_nb is not None)
        self._finish(self._post(send))

    def _replan(self):
        """message sizes follow the band occupancy: every rank learns the largest counts seen
        anywhere since the last re-plan and all switch to the same new capacities"""
        st = self.engine.status(reset_high_water=True)
        hw = torch.tensor([st[0], st[1], st[2], st[3]], dtype=torch.int64, device=self.comm_dev)
        dist.all_reduce(hw, op=dist.ReduceOp.MAX, group=self.group)
        overflow, missed, hw_full, hw_x = (int(v) for v in hw.cpu())
        # every rank sees the same reduced words, so every rank raises (or none does)
        if overflow:
            raise SlabOverflow(f"slab exchange overflow: {overflow} records did not fit a band message or the particle "
                               "capacity on some rank; particles were lost (enlarge cap_full / cap_x / capacity)")
        if missed:
            raise SlabOverflow("split slab step: a particle outran the margin on some rank; ghosts were missed")
        want_full, want_x = int(1.15 * hw_full) + 1024, int(1.15 * hw_x) + 1024
        max_full, max_x = getattr(self.engine, "max_full", want_full), getattr(self.engine, "max_x", want_x)
        if hw_full > max_full or hw_x > max_x:
            raise SlabOverflow(f"band occupancy ({hw_full} full, {hw_x} position-only records) outgrew the message "
                               f"buffers ({max_full}, {max_x})")
        self.engine.set_caps(want_full, want_x)

    # -- the exchange behind the C ABI (dsl_slab_attach & co.: RCCL inside libdslsph.so) --------
    def attach_native(self, comm=None, lo_rank=None, hi_rank=None):
        """Hands the whole slab step to the library: the RCCL group send/recv, the split step, the
        re-plan.  This driver is then a thin caller (wcsph_step / pcisph_step are one C call each)."""
        from .engine import Comm, HostStagedComm
        core = self.engine_core
        if comm is None and self.world > 1 and self.backend != "nccl":
            # no RCCL between these ranks (gloo: several ranks on one device, CPU rehearsals): the library drives
            # the very same step, its transport calls go through the host's table (dsl_comm_create_custom)
            comm = HostStagedComm(self.world, self.rank, core.device, group=self.group)
        elif comm is None and self.world > 1:
            def bcast(raw):
                objs = [raw]
                dist.broadcast_object_list(objs, src=0, group=self.group)
                return objs[0]
            comm = Comm(self.world, self.rank, core.device, bcast)
        lo_nb, hi_nb = self._neighbours()
        if lo_rank is not None or hi_rank is not None:  # (tests: a rank that is its own periodic neighbour)
            lo_nb, hi_nb = lo_rank, hi_rank
        core.use_own_stream()  # the library orders its own streams; nothing of torch's is involved any more
        core.slab_attach(comm, -1 if lo_nb is None else lo_nb, -1 if hi_nb is None else hi_nb, self.width_full,
                         self.width, self.engine.cap_full, self.engine.cap_x, self.overlap)
        self.native = True
        self.native_comm = comm
        return comm

    def wcsph_step(self, nsteps: int = 1):
        if getattr(self, "native", False):
            try:
                self.engine_core.slab_wcsph_step(nsteps)
            except DslError as e:
                raise SlabOverflow(str(e)) if "overflow" in str(e) or "margin" in str(e) else e
            self.steps += nsteps
            return
        e = self.engine
        lo_nb, hi_nb = self._neighbours()
        for _ in range(nsteps):
            if not self._ghosts_in:
                self.exchange()      # first step: migrants + 2h ghosts from both neighbours
            e.nn()                   # counting sort; drops the previous step's ghosts
            e.density_all()          # owned + ghosts
            self._ghosts_in = False
            if self.world == 1:
                e.force_pass()
            elif self.overlap:
                e.force_band()       # owned particles of the cell layers near the planes
                send = e.pack_band(self.width_full, lo_nb is not None, hi_nb is not None)
                e.comm_stream.wait_stream(torch.cuda.current_stream(e.dev))  # ... the pack, not what follows
                with torch.cuda.stream(e.comm_stream):
                    posted = self._post(send)
                e.force_inner()      # the rest, concurrently with the transfer
                self._finish(posted)
            else:
                e.force_pass()       # owned only; ghosts are marked for removal
                self.exchange()
            self.steps += 1
            if self.world > 1 and self.steps % self.REPLAN_EVERY == 0:
                self._replan()

    def pcisph_step(self, nsteps: int = 1):
        """PCISPH across slabs.  DensityF uses the neighbours' CURRENT positions and the ghosts'
        densities are recomputed locally, so the correction iterations need no halo exchange; what
        has to be global is the iteration's max density error (the early-out of
        pcisph_darwin.go:95-98), one 4-byte MAX all-reduce per iteration.  Migrants carry their
        predictor state in the message (13-float records)."""
        if getattr(self, "native", False):
            try:
                self.engine_core.slab_pcisph_step(nsteps)
            except DslError as e:
                raise SlabOverflow(str(e)) if "overflow" in str(e) or "margin" in str(e) else e
            self.steps += nsteps
            return
        e = self.engine
        core = self.engine_core
        iters = int(core.params.pci_max_iters)
        if getattr(self, "_err", None) is None:
            self._err = torch.zeros(1, dtype=torch.int32, device=e.dev)
        for _ in range(nsteps):
            if not self._ghosts_in:
                self.exchange()
            self._ghosts_in = False
            core.pcisph_phase(0)            # NN, DensityAll, ViscousAll
            for _it in range(iters):
                core.pcisph_phase(1)        # predict, DensityF, gradient force
                if self.world > 1:
                    core.pcisph_error_word(self._err.data_ptr(), store=False)
                    if self.comm_dev.type == "cuda":
                        dist.all_reduce(self._err, op=dist.ReduceOp.MAX, group=self.group)
                    else:
                        t = self._err.cpu()
                        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                        self._err.copy_(t)
                    core.pcisph_error_word(self._err.data_ptr(), store=True)
                core.pcisph_phase(2)        # convergence check
            core.pcisph_phase(3)            # Update; ghosts are marked for removal
            self.exchange()
            self.steps += 1
            if self.world > 1 and self.steps % self.REPLAN_EVERY == 0:
                self._replan()

    # -- validation helper ---------------------------------------------------------------
    def gather_state(self, n_total: int):
        """Owned particles of every rank assembled by global id on rank 0 (tests only)."""
        ids, pos, vel = self.engine.owned_state(self.axis, self.lo, self.hi)
        if self.world == 1:
            parts = [(ids, pos, vel)]
        else:
            parts = [None] * self.world if self.rank == 0 else None
            dist.gather_object((ids, pos, vel), parts, dst=0, group=self.group)
        if self.rank != 0:
            return None
        gp = np.full((n_total, 3), np.nan, dtype=np.float32)
        gv = np.full((n_total, 3), np.nan, dtype=np.float32)
        seen = np.zeros(n_total, dtype=np.int32)
        for i, p, v in parts:
            gp[i], gv[i] = p, v
            seen[i] += 1
        return gp, gv, seen

    # -- the bench / test scene ---------------------------------------------------------------
    @classmethod
    def dambreak(cls, n3: int, math_mode: int = 1, device: int = 0, axis: int = 2, rank=None, world=None,
                 engine_factory=None, group=None, vel_fn=None, overlap=None, tile_align=True, params_hook=None,
                 pcisph=False, native=None, **scene_kw):
        """Dam-break of n3^3 particles split into `world` slabs along `axis` (default z: the
        collapse is symmetric in z, so the slabs stay balanced without re-planning)."""
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        p, _ = scenes.dambreak_scene(n3, math_mode=math_mode, positions=False, **scene_kw)
        if params_hook is not None:
            params_hook(p)
        L = p.box_max[2]
        dx = L / n3
        h = p.h
        width = 2.0 * h
        # Split the n3 lattice layers along the axis; planes sit between layers.  Slab thicknesses
        # are multiples of 4 grid cells (8 layers at h = 2dx) where that costs little balance, so
        # that both planes of a rank fall on the planes of its 4-cell tiles (see below).
        per = max(1, round(4.0 * h / dx))
        layer = [round(r * n3 / world) for r in range(world + 1)]
        if tile_align and n3 >= 2 * per * world:
            layer = [min(n3, per * round(r * n3 / (world * per))) for r in range(world)] + [n3]
        planes = [l * dx for l in layer]
        k0, k1 = layer[rank], layer[rank + 1]
        ids = scenes.dambreak_slab_ids(n3, axis, k0, k1)
        pos = scenes.dambreak_positions_ids(n3, dx, ids, scene_kw.get("jitter", 0.05), scene_kw.get("seed", 1234))
        n_local = ids.shape[0]
        width_full = h
        cap_full = int(1.25 * n3 * n3 * math.ceil(width_full / dx)) + 1024
        cap_x = int(1.25 * n3 * n3 * math.ceil((width - width_full) / dx)) + 1024
        max_scale = 2.0
        p.n_particles = n_local
        p.capacity = int(1.25 * n_local) + int(2 * max_scale * (cap_full + cap_x)) + 1024
        # The neighbour grid only has to cover this slab plus its ghost band.  It starts a whole
        # number of 4-cell tiles below the lower plane (two empty cell layers, then the two ghost
        # layers), so tile planes coincide with the slab planes: the force pass stages no tile that
        # is half ghosts, and the band layers of the split step (the cells within width + margin =
        # 4 cells of a plane, include/dslsph.h) are whole tile layers.
        margin = 2.0 * h
        if world > 1:
            tile = 4.0 * h
            if rank > 0:
                gmin = planes[rank] - tile
            else:  # first rank: align with its upper plane instead
                gmin = planes[1] - tile * math.ceil((planes[1] - p.grid_min[axis]) / tile - 1e-6)
            gmax = (planes[rank + 1] + width) if rank < world - 1 else p.grid_max[axis]
            p.grid_min[axis] = gmin
            p.grid_max[axis] = min(gmax, p.grid_max[axis])
        engine = (engine_factory or HipSlabEngine)(p, device, cap_full, cap_x)
        eng = engine.eng if hasattr(engine, "eng") else engine
        eng.upload("positions", pos)
        if vel_fn is not None:
            eng.upload("velocities", np.ascontiguousarray(vel_fn(ids, pos), dtype=np.float32))
        eng.set_ids(ids)
        eng.reset_forces()
        if pcisph:
            eng.pcisph_begin()  # predictor state = the initial positions / velocities (pcisph_darwin.go:28-41)
            overlap = False     # the PCISPH step has no split force pass
        drv = cls(engine, rank, world, axis, planes, width, width_full, group=group, overlap=overlap, margin=margin)
        eng.slab_config(axis, drv.lo, drv.hi)
        if drv.overlap:
            engine.split(width, margin)
        drv.params = p
        drv.n_total = n3 ** 3
        # real GPUs, one per rank: the library drives the step, RCCL included.  (gloo rehearsals and the
        # CPU logic tests keep the Python protocol above: RCCL refuses several ranks on one device.)
        if native is None:
            native = drv.backend == "nccl" and engine_factory is None
        if native:
            drv.attach_native()
        return drv

    @property
    def engine_core(self):
        return self.engine.eng if hasattr(self.engine, "eng") else self.engine
