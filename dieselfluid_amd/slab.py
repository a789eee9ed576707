"""Spatial-slab decomposition of the particle domain, one process per GPU.

The reference has no multi-process path; this is the build's own
host logic above the C ABI (dsl_slab_* in include/dslsph.h).  torch.distributed is the
transport only (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).

Per step and per neighbour ONE message: every owned particle within 2h of the slab plane
(migrants that crossed the plane included), 28 bytes each (x, v, global id).  The
receiver keeps those inside its own [lo,hi) as newly owned and the rest as ghosts.  A 2h
band lets the receiver recompute the ghosts' densities itself, so the force pass needs no
second exchange: ghosts within h of the plane see their full neighbourhood, and only
those contribute to owned particles.

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
from .engine import SPHEngine

RECORD = 7  # x,y,z,vx,vy,vz,id-bits


def message_count(msg: torch.Tensor) -> int:
    """record count stored (as int32 bits) in the header record of a message"""
    return int(msg[0, :1].cpu().contiguous().view(torch.int32).item())


class HipSlabEngine:
    """SPHEngine + device message buffers (torch tensors used as plain device memory).
    A message is (capacity+1) x 7 floats: header record (count) + records."""

    def __init__(self, params, device: int, band_capacity: int):
        self.eng = SPHEngine(params, device=device)
        self.dev = torch.device("cuda", device)
        self.band_capacity = int(band_capacity)
        self._send = [torch.zeros((self.band_capacity + 1, RECORD), dtype=torch.float32, device=self.dev)
                      for _ in range(2)]
        # kernels and NCCL ops are ordered through torch's current stream
        self.eng.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)

    # -- protocol -------------------------------------------------------------------
    def pack(self, width: float, want_lo: bool, want_hi: bool):
        """Asynchronous: both band messages are filled on the device, counts included."""
        self.eng.slab_pack(width, self._send[0].data_ptr() if want_lo else 0,
                           self._send[1].data_ptr() if want_hi else 0, self.band_capacity)
        return (self._send[0] if want_lo else None, self._send[1] if want_hi else None)

    def append(self, msg: torch.Tensor):
        m = msg if msg.device == self.dev else msg.to(self.dev)
        # the engine runs on torch's current stream, so the caching allocator's stream-ordered
        # reuse keeps m's memory valid until the append kernel has run
        self.eng.slab_append(m.contiguous().data_ptr(), self.band_capacity)

    def nn(self):
        self.eng.nn()

    def density_all(self):
        self.eng.density_all()

    def force_pass(self):
        self.eng.force_pass()

    def owned_state(self, axis, lo, hi):
        """Particles this rank is responsible for between steps: after a step the ghosts
        carry NaN positions, so everything finite is owned or has just crossed a plane
        (and will be handed over at the next exchange)."""
        ids = self.eng.download_ids()
        pos = self.eng.download("positions", sorted_order=True)
        vel = self.eng.download("velocities", sorted_order=True)
        own = np.isfinite(pos).all(axis=1)
        return ids[own], pos[own], vel[own]

    @property
    def n(self):
        return self.eng.n


class SlabDriver:
    """Runs the WCSPH step on one slab and exchanges the 2h band with the two neighbours."""

    def __init__(self, engine, rank: int, world: int, axis: int, planes, width: float, group=None):
        assert len(planes) == world + 1
        self.engine, self.rank, self.world, self.axis, self.width, self.group = engine, rank, world, axis, width, group
        self.lo = -math.inf if rank == 0 else float(planes[rank])
        self.hi = math.inf if rank == world - 1 else float(planes[rank + 1])
        self.backend = dist.get_backend(group) if world > 1 else "none"
        self.comm_dev = torch.device("cpu") if self.backend != "nccl" else getattr(engine, "dev", torch.device("cuda"))
        self.steps = 0
        self._recv = None

    # -- halo + migration exchange ----------------------------------------------------
    def _neighbours(self):
        return (self.rank - 1 if self.rank > 0 else None, self.rank + 1 if self.rank < self.world - 1 else None)

    def exchange(self):
        """One fixed-size message per neighbour and direction; the record count travels in the
        message header, so nothing here waits for the GPU (with NCCL, Work.wait() only orders
        the current stream behind the transfer)."""
        if self.world == 1:
            return
        lo_nb, hi_nb = self._neighbours()
        send = self.engine.pack(self.width, lo_nb is not None, hi_nb is not None)
        nbs = [lo_nb, hi_nb]
        if self._recv is None:
            cap = self.engine.band_capacity
            self._recv = [torch.zeros((cap + 1, RECORD), dtype=torch.float32, device=self.comm_dev) for _ in range(2)]
        sbuf = [None if s is None else (s if s.device == self.comm_dev else s.to(self.comm_dev)) for s in send]
        ops = []
        for k in range(2):
            if nbs[k] is not None:
                ops.append(dist.P2POp(dist.isend, sbuf[k], nbs[k], group=self.group))
                ops.append(dist.P2POp(dist.irecv, self._recv[k], nbs[k], group=self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for k in range(2):
            if nbs[k] is not None:
                self.engine.append(self._recv[k])

    def wcsph_step(self, nsteps: int = 1):
        for _ in range(nsteps):
            self.exchange()             # migrants + 2h ghosts from both neighbours
            self.engine.nn()            # counting sort; drops the previous step's ghosts
            self.engine.density_all()   # owned + ghosts
            self.engine.force_pass()    # owned only; ghosts are marked for removal
            self.steps += 1

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
                 engine_factory=None, group=None, vel_fn=None, **scene_kw):
        """Dam-break of n3^3 particles split into `world` slabs along `axis` (default z: the
        collapse is symmetric in z, so the slabs stay balanced without re-planning)."""
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
        p, _ = scenes.dambreak_scene(n3, math_mode=math_mode, positions=False, **scene_kw)
        L = p.box_max[2]
        dx = L / n3
        h = p.h
        width = 2.0 * h
        # split the n3 lattice layers along the axis evenly; planes sit between layers
        layer = [round(r * n3 / world) for r in range(world + 1)]
        planes = [l * dx for l in layer]
        k0, k1 = layer[rank], layer[rank + 1]
        ids = scenes.dambreak_slab_ids(n3, axis, k0, k1)
        pos = scenes.dambreak_positions_ids(n3, dx, ids, scene_kw.get("jitter", 0.05), scene_kw.get("seed", 1234))
        n_local = ids.shape[0]
        band = int(1.5 * n3 * n3 * math.ceil(width / dx)) + 1024
        p.n_particles = n_local
        p.capacity = int(1.25 * n_local) + 2 * band + 1024
        # the neighbour grid only has to cover this slab plus its ghost band
        if world > 1:
            gmin = (planes[rank] - width - h) if rank > 0 else p.grid_min[axis]
            gmax = (planes[rank + 1] + width + h) if rank < world - 1 else p.grid_max[axis]
            p.grid_min[axis] = max(gmin, p.grid_min[axis])
            p.grid_max[axis] = min(gmax, p.grid_max[axis])
        engine = (engine_factory or HipSlabEngine)(p, device, band)
        eng = engine.eng if hasattr(engine, "eng") else engine
        eng.upload("positions", pos)
        if vel_fn is not None:
            eng.upload("velocities", np.ascontiguousarray(vel_fn(ids, pos), dtype=np.float32))
        eng.set_ids(ids)
        eng.reset_forces()
        drv = cls(engine, rank, world, axis, planes, width, group=group)
        eng.slab_config(axis, drv.lo, drv.hi)
        drv.params = p
        drv.n_total = n3 ** 3
        return drv

    @property
    def engine_core(self):
        return self.engine.eng if hasattr(self.engine, "eng") else self.engine
