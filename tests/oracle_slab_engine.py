"""Oracle-backed stand-in for HipSlabEngine (TEST INFRASTRUCTURE): implements the engine
protocol SlabDriver drives (pack / append / nn / density_all / force_pass / owned_state and
the set-up calls) with numpy + the CPU oracle, so the slab logic -- band selection,
migration, ghost classification, global-id bookkeeping -- runs on CPU under gloo."""
import numpy as np
import torch

import helpers
from oracle import pyoracle as po

RECORD = 7


class OracleSlabEngine:
    def __init__(self, params, device, band_capacity):
        self.p = params
        self.q = helpers.oracle_params(params)
        self.capacity = params.capacity or params.n_particles
        self.band_capacity = band_capacity
        self.pos = np.zeros((0, 3), np.float32)
        self.vel = np.zeros((0, 3), np.float32)
        self.ids = np.zeros((0,), np.int32)
        self.axis, self.lo, self.hi = -1, -np.inf, np.inf
        self.keep_from = 0
        self._sys = None

    # -- set-up (mirrors SPHEngine) -------------------------------------------------
    def upload(self, name, arr):
        a = np.ascontiguousarray(arr, np.float32).reshape(-1, 3)
        if name == "positions":
            self.pos = a.copy()
            if self.vel.shape[0] != a.shape[0]:
                self.vel = np.zeros_like(a)
            self.ids = np.arange(a.shape[0], dtype=np.int32)
            self.keep_from = a.shape[0]
        elif name == "velocities":
            self.vel = a.copy()
        else:
            raise KeyError(name)

    def set_ids(self, ids):
        self.ids = np.ascontiguousarray(ids, np.int32).copy()

    def reset_forces(self):
        pass

    def slab_config(self, axis, lo, hi):
        self.axis, self.lo, self.hi = axis, lo, hi

    def _owned(self, pos):
        with np.errstate(invalid="ignore"):
            return (pos[:, self.axis] >= self.lo) & (pos[:, self.axis] < self.hi)

    # -- protocol ---------------------------------------------------------------------
    def _message(self, take):
        """(capacity+1) x 7 message: header record holds the count as int32 bits"""
        n = int(take.sum())
        assert n <= self.band_capacity
        msg = np.zeros((self.band_capacity + 1, RECORD), np.float32)
        msg[0, 0] = np.array([n], np.int32).view(np.float32)[0]
        msg[1:n + 1, 0:3], msg[1:n + 1, 3:6] = self.pos[take], self.vel[take]
        msg[1:n + 1, 6] = self.ids[take].view(np.float32)
        return torch.from_numpy(msg)

    def pack(self, width, want_lo, want_hi):
        a = self.pos[:, self.axis]
        finite = np.isfinite(self.pos).all(axis=1)
        with np.errstate(invalid="ignore"):
            lo = finite & (a < np.float32(self.lo + width))
            hi = finite & (a >= np.float32(self.hi - width))
        return (self._message(lo) if want_lo else None, self._message(hi) if want_hi else None)

    def append(self, msg):
        m = msg.cpu().numpy()
        n = int(m[0, :1].view(np.int32)[0])
        if n == 0:
            return
        r = m[1:n + 1]
        assert self.pos.shape[0] + n <= self.capacity
        self.pos = np.concatenate([self.pos, r[:, 0:3]])
        self.vel = np.concatenate([self.vel, r[:, 3:6]])
        self.ids = np.concatenate([self.ids, np.ascontiguousarray(r[:, 6]).view(np.int32)])

    def nn(self):
        # stale ghosts carry NaN; a particle that just crossed the plane stays one more
        # step as a ghost (the neighbour packed its band before receiving it)
        keep = np.isfinite(self.pos).all(axis=1)
        self.pos, self.vel, self.ids = self.pos[keep], self.vel[keep], self.ids[keep]

    def density_all(self):
        frc = np.tile(np.array(self.p.force_reset[:], np.float32), (self.pos.shape[0], 1))
        self._sys = po.OracleSPH.from_state(self.q, self.pos, vel=self.vel, force=frc)

    def force_pass(self):
        s = self._sys
        s.wcsph_step(1)  # sampler rebuild, D, [G], [V], X, PR, U -- ghosts are integrated too, then discarded
        own = self._owned(self.pos)
        newp, newv = s.positions(), s.velocities()
        self.pos = np.where(own[:, None], newp, np.float32(np.nan)).astype(np.float32)
        self.vel = np.where(own[:, None], newv, self.vel).astype(np.float32)
        self._sys = None

    def owned_state(self, axis, lo, hi):
        own = np.isfinite(self.pos).all(axis=1)  # ghosts carry NaN after a step
        return self.ids[own], self.pos[own], self.vel[own]

    @property
    def n(self):
        return self.pos.shape[0]
