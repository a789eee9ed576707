"""Oracle-backed stand-in for HipSlabEngine (TEST INFRASTRUCTURE): implements the engine
protocol SlabDriver drives (pack / append / nn / density_all / force_pass / owned_state and
the set-up calls) with numpy + the CPU oracle, so the slab logic -- band selection,
migration, ghost classification, global-id bookkeeping -- runs on CPU under gloo."""
import numpy as np
import torch

import helpers
from oracle import pyoracle as po

RECORD, RECORD_X = 7, 4


class OracleSlabEngine:
    supports_split = False

    def __init__(self, params, device, cap_full, cap_x):
        self.p = params
        self.q = helpers.oracle_params(params)
        self.capacity = params.capacity or params.n_particles
        self.cap_full, self.cap_x = int(cap_full), int(cap_x)
        self.max_full, self.max_x = 2 * self.cap_full, 2 * self.cap_x
        self.hw = [0, 0]
        self.pos = np.zeros((0, 3), np.float32)
        self.vel = np.zeros((0, 3), np.float32)
        self.ids = np.zeros((0,), np.int32)
        self.axis, self.lo, self.hi = -1, -np.inf, np.inf
        self.keep_from = 0
        self._sys = None

    def set_caps(self, cap_full, cap_x):
        self.cap_full, self.cap_x = min(int(cap_full), self.max_full), min(int(cap_x), self.max_x)

    def message_floats(self):
        return (self.cap_full + 1) * RECORD + self.cap_x * RECORD_X

    def status(self, reset_high_water=False):
        st = (0, 0, self.hw[0], self.hw[1])
        if reset_high_water:
            self.hw = [0, 0]
        return st

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
    def _message(self, full, xonly):
        """header (counts as int32 bits) + cap_full full records + cap_x position-only records"""
        nf, nx = int(full.sum()), int(xonly.sum())
        assert nf <= self.cap_full and nx <= self.cap_x
        self.hw = [max(self.hw[0], nf), max(self.hw[1], nx)]
        msg = np.zeros(self.message_floats(), np.float32)
        msg[0:2] = np.array([nf, nx], np.int32).view(np.float32)
        rec = msg[RECORD:RECORD * (1 + self.cap_full)].reshape(-1, RECORD)
        rec[:nf, 0:3], rec[:nf, 3:6] = self.pos[full], self.vel[full]
        rec[:nf, 6] = self.ids[full].view(np.float32)
        xrec = msg[RECORD * (1 + self.cap_full):].reshape(-1, RECORD_X)
        xrec[:nx, 0:3] = self.pos[xonly]
        xrec[:nx, 3] = self.ids[xonly].view(np.float32)
        return torch.from_numpy(msg)

    def pack(self, width_full, width, want_lo, want_hi):
        a = self.pos[:, self.axis]
        finite = np.isfinite(self.pos).all(axis=1)
        with np.errstate(invalid="ignore"):
            lo_f = finite & (a < np.float32(self.lo + width_full))
            lo_x = finite & ~lo_f & (a < np.float32(self.lo + width))
            hi_f = finite & (a >= np.float32(self.hi - width_full))
            hi_x = finite & ~hi_f & (a >= np.float32(self.hi - width))
        return (self._message(lo_f, lo_x) if want_lo else None, self._message(hi_f, hi_x) if want_hi else None)

    def append(self, msg):
        m = msg.cpu().numpy()
        nf, nx = (int(v) for v in m[0:2].view(np.int32))
        if nf + nx == 0:
            return
        assert self.pos.shape[0] + nf + nx <= self.capacity
        r = m[RECORD:RECORD * (1 + self.cap_full)].reshape(-1, RECORD)[:nf]
        x = m[RECORD * (1 + self.cap_full):].reshape(-1, RECORD_X)[:nx]
        self.pos = np.concatenate([self.pos, r[:, 0:3], x[:, 0:3]])
        self.vel = np.concatenate([self.vel, r[:, 3:6], np.zeros_like(x[:, 0:3])])
        self.ids = np.concatenate([self.ids, np.ascontiguousarray(r[:, 6]).view(np.int32),
                                   np.ascontiguousarray(x[:, 3]).view(np.int32)])

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
        # old ghosts carry NaN after a step; fresh ghosts and departed particles lie outside [lo,hi)
        with np.errstate(invalid="ignore"):
            own = np.isfinite(self.pos).all(axis=1) & (self.pos[:, axis] >= np.float32(lo)) & \
                (self.pos[:, axis] < np.float32(hi))
        return self.ids[own], self.pos[own], self.vel[own]

    @property
    def n(self):
        return self.pos.shape[0]
