"""One-GPU estimate of a MIDDLE slab rank's step cost (tools only, not part of the product).

The rank's two neighbours are emulated by periodic images of the rank itself: the band it
packs for the lower neighbour comes back, shifted by the slab thickness, as the ghosts of the
upper neighbour and vice versa.  Everything a real middle rank does per step (pack, append,
ghost sort, ghost densities, split force pass) runs unchanged; only the RCCL transfer is
replaced by a device copy on the same stream.  Usage:
  python tools/slab_periodic_bench.py [--n3 252] [--world 8] [--steps 50] [--no-overlap]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist

from dieselfluid_amd import slab


class PeriodicDriver(slab.SlabDriver):
    def _shift(self, msg, dz):
        e = self.engine
        out = msg.clone()
        full = out[slab.RECORD:slab.RECORD * (1 + e.cap_full)].view(-1, slab.RECORD)
        xo = out[slab.RECORD * (1 + e.cap_full):slab.message_floats(e.cap_full, e.cap_x)].view(-1, slab.RECORD_X)
        full[:, self.axis] += dz
        xo[:, self.axis] += dz
        return out

    def _post(self, send):
        if self.use_nccl:
            return self._post_nccl(send)
        T = self.hi - self.lo
        # my lo band is what the (image) upper neighbour receives from below, and vice versa
        self._images = [self._shift(send[1], -T), self._shift(send[0], +T)]
        return None

    def _post_nccl(self, send):
        """real RCCL point-to-point calls, addressed to this very rank: the same batch_isend_irecv /
        Work.wait() / stream ordering as the multi-GPU driver, minus the wire"""
        n = self.engine.message_floats()
        if self._recv is None:
            full = slab.message_floats(self.engine.max_full, self.engine.max_x)
            self._recv = [torch.zeros(full, dtype=torch.float32, device=self.engine.dev) for _ in range(2)]
        ops = [dist.P2POp(dist.isend, send[1], 0), dist.P2POp(dist.irecv, self._recv[0][:n], 0),
               dist.P2POp(dist.isend, send[0], 0), dist.P2POp(dist.irecv, self._recv[1][:n], 0)]
        return dist.batch_isend_irecv(ops), n

    def _finish(self, posted):
        if self.use_nccl:
            works, n = posted
            for w in works:
                w.wait()
            T = self.hi - self.lo
            self._images = [self._shift(self._recv[0][:n], -T), self._shift(self._recv[1][:n], +T)]
            for m in self._images:
                self.engine.append(m)
            self._ghosts_in = True
            return
        if self.overlap:  # the main stream waits for the side stream's pack + copies
            torch.cuda.current_stream().wait_stream(self.engine.comm_stream)
        for m in self._images:
            self.engine.append(m)
        self._ghosts_in = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n3", type=int, default=252)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--no-exchange", action="store_true", help="same slab and grid, no ghosts at all")
    ap.add_argument("--no-timing", action="store_true")
    ap.add_argument("--null-stream", action="store_true", help="run on torch's default (null) stream")
    ap.add_argument("--no-replan", action="store_true", help="no message re-sizing (and so no host sync) in the timed loop")
    ap.add_argument("--dump", default="", help="write the owned state (sorted by id) to this .npz after the run")
    ap.add_argument("--nccl", action="store_true", help="send the bands through RCCL (to this same rank)")
    ap.add_argument("--native", action="store_true",
                    help="the library's own step driver (dsl_slab_wcsph_step): RCCL group send/recv to this same rank, "
                         "issued from C, periodic image shift in the append kernel")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="a library option (include/dslsph.h DSL_OPT_*)")
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    if a.nccl:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
    if not a.null_stream:
        torch.cuda.set_stream(torch.cuda.Stream())
    drv = PeriodicDriver.dambreak(a.n3, math_mode=1, device=0, rank=a.rank, world=a.world,
                                  overlap=not a.no_overlap, native=False)
    if a.native:
        from dieselfluid_amd.engine import Comm
        comm = Comm(1, 0, 0)
        drv.attach_native(comm, 0, 0)
        T = drv.hi - drv.lo
        drv.engine_core.slab_image_shift(-T, +T)
    drv.comm_dev = torch.device("cuda", 0) if a.nccl else torch.device("cpu")
    drv.use_nccl = a.nccl
    if a.no_replan:
        PeriodicDriver.REPLAN_EVERY = 10 ** 9
    if a.no_exchange:
        drv.world = 1
    eng = drv.engine_core
    for kv in a.opt:
        eng.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    drv.wcsph_step(a.warmup)
    eng.timing_reset()
    eng.timing_enable(not a.no_timing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    drv.wcsph_step(a.steps)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.timing_enable(False)
    st = drv.engine.status()
    stats = eng.stats()
    n_live, n_owned = eng.n, eng.n_owned()
    out = {"options": a.opt, "ms_per_step": round(dt / a.steps * 1e3, 4), "host_enqueue_ms_per_step": round(t_host / a.steps * 1e3, 4), "owned": n_owned, "live_with_ghosts": n_live,
           "overlap": drv.overlap, "driver": "native" if a.native else ("python+rccl" if a.nccl else "python+copy"), "max_cell_count": stats.max_cell_count, "grid": list(stats.grid_dims), "status": st, "caps": [drv.engine.cap_full, drv.engine.cap_x],
           "message_MB": round(drv.engine.message_floats() * 4 / 1e6, 3) if not a.native else None,
           "kernels_ms_per_step": {k: round(eng.timing(k)[0] * eng.timing(k)[1] / a.steps, 4) for k in
                                   ("cell_rank", "scan", "scatter", "tile_list", "density", "force_integrate")},
           "ideal_ms_at_1gpu_rate": None}
    print(json.dumps(out))
    if a.dump:
        import numpy as np
        ids, pos, vel = drv.engine.owned_state(drv.axis, drv.lo, drv.hi)
        o = np.argsort(ids)
        np.savez(a.dump, ids=ids[o], pos=pos[o], vel=vel[o])
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
