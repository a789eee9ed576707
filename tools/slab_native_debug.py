import os, sys
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from slab_periodic_bench import PeriodicDriver
from dieselfluid_amd.engine import Comm
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29593")
dist.init_process_group("gloo", rank=0, world_size=1)
for steps in (1, 2, 5, 9):
    res = {}
    for kind in ("python", "native"):
        drv = PeriodicDriver.dambreak(64, math_mode=1, device=0, rank=1, world=4, overlap=False, native=False)
        drv.comm_dev, drv.use_nccl = torch.device("cpu"), False
        if kind == "native":
            comm = Comm(1, 0, 0)
            drv.attach_native(comm, 0, 0)
            T = drv.hi - drv.lo
            drv.engine_core.slab_image_shift(-T, +T)
        drv.wcsph_step(steps)
        torch.cuda.synchronize()
        n_live, n_owned = drv.engine_core.n, drv.engine_core.n_owned()
        ids, pos, vel = drv.engine.owned_state(drv.axis, drv.lo, drv.hi)
        o = np.argsort(ids)
        res[kind] = (ids[o], pos[o], vel[o], n_live, n_owned, drv.engine.status())
        drv.engine_core.close()
    a, b = res["python"], res["native"]
    d = np.abs(a[1].astype(np.float64) - b[1]).max(axis=1)
    bad = np.nonzero(d > 0)[0]
    print(f"steps {steps}: live {a[3]} vs {b[3]} owned {a[4]} vs {b[4]} status {a[5]} {b[5]} differing {bad.size} max {d.max():.3e}",
          "z of differing:", np.unique(np.round(a[1][bad][:, 2], 2))[:12] if bad.size else "")
dist.destroy_process_group()
