import functools, os, sys, numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.distributed as dist
import torch.multiprocessing as mp
from test_slab_cpu import _free_port, _vel_fn

def worker(rank, world, port, math_mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch; torch.cuda.set_device(0)
    from dieselfluid_amd.slab import SlabDriver, message_count
    drv = SlabDriver.dambreak(16, math_mode=math_mode, device=0, axis=2, vel_fn=functools.partial(_vel_fn, axis=2))
    e = drv.engine_core
    for step in range(4):
        drv.exchange()
        n_after_append = e.n
        rc = [message_count(m) if m is not None else -1 for m in drv._recv]
        drv.engine.nn()
        n_live = e.n
        pos = e.download("positions", sorted_order=True)
        fin0 = int(np.isfinite(pos).all(axis=1).sum())
        drv.engine.density_all()
        rho = e.download("densities", sorted_order=True)
        drv.engine.force_pass()
        pos = e.download("positions", sorted_order=True); vel = e.download("velocities", sorted_order=True)
        fin = np.isfinite(pos).all(axis=1)
        own = (pos[:, 2] >= drv.lo) & (pos[:, 2] < drv.hi)
        print(f"mode {math_mode} rank {rank} step {step}: recv {rc} n_append {n_after_append} live {n_live} finite_before {fin0} "
              f"rho_nan {int(np.isnan(rho).sum())} rho_min {np.nanmin(rho):.3g} finite_after {int(fin.sum())} owned_finite {int((fin&own).sum())} velnan {int(np.isnan(vel).any(axis=1).sum())} ovf {e.slab_overflow()}", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    for mode in (0, 1):
        mp.spawn(worker, args=(2, _free_port(), mode), nprocs=2, join=True)
