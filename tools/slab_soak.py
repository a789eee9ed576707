"""Soak test of the slab step (tools only): a middle rank with periodic-image neighbours for
thousands of steps; prints particle conservation and the slab status words as the flow develops."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
from slab_periodic_bench import PeriodicDriver

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 126
total = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 500
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29578")
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)
torch.cuda.set_stream(torch.cuda.Stream())
drv = PeriodicDriver.dambreak(n3, math_mode=1, device=0, rank=1, world=4, overlap=True)
drv.comm_dev = torch.device("cpu")
drv.use_nccl = False
eng = drv.engine_core
n0 = eng.n_owned()
done = 0
while done < total:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    drv.wcsph_step(chunk)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done += chunk
    st = drv.engine.status()
    s = eng.stats()
    print(json.dumps({"steps": done, "ms_per_step": round(dt / chunk * 1e3, 4), "owned": eng.n_owned(), "owned0": n0,
                      "live": eng.n, "overflow": st[0], "band_missed": st[1], "hw": [st[2], st[3]],
                      "caps": [drv.engine.cap_full, drv.engine.cap_x], "max_vel": round(s.max_vel, 3),
                      "max_cell": s.max_cell_count}), flush=True)
dist.destroy_process_group()
