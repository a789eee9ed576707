"""GPU: two slab ranks (both on cuda:0, transport gloo) drive the HIP engine through the
dsl_slab_* C ABI; the assembled result must match a single-engine run of the same scene.
The RCCL transport itself cannot be exercised on a one-GPU box; the driver code path is
identical except for the tensors' device."""
import functools
import os

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers
from test_slab_cpu import _free_port, _vel_fn

pytestmark = pytest.mark.gpu

N3 = 16
STEPS = 8


def _worker(rank, world, port, math_mode, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    torch.cuda.set_device(0)
    from dieselfluid_amd.slab import SlabDriver
    drv = SlabDriver.dambreak(N3, math_mode=math_mode, device=0, axis=2,
                              vel_fn=functools.partial(_vel_fn, axis=2))
    drv.wcsph_step(STEPS)
    res = drv.gather_state(N3 ** 3)
    counts = [None] * world if rank == 0 else None
    dist.gather_object((drv.engine.n, drv.engine_core.n_owned(), drv.engine_core.slab_overflow()), counts, dst=0)
    if rank == 0:
        gp, gv, seen = res
        np.savez(out, pos=gp, vel=gv, seen=seen, counts=np.array(counts))
    dist.destroy_process_group()


@pytest.mark.parametrize("math_mode,tol_x,tol_v", [(0, 2e-6, 5e-5), (1, 1e-5, 1e-3)])
def test_two_hip_slabs_match_single_engine(tmp_path, math_mode, tol_x, tol_v):
    from dieselfluid_amd import SPHEngine, scenes
    out = str(tmp_path / "slab_gpu.npz")
    mp.spawn(_worker, args=(2, _free_port(), math_mode, out), nprocs=2, join=True)
    z = np.load(out)
    assert np.all(z["seen"] == 1)
    assert z["counts"][:, 0].sum() > N3 ** 3  # ghosts are present on both ranks
    assert np.all(z["counts"][:, 2] == 0)     # no message / capacity overflow
    p, pos = scenes.dambreak_scene(N3, math_mode=math_mode)
    eng = SPHEngine(p)
    eng.upload("positions", pos)
    eng.upload("velocities", _vel_fn(np.arange(N3 ** 3), pos, 2))
    eng.reset_forces()
    eng.wcsph_step(STEPS)
    assert helpers.rel_err(z["pos"], eng.download("positions")) < tol_x
    assert helpers.rel_err(z["vel"], eng.download("velocities")) < tol_v
