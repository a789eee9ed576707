"""Diagnostic (GPU box): where do EXACT slab ranks and the single engine part?  Runs the 2- or 3-rank EXACT case
of tests/test_gpu_slab.py for 1, 2, ... steps and prints the particles whose bits differ, with their distance to
the slab planes and to the nearest cell plane of either grid.
  python tools/slab_exact_diff.py [world] [n3] [max_steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    n3 = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    max_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    import torch.multiprocessing as mp
    for steps in range(1, max_steps + 1):
        os.environ["DSL_SLAB_TEST_STEPS"] = str(steps)
        for m in ("test_gpu_slab", "test_slab_cpu"):
            sys.modules.pop(m, None)
        import test_gpu_slab as T
        out = f"/tmp/slab_exact_{steps}.npz"
        mp.spawn(T._worker, args=(world, T._free_port(), 0, n3, None, 1.0, out), nprocs=world, join=True)
        z = np.load(out)
        pos, vel = T._single(n3, 0, 1.0, steps=steps)
        dx_ = z["pos"].view(np.uint32) != pos.view(np.uint32)
        dv_ = z["vel"].view(np.uint32) != vel.view(np.uint32)
        bad = np.nonzero(dx_.any(axis=1) | dv_.any(axis=1))[0]
        print(f"steps {steps}: {dx_.sum()} position words, {dv_.sum()} velocity words, {bad.size} particles differ", flush=True)
        if bad.size:
            from dieselfluid_amd import scenes
            p, _ = scenes.dambreak_scene(n3, math_mode=0)
            h = float(p.h)
            for i in bad[:40]:
                zz = float(pos[i, 2])
                cellfrac = [((float(pos[i, a]) + h) / h) % 1.0 for a in range(3)]
                print(f"  id {i} pos {pos[i]} dpos {z['pos'][i] - pos[i]} dvel {z['vel'][i] - vel[i]} z/h {zz / h:.4f} "
                      f"cell fractions {cellfrac[0]:.5f} {cellfrac[1]:.5f} {cellfrac[2]:.5f}", flush=True)
            break


if __name__ == "__main__":
    main()
