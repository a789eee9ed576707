"""Where should the PCISPH solver start binning its queries (tools only)?  The 4M scene in chunks of 8 steps, once with
the binning forbidden and once forced: DensityF iteration cost of both forms against the fraction of tile leavers."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dieselfluid_amd import SPHEngine, scenes

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 160
total = int(sys.argv[2]) if len(sys.argv) > 2 else 96
for mode in (-1, 1):
    p, pos = scenes.dambreak_scene(n3, math_mode=1)
    p.pci_max_iters = 4
    p.eos_w = p.eos_w / 4
    p.delta = 1.0e-7
    p.pci_max_error = -1.0
    eng = SPHEngine(p)
    eng.pcisph_set_binning(mode)
    eng.upload("positions", pos)
    eng.reset_forces()
    eng.pcisph_begin()
    g0 = np.array(p.grid_min[:], dtype=np.float32)
    done = 0
    while done < total:
        eng.timing_reset(); eng.timing_enable(True)
        eng.sync(); t0 = time.perf_counter()
        eng.pcisph_step(8)
        eng.sync(); dt = time.perf_counter() - t0
        eng.timing_enable(False)
        done += 8
        x, xp = eng.download("positions"), eng.download("pci_positions")
        tile = lambda a: np.floor(np.clip(np.floor((a - g0) / np.float32(p.h)), 0, None) / 4)
        left = float(np.any(tile(x) != tile(xp), axis=1).mean())
        print(json.dumps({"mode": mode, "steps": done, "ms_per_step": round(dt / 8 * 1e3, 4),
                          "iteration_ms": round(eng.timing("pci_density")[0] + eng.timing("pci_predict")[0], 4),
                          "tile_leavers": round(left, 4)}), flush=True)
    eng.close()
