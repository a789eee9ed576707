"""Save / replay a drifted PCISPH state of the 4M scene (tools only): `save N file` runs N steps and stores positions,
velocities, forces and the predictor state; `run file steps [binning]` uploads it and times `steps` steps."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dieselfluid_amd import SPHEngine, scenes


def scene(n3):
    p, pos = scenes.dambreak_scene(n3, math_mode=1)
    p.pci_max_iters = 4
    p.eos_w = p.eos_w / 4
    p.delta = 1.0e-7
    p.pci_max_error = -1.0
    return p, pos


if sys.argv[1] == "save":
    n3, steps, path = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    p, pos = scene(n3)
    eng = SPHEngine(p)
    eng.upload("positions", pos)
    eng.reset_forces()
    eng.pcisph_begin()
    eng.pcisph_step(steps)
    np.savez(path, n3=n3, **{k: eng.download(k) for k in ("positions", "velocities", "forces", "pci_positions", "pci_velocities")})
    print("saved", path, eng.pcisph_binning())
else:
    path, steps = sys.argv[2], int(sys.argv[3])
    binning = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    z = np.load(path)
    p, _ = scene(int(z["n3"]))
    eng = SPHEngine(p)
    eng.pcisph_set_binning(binning)
    eng.upload("positions", z["positions"])
    eng.upload("velocities", z["velocities"])
    eng.upload("forces", z["forces"])
    eng.pcisph_begin()
    eng.upload("pci_positions", z["pci_positions"])
    eng.upload("pci_velocities", z["pci_velocities"])
    eng.pcisph_step(3)
    eng.timing_reset(); eng.timing_enable(True)
    eng.sync(); t0 = time.perf_counter()
    eng.pcisph_step(steps)
    eng.sync(); dt = time.perf_counter() - t0
    eng.timing_enable(False)
    print(json.dumps({"ms_per_step": round(dt / steps * 1e3, 4), "binning": eng.pcisph_binning(),
                      "ms": {k: round(eng.timing(k)[0], 4) for k in ("cell_rank", "scan", "scatter", "tile_list", "density", "viscous", "pci_predict", "pci_density", "update")}}))
