"""Determinism soak of the binned PCISPH path (tools only): the 4M scene for N steps, printed are a checksum of positions,
velocities and predictor positions and the step count at which the queries were first binned; repeated runs must agree."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dieselfluid_amd import SPHEngine, scenes

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 160
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
p, pos = scenes.dambreak_scene(n3, math_mode=1)
p.pci_max_iters = 4
p.eos_w = p.eos_w / 4
p.delta = 1.0e-7
p.pci_max_error = -1.0
eng = SPHEngine(p)
eng.upload("positions", pos)
eng.reset_forces()
eng.pcisph_begin()
first = None
done = 0
while done < steps:
    eng.pcisph_step(4)
    done += 4
    if first is None and eng.pcisph_binning()[1]:
        first = done
h = hashlib.sha256()
for k in ("positions", "velocities", "pci_positions", "pressures"):
    a = eng.download(k)
    assert np.isfinite(a).all() or k == "pci_positions"
    h.update(a.tobytes())
print(json.dumps({"n3": n3, "steps": done, "binned_from": first, "sha256": h.hexdigest()[:16], "max_vel": eng.stats().max_vel}))
