"""How the PCISPH step evolves over a long run (tools only): the reference never brings its predictor state back to the
particles (pcisph_darwin.go:28-41), so the predicted positions -- the query points of DensityF -- drift away from them."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dieselfluid_amd import SPHEngine, scenes

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 160
total = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 50
p, pos = scenes.dambreak_scene(n3, math_mode=1)
p.pci_max_iters = 4
p.eos_w = p.eos_w / 4
p.delta = 1.0e-7
p.pci_max_error = -1.0
eng = SPHEngine(p)
eng.upload("positions", pos)
eng.reset_forces()
eng.pcisph_begin()
done = 0
lo, hi = np.array(p.box_min[:]), np.array(p.box_max[:])
while done < total:
    eng.timing_reset(); eng.timing_enable(True)
    eng.sync(); t0 = time.perf_counter()
    eng.pcisph_step(chunk)
    eng.sync(); dt = time.perf_counter() - t0
    eng.timing_enable(False)
    done += chunk
    st = eng.stats()
    x, xp = eng.download("positions"), eng.download("pci_positions")
    d = np.linalg.norm((xp - x).astype(np.float64), axis=1) / p.h
    outside = np.any((xp < lo - p.h) | (xp > hi + p.h), axis=1)
    print(json.dumps({"steps": done, "ms_per_step": round(dt / chunk * 1e3, 4),
                      "pci_density_ms": round(eng.timing("pci_density")[0], 4), "pci_predict_ms": round(eng.timing("pci_predict")[0], 4),
                      "binned": eng.pcisph_binning()[1],
                      "other_ms": {k: round(eng.timing(k)[0], 4) for k in ("cell_rank", "scan", "scatter", "tile_list", "density", "viscous", "update")},
                      "drift_h_median": round(float(np.median(d)), 3), "drift_h_p99": round(float(np.percentile(d, 99)), 3),
                      "frac_beyond_1h": round(float((d > 1).mean()), 4), "frac_beyond_4h": round(float((d > 4).mean()), 4),
                      "frac_outside_box": round(float(outside.mean()), 4),
                      "max_vel": round(st.max_vel, 3), "finite": bool(np.isfinite(x).all()), "pci_finite": bool(np.isfinite(xp).all())}), flush=True)
