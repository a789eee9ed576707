"""How the step time evolves as the dam-break lattice melts (tools only)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dieselfluid_amd import SPHEngine, scenes

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 126
total = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 500
p, pos = scenes.dambreak_scene(n3, math_mode=1)
eng = SPHEngine(p)
eng.upload("positions", pos)
eng.reset_forces()
done = 0
while done < total:
    eng.timing_reset(); eng.timing_enable(True)
    eng.sync(); t0 = time.perf_counter()
    eng.wcsph_step(chunk)
    eng.sync(); dt = time.perf_counter() - t0
    eng.timing_enable(False)
    done += chunk
    st = eng.stats()
    x = eng.download("positions")
    print(json.dumps({"steps": done, "t_sim": round(done * p.dt, 4), "ms_per_step": round(dt / chunk * 1e3, 4),
                      "density_ms": round(eng.timing("density")[0], 4), "force_ms": round(eng.timing("force_integrate")[0], 4),
                      "grid_ms": round(sum(eng.timing(k)[0] for k in ("cell_rank", "scan", "scatter", "tile_list")), 4),
                      "max_cell": st.max_cell_count, "max_vel": round(st.max_vel, 3),
                      "x_front": round(float(x[:, 0].max()), 3), "finite": bool(np.isfinite(x).all())}), flush=True)

# -- occupancy statistics of the final state ---------------------------------------------
from scipy.spatial import cKDTree
h = p.h
g0 = np.array(p.grid_min[:], dtype=np.float64)
cell = np.floor((x.astype(np.float64) - g0) / h).astype(np.int64)
dims = cell.max(axis=0) + 1
lin = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
cnt = np.bincount(lin, minlength=int(dims.prod())).reshape(dims[2], dims[1], dims[0])
occ = cnt[cnt > 0]
run3 = cnt[:, :, :-2] + cnt[:, :, 1:-1] + cnt[:, :, 2:]          # x-run of 3 cells centred on each cell
pad = np.zeros((dims[2] + 2, dims[1] + 2, dims[0]), dtype=np.int64)
pad[1:-1, 1:-1, 1:-1] = run3
over = (pad > 32)
any_over = np.zeros_like(run3, dtype=bool)
for dz in range(3):
    for dy in range(3):
        any_over |= over[dz:dz + dims[2], dy:dy + dims[1], 1:-1]
w = cnt[:, :, 1:-1]
tree = cKDTree(x)
idx = np.random.default_rng(0).choice(x.shape[0], 20000, replace=False)
nb = np.array([len(v) for v in tree.query_ball_point(x[idx], h)]) - 1
print(json.dumps({"cells_nonempty": int(occ.size), "occ_mean": float(occ.mean()), "occ_std": float(occ.std()),
                  "occ_p99": float(np.percentile(occ, 99)), "run3_mean_weighted": float((run3 * w).sum() / w.sum()),
                  "frac_targets_run_over32": float((over[1:-1, 1:-1, 1:-1] * w).sum() / w.sum()),
                  "frac_targets_any_of_9_runs_over32": float((any_over * w).sum() / w.sum()),
                  "neighbours_mean": float(nb.mean()), "neighbours_p99": float(np.percentile(nb, 99))}))
