"""Targets per 4x4x4-cell tile in the developed 16M flow (tools only): how many tiles need a second pass of the 512-thread
workgroup, and how long that pass is (the remainder decides how many lanes share a target)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from dieselfluid_amd import SPHEngine, scenes

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 252
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
p, pos = scenes.dambreak_scene(n3, math_mode=1)
eng = SPHEngine(p)
eng.upload("positions", pos)
eng.reset_forces()
for _ in range(steps // 500):
    eng.wcsph_step(500)
x = eng.download("positions")
g0 = np.array(p.grid_min[:], dtype=np.float32)
c = np.floor((x - g0) / np.float32(p.h)).astype(np.int64) // 4
dims = c.max(axis=0) + 1
t = (c[:, 2] * dims[1] + c[:, 1]) * dims[0] + c[:, 0]
cnt = np.bincount(t)
cnt = cnt[cnt > 0]
rem = np.where(cnt > 512, cnt - 512, 0)
out = {"tiles": int(cnt.size), "mean": float(cnt.mean()), "p50": float(np.median(cnt)), "frac_over_512": float((cnt > 512).mean()),
       "rem_hist": {k: float(((rem > lo) & (rem <= hi)).mean()) for k, (lo, hi) in
                    {"1-16": (0, 16), "17-32": (16, 32), "33-64": (32, 64), "65-128": (64, 128), "129-256": (128, 256), ">256": (256, 10**9)}.items()},
       "frac_256_512": float(((cnt > 256) & (cnt <= 512)).mean()), "frac_le_256": float((cnt <= 256).mean())}
print(json.dumps(out))
