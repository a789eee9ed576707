"""wall time per WCSPH step with and without the per-kernel HIP events (tools only)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dieselfluid_amd import SPHEngine, scenes

n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 252
p, pos = scenes.dambreak_scene(n3, math_mode=1)
eng = SPHEngine(p)
eng.upload("positions", pos)
eng.reset_forces()
eng.wcsph_step(5)
for timing in (False, True, False, True):
    eng.timing_reset()
    eng.timing_enable(timing)
    eng.sync()
    t0 = time.perf_counter()
    eng.wcsph_step(20)
    eng.sync()
    dt = time.perf_counter() - t0
    print("events" if timing else "plain ", round(dt / 20 * 1e3, 4), "ms/step", round(n3 ** 3 * 20 / dt / 1e6, 1), "M/s")
eng.timing_enable(False)
