"""Soak (GPU box): the 16M bench scene stepped into its developed state, max |v| and the fullest cell printed every
500 steps; stops at the first non-finite or absurd value.  Looks for rare faults that only a long run meets: repeated runs
must end in the same bits (state_sha1 of positions + velocities), skin step, suspensions and retries included.
  python tools/soak_developed.py [n3] [steps] [tag]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n3 = int(sys.argv[1]) if len(sys.argv) > 1 else 252
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10500
    tag = sys.argv[3] if len(sys.argv) > 3 else ""
    import numpy as np
    from dieselfluid_amd import SPHEngine, scenes
    p, pos = scenes.dambreak_scene(n3, math_mode=1)
    eng = SPHEngine(p, device=0)
    eng.upload("positions", pos)
    eng.reset_forces()
    t0 = time.perf_counter()
    done, bad_at = 0, None
    hist = []
    while done < steps:
        eng.wcsph_step(500)
        done += 500
        st = eng.stats()
        hist.append((done, round(float(st.max_vel), 4), int(st.max_cell_count)))
        if not np.isfinite(st.max_vel) or st.max_vel > 500.0 or st.max_cell_count > 200:
            bad_at = done
            break
    import hashlib
    digest = hashlib.sha1(eng.download("positions").tobytes() + eng.download("velocities").tobytes()).hexdigest()[:16]
    skin = {k: eng.get_option(k) for k in ("skin", "skin_steps", "skin_rebuilds", "skin_suspensions", "skin_list_overflow")}
    out = {"tag": tag, "lib": os.environ.get("DSL_LIB", "default"), "n3": n3, "steps": done, "bad_at": bad_at,
           "state_sha1": digest, "skin": skin,
           "seconds": round(time.perf_counter() - t0, 1), "last": hist[-3:], "env": {k: v for k, v in os.environ.items() if k.startswith("DSL_")}}
    print(json.dumps(out), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
