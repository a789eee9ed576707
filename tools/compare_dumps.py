"""compares two owned-state dumps of tools/slab_periodic_bench.py (tools only)"""
import sys
import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
same_ids = np.array_equal(a["ids"], b["ids"])
ex = np.abs(a["pos"] - b["pos"]).max() if same_ids else float("nan")
ev = np.abs(a["vel"] - b["vel"]).max() if same_ids else float("nan")
print("particles", a["ids"].shape[0], b["ids"].shape[0], "same ids", same_ids, "max |dx|", ex, "max |dv|", ev)
