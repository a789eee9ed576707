"""prints value, ms/step and the per-kernel times of a bench.py JSON line read from stdin (tools only)"""
import json
import sys

d = json.loads(sys.stdin.read())
print(" ".join(sys.argv[1:]), d["value"], d["ms_per_step"], d.get("kernels_ms"))
