"""prints value, ms/step and the per-kernel times of bench.py JSON lines (files given as arguments, or stdin)"""
import json
import sys


def show(tag, text):
    lines = [l for l in text.strip().splitlines() if l.startswith("{")]
    if not lines:
        print(tag, "NO JSON LINE")
        return
    d = json.loads(lines[-1])
    extra = ""
    if d.get("developed"):
        extra += f" developed {d['developed']['value']} {d['developed']['kernels_ms']}"
    if d.get("exact"):
        extra += f" exact {d['exact']['value']}"
    print(tag, d["value"], d["ms_per_step"], d.get("kernels_ms"), "pass68", d["roofline"].get("pass_frac_68B"), extra)


if len(sys.argv) > 1 and all(a.endswith(".json") for a in sys.argv[1:]):
    for f in sys.argv[1:]:
        show(f, open(f).read())
else:
    show(" ".join(sys.argv[1:]), sys.stdin.read())
