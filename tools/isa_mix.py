"""Instruction mix of kernels in an assembly file written by tools/kernel_resources.py --asm:
  python tools/isa_mix.py /tmp/dsl_device.s SUBSTR [SUBSTR ...]
prints, per kernel whose mangled name contains every SUBSTR, the instruction count and the memory / barrier ops."""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
want = sys.argv[2:]
for m in re.finditer(r"^(_ZN3dsl\w+):[^\n]*\n", s, re.M):
    name = m.group(1)
    if not all(w in name for w in want):
        continue
    end = s.index("s_endpgm", m.end())
    c = Counter()
    for line in s[m.end():end].splitlines():
        t = line.strip().split()
        if t and re.match(r"^[a-z_0-9]+$", t[0]):
            c[t[0]] += 1
    keep = {k: v for k, v in sorted(c.items()) if k.startswith(("flat_", "ds_", "global_", "scratch_", "buffer_", "s_barrier"))}
    print(name[:60], "instructions", sum(c.values()), "valu", sum(v for k, v in c.items() if k.startswith("v_")), keep)
