"""ISA audit of libdslsph's device code (no GPU needed: hipcc cross-compiles gfx950).

Rule checked: every `s_barrier` is preceded, in straight-line code with no LDS operation or branch in between, by an
`s_waitcnt ... lgkmcnt(0)`.  Round 3 lost days of soak runs to a barrier at the HEAD of a loop whose body ENDED in
LDS writes: the compiler's waitcnt pass put no wait in front of it (the barrier is reached over the back edge), a
wave could pass it while a sibling's records were still in flight, and one 16M run in three met a stale record.
kernels_tiled.hpp: sync_lds() states the wait explicitly; this audit keeps it that way.

  python tools/isa_audit.py            # prints the offenders, exit code 1 if any
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(extra_flags=()):
    sys.path.insert(0, ROOT)
    from dieselfluid_amd import _lib
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-fPIC", "-shared", "-Wall")]
    out = os.path.join(tempfile.mkdtemp(prefix="dsl_isa_"), "dslsph.s")
    cmd = ["/opt/rocm/bin/hipcc"] + flags + list(extra_flags) + ["-S", "--cuda-device-only", "-o", out,
                                                                os.path.join(ROOT, "dieselfluid_amd", "csrc", "dslsph.hip")]
    subprocess.run(cmd, check=True, capture_output=True)
    return open(out).read()


def unprotected_barriers(asm: str):
    """[(kernel, index, the instructions in front of the barrier)] for barriers without a drained LDS queue"""
    bad, total = [], 0
    for m in re.finditer(r"^(_ZN3dsl\w+):", asm, re.M):
        end = asm.find("s_endpgm", m.end())
        body = [l.strip() for l in asm[m.end():end].splitlines() if l.strip() and not l.strip().startswith(";")]
        for i, l in enumerate(body):
            if not l.startswith("s_barrier"):
                continue
            total += 1
            ok = False
            for x in reversed(body[max(0, i - 16):i]):
                if "lgkmcnt(0)" in x:
                    ok = True
                    break
                if x.startswith(("ds_", "s_cbranch", "s_branch")) or x.endswith(":"):
                    break
            if not ok:
                bad.append((m.group(1), i, body[max(0, i - 6):i]))
    return bad, total


if __name__ == "__main__":
    bad, total = unprotected_barriers(device_asm())
    for k, i, prev in bad:
        print(k[:70], i, prev)
    print(f"{total} barriers, {len(bad)} without an LDS drain in front")
    sys.exit(1 if bad else 0)
