#!/usr/bin/env python3
"""Compile dslsph.hip for gfx950 (device side only) and print VGPR / SGPR / occupancy / LDS per
kernel from -Rpass-analysis=kernel-resource-usage; optionally keep the assembly.

  python tools/kernel_resources.py [--filter SUBSTR] [--asm /tmp/dsl.s] [-D MACRO ...]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), text=True,
                             capture_output=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--asm", default="/tmp/dsl_device.s")
    ap.add_argument("-D", action="append", default=[])
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    from dieselfluid_amd import _lib
    flags = [f for f in _lib.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    cmd = ["/opt/rocm/bin/hipcc"] + flags + ["-S", "--cuda-device-only", "-o", a.asm,
                                            os.path.join(ROOT, "dieselfluid_amd", "csrc", "dslsph.hip"),
                                            "-Rpass-analysis=kernel-resource-usage"] + ["-D" + d for d in a.D]
    r = subprocess.run(cmd, text=True, capture_output=True)
    if r.returncode:
        sys.stderr.write(r.stderr)
        raise SystemExit(r.returncode)
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    dm = demangle([r_["name"] for r_ in rows])
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scr':>4} {'occ':>3} {'LDS':>7}  kernel")
    for r_ in rows:
        name = dm[r_["name"]]
        name = re.sub(r"\(.*", "", name)
        if a.filter and a.filter not in name:
            continue
        print(f"{r_.get('VGPRs', 0):5d} {r_.get('AGPRs', 0):5d} {r_.get('TotalSGPRs', 0):5d} "
              f"{r_.get('ScratchSize', 0):4d} {r_.get('Occupancy', 0):3d} {r_.get('LDS', 0):7d}  {name}")


if __name__ == "__main__":
    main()
