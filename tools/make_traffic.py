#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc_summary.py table (rocprofv3 --pmc passes of the bench command).

  python tools/make_traffic.py gpurun_out/r4/prof_final/pmc.md profiles/traffic.json --particles 16003008 --source "..."

Per hot kernel: HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE counts half of a wide streaming
read on gfx950: MI355X_MICROARCH.md, HBM section; calibrated on k_cell_rank in round 1), the vector-instruction count,
and two readings of how busy the vector ALUs were:
  valu_issue_frac = SQ_INSTS_VALU x 2 / (1024 SIMDs x clocks)   -- VERDICT r03's definition (2 clocks per wave64 op)
  valu_busy_frac  = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x clocks) -- the SQ's own busy count (quad-cycles)
with clocks = GRBM_GUI_ACTIVE / 8 (the counter sums the eight XCDs).  Averages are per launch over every launch of the
kernel in the profiled command; only kernels every launch of which does work are listed."""
import argparse
import json
import re

ALIAS = {"k_force_list": "k_force_list", "k_density_list": "k_density_list", "k_force_integrate_tiled": "k_force_integrate",
         "k_density_pair": "k_density", "k_pci_density_tiled": "k_pci_density"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc_md")
    ap.add_argument("out")
    ap.add_argument("--particles", type=int, required=True)
    ap.add_argument("--source", default="")
    a = ap.parse_args()
    kern, cur = {}, None
    for line in open(a.pmc_md):
        m = re.match(r"## (?:dsl::)?(\w+)(<[^>]*>)?", line)
        if m:
            cur = (m.group(1), m.group(2) or "")
            kern[cur] = {}
            continue
        m = re.match(r"\| (\w+) \| ([0-9.e+\-]+) \| n=(\d+) \|", line)
        if m and cur:
            kern[cur][m.group(1)] = float(m.group(2))
    out = {"source": a.source, "particles": a.particles,
           "definitions": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch; valu_issue_frac = SQ_INSTS_VALU x 2 / (1024 x clocks); "
                          "valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (1024 x clocks); clocks = GRBM_GUI_ACTIVE / 8"}
    for (name, targs), c in kern.items():
        key = ALIAS.get(name)
        if not key or "FETCH_SIZE" not in c or c.get("SQ_INSTS_VALU", 0) < 1e6:
            continue  # (an instantiation that returns at once, a gated launch)
        if key in out and out[key]["insts_valu"] > c["SQ_INSTS_VALU"]:
            continue
        clocks = c["GRBM_GUI_ACTIVE"] / 8.0
        out[key] = {"kernel": name + targs, "fetch_kib": round(c["FETCH_SIZE"]), "write_kib": round(c["WRITE_SIZE"]),
                    "bytes": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "insts_valu": c["SQ_INSTS_VALU"],
                    "clocks": clocks, "valu_issue_frac": round(c["SQ_INSTS_VALU"] * 2 / (1024 * clocks), 4),
                    "valu_busy_frac": round(c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * clocks), 4),
                    "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "insts_lds": c.get("SQ_INSTS_LDS")}
    for listed, swept in (("k_density_list", "k_density"), ("k_force_list", "k_force_integrate")):
        if listed in out:  # (the skin step: the sweeping kernels are gated launches there, their averages mean nothing)
            out.pop(swept, None)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
