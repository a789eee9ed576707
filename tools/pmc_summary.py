#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: average per dispatch and kernel.
usage: tools/pmc_summary.py <dir-or-csv>... [--filter substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv:
        flt = sys.argv[sys.argv.index("--filter") + 1]
        args = [a for a in args if a != flt]
    files = []
    for a in args:
        files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(files):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if flt and not any(f_ in name for f_ in flt.split(",")):  # (--filter a,b: any of the substrings)
                continue
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"## {k}")
        for c, vals in sorted(v.items()):
            print(f"| {c} | {sum(vals) / len(vals):.5g} | n={len(vals)} |")
        print()


if __name__ == "__main__":
    main()
