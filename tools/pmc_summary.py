#!/usr/bin/env python3
"""Pivot rocprofv3 --pmc counter_collection.csv files: one row per (kernel, grid) with the mean of
each counter over its dispatches.  Usage: python tools/pmc_summary.py gpurun_out/pmc_*/*/*_counter_collection.csv"""
import csv
import re
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            name = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            key = (name[:44], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    ctrs = sorted({c for v in acc.values() for c in v})
    for key in sorted(acc):
        if "conv" not in key[0] and len(sys.argv) and "--all" not in sys.argv:
            continue
        print(f"{key[0]} wgs={key[1]}")
        for c in ctrs:
            if c in acc[key]:
                v = acc[key][c]
                print(f"    {c:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")


if __name__ == "__main__":
    main([a for a in sys.argv[1:] if not a.startswith("--")])
