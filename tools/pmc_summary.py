#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel (mean counter value per dispatch).

usage: pmc_summary.py <dir-with-*_counter_collection.csv> [more dirs...]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    if not name.startswith(("void rt::", "rt::")):
                        continue
                    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for name, counters in sorted(acc.items()):
        for cname, vals in sorted(counters.items()):
            print(f"{name[:60]:60s} {cname:12s} n={len(vals):3d} mean={sum(vals)/len(vals):.1f} min={min(vals):.1f} max={max(vals):.1f}")


if __name__ == "__main__":
    main()
