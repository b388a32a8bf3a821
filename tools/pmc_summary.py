#!/usr/bin/env python3
"""Averages rocprofv3 --pmc counter CSVs per kernel name: `python tools/pmc_summary.py <dir> [name filter]`."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            acc[name.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        v = acc[name][c]
        print(f"    {c:36s} {sum(v) / len(v):16.1f}   (n={len(v)})")
