"""Averages rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py <dir> [name-filter]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if flt and flt not in k:
            continue
        a = acc[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
for k, cs in sorted(acc.items()):
    print(k[:110])
    for c, (v, n) in sorted(cs.items()):
        print(f"    {c:32s} {v / n:16.1f}   (n={n})")
