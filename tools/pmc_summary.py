#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection csv: per kernel name, mean of each counter per launch."""
import csv, glob, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if pat and pat not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k[:70])
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
