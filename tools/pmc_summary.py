#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc counter_collection.csv to per-kernel averages (one row per kernel name
and grid size): launches, mean of every counter.  Usage: pmc_summary.py IN.csv OUT.csv [name-filter]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[3] if len(sys.argv) > 3 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if flt and flt not in k:
        continue
    agg[(k, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["kernel", "grid", "dispatches"] + names)
for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values())):
    n = max(len(x) for x in v.values())
    w.writerow([k[:160], g, n] + [round(sum(v[c]) / len(v[c]), 2) if c in v else "" for c in names])
