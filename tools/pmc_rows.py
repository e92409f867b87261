#!/usr/bin/env python3
"""Mean of every counter per kernel name from a rocprofv3 `*counter_collection.csv` (--pmc ... --kernel-trace --output-format csv).
    python tools/pmc_rows.py <counter_collection.csv> [substring of the kernel name]"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    name = r.get("Kernel_Name") or r.get("Kernel Name")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    key = (r.get("Dispatch_Id"), r.get("Counter_Name"))
    rows[(name[:70], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), c in rows.items():
    print(name, "grid", grid)
    for k, v in sorted(c.items()):
        print("   %-28s %16.1f  (n=%d)" % (k, sum(v) / len(v), len(v)))
