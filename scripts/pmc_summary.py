#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 counter_collection.csv (one row per
dispatch and counter)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").split("(")[0][-60:]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in sorted(acc.items()):
    n = max(len(v) for v in ctrs.values())
    print(f"{name}  dispatches={n}")
    for c, v in sorted(ctrs.items()):
        print(f"    {c:28s} avg={sum(v)/len(v):.6g}")
