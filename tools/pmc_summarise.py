#!/usr/bin/env python3
"""Sums a rocprofv3 --pmc counter_collection CSV per kernel name: usage pmc_summarise.py <counter_collection.csv> [counter]
Prints kernel, dispatches, and the counter's sum and per-dispatch mean (the counters FETCH_SIZE / WRITE_SIZE are in KB)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else None
acc = defaultdict(lambda: [0, 0.0])
disp = defaultdict(set)
with open(path, newline="") as f:
    for r in csv.DictReader(f):
        if want and r.get("Counter_Name") != want:
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        key = (name, r.get("Counter_Name"))
        acc[key][1] += float(r["Counter_Value"])
        disp[key].add(r["Dispatch_Id"])
for (name, c), v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    n = len(disp[(name, c)])
    print("%-90s %-12s n=%4d sum=%16.1f mean=%14.1f" % (name, c, n, v[1], v[1] / max(n, 1)))
