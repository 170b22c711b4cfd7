#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (CSV), attributed to the kernel that FOLLOWS the gap.
usage: gap_report.py <kernel_trace.csv> [min_gap_us] [from_ms] [to_ms]  -- prints, per (previous -> following) kernel pair, count / total /
mean idle time; the window (ms after the first kernel) lets the warm-up step, which grows every buffer, be left out."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_first = int(rows[0]["Start_Timestamp"])
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e18
rows = [r for r in rows if lo <= (int(r["Start_Timestamp"]) - t_first) / 1e6 <= hi]
short = lambda n: re.sub(r"\(.*", "", re.sub(r"void |\(anonymous namespace\)::", "", n))[:48]
gaps = defaultdict(lambda: [0, 0.0])
busy = 0.0
idle = 0.0
end = int(rows[0]["Start_Timestamp"])
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - end) / 1e3
    if g > thr and g < 2e5:  # > 0.2 s: between bench phases (data generation, accuracy legs)
        key = (short(rows[i - 1]["Kernel_Name"]) if i else "-") + " -> " + short(r["Kernel_Name"])
        gaps[key][0] += 1
        gaps[key][1] += g
        idle += g
    busy += (e - s) / 1e3
    end = max(end, e)
print("kernels %d, busy %.1f ms, idle (gaps < 0.2 s) %.1f ms" % (len(rows), busy / 1e3, idle / 1e3))
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.2f ms %6d x %8.1f us  %s" % (t / 1e3, n, t / n, k))
