#!/usr/bin/env python3
"""Ordered kernel list of ONE coupled step from a rocprofv3 kernel trace of bench.py (development tool):
   step_trace.py <kernel_trace.csv> [step_index_from_end=2]  -> name, duration, gap to the previous kernel"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts at the particle phase's first kernel (k_tile_caps; k_courant has been folded into the velocity correction)
starts = [i for i, n in enumerate(names) if "k_tile_caps" in n]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = starts[-which - 1], starts[-which]
prev_end = None
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"fy::\(anonymous namespace\)::", "", r["Kernel_Name"]); nm = re.sub(r"\(.*", "", nm)[:40]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{nm:42s} {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', ''))}")
    prev_end = e
    tot += e - s
print("kernels", b - a, "sum", tot / 1e6, "ms; wall", (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6, "ms")
