#!/usr/bin/env python3
"""rocprofv3 kernel trace -> per-(kernel, grid size) statistics, so that launches of one symbol on different multigrid levels (k_mg_smooth on
160^3 vs 80^3 ...) are told apart: kernel_stats_by_grid.py <kernel_trace.csv> <out.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    nm = re.sub(r"fy::\(anonymous namespace\)::", "", r["Kernel_Name"])
    nm = re.sub(r"\(.*", "", nm)
    grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * max(int(r.get("Grid_Size_Y", 1) or 1), 1) * max(int(r.get("Grid_Size_Z", 1) or 1), 1)
    agg[(nm, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = csv.writer(open(sys.argv[2], "w"))
out.writerow(["Name", "GridSize", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
for (nm, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    out.writerow([nm, grid, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v)])
