#!/usr/bin/env python3
"""Compact a rocprofv3 counter_collection.csv (hundreds of MB of text at C3) to one row per (kernel, grid, counter):
launches, mean counter value, mean duration.  usage: pmc_compact.py <counter_collection.csv> <out.csv>"""
import collections
import csv
import re
import sys

agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    nm = re.sub(r"fy::\(anonymous namespace\)::", "", r["Kernel_Name"])
    nm = re.sub(r"^void ", "", nm)
    nm = re.sub(r"\(.*", "", nm)
    key = (nm, int(r["Grid_Size"]), r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
    a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "counter", "launches", "mean_value", "mean_us"])
    for (nm, grid, cn), (n, v, us) in agg.items():
        w.writerow([nm, grid, cn, n, f"{v / n:.3f}", f"{us / n:.2f}"])
