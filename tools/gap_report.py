#!/usr/bin/env python3
"""Where the GPU idles inside a coupled step: gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (one steady step,
delimited by k_force_gaussian launches), grouped by the kernel that precedes the gap.  usage: gap_report.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
idx = [i for i, e in enumerate(ev) if "k_force_gaussian" in e[2]]
a, b = idx[-3], idx[-2]
seg = ev[a:b + 1]
short = lambda n: re.sub(r"\(.*", "", n.replace("fy::(anonymous namespace)::", "").replace("void ", ""))[:34]
by = collections.defaultdict(lambda: [0, 0])
tot = 0
for x, y in zip(seg[:-1], seg[1:]):
    g = y[0] - x[1]
    if g > 2000:
        by[short(x[2]) + " -> " + short(y[2])][0] += 1
        by[short(x[2]) + " -> " + short(y[2])][1] += g
        tot += g
print(f"step {(seg[-1][0] - seg[0][0]) / 1e6:.3f} ms, {len(seg) - 1} kernels, kernel time {sum(e[1] - e[0] for e in seg[:-1]) / 1e6:.3f} ms, gaps > 2 us: {tot / 1e6:.3f} ms")
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / 1e3:8.1f} us  x{n:<3d} {k}")
