#!/usr/bin/env python3
"""FV pass table of ONE coupled C3 step (task 1a): every kernel of the step in launch order with its duration, the HBM
bytes the counters saw (FETCH_SIZE doubled per the guide's gfx950 rule, and as counted; WRITE_SIZE), the bytes its arrays hold (the design's
own count: every array the kernel names once) and the SURVEY.md 8(d) row it belongs to -- so that the excess over 8(d)'s 280 + nCorr x 504
B/cell has addresses.

usage: fv_pass_table.py <dir with kernel_trace.csv and pmc_*.csv of tools/r05/baseline.sh> [cells] > profiles/r05_fv_pass_table.txt"""
import csv
import os
import re
import sys

d = sys.argv[1]
NC = int(sys.argv[2]) if len(sys.argv) > 2 else 4096000

# (design bytes per cell) per kernel name; None = not one of 8(d)'s non-solver FV passes
DESIGN = {}


def reg(name, b, row):
    DESIGN[name] = (b, row)


# filled from the file next to this script so that the table and the kernels stay together
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fv_pass_bytes.py")).read())


def short(n):
    n = re.sub(r"fy::\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


pmc = {}
for f in os.listdir(d):
    if f.startswith("pmc_") and f.endswith(".csv"):
        for r in csv.DictReader(open(os.path.join(d, f))):
            pmc.setdefault((r["kernel"], int(r["grid"])), {})[r["counter"]] = float(r["mean_value"])

rows = list(csv.DictReader(open(os.path.join(d, "kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [short(r["Kernel_Name"]) for r in rows]
starts = [i for i, n in enumerate(names) if n == "k_tile_caps"]
a, b = starts[-3], starts[-2]
gkey = "Grid_Size_X" if "Grid_Size_X" in rows[0] else "Grid_Size"
print(f"# one coupled step (C3: {NC} cells), kernels in launch order; us = this launch; rdMB / wrMB = mean HBM bytes per launch of that kernel at that grid, PMC passes")
print("# (rd = 2 x FETCH_SIZE KiB, the guide's gfx950 rule; rd1 = FETCH_SIZE as counted); B/cell = (rd + wr) / cells; design = the arrays the kernel names, each once;")
print("# GB/s = (rd + wr) / us; wait% = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES")
print(f"{'kernel':34s} {'us':>7s} {'rdMB':>7s} {'rd1MB':>7s} {'wrMB':>7s} {'B/cell':>7s} {'design':>7s} {'GB/s':>6s} {'wait%':>6s} 8(d) row")
tot = {}
sum_us = 0.0
for r in rows[a:b]:
    nm = short(r["Kernel_Name"]); grid = int(r[gkey]); us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = pmc.get((nm, grid), {})
    rd1 = c.get("FETCH_SIZE", float("nan")) * 1024 / 1e6; wr = c.get("WRITE_SIZE", float("nan")) * 1024 / 1e6
    rd = 2 * rd1
    wait = 100.0 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
    des, row = DESIGN.get(nm, (None, "-"))
    bc = (rd + wr) * 1e6 / NC
    print(f"{nm:34s} {us:7.1f} {rd:7.1f} {rd1:7.1f} {wr:7.1f} {bc:7.1f} {des if des is not None else '':>7} {(rd + wr) / us * 1e3 if us else 0:6.0f} {wait:6.1f} {row}")
    t = tot.setdefault(row, [0.0, 0.0, 0.0, 0])
    t[0] += us; t[1] += (rd + wr) if rd == rd else 0.0; t[2] += (des or 0); t[3] += 1
    sum_us += us
print()
print("# totals by 8(d) row: launches, us, counter MB (reads doubled), counter B/cell, design B/cell")
for k, (us, mb, des, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:34s} {n:3d} {us:8.1f} {mb:9.1f} {mb * 1e6 / NC:8.1f} {des:8.0f}")
print(f"# step: {b - a} kernels, {sum_us / 1e3:.3f} ms of kernels")
print("# SURVEY.md 8(d) budget of the non-solver FV passes: 280 + nCorr x 504 = 1288 B/cell per step")
