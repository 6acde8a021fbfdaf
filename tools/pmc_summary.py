#!/usr/bin/env python3
"""Per-kernel summary of a directory of rocprofv3 --pmc passes (one sub-directory per counter set, csv output):
duration, HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, guide's gfx950 rule), VALU instructions per wave, wait share, L2 hit rate.
usage: pmc_summary.py <dir> [cells]"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 4096000
data = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(root + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_\w+(<\w+>)?)', r['Kernel_Name'])
        if not m:
            continue
        nm = m.group(1)
        if nm.startswith('k_mg_') and int(r['Grid_Size']) < nc:
            continue
        data[nm][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] == 'FETCH_SIZE':
            dur[nm].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(f"{'kernel':24s} {'us':>7s} {'rdMB':>7s} {'wrMB':>7s} {'GB/s':>6s} {'VALU/wv':>8s} {'SALU/wv':>8s} {'VMEMrd/wv':>9s} {'busy%':>6s} {'L2hit%':>6s}")
rows = []
for k, d in data.items():
    av = lambda n: sum(d[n]) / len(d[n]) if d.get(n) else float('nan')
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else float('nan')
    f = av('FETCH_SIZE') * 1024 * 2 / 1e6
    w = av('WRITE_SIZE') * 1024 / 1e6
    hit, miss = av('TCC_HIT_sum'), av('TCC_MISS_sum')
    grid = None
    rows.append((us, k, f, w, av('SQ_INSTS_VALU'), av('SQ_INSTS_SALU'), av('SQ_INSTS_VMEM_RD'), av('SQ_ACTIVE_INST_VALU'), av('SQ_BUSY_CYCLES'), av('SQ_WAVE_CYCLES'), hit, miss, av('SQ_WAIT_INST_ANY')))
for us, k, f, w, valu, salu, vrd, act, busy, wc, hit, miss, wait in sorted(rows, reverse=True)[:40]:
    nw = nc / 64
    print(f"{k:24s} {us:7.1f} {f:7.1f} {w:7.1f} {(f + w) / us * 1e3 if us == us else 0:6.0f} {valu / nw:8.1f} {salu / nw:8.1f} {vrd / nw:9.1f} {100 * wait / wc if wc == wc else 0:6.1f} {100 * hit / (hit + miss) if hit == hit else 0:6.1f}")
