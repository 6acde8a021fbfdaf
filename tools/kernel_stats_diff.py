#!/usr/bin/env python3
"""Per-kernel average duration of a rocprofv3 --kernel-trace --stats run next to a committed profile.
usage: kernel_stats_diff.py <new kernel_stats.csv> <old kernel_stats.csv> <steps in the new run>"""
import csv
import re
import sys


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r'(k_\w+(<\w+>)?)', r['Name'])
        nm = m.group(1) if m else r['Name'][:30]
        out[nm] = (float(r['TotalDurationNs']), int(r['Calls']), float(r['AverageNs']) / 1e3)
    return out


new, old, steps = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
print(f"{'kernel':32s} {'calls':>6s} {'avg_us':>8s} {'old_us':>8s} {'ms/step':>8s}")
for nm, (t, c, a) in sorted(new.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{nm:32s} {c:6d} {a:8.1f} {old.get(nm, (0, 0, 0))[2]:8.1f} {t / 1e6 / steps:8.3f}")
print('kernel time per step [ms]:', sum(v[0] for v in new.values()) / 1e6 / steps)
