#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes of `bench.py` (FETCH_SIZE in one, WRITE_SIZE in the other; the guide's HBM recipe) into
profiles/rNN_pmc_traffic.json: average HBM bytes per launch of the kernels bench.py reports rooflines for.

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE counts wide coalesced streaming reads
at half their size, so reads = 2 x FETCH_SIZE x 1024.  Calibrated on our own kernels: k_mg_smooth at 160^3 writes exactly
4,096,000 x 8 B = 32,000 KiB (WRITE_SIZE = 32000.0) and its doubled FETCH_SIZE (201.6 MB) matches its 196.6 MB of algorithmic reads
within 2.5 %.  For gather-dominated kernels (k_force_gaussian, k_deposit) the doubling is an upper bound and is flagged as such.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [cells_per_rank]
"""
import collections
import csv
import json
import sys


def classify(name, grid, nc):
    if "k_locate_deposit" in name: return "k_locate_deposit"
    if "k_locate_lists" in name: return "k_locate_lists"
    if "k_locate" in name: return "k_locate(walk)"
    if "k_force_gaussian" in name: return "k_force_gaussian"
    if "k_deposit" in name: return "k_deposit"
    if "k_p_apply_dot" in name: return "k_p_apply_dot"
    if "k_mom_pass" in name: return "k_mom_pass"
    if ("k_mg_smooth(" in name or "k_mg_smooth_dot(" in name) and grid >= nc: return "k_mg_smooth(level 0)"
    if "k_tile_reduce" in name: return "k_tile_reduce"
    if "k_point_force" in name: return "k_point_force"
    if "k_bin_gather" in name: return "k_bin_gather"
    return None


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    nc = int(sys.argv[4]) if len(sys.argv) > 4 else 4096000
    res = {}
    for tag, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != tag:
                continue
            k = classify(r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0))), nc)
            if k:
                agg[k].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            res.setdefault(k, {})[tag + "_KiB_per_launch"] = sum(v) / len(v)
            res[k][tag + "_launches"] = len(v)
    for k, v in res.items():
        f, w = v.get("FETCH_SIZE_KiB_per_launch", 0.0), v.get("WRITE_SIZE_KiB_per_launch", 0.0)
        v["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        v["read_correction"] = "x2 (gfx950 wide-coalesced rule)" + ("; upper bound: gather-dominated" if k in ("k_force_gaussian", "k_deposit", "k_locate_deposit", "k_locate_lists", "k_locate(walk)") else "")
    json.dump({"unit": "bytes", "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of bench.py --steps 10 --warmup 2 (12 launches per kernel and step; the first step of a particle population flushes its tables with global atomics and is in the averages)", "kernels": res},
              open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
