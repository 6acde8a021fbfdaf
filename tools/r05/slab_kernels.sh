#!/bin/bash
# kernel time of two virtual slabs against the single domain, kernel by kernel (both include 2 warm-up steps: steps+2 in the divisor).
# FOAMYADE_LOCALCOMM_TURNS=1: the slabs take turns between collectives, so every kernel is traced at the duration it has with the GPU to itself
cd /root/repo; export TMPDIR=/tmp
KSTATS_TOP=0 bash tools/kstats.sh s1 -- python /root/repo/tools/r05/slab_run.py 1 8
KSTATS_TOP=0 bash tools/kstats.sh s2 FOAMYADE_LOCALCOMM_TURNS=1 -- python /root/repo/tools/r05/slab_run.py 2 8
grep "ms/step" gpurun_out/ks_s1/run.log gpurun_out/ks_s2/run.log
python - <<'PY'
import csv, glob, re
def load(d):
    f = glob.glob(f"gpurun_out/ks_{d}/**/*kernel_stats.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        nm = re.sub(r"fy::(gr::)?(\(anonymous namespace\)::)?", "", r["Name"]); nm = re.sub(r"\(.*", "", nm)[:50]
        t, c = out.get(nm, (0.0, 0)); out[nm] = (t + float(r["TotalDurationNs"]) / 1e6, c + int(r["Calls"]))
    return out
a, b = load("s1"), load("s2")
rows = sorted(set(a) | set(b), key=lambda k: -(b.get(k, (0, 0))[0] - 2 * a.get(k, (0, 0))[0]))
print(f"{'kernel':52s} {'1 slab ms':>10s} {'calls':>6s} {'2 slabs ms':>10s} {'calls':>6s} {'excess ms/step/slab':>20s}")
tot = 0.0
for k in rows:
    ta, ca = a.get(k, (0, 0)); tb, cb = b.get(k, (0, 0)); ex = (tb - 2 * ta) / 10 / 2; tot += ex
    if abs(ex) > 0.004: print(f"{k:52s} {ta:10.2f} {ca:6d} {tb:10.2f} {cb:6d} {ex:20.3f}")
print("sum of kernel time per step: 1 slab", sum(v[0] for v in a.values()) / 10, " 2 slabs (per slab)", sum(v[0] for v in b.values()) / 20, " excess", tot)
PY
