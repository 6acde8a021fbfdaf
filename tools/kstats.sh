#!/bin/bash
# usage: tools/kstats.sh NAME [ENV=VAL ...] -- command ...   (on the GPU box): rocprofv3 kernel-trace stats of the command, top kernels printed
name=$1; shift
envs=()
while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ks_$name
rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && env "${envs[@]}" rocprofv3 --kernel-trace --stats --output-format csv -d $O -- "$@" > $O/run.log 2>&1 )
f=$(find $O -name "*kernel_stats.csv" | head -1)
echo "=== $name (${envs[*]})"
python - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(__import__("os").environ.get("KSTATS_TOP", "14"))]:
    nm = re.sub(r"fy::\(anonymous namespace\)::", "", r["Name"])
    nm = re.sub(r"\(.*", "", nm)[:44]
    print(f"{nm:46s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
