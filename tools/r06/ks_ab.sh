#!/bin/bash
# per-kernel averages of the C3 bench under rocprofv3 for two library builds
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_$v.so
  [ "$v" = "default" ] && L=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/libfoamyade_hip.so
  O=$GRAFT_REPO_ROOT/gpurun_out/ksab_$v; rm -rf $O; mkdir -p $O
  FOAMYADE_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras --steps 20 --warmup 5 > $O/log 2>&1
  echo "== $v"; grep -h -E "k_locate_deposit|k_force_gaussian|k_tile_reduce" $(find $O/ks -name "*kernel_stats.csv") | awk -F, '{print $1, $2, $4}' | cut -c1-40,400-
  python - <<PY
import csv,glob
f=glob.glob("$O/ks/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_locate_deposit","k_force_gaussian")): print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3)
PY
done
