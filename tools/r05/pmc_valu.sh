#!/bin/bash
# VALU / SALU instruction counts and wait share per kernel (one PMC pass each) of the C3 headline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_valu_${TAG:-x}; rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/$tag -- $B --steps 3 --warmup 2 > $O/pmc_$tag.log 2>&1
  f=$(find $O/pmc/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/r05/pmc_compact.py $f $O/pmc_$tag.csv
done
rm -rf $O/pmc
python - <<PY
import csv,glob
d={}
for f in glob.glob("$O/pmc_*.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["grid"])==4096000:
            d.setdefault(r["kernel"],{})[r["counter"]]=float(r["mean_value"]); d[r["kernel"]]["us"]=float(r["mean_us"])
nw=4096000/64
print(f"{'kernel':30s} {'us':>7s} {'VALU/wv':>8s} {'SALU/wv':>8s} {'VMEMrd':>7s} {'VMEMwr':>7s} {'valu_busy%':>10s} {'wait%':>6s}")
for k,v in sorted(d.items(), key=lambda kv:-kv[1]["us"]):
    g=lambda n: v.get(n,0.0)
    print(f"{k:30s} {v['us']:7.1f} {g('SQ_INSTS_VALU')/nw:8.1f} {g('SQ_INSTS_SALU')/nw:8.1f} {g('SQ_INSTS_VMEM_RD')/nw:7.1f} {g('SQ_INSTS_VMEM_WR')/nw:7.1f} {100*g('SQ_ACTIVE_INST_VALU')/max(g('SQ_BUSY_CYCLES'),1):10.1f} {100*g('SQ_WAIT_INST_ANY')/max(g('SQ_WAVE_CYCLES'),1):6.1f}")
PY
