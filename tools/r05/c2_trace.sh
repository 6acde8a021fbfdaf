#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_c2; rm -rf $O; mkdir -p $O
B="python $R/bench.py --config c2 --no-cpu-baseline --wire 0 --pmc 0"
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- $B --steps 6 --warmup 3 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/kt
python - $O/kernel_trace.csv <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
short=lambda n: re.sub(r"\(.*","",re.sub(r"fy::\(anonymous namespace\)::","",re.sub(r"^void ","",n)))[:34]
idx=[i for i,r in enumerate(rows) if "k_set_source_zero" in r["Kernel_Name"]]
a,b=idx[-2]+1,idx[-1]+1
prev=None; tot=0; gaps=0
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"]); g=(s-prev)/1e3 if prev else 0
    print(f"{short(r['Kernel_Name']):36s} {(e-s)/1e3:7.1f} us gap {g:6.1f} grid {r.get('Grid_Size_X', r.get('Grid_Size',''))}")
    prev=e; tot+=e-s; gaps+=max(g,0)
print("kernels",b-a,"sum ms",tot/1e6,"gaps ms",gaps/1e3,"span ms",(int(rows[b-1]["End_Timestamp"])-int(rows[a]["Start_Timestamp"]))/1e6)
PY
