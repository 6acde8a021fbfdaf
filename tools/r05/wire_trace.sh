#!/bin/bash
cd /tmp
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
A="$R/tools/native/wire_bench 160 10000000 4 1e-4 - 7"
O=$R/gpurun_out/r05_wt; rm -rf $O; mkdir -p $O
/opt/conda/bin/mpiexec -n 1 $A : -n 6 $A : -n 1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/kt -- $A : -n 6 $A > $O/run.log 2>&1
tail -2 $O/run.log
kt=$(find $O/kt -name "*kernel_trace.csv" | head -1); mc=$(find $O/kt -name "*memory_copy_trace.csv" | head -1)
python - "$kt" "$mc" <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
short=lambda n: re.sub(r"\(.*","",re.sub(r"fy::\(anonymous namespace\)::","",re.sub(r"^void ","",n)))[:30]
idx=[i for i,r in enumerate(rows) if "k_set_source_zero" in r["Kernel_Name"]]
a,b=idx[-2]+1,idx[-1]+1
t0=int(rows[a]["Start_Timestamp"]); t1=int(rows[b-1]["End_Timestamp"])
ev=[]
for r in rows[a:b]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    nm=short(r["Kernel_Name"])
    if "tile_caps" in nm or "force_gaussian" in nm or "pre_coupling" in nm or "corr_back" in nm: ev.append((s,e,"K "+nm))
for r in csv.DictReader(open(sys.argv[2])):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if t0-2_000_000<=s<=t1+15_000_000 and e-s>100000: ev.append((s,e,"C "+r.get("Direction","?").replace("MEMORY_COPY_","")))
ev.sort()
# compress runs of H2D copies
out=[]; run=None
for s,e,nm in ev:
    if nm.startswith("C HOST_TO"):
        if run and s-run[1]<300000: run=(run[0],e,run[2]+1)
        else:
            if run: out.append((run[0],run[1],"H2D x%d"%run[2]))
            run=(s,e,1)
    else:
        if run: out.append((run[0],run[1],"H2D x%d"%run[2])); run=None
        out.append((s,e,nm))
if run: out.append((run[0],run[1],"H2D x%d"%run[2]))
out.sort()
for s,e,nm in out: print(f"{(s-t0)/1e6:8.2f} -> {(e-t0)/1e6:8.2f} ms  {nm}")
PY
rm -rf $O/kt
