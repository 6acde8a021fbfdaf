#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
$R/tools/micro/vec3_layout 160
$R/tools/micro/vec3_layout 320
O=$R/gpurun_out/r05_fused; rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras"
for v in cells arrays; do
  e=""; [ $v = arrays ] && e="FOAMYADE_FACES_FROM_ARRAYS=1"
  env $e rocprofv3 --kernel-trace --output-format csv -d $O/kt_$v -- $B --steps 6 --warmup 3 > $O/kt_$v.log 2>&1
  cp $(find $O/kt_$v -name "*kernel_trace.csv" | head -1) $O/kernel_trace_$v.csv; rm -rf $O/kt_$v
  python $R/tools/step_trace.py $O/kernel_trace_$v.csv 2 > $O/step_trace_$v.txt 2>&1
  echo "== $v"; grep -v "k_mg_\|k_reduce\|k_pcg\|k_p_apply" $O/step_trace_$v.txt
done
