#!/bin/bash
# kernel trace of the general-mesh solver at the C3 size (one step, launch by launch)
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_ldu; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/tools/ldu_bench.py 160 3 wavy 10000000 mg 1e-6 pimple > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/kt
grep "^{" $O/kt.log | cut -c1-600
python $GRAFT_REPO_ROOT/tools/step_trace.py $O/kernel_trace.csv 2 > $O/step_trace.txt; grep -v "k_amg\|k_ell\|k_reduce_fin" $O/step_trace.txt | tail -60
