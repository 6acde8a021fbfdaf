#!/bin/bash
# kernel trace of one coupled step at C2 (1 M cells / 1 M particles): where a launch-bound step spends its time
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_c2; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --config c2 --wire 0 --no-cpu-baseline --no-extras --pmc 0 --steps 20 --warmup 5 > $O/kt.log 2>&1
f=$(ls -S $(find $O/kt -name "*kernel_trace.csv") | head -1); cp $f $O/kernel_trace.csv; rm -rf $O/kt
tail -1 $O/kt.log | cut -c1-300
python $GRAFT_REPO_ROOT/tools/step_trace.py $O/kernel_trace.csv 3 > $O/step_trace.txt; tail -3 $O/step_trace.txt
