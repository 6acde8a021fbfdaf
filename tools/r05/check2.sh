#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fv_parity.py tests/test_slabs.py tests/test_fv_known_answers_gpu.py tests/test_graded_mesh.py tests/test_foam_case.py -m gpu -x -q > gpurun_out/r05_fused_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05_fused_tests.log
tail -5 gpurun_out/r05_fused_tests.log
bash tools/r05/ab.sh - FOAMYADE_NO_FUSED_CORRECTOR=1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras"
O=$GRAFT_REPO_ROOT/gpurun_out/r05_q; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- $B --steps 6 --warmup 3 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/kt
python $GRAFT_REPO_ROOT/tools/step_trace.py $O/kernel_trace.csv 2 | grep -v "k_mg_\|k_reduce\|k_pcg\|k_p_apply"
