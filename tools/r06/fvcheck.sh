#!/bin/bash
# FV-side check after a solver change: the FV / slab / case parity tests, then the C3 line at rest and moving, then a kernel trace of two steps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_fv_parity.py tests/test_slabs.py tests/test_slabs_multiprocess.py tests/test_fv_known_answers_gpu.py tests/test_graded_mesh.py tests/test_bench_size_parity.py tests/test_case_vs_oracle.py -m gpu -x -q > gpurun_out/r06_fv_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r06_fv_tests.log
grep -E "passed|failed|error|rc=" gpurun_out/r06_fv_tests.log | tail -4
B="python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-extras"
$B --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', d['value'], d['ms_per_step'], d.get('per_step_ms'), d.get('p_iters_per_step')); m=d.get('moving_cloud') or {}; print('moving', {k: m.get(k) for k in ('value','ms_per_step','per_step_ms','p_iters_per_step')})"
cd /tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06_q; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras --steps 6 --warmup 3 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/kt
python $GRAFT_REPO_ROOT/tools/step_trace.py $O/kernel_trace.csv 2 > $O/step_trace.txt; tail -1 $O/step_trace.txt
