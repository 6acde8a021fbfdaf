#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fv_parity.py tests/test_slabs.py tests/test_fv_known_answers_gpu.py tests/test_graded_mesh.py -m gpu -x -q > gpurun_out/r05_fused_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05_fused_tests.log
tail -5 gpurun_out/r05_fused_tests.log
timeout 900 python -m pytest tests/test_bench_size_parity.py -m gpu -x -q -s > gpurun_out/r05_size_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r05_size_parity.log
tail -8 gpurun_out/r05_size_parity.log
B="python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras"
for v in "" "FOAMYADE_FACES_FROM_ARRAYS=1" "FOAMYADE_NO_FUSED_CORRECTOR=1"; do
  env $v $B --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['per_step_ms'], d['p_iters_per_step'])"
done
