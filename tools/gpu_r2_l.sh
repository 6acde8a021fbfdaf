#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_fv_parity.py tests/test_fv_known_answers_gpu.py tests/test_slabs.py tests/test_foam_case.py tests/test_full_size_properties.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
for e in "" FOAMYADE_NO_DIAG_FUSION=1; do
  echo "== $e"
  env $e python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wire 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'])"
done
done
