#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_fv_parity.py tests/test_slabs.py tests/test_fv_known_answers_gpu.py tests/test_full_size_properties.py tests/test_slabs_multiprocess.py -x -q 2>&1 | tail -8
for f in "" "--moving"; do timeout 600 python bench.py --no-cpu-baseline --wire 0 --steps 16 --warmup 3 $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['ms_per_step'], d['per_step_ms'], d['p_iters_per_step'], d['u_iters_per_step'])" $f; done
