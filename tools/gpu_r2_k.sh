#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_fv_parity.py tests/test_slabs.py tests/test_fv_known_answers_gpu.py -x -q -k "not ghia" 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --wire 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'], d['p_iters_per_step'])"
