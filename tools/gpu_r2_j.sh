#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_fv_parity.py tests/test_full_size_properties.py tests/test_foam_case.py tests/test_fv_known_answers_gpu.py -x -q 2>&1 | tail -15
