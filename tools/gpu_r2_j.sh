#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_fv_known_answers_gpu.py -x -q -k "hip" 2>&1 | tail -15
