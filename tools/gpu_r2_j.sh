#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_slabs_multiprocess.py -x -q 2>&1 | tail -30
