#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_mpi_e2e.py -x -q 2>&1 | tail -40
