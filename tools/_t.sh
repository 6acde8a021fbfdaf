#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4i; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_ldu_parity.py -x -q -m gpu > $O/tests.txt 2>&1; tail -25 $O/tests.txt
