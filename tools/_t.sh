#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_slabs.py -x -q -m gpu -k "deep or budget" > $O/tests.txt 2>&1; tail -15 $O/tests.txt
