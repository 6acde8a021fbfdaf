#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -6 $O/tests.txt
