#!/bin/bash
# general-mesh path: parity tests, then the C3-size bench line + kernel stats (TAG names the outputs)
cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ldu_parity.py tests/test_ldu_case.py tests/test_case_vs_oracle.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
TAG=${TAG:-now} bash tools/r05/ldu_c3.sh 2>&1 | head -${HEADN:-14}
