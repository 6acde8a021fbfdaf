#!/bin/bash
cd /root/repo
python -m pytest tests/test_ldu_parity.py tests/test_ldu_case.py -m gpu -x -q 2>&1 | grep -E "^E  |passed|failed" | head -14
