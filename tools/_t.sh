#!/bin/bash
cd /root/repo
python -m pytest tests/test_ldu_parity.py -m gpu -x -q -k bench_size 2>&1 | tail -15
