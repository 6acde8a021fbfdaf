#!/bin/bash
cd /root/repo
python -m pytest tests/test_long_runs.py -m gpu -x -q -k general 2>&1 | tail -12
