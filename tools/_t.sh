#!/bin/bash
cd /root/repo
python -m pytest tests/test_slabs.py -m gpu -x -q -k "rccl" 2>&1 | tail -12
