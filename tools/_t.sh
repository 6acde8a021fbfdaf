#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_slabs.py tests/test_slabs_multiprocess.py -m gpu -x -q 2>&1 | tail -6
timeout 600 python tools/virtual_slab_bench.py 2 10 2>&1 | tail -1
timeout 900 python tools/virtual_strong_bench.py 8 5 2>&1 | tail -1
