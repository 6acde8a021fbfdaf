#!/bin/bash
# slab paths: tests, then the virtual-slab ratio (2 slabs) and the 8-slab strong run
cd /root/repo; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_slabs.py tests/test_slabs_multiprocess.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do timeout 600 python tools/virtual_slab_bench.py 2 8 2>&1 | grep "single slab"; done
timeout 600 python tools/virtual_strong_bench.py 8 5 2>&1 | grep "configs"
