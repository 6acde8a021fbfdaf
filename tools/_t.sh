#!/bin/bash
# scratch driver for one gpurun call (development)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_locate_paths.py -m gpu -x -q > gpurun_out/gpu_lp.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_lp.log
grep -E "passed|failed|rc=|^E  " gpurun_out/gpu_lp.log | tail -8
