#!/bin/bash
# scratch driver for one gpurun call (development)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ldu_parity.py -m gpu -x -q > gpurun_out/gpu_ldu.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_ldu.log
grep -E "passed|failed|rc=|^E  " gpurun_out/gpu_ldu.log | tail -8
