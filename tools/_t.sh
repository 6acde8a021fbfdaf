#!/bin/bash
# scratch driver for one gpurun call (development): the GPU suite and the smoke entry
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite.log
grep -E "passed|failed|rc=" gpurun_out/gpu_suite.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
