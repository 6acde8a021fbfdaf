#!/bin/bash
# scratch driver for one gpurun call (development)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ldu_parity.py tests/test_ldu_case.py tests/test_long_runs.py -m gpu -x -q > gpurun_out/gpu_ldu.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_ldu.log
grep -E "passed|failed|rc=|^E  " gpurun_out/gpu_ldu.log | tail -8
KSTATS_TOP=30 bash tools/kstats.sh ldu_c3 -- python /root/repo/tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple 2>&1 | grep -E "p_cells|p_faces"
grep '"tool"' gpurun_out/ks_ldu_c3/run.log | cut -c1-330
rm -rf gpurun_out/ks_ldu_c3
