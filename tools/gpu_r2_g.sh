#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
KSTATS_TOP=60 tools/kstats.sh bench -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/ks_bench.txt
python tools/step_trace.py $(find gpurun_out/ks_bench -name "*kernel_trace.csv" | head -1) 2 > gpurun_out/step_trace.txt
tail -5 gpurun_out/step_trace.txt
