#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite.log
tail -4 gpurun_out/gpu_suite.log
RND=r04 bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -c 400 gpurun_out/r04/bench_line.json
rm -f gpurun_out/ldu_bench.jsonl
for cfg in "128 10 lattice 0 mg" "128 10 wavy 0 mg" "128 10 wavy 0 diag" "96 10 prisms 0 mg" "128 5 wavy 1000000 mg"; do
  timeout 600 python tools/ldu_bench.py $cfg 2>&1 | tail -1 >> gpurun_out/ldu_bench.jsonl
done
KSTATS_TOP=30 bash tools/kstats.sh ldu -- python tools/ldu_bench.py 128 10 wavy 0 mg > gpurun_out/ldu_kstats.txt 2>&1
cat gpurun_out/ldu_kstats.txt | head -40
