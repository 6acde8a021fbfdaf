#!/bin/bash
# scratch driver for one gpurun call
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_ldu_case.py tests/test_ldu_parity.py -m gpu -x -q > gpurun_out/ldu_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/ldu_tests.log
tail -5 gpurun_out/ldu_tests.log
for cfg in "64 10 lattice" "64 10 wavy" "128 10 lattice" "128 10 wavy" "96 10 prisms" "128 5 wavy 1000000"; do
  timeout 600 python tools/ldu_bench.py $cfg 2>&1 | tail -1 | tee -a gpurun_out/ldu_bench.jsonl
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/ldu_prof -o ldu -- python /root/repo/tools/ldu_bench.py 128 10 wavy > /root/repo/gpurun_out/ldu_prof.log 2>&1
cd /root/repo; ls gpurun_out/ldu_prof | head
