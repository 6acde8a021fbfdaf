#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
FOAMYADE_AMG_VERBOSE=1 timeout 600 python tools/_amg_probe.py 2>&1 | grep -v "amg level" | tail -4
python -m pytest tests/test_ldu_case.py tests/test_ldu_parity.py -m gpu -x -q 2>&1 | tail -3
rm -f gpurun_out/ldu_bench.jsonl
for cfg in "64 10 lattice 0 mg" "128 10 lattice 0 mg" "128 10 wavy 0 mg" "96 10 prisms 0 mg"; do
  timeout 600 python tools/ldu_bench.py $cfg 2>&1 | tail -1 | tee -a gpurun_out/ldu_bench.jsonl
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/ldu_prof -o ldu -- python /root/repo/tools/ldu_bench.py 128 10 wavy 0 mg > /root/repo/gpurun_out/ldu_prof.log 2>&1
