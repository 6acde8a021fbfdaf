#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_ldu_case.py tests/test_ldu_parity.py -m gpu -x -q 2>&1 | tail -3
rm -f gpurun_out/ldu_bench.jsonl
for cfg in "128 10 lattice 0 mg" "128 10 wavy 0 mg" "96 10 prisms 0 mg"; do
  timeout 600 python tools/ldu_bench.py $cfg 2>&1 | tail -1 >> gpurun_out/ldu_bench.jsonl
done
python -c "
import json
for l in open('gpurun_out/ldu_bench.jsonl'):
    d=json.loads(l); print(d['kind'],d['cells'],'its',d['pcg_iters_per_step'],'ms',round(d['ms_per_step_wall'],2))
"
KSTATS_TOP=40 bash tools/kstats.sh ldu -- python /root/repo/tools/ldu_bench.py 128 10 wavy 0 mg > gpurun_out/ldu_kstats.txt 2>&1
head -45 gpurun_out/ldu_kstats.txt
timeout 900 python tools/virtual_strong_bench.py 8 5 2>&1 | tail -1 | tee gpurun_out/virtual_strong_8.txt
