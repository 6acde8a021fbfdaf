#!/bin/bash
# scratch driver for one gpurun call (development): the GPU suite, then the round's profile set
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite.log
tail -4 gpurun_out/gpu_suite.log
RND=${RND:-r04} bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r04/bench_line.json'))
print(d['value'], d['ms_per_step'], d.get('c2',{}).get('value'), d.get('general_mesh',{}).get('value'), d.get('moving',{}).get('value'))
"
rm -f gpurun_out/ldu_bench.jsonl
for cfg in "128 10 lattice 0 mg" "128 10 wavy 0 mg" "96 10 prisms 0 mg" "128 10 wavy 2500000 mg 1e-6 pimple" "96 10 prisms 1000000 mg 1e-6 pimple"; do
  timeout 600 python tools/ldu_bench.py $cfg 2>&1 | tail -1 >> gpurun_out/ldu_bench.jsonl
done
python -c "
import json
for l in open('gpurun_out/ldu_bench.jsonl'):
    d=json.loads(l); print(d['kind'],d['cells'],d['solver'],d['particles'],'its',d['pcg_iters_per_step'],'ms',round(d['ms_per_step_wall'],2))
"
