#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_suite.log
tail -4 gpurun_out/gpu_suite.log
RND=r04 bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
python -c "
import json
d=json.load(open('gpurun_out/r04/bench_line.json'))
print(d['value'], d['ms_per_step'], d.get('c2',{}).get('value'), d.get('general_mesh'))
"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
