#!/bin/bash
cd /root/repo
python -m pytest tests/test_ldu_parity.py -m gpu -x -q 2>&1 | tail -2
for cfg in "64 10 lattice 0 mg" "128 10 wavy 0 mg"; do
  timeout 600 python tools/ldu_bench.py $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kind'],d['cells'],'its',d['pcg_iters_per_step'],'ms',round(d['ms_per_step_wall'],2))"
done
KSTATS_TOP=40 bash tools/kstats.sh inv -- python /root/repo/tools/ldu_bench.py 64 10 lattice 0 mg 2>&1 | grep -i "invert\|tail"
