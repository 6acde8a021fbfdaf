#!/bin/bash
# scratch driver for one gpurun call (development): the general-mesh bench set
cd /root/repo
mkdir -p gpurun_out
rm -f gpurun_out/ldu_bench.jsonl gpurun_out/ldu_bench_pimple.jsonl
for cfg in "64 10 lattice 0 mg" "128 10 lattice 0 mg" "128 10 wavy 0 mg" "128 10 wavy 0 diag" "96 10 prisms 0 mg"; do
  timeout 600 python tools/ldu_bench.py $cfg 2>&1 | tail -1 >> gpurun_out/ldu_bench.jsonl
done
for cfg in "64 10 wavy 300000 mg 1e-6 pimple" "128 10 wavy 2500000 mg 1e-6 pimple" "128 10 lattice 2500000 mg 1e-6 pimple" "96 10 prisms 1000000 mg 1e-6 pimple"; do
  timeout 900 python tools/ldu_bench.py $cfg 2>&1 | tail -1 >> gpurun_out/ldu_bench_pimple.jsonl
done
python -c "
import json
for fn in ('gpurun_out/ldu_bench.jsonl','gpurun_out/ldu_bench_pimple.jsonl'):
    for l in open(fn):
        d=json.loads(l); print(d['kind'],d['cells'],d['solver'],d['p_solver'],d['particles'],'its',d['pcg_iters_per_step'],'ms',round(d['ms_per_step_wall'],2),d.get('structured_ms_per_step'))
"
