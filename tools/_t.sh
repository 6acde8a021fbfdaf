#!/bin/bash
cd /root/repo
rm -f gpurun_out/ldu_bench_pimple.jsonl
for cfg in "64 10 wavy 300000 mg 1e-6 pimple" "128 10 wavy 2500000 mg 1e-6 pimple" "128 10 lattice 2500000 mg 1e-6 pimple" "96 10 prisms 1000000 mg 1e-6 pimple"; do
  timeout 900 python tools/ldu_bench.py $cfg 2>&1 | tail -1 >> gpurun_out/ldu_bench_pimple.jsonl
done
python -c "
import json
for l in open('gpurun_out/ldu_bench_pimple.jsonl'):
    try: d=json.loads(l)
    except Exception: print('BAD', l[:300]); continue
    print(d['kind'],d['cells'],d['particles'],'its',d['pcg_iters_per_step'],'ms',round(d['ms_per_step_wall'],2),'particle ms',round(d['ms_particle'],2),'create',round(d['create_s'],1))
"
