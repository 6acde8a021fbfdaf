#!/bin/bash
# scratch driver for one gpurun call (development)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ldu_parity.py tests/test_ldu_case.py tests/test_long_runs.py -m gpu -x -q > gpurun_out/gpu_ldu.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_ldu.log
grep -E "passed|failed|rc=|^E  " gpurun_out/gpu_ldu.log | tail -8
KSTATS_TOP=30 bash tools/kstats.sh ldu_c3 -- python /root/repo/tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple 2>&1 | grep -E "grad_vec|grad_scalar|reconstruct"
rm -rf gpurun_out/ks_ldu_c3
rm -f gpurun_out/ldu_bench_pimple.jsonl gpurun_out/ldu_bench_c3.jsonl gpurun_out/ldu_bench.jsonl
python tools/ldu_bench.py 64 10 wavy 300000 mg 1e-6 pimple | grep '"tool"' >> gpurun_out/ldu_bench_pimple.jsonl
python tools/ldu_bench.py 128 10 wavy 2500000 mg 1e-6 pimple | grep '"tool"' >> gpurun_out/ldu_bench_pimple.jsonl
python tools/ldu_bench.py 128 10 lattice 2500000 mg 1e-6 pimple | grep '"tool"' >> gpurun_out/ldu_bench_pimple.jsonl
python tools/ldu_bench.py 96 10 prisms 1000000 mg 1e-6 pimple | grep '"tool"' >> gpurun_out/ldu_bench_pimple.jsonl
python tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple | grep '"tool"' >> gpurun_out/ldu_bench_c3.jsonl
python tools/ldu_bench.py 160 5 lattice 10000000 mg 1e-6 pimple | grep '"tool"' >> gpurun_out/ldu_bench_c3.jsonl
python tools/ldu_bench.py 128 10 lattice 0 mg 1e-4 ico | grep '"tool"' >> gpurun_out/ldu_bench.jsonl
python tools/ldu_bench.py 128 10 wavy 0 mg 1e-4 ico | grep '"tool"' >> gpurun_out/ldu_bench.jsonl
python tools/ldu_bench.py 96 10 prisms 0 mg 1e-4 ico | grep '"tool"' >> gpurun_out/ldu_bench.jsonl
python3 -c "
import json
for f in ('gpurun_out/ldu_bench_pimple.jsonl','gpurun_out/ldu_bench_c3.jsonl','gpurun_out/ldu_bench.jsonl'):
    for l in open(f):
        d=json.loads(l); print(d['solver'], d['kind'], d['cells'], d['particles'], round(d['ms_per_step_stream'],2), round(d['ms_particle'],2), d.get('structured_ms_per_step'), d['pcg_iters_per_step'])
"
