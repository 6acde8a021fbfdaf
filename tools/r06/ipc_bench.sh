#!/bin/bash
# bench.py with N > 1 ranks on the ONE GPU of the box: the peer-store transport (FOAMYADE_COMM=ipc, chosen by itself when the ranks outnumber the GPUs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for cfg in "--gpus 2" "--gpus 2 --strong" "--gpus 4 --strong" "--config c5 --gpus 2"; do
  echo "== bench.py $cfg"
  timeout 900 python bench.py $cfg --steps 8 --warmup 3 --no-cpu-baseline --wire 0 --pmc 0 --no-extras 2> gpurun_out/ipc_bench.err | tail -1 | python -c "
import json,sys
l=sys.stdin.read().strip()
try:
    d=json.loads(l); print({k:d.get(k) for k in ('value','ms_per_step','n_gpus','scaling','per_step_ms','p_iters_per_step')}); print(d['config'].get('parallelism')); print(json.dumps(d.get('exchange_wait'))[:1500])
except Exception as e: print('NOT JSON', e, l[-500:])
"
  grep -v "^\[W\|Gloo\|amdgpu.ids\|^$" gpurun_out/ipc_bench.err | tail -5
done
