#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in "default:" "noside:FOAMYADE_NO_SIDE_STREAM=1" "notile:FOAMYADE_NO_TILE_FLUSH=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "=== $name ($envs)"
  env $envs timeout 300 python tools/bench_particles.py --steps 6 2>&1 | tail -2
done
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'], d['p_iters_per_step'])"
