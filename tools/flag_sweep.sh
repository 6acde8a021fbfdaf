#!/bin/bash
# usage: tools/flag_sweep.sh "flags A" "flags B" ... -- runs bench.py once per flag set and prints value + per-step phases
for x in "$@"; do
  line=$(python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline $x 2>&1 | tail -1)
  echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$x]', d['value'], d['per_step_ms'], d.get('p_iters_per_step'))" || echo "$line" | tail -c 600
done
