#!/bin/bash
# usage: tools/env_sweep.sh VAR v1 v2 ... -- runs the default bench once per value of VAR and prints value + per-step phases
var=$1; shift
for x in "$@"; do
  line=$(env $var=$x python bench.py --steps ${STEPS:-6} --warmup 2 2>&1 | tail -1)
  echo "$line" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$x', d['value'], d['per_step_ms'])"
done
