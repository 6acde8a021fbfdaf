#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1:], d['value'], d['ms_per_step'], d['per_step_ms']['pressure'], d['p_iters_per_step'])" "$@"; }
for deg in 2 4; do
export FOAMYADE_MG_DEGREE=$deg
for f in "" "--moving" "--config c2"; do timeout 600 python bench.py --no-cpu-baseline --wire 0 --steps 16 --warmup 3 $f 2>/dev/null | show deg$deg $f; done
done
