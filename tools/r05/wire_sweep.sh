#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for wk in "4 4" "6 6" "8 8" "8 4" "12 8"; do
  set -- $wk
  python bench.py --no-cpu-baseline --no-extras --no-moving --pmc 0 --steps 5 --warmup 3 --wire 3 --wire-workers $1 --wire-helpers $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['drop_in_path']; print('workers $1 helpers $2', p.get('ms_per_step'), p.get('per_step_ms'), p.get('error'))"
done
nproc; cat /sys/fs/cgroup/cpu.max
