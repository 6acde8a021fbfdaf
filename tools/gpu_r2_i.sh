#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --wire 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'], d['p_iters_per_step'])"
done
timeout 600 python bench.py --no-cpu-baseline --wire 0 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'], d['p_iters_per_step'])"
