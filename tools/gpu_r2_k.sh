#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python tools/bench_particles.py --steps 5 2>&1 | tail -2
timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_slabs.py -x -q 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --wire 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'], d['p_iters_per_step'])"
