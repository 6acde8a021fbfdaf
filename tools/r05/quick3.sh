#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_slabs.py tests/test_bench_size_parity.py tests/test_mpi_e2e.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
B="python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-extras"
for i in 1 2 3; do $B --steps 64 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['per_step_ms'].get('particle'), d['per_step_ms'].get('pressure'), 'moving', d.get('moving', {}).get('value'))"; done
