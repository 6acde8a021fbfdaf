#!/bin/bash
# FV parity + slab tests, then the C3 headline twice (pressure ms in the third column) and a kernel trace of one step
cd /root/repo; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_fv_parity.py tests/test_slabs.py tests/test_case_vs_oracle.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
B="python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-extras"
for i in 1 2 3; do $B --steps 64 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['per_step_ms'].get('pressure'), 'moving', d.get('moving', {}).get('value'))"; done
python bench.py --config c2 --no-cpu-baseline --wire 0 --pmc 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'], d['ms_per_step'])"
