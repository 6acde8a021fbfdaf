#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_fv_parity.py tests/test_slabs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -8
B="python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-extras"
for i in 1 2; do
  for m in 0 1; do
    echo -n "NO_PAIRS=$m: "; FOAMYADE_NO_PAIRS=$m $B --steps 64 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['per_step_ms'].get('pressure'), 'moving', d.get('moving', {}).get('value'), d['roofline_pEqn_laplacian']['frac'])"
  done
done
for m in 0 1; do echo -n "c2 NO_PAIRS=$m: "; FOAMYADE_NO_PAIRS=$m python bench.py --config c2 --no-cpu-baseline --wire 0 --pmc 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
