#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_mpi_e2e.py tests/test_wire_protocol.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
for i in 1 2; do
python bench.py --no-cpu-baseline --no-extras --no-moving --pmc 0 --steps 5 --warmup 3 --wire 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['drop_in_path']; print(p.get('solver_side_ranks'), p.get('ms_per_step'), p.get('per_step_ms'), p.get('error')); q=d.get('drop_in_path_one_receiving_rank',{}); print('one rank', q.get('ms_per_step'))"
done
