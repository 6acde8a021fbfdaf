#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_particle_parity.py tests/test_bench_size_parity.py tests/test_wire_protocol.py tests/test_async_dt.py tests/test_mpi_e2e.py tests/test_fibre_coupling.py tests/test_long_runs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
B="python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-extras --steps 40 --warmup 5"
for e in "-" "FOAMYADE_NO_DEFER_FORCE=1" "-" "FOAMYADE_NO_DEFER_FORCE=1"; do
  v="$e"; [ "$e" = "-" ] && v=""
  env $v $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('moving') or {}; print('%-28s' % '$e', d['value'], d['ms_per_step'], d['per_step_ms'], '| moving', m.get('value'), m.get('ms_per_step'))"
done
