#!/bin/bash
# does the step time hold over a long timed region?  (per_step_ms of 64 / 320 / 1280 steps, twice)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for n in 64 320 1280; do
  python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-extras --no-moving --steps $n --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['per_step_ms']
print($n, d['value'], d['ms_per_step'], 'rebuilds', d.get('rebuilds_in_region'), {k:p[k] for k in ('particle','locate_deposit','force','momentum','pressure','other')}, d.get('placement_rebuild'))"
done; done
