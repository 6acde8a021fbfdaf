#!/bin/bash
# A/B of environment switches on the C3 headline: tools/r05/ab.sh "<env1>" "<env2>" ...   ("-" = no variables)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras"
for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e=""
  env $e $B --steps ${STEPS:-40} --warmup 5 ${EXTRA} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s' % '$v', d['value'], d['ms_per_step'], d['per_step_ms'], d['p_iters_per_step'])"
done
