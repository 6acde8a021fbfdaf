#!/bin/bash
# A/B of environment switches on the C3 line (+ the moving bed): tools/r06/ab_env.sh "<env1>" "<env2>" ...   ("-" = no variables)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e=""
  env $e python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-extras --steps ${STEPS:-40} --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d.get('moving') or {}; p=d['per_step_ms']; q=m.get('per_step_ms') or {}
print('%-26s' % '$v', d['value'], d['ms_per_step'], 'mom', p['momentum'], 'pres', p['pressure'], '| moving', m.get('value'), m.get('ms_per_step'), 'mom', q.get('momentum'), 'pres', q.get('pressure'), 'iters', m.get('p_iters_per_step'))"
done
