#!/bin/bash
# A/B of library build variants on the C3 line (per-phase ms): tools/r06/ab_fv.sh default back5 ...   (names under lib/variants; "default" = the shipped build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_$v.so
  [ "$v" = "default" ] && L=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/libfoamyade_hip.so
  FOAMYADE_HIP_LIB=$L python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras --steps ${STEPS:-40} --warmup 5 2>/dev/null | V=$v python -c "
import json,sys,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['per_step_ms']
print('%-10s' % os.environ['V'], d['value'], d['ms_per_step'], 'ld', p['locate_deposit'], 'force', p['force'], 'particle', p['particle'], 'momentum', p['momentum'], 'pressure', p['pressure'], 'other', p['other'])"
done
