#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
true
V=$R/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_prev.so
for rep in 1 2; do
for lib in "" $V $R/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_cache.so; do
  echo "== lib=$lib"
  FOAMYADE_HIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wire 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'])"
done
done
