#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_fibre_coupling.py tests/test_fv_parity.py -x -q -m gpu 2>&1 | tail -2
V=$R/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_oldlayout.so
for rep in 1 2; do
for lib in "" $V; do
  echo "== lib=$lib"
  FOAMYADE_HIP_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --wire 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['per_step_ms'])"
done
done
