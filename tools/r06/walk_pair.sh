#!/bin/bash
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants
bash tools/r06/walk_ab.sh "$@"
for v in "$@"; do
  echo "== parity with $v"
  FOAMYADE_HIP_LIB=$V/libfoamyade_hip_$v.so timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_graded_mesh.py -m gpu -x -q 2>&1 | tail -3
done
