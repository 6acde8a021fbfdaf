#!/bin/bash
# k_deposit (explicit-tree path): shipped build and named variants, interleaved twice; then the explicit-tree parity tests with each variant
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants
CMD="python $GRAFT_REPO_ROOT/tools/bench_particles.py --mesh wavy --steps 5 ${DEP_ARGS}"
for rep in 1 2; do
  KSTATS_TOP=3 bash tools/kstats.sh ddef -- $CMD
  for v in "$@"; do KSTATS_TOP=3 bash tools/kstats.sh d$v FOAMYADE_HIP_LIB=$V/libfoamyade_hip_$v.so -- $CMD; done
done
timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_graded_mesh.py tests/test_ldu_parity.py -m gpu -x -q 2>&1 | tail -2
for v in "$@"; do
  FOAMYADE_HIP_LIB=$V/libfoamyade_hip_$v.so timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_graded_mesh.py tests/test_ldu_parity.py -m gpu -x -q 2>&1 | tail -2
done
