#!/bin/bash
# explicit-tree walk with 8-byte stack entries (variant e8): kernel time at the depth the histogram picks and at forced depths, then the explicit-tree parity tests with the variant
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants
CMD="python $GRAFT_REPO_ROOT/tools/bench_particles.py --mesh wavy --steps 4"
for rep in 1 2; do
  KSTATS_TOP=1 bash tools/kstats.sh wdef -- $CMD
  KSTATS_TOP=1 bash tools/kstats.sh we8 FOAMYADE_HIP_LIB=$V/libfoamyade_hip_e8.so -- $CMD
done
for d in 11 14 16 20; do
  KSTATS_TOP=1 bash tools/kstats.sh we8_$d FOAMYADE_HIP_LIB=$V/libfoamyade_hip_e8.so FOAMYADE_LOCATE_STACK=$d -- $CMD
done
KSTATS_TOP=1 bash tools/kstats.sh wdef_11 FOAMYADE_LOCATE_STACK=11 -- $CMD
echo "== parity with e8"
FOAMYADE_HIP_LIB=$V/libfoamyade_hip_e8.so timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_graded_mesh.py tests/test_ldu_parity.py -m gpu -x -q 2>&1 | tail -3
