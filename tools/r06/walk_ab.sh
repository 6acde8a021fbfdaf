#!/bin/bash
# explicit-tree walk: kernel times of the shipped build and of named variants (lib/variants), interleaved twice
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants
CMD="python $GRAFT_REPO_ROOT/tools/bench_particles.py --mesh wavy --steps 4"
for rep in 1 2; do
  KSTATS_TOP=3 bash tools/kstats.sh wdef -- $CMD
  for v in "$@"; do
    KSTATS_TOP=3 bash tools/kstats.sh w$v FOAMYADE_HIP_LIB=$V/libfoamyade_hip_$v.so -- $CMD
    grep "^step 3" gpurun_out/ks_w$v/run.log
  done
done
