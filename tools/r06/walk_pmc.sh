#!/bin/bash
# the explicit-tree walk (general meshes) at the C3 size: kernel times of the shipped build and of a build with one more L1 access per visit, then the unit counters
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants
CMD="python $GRAFT_REPO_ROOT/tools/bench_particles.py --mesh wavy --steps 4"
KSTATS_TOP=8 bash tools/kstats.sh wdef -- $CMD
grep "^step" gpurun_out/ks_wdef/run.log | tail -2
for v in "$@"; do
  KSTATS_TOP=8 bash tools/kstats.sh w$v FOAMYADE_HIP_LIB=$V/libfoamyade_hip_$v.so -- $CMD
done
PMCP_NAME=pmcp_walk PMCP_CMD="$CMD" bash tools/pmc_particles.sh > /dev/null 2>&1
PMCP_KERNELS=k_locate,k_deposit,k_force_gaussian python tools/pmc_particles_report.py gpurun_out/pmcp_walk
