#!/bin/bash
# A/B of particle-kernel builds on the C3 particle phase: tools/r06/ab_particles.sh default noquad ...   (names under lib/variants; "default" = the shipped build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "$@"; do
  L=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_$v.so
  [ "$v" = "default" ] && L=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/libfoamyade_hip.so
  for vel in ${VELS:-0 0.1}; do
    echo "== $v vel=$vel"
    FOAMYADE_HIP_LIB=$L timeout 600 python tools/bench_particles.py --steps ${STEPS:-5} --vel $vel 2>&1 | grep "^step" | tail -3
  done
done
