#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_${1:-ldloop}.so
FOAMYADE_HIP_LIB=$V timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_bench_size_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
VELS="0 0.1" STEPS=4 bash tools/r06/ab_particles.sh default ${1:-ldloop} default ${1:-ldloop}
