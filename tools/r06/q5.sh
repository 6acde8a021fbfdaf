#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_slabs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
VELS="0 0.1" STEPS=4 bash tools/r06/ab_particles.sh default norow rowonly noquad default
