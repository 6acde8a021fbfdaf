#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py -m gpu -x -q 2>&1 | tail -4
bash tools/r06/ab_particles.sh default noquad fqonly t256 t256l9 tvnogather tvnoscatter
