#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_particle_parity.py -x -q -k "nearest" 2>&1 | tail -12
FOAMYADE_EXPLICIT_TREE=1 timeout 600 python -m pytest tests/test_particle_parity.py -x -q -k "nearest" 2>&1 | tail -3
timeout 300 python bench.py --force-rccl --steps 3 --warmup 1 --no-cpu-baseline --wire 0 2>&1 | tail -c 300
