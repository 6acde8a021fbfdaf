#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/yade-openfoam-coupling_amd/lib/variants
KSTATS_TOP=12 tools/kstats.sh fused_f10 FOAMYADE_HIP_LIB=$V/libfoamyade_hip_f10.so -- python $R/tools/bench_particles.py --steps 6
KSTATS_TOP=12 tools/kstats.sh split_f10 FOAMYADE_FORCE_SPLIT=1 FOAMYADE_HIP_LIB=$V/libfoamyade_hip_f10.so -- python $R/tools/bench_particles.py --steps 6
