#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/yade-openfoam-coupling_amd/lib/variants
for v in SCAT_NOHASH SCAT_NOADD SCAT_ONEADD SCAT_NOFLUSH GATHER_SAMECELL GATHER_NOLAW GATHER_NOFSTORE; do
  KSTATS_TOP=5 tools/kstats.sh exp_$v FOAMYADE_HIP_LIB=$V/libfoamyade_hip_exp_$v.so -- python $R/tools/bench_particles.py --steps 4 | grep -E "===|scatter|gather_pairs"
done
