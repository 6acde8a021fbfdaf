#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/yade-openfoam-coupling_amd/lib/variants
for v in LD_NOEXP LD_NOSTORE LD_NODEPOSIT; do
  KSTATS_TOP=5 tools/kstats.sh exp_$v FOAMYADE_HIP_LIB=$V/libfoamyade_hip_exp_$v.so -- python $R/tools/bench_particles.py --steps 4 | grep -E "===|locate_deposit"
done
KSTATS_TOP=5 tools/kstats.sh base -- python $R/tools/bench_particles.py --steps 4 | grep -E "===|force_gaussian|locate_deposit"
