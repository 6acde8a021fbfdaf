#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$R/yade-openfoam-coupling_amd/lib/variants
for v in NOLAW NOFSTORE NOHASH SAMECELL; do
  KSTATS_TOP=5 tools/kstats.sh exp_$v FOAMYADE_HIP_LIB=$V/libfoamyade_hip_exp_$v.so -- python $R/tools/bench_particles.py --steps 4 | grep -E "===|force_gaussian"
done
KSTATS_TOP=5 tools/kstats.sh base -- python $R/tools/bench_particles.py --steps 4 | grep -E "===|force_gaussian|locate_deposit"
export PMCP_NAME=pmcp_r2; bash $R/tools/pmc_particles.sh > /dev/null 2>&1
python tools/pmc_particles_report.py gpurun_out/pmcp_r2 | grep -A8 -E "k_force_gaussian|k_locate_deposit"
