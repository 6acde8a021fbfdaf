#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export FOAMYADE_FORCE_SPLIT=1 FOAMYADE_HIP_LIB=$R/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_f10.so PMCP_NAME=pmcp_split
bash $R/tools/pmc_particles.sh > /dev/null 2>&1
cd $R && python tools/pmc_particles_report.py gpurun_out/pmcp_split
