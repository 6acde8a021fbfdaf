#!/bin/bash
# PMC passes over the particle kernels of the shipped build and of the two quad variants (tools/pmc_particles.sh per build)
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/variants
for name in default fq noquad; do
  L=$V/libfoamyade_hip_$name.so; [ "$name" = "default" ] && L=$GRAFT_REPO_ROOT/yade-openfoam-coupling_amd/lib/libfoamyade_hip.so
  FOAMYADE_HIP_LIB=$L PMCP_NAME=pmcp_$name bash tools/pmc_particles.sh > /dev/null 2>&1
  echo "===== build: $name"
  PMCP_KERNELS=k_force_gaussian,k_locate_deposit python tools/pmc_particles_report.py gpurun_out/pmcp_$name
done
