#!/bin/bash
# unit-level counters of the general-mesh particle kernels (explicit-tree walk, deposit, force) at the C3 size
cd /root/repo; export TMPDIR=/tmp
export PMCP_NAME=pmc_ldu PMCP_CMD="python /root/repo/tools/ldu_bench.py 160 2 wavy 10000000 mg 1e-6 pimple"
bash tools/pmc_particles.sh > /dev/null 2>&1
PMCP_KERNELS=k_locate,k_deposit,k_force_gaussian,k_ldu_pre_coupling python tools/pmc_particles_report.py gpurun_out/pmc_ldu | tee gpurun_out/pmc_ldu_report.txt
