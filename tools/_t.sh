#!/bin/bash
# scratch driver for one gpurun call (development tool)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
tools/micro/lds_atomic_rate > $O/lds_micro.txt 2>&1
timeout 900 python -m pytest tests/test_particle_parity.py -x -q -m gpu > $O/parity.txt 2>&1; tail -3 $O/parity.txt
for rep in 1 2; do
  for v in soa aos; do
    if [ $v == aos ]; then export FOAMYADE_HIP_LIB=$R/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_aos.so; else unset FOAMYADE_HIP_LIB; fi
    timeout 300 python tools/bench_particles.py --steps 6 > $O/bp_${v}_$rep.txt 2>&1
    echo "== $v rest rep $rep"; tail -2 $O/bp_${v}_$rep.txt
    timeout 300 python tools/bench_particles.py --steps 6 --vel 0.05 > $O/bpm_${v}_$rep.txt 2>&1
    echo "== $v moving rep $rep"; tail -2 $O/bpm_${v}_$rep.txt
  done
done
unset FOAMYADE_HIP_LIB
cd /tmp && export TMPDIR=/tmp
for v in soa aos; do
  if [ $v == aos ]; then export FOAMYADE_HIP_LIB=$R/yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_aos.so; else unset FOAMYADE_HIP_LIB; fi
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_$v -- python $R/tools/bench_particles.py --steps 3 > $O/pmc_$v.log 2>&1
done
cat $O/lds_micro.txt
