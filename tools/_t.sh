#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
cd $R
export FOAMYADE_TREE_CACHE_DIR=/dev/shm
V=$R/yade-openfoam-coupling_amd/lib/variants
for rep in 1 2; do
for v in cells base zm_u4 zm_u8 zm_p32u4 zm_p8u4; do
  if [ $v == cells ]; then export FOAMYADE_ZMARCH_MIN_CELLS=100000000000; unset FOAMYADE_HIP_LIB;
  elif [ $v == base ]; then export FOAMYADE_ZMARCH_MIN_CELLS=1; unset FOAMYADE_HIP_LIB;
  else export FOAMYADE_ZMARCH_MIN_CELLS=1; export FOAMYADE_HIP_LIB=$V/libfoamyade_hip_$v.so; fi
  echo -n "$v: "; timeout 300 python bench.py --laplacian-probe 320 --laplacian-reps 40 2>/dev/null | tail -1
done; done
