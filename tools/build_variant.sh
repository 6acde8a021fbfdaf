#!/bin/bash
# A/B builds of the particle kernels with other compile-time constants (development tool):
#   tools/build_variant.sh NAME "-DFY_FORCE_LOG2=10 ..."   ->  yade-openfoam-coupling_amd/lib/variants/libfoamyade_hip_NAME.so
# select at run time with FOAMYADE_HIP_LIB=<that path>.  Only particle_kernels.hip is recompiled; the other objects are the default build's.
set -e
cd "$(dirname "$0")/../yade-openfoam-coupling_amd/csrc"
name=$1; shift
make -s all
mkdir -p ../build/variants ../lib/variants
src=${VARIANT_SRC:-particle_kernels.hip}
repl=${VARIANT_REPLACES:-$src}      # the default object the variant stands in for (VARIANT_SRC may be another file, e.g. an older revision)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-result "$@" -x hip -c $src -o ../build/variants/${src}_$name.o
objs=""
for f in coupling.cpp comm.cpp kdtree.cpp particle_kernels.hip fv_kernels.hip fv_kernels_graded.hip fv_solver.cpp fv_pressure.cpp fv_solver_api.cpp foam_dict.cpp foam_case.cpp ldu_mesh.cpp ldu_kernels.hip ldu_amg.hip ldu_solver.cpp; do
  if [ "$f" == "$repl" ]; then objs="$objs ../build/variants/${src}_$name.o"; else objs="$objs ../build/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/libfoamyade_hip_$name.so $objs -pthread -ldl
echo built ../lib/variants/libfoamyade_hip_$name.so
