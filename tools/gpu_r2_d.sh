#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_full_size_properties.py tests/test_slabs.py tests/test_wire_protocol.py -x -q 2>&1 | tail -15
V=$R/yade-openfoam-coupling_amd/lib/variants
for cfg in "tiles:" "dep10:FOAMYADE_HIP_LIB=$V/libfoamyade_hip_dep10.so" "f11:FOAMYADE_HIP_LIB=$V/libfoamyade_hip_f11.so" "notile:FOAMYADE_NO_TILE_FLUSH=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "=== $name ($envs)"
  env $envs timeout 300 python tools/bench_particles.py --steps 6 2>&1 | tail -3
done
KSTATS_TOP=18 tools/kstats.sh tiles -- python $R/tools/bench_particles.py --steps 6
