#!/bin/bash
# round-2 GPU session A: parity of the packed-record force pass, then A/B timings
mkdir -p gpurun_out/r2a
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/pytest.log
V=yade-openfoam-coupling_amd/lib/variants
for cfg in "default:" "split:FOAMYADE_FORCE_SPLIT=1" "f10:FOAMYADE_HIP_LIB=$V/libfoamyade_hip_f10.so" "f10t256:FOAMYADE_HIP_LIB=$V/libfoamyade_hip_f10t256.so" "f12t1024:FOAMYADE_HIP_LIB=$V/libfoamyade_hip_f12t1024.so" "split_f10:FOAMYADE_FORCE_SPLIT=1 FOAMYADE_HIP_LIB=$V/libfoamyade_hip_f10.so"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "=== $name ($envs)"
  env $envs timeout 300 python tools/bench_particles.py --steps 6 2>&1 | tail -4
done > gpurun_out/r2a/particles.log 2>&1
cat gpurun_out/r2a/particles.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
tail -c 3000 gpurun_out/r2a/bench.json
