#!/bin/bash
# same-box A/B of the V-cycle tail's batch size (1 = one cell at a time, the old loops)
cd /root/repo; export TMPDIR=/tmp
L=/root/repo/yade-openfoam-coupling_amd/lib/variants
for i in 1 2; do
for v in "-" "FOAMYADE_HIP_LIB=$L/libfoamyade_hip_tail1.so" "FOAMYADE_HIP_LIB=$L/libfoamyade_hip_tail2.so"; do
  e="$v"; [ "$v" = "-" ] && e=""
  env $e python bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras --steps 64 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-30s' % '$v'[-22:], d['value'], d['ms_per_step'], d['per_step_ms']['pressure'])"
  env $e python bench.py --config c2 --no-cpu-baseline --wire 0 --pmc 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('     c2', d['value'], d['ms_per_step'])"
done; done
