#!/bin/bash
# stream-ordered in-process communicator against the host-synchronous one: slab tests, then the virtual-slab ratio both ways
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_slabs.py tests/test_slabs_multiprocess.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -5
for m in 0 1; do
  for ov in 1 0; do
    echo "== LOCALCOMM_SYNC=$m HALO_OVERLAP=$ov"
    FOAMYADE_LOCALCOMM_STREAM=$m FOAMYADE_HALO_OVERLAP=$ov timeout 600 python tools/virtual_slab_bench.py 2 8 2>&1 | grep -v "^\[" | tail -4
  done
done
echo "== 4 slabs"
for m in 0 1; do FOAMYADE_LOCALCOMM_STREAM=$m timeout 600 python tools/virtual_strong_bench.py 8 5 2>&1 | grep -v "^\[" | tail -6; done
