#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1 HSA_ENABLE_IPC_MODE_LEGACY=0 FOAMYADE_IPC_TIMEOUT_MS=5000
for w in 2 3; do
  timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29800+w)) tests/native/ipc_selftest_worker.py 2>&1 | grep -v "^W0\|^\*\*\*\|OMP_NUM" | tail -8
  echo "world $w rc=${PIPESTATUS[0]}"
done
