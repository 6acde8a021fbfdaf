#!/bin/bash
# usage: tools/r05/gpu.sh <timeout_s> <logname> '<command>'   -- gpurun with retries while the pool is busy (exit code 3)
T=$1; L=$2; shift 2
mkdir -p /root/repo/gpurun_out
for try in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > /root/repo/gpurun_out/$L.log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
