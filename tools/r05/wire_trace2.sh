#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 FOAMYADE_WIRE_TRACE=1 WIRE_BENCH_VERBOSE=1
A="tools/native/wire_bench 160 10000000 4 1e-4 - 7"
/opt/conda/bin/mpiexec -n 1 $A : -n 6 $A : -n 7 $A 2>&1 | tail -9
