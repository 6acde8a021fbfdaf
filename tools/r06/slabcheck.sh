#!/bin/bash
# after a change to the slab path: the slab tests (in-process, multi-process over both transports, the full-size cuts), then the kernel-excess table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_slabs.py tests/test_slabs_multiprocess.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 1800 python -m pytest tests/test_full_size_properties.py -m gpu -x -q -k "slabs" 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/r05/slab_kernels.sh 2>&1 | tail -48
