#!/bin/bash
export RND=r04
bash $GRAFT_REPO_ROOT/tools/profile_round.sh > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python tools/comm_count.py 160 2 10000000 4 > gpurun_out/r04/collectives_2x160.txt 2>&1
timeout 600 python tools/comm_count.py 160 2 10000000 4 0.05 > gpurun_out/r04/collectives_2x160_moving.txt 2>&1
timeout 900 python tools/virtual_slab_bench.py 2 6 > gpurun_out/r04/virtual_slabs_2.txt 2>&1
ls -la gpurun_out/r04; tail -c 400 gpurun_out/r04/bench_line.json
