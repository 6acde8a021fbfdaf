#!/bin/bash
cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/tools/bench_particles.py
KSTATS_TOP=4 bash tools/kstats.sh wavy -- python $B --mesh wavy --steps 6
grep "^step" gpurun_out/ks_wavy/run.log | tail -2
KSTATS_TOP=6 bash tools/kstats.sh block -- python $B --steps 6 --vel 0.05
grep "^step" gpurun_out/ks_block/run.log | tail -2
timeout 1500 python -m pytest tests/test_particle_parity.py tests/test_locate_paths.py tests/test_graded_mesh.py tests/test_ldu_parity.py tests/test_bench_size_parity.py -m gpu -x -q 2>&1 | tail -3
python tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple 2>&1 | tail -1 | cut -c1-400
