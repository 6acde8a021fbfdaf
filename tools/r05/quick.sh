#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fv_parity.py -m gpu -x -q -k "cavity or coupled_steps_match or c2_channel or c5_fluid or corrector_counts or slip or graded_cavity" 2>&1 | tail -4
bash tools/r05/ab.sh - FOAMYADE_FACES_FROM_ARRAYS=1 FOAMYADE_NO_FUSED_CORRECTOR=1
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras"
O=$GRAFT_REPO_ROOT/gpurun_out/r05_q; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -- $B --steps 6 --warmup 3 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv; rm -rf $O/kt
python $GRAFT_REPO_ROOT/tools/step_trace.py $O/kernel_trace.csv 2 | grep -v "k_mg_\|k_reduce\|k_pcg\|k_p_apply"
