#!/bin/bash
# round-5 baseline (on the GPU box): bench line of the C3 headline alone, kernel trace of it, and the PMC passes per FV kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-base}
O=$R/gpurun_out/r05_$TAG
rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving --no-extras"
$B --steps 40 --warmup 5 > $O/bench_line.json 2> $O/bench.err
tail -c 1500 $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B --steps 6 --warmup 3 > $O/kt.log 2>&1
cp $(find $O/kt -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/kt
python $R/tools/step_trace.py $O/kernel_trace.csv 2 > $O/step_trace.txt 2>&1
tail -3 $O/step_trace.txt
if [ "${PMC:-1}" = "1" ]; then
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc/$tag -- $B --steps 3 --warmup 2 > $O/pmc_$tag.log 2>&1
  f=$(find $O/pmc/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/r05/pmc_compact.py $f $O/pmc_$tag.csv
done
rm -rf $O/pmc
fi
ls -la $O
