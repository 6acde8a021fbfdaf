cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 600 $O/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu-baseline > $O/ks.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
find $O -name "*.csv" | head -20
