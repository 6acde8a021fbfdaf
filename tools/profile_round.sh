# round profile: bench line, rocprofv3 kernel stats (+ per-grid-size table), PMC traffic passes; outputs under gpurun_out/$RND, copied to profiles/ by hand
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RND=${RND:-r02}
O=$R/gpurun_out/$RND
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 400 $O/bench_line.json
python $R/bench.py --config c2 > $O/bench_line_c2.json 2> $O/bench_c2.err
python $R/bench.py --moving --no-cpu-baseline --wire 0 > $O/bench_line_moving.json 2> $O/bench_moving.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu-baseline --wire 0 > $O/ks.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --wire 0 > $O/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --wire 0 > $O/write.log 2>&1
cd $R
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python tools/kernel_stats_by_grid.py $(find $O/ks -name "*kernel_trace.csv" | head -1) $O/bench_kernel_stats_by_grid.csv
python tools/pmc_traffic.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > /dev/null
rm -rf $O/ks $O/fetch $O/write
PMCP_NAME=$RND/pmcp bash tools/pmc_particles.sh > /dev/null 2>&1
python tools/pmc_particles_report.py $O/pmcp > $O/pmc_particles.txt 2>&1
rm -rf $O/pmcp
ls -la $O
