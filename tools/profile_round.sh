# round profile (run on the GPU box through gpurun): bench lines of the three single-GPU configurations, rocprofv3 kernel stats of the default
# command (+ per-grid-size table), HBM traffic from PMC passes (bench.py collects them itself: --pmc); outputs under gpurun_out/$RND,
# copied to profiles/ by hand
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RND=${RND:-r03}
O=$R/gpurun_out/$RND
rm -rf $O; mkdir -p $O
python $R/bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 300 $O/bench_line.json
python $R/bench.py --config c2 --wire 0 > $O/bench_line_c2.json 2> $O/bench_c2.err
python $R/bench.py --config c5 --steps 5 --warmup 2 --pmc 1 --wire 0 > $O/bench_line_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu-baseline --wire 0 --pmc 0 --no-moving > $O/ks.log 2>&1
cd $R
cp $(find $O/ks -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python tools/kernel_stats_by_grid.py $(find $O/ks -name "*kernel_trace.csv" | head -1) $O/bench_kernel_stats_by_grid.csv
rm -rf $O/ks
PMCP_NAME=$RND/pmcp bash tools/pmc_particles.sh > /dev/null 2>&1
python tools/pmc_particles_report.py $O/pmcp > $O/pmc_particles.txt 2>&1
rm -rf $O/pmcp
ls -la $O
