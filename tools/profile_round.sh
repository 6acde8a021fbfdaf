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
# bench.py runs its sub-records (C2, general mesh, Laplacian probe) in child processes, each with its own csv: the timed loop's is the one that holds k_locate_deposit most often
main=$(python - $O/ks <<'PY'
import csv, glob, sys
best = (-1, "")
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    calls = sum(int(r["Calls"]) for r in csv.DictReader(open(f)) if "k_locate_deposit(" in r["Name"])
    best = max(best, (calls, f))
print(best[1])
PY
)
echo "kernel stats of the timed loop: $main"
cp $main $O/bench_kernel_stats.csv
python tools/kernel_stats_by_grid.py ${main%kernel_stats.csv}kernel_trace.csv $O/bench_kernel_stats_by_grid.csv
rm -rf $O/ks
PMCP_NAME=$RND/pmcp bash tools/pmc_particles.sh > /dev/null 2>&1
python tools/pmc_particles_report.py $O/pmcp > $O/pmc_particles.txt 2>&1
rm -rf $O/pmcp
ls -la $O
