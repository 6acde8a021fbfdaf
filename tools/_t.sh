#!/bin/bash
# scratch driver for one gpurun call (development)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_locate_paths.py tests/test_graded_mesh.py -m gpu -x -q > gpurun_out/gpu_lp.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_lp.log
grep -E "passed|failed|rc=|^E  " gpurun_out/gpu_lp.log | tail -6
for cap in -1 8 10 12 14; do
  envs=""; [ $cap -ge 0 ] && envs="FOAMYADE_LOCATE_STACK=$cap"
  KSTATS_TOP=40 bash tools/kstats.sh ldu_c3_$cap $envs -- python /root/repo/tools/ldu_bench.py 160 5 wavy 10000000 mg 1e-6 pimple 2>&1 | grep -E "===|k_locate"
  grep '"tool"' gpurun_out/ks_ldu_c3_$cap/run.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   ms_particle', d['ms_particle'], 'ms_step', d['ms_per_step_stream'])"
  python3 - gpurun_out/ks_ldu_c3_$cap <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
d=[ (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(f)) if "k_locate" in r["Kernel_Name"]]
print("   launches (us):", [round(x) for x in d][:8])
PY
  rm -rf gpurun_out/ks_ldu_c3_$cap
done
