#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
( time timeout 900 python bench.py ) > $O/bench.txt 2> $O/bench.err; tail -1 $O/bench.txt > $O/bench_line.json; tail -5 $O/bench.err
python - <<'PY'
import json
j=json.load(open("gpurun_out/r4g/bench_line.json"))
print("value", j["value"], "ms", j["ms_per_step"], j["per_step_ms"])
print("roofline", {k:j["roofline"][k] for k in ("kernel","frac","compulsory_frac","traffic","traffic_reads_undoubled","avg_launch_ms")})
print("lap", {k:j["roofline_pEqn_laplacian"].get(k) for k in ("kernel","frac","traffic","avg_launch_ms")})
print("past", j["roofline_pEqn_laplacian"].get("past_infinity_cache"))
print("c2", j.get("c2"))
print("moving", j.get("moving",{}).get("value"), j.get("moving",{}).get("per_step_ms"))
for k in ("drop_in_path","drop_in_path_one_receiving_rank","drop_in_path_in_process_peer"):
    d=j.get(k,{}); print(k, d.get("ms_per_step"), d.get("per_step_ms"), d.get("error"))
print("cpu", j.get("cpu_baseline",{}).get("value"))
PY
