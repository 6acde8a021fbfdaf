#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r2h
timeout 900 python bench.py > gpurun_out/r2h/c3.json 2> gpurun_out/r2h/c3.err; echo "c3 rc $?"; tail -3 gpurun_out/r2h/c3.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2h/c3.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "per_step_ms", "p_iters_per_step", "whole_step", "drop_in_path", "cpu_baseline"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms")})
print("lap", {k: d["roofline_pEqn_laplacian"][k] for k in ("kernel", "frac", "avg_launch_ms")})
PY
timeout 600 python bench.py --config c2 --steps 10 --warmup 3 > gpurun_out/r2h/c2.json 2> gpurun_out/r2h/c2.err; echo "c2 rc $?"; tail -3 gpurun_out/r2h/c2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2h/c2.json").read().strip().splitlines()[-1])
for k in ("metric", "value", "ms_per_step", "per_step_ms", "p_iters_per_step", "u_iters_per_step", "whole_step", "drop_in_path", "cpu_baseline", "kernels"):
    print(k, d.get(k))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms")})
PY
