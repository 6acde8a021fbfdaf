"""N > 1 path of bench.py on CPU: two gloo ranks, the contract's timing rule (max over ranks) and the whole-job aggregate.  (The sharded
solver itself -- N z-slabs, one process each -- needs a GPU: tests/test_slabs.py and tests/test_slabs_multiprocess.py.)"""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo")
    r = dist.get_rank()
    elapsed = 1.0 + 0.5 * r                      # rank 1 is the slow one
    e = bench.max_over_ranks(elapsed, dist, torch.device("cpu"))
    v = bench.aggregate_value(dist.get_world_size(), 10, e)
    if r == 0:
        print(json.dumps({"elapsed": e, "value": v}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_gloo_ranks_take_the_max(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29617", str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1                        # only rank 0 prints
    j = json.loads(line[0])
    assert j["elapsed"] == 1.5
    assert abs(j["value"] - 2 * 10 / 1.5) < 1e-12


def test_plain_command_line_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher (the shape of the driver's N = 1 command) re-runs itself under torch.distributed.run,
    one rank per GPU, with its own arguments -- instead of dying on WORLD_SIZE != --gpus"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["FOAMYADE_BENCH_LAUNCH_PROBE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    got = sorted((json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")), key=lambda j: j["rank"])
    assert [j["rank"] for j in got] == [0, 1] and all(j["world"] == 2 and j["gpus"] == 2 and j["steps"] == 7 for j in got)
