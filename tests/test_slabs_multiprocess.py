"""The z-slab solver with one OS PROCESS per slab: same code path as the RCCL deployment except for the transport, which is
the host-staged callback communicator over gloo (fy_comm_create_host) because all ranks share the one GPU of the test box.  What this adds
to tests/test_slabs.py: the in-process communicator serialises every collective behind host barriers, so it cannot show a rank that
issues its collectives in another order or number than its neighbours; separate processes do (a hang = the test's time-out)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run(world, solver, steps, migrate, port, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "native", "slab_worker.py"), str(solver), str(steps), str(migrate)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    assert len(line) == 1, out.stdout[-2000:]
    return json.loads(line[0][7:])


def ipc_selftest(world, port, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", FOAMYADE_IPC_TIMEOUT_MS="8000", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "native", "ipc_selftest_worker.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    import re
    assert sorted(re.findall(r"IPC-SELFTEST OK (\d)", out.stdout)) == [str(r) for r in range(world)], out.stdout[-2000:]      # (two ranks may share a line)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ipc_peer_store_communicator_known_answers(world):
    """fy_comm_create_ipc (direct peer stores into hipIpc-mapped windows) with `world` OS processes on the one GPU of the test box -- the transport an
    xGMI node would run, executed with N > 1: fy_comm_selftest's known answers three times over (the two slots of every channel are reused), then once more
    with 16 KB slots, so that the 200 KB planes of the self-test travel in 13 chunks each way"""
    ipc_selftest(world, 29750 + world)
    ipc_selftest(world, 29760 + world, FOAMYADE_IPC_SLOT_KB="16")


@pytest.mark.parametrize("transport", ["host", "ipc"])
@pytest.mark.parametrize("world,solver", [(2, 1), (3, 1), (2, 0)])
def test_slab_processes_match_the_single_domain(world, solver, transport):
    r = run(world, solver, 3, 0, 29640 + 3 * world + solver + (40 if transport == "ipc" else 0), FOAMYADE_TEST_COMM=transport)
    for s in range(3):
        assert r[f"force_err_s{s}"] <= 1e-6, r
    for nm in (("U", "p", "alpha") if solver else ("U", "p")):
        assert r[f"{nm}_err"] <= (1e-9 if nm == "alpha" else 1e-5), r
    assert r["p_iters_same_on_all_ranks"] and r["comm"]["exchanges"] > 0 and r["comm"]["allreduces"] > 0


def test_migration_with_an_idle_middle_rank():
    """three slabs; particles cross from slab 0 into slab 1 only: the top rank has nothing to send or receive in the migration while its
    neighbour has, and takes part in the collectives all the same (comm.hpp: neighbour_exchange_sized must not return early for a rank
    whose four sizes are zero)"""
    r = run(3, 1, 1, 1, 29671)
    assert r["crossed"] > 20 and r["migrated_total"] == r["n_records"] and r["everybody_on_its_owner"], r
    assert r["force_err_s1"] <= 1e-6 and r["p_iters_same_on_all_ranks"], r


def test_slab_processes_next_to_a_parallel_yade():
    """three slab PROCESSES and two Yade workers whose particles intersect different slabs (the per-rank count of non-empty batches
    differs; in step 1 worker 1 reaches slab 0 only): every rank still walks one batch per worker through its collectives -- no hang -- and
    what the workers get back equals the single domain's answer to the same two workers"""
    r = run(3, 1, 3, 2, 29689)
    for s in range(3):
        assert r[f"force_err_s{s}"] <= 1e-6 and r[f"found_same_s{s}"] and r[f"answers_s{s}"] == [1, 2], r
    assert r["ranks_with_an_empty_batch_s1"] >= 1, r
    for nm in ("U", "p"):
        assert r[f"{nm}_err"] <= 1e-5, r
    assert r["p_iters_same_on_all_ranks"], r


@pytest.mark.parametrize("solver", [1, 0])
def test_overlapped_exchanges_equal_the_serial_schedule_across_processes(solver):
    """round 5: every slab exchange runs beside the interior planes of the sweep that consumes it (or beside independent work); FOAMYADE_HALO_OVERLAP=0
    is the exchange-then-consume schedule.  Fluid only, three processes: the gathered U / p / phi_z are the same BITS either way, the pressure
    solver takes the same iterations, the number of exchanges is the same, and the coupled run (particles) still matches the single domain with
    the serial schedule too"""
    a = run(3, solver, 3, 3, 29711 + solver)
    b = run(3, solver, 3, 3, 29721 + solver, FOAMYADE_HALO_OVERLAP="0")
    assert a["fields_sha"] == b["fields_sha"] and a["p_iters"] == b["p_iters"] and a["p_iters"] > 0, (a, b)
    assert a["comm"]["exchanges"] == b["comm"]["exchanges"] and a["comm"]["allreduces"] == b["comm"]["allreduces"], (a["comm"], b["comm"])
    for nm in ("U", "p"):
        assert a[f"{nm}_err"] <= 1e-5 and b[f"{nm}_err"] <= 1e-5, (a, b)
    assert sum(v[1] for v in a["exchange_wait"].values()) > 0, a["exchange_wait"]        # the waits were sampled
    r = run(2, 1, 2, 0, 29731, FOAMYADE_HALO_OVERLAP="0")
    assert r["force_err_s1"] <= 1e-6 and r["U_err"] <= 1e-5, r


@pytest.mark.parametrize("solver", [1, 0])
def test_peer_store_transport_gives_the_host_staged_transport_s_bits(solver):
    """the same slab processes over the two inter-process transports -- planes staged through host memory and gloo (fy_comm_create_host) against kernels
    storing straight into the neighbour's device window (fy_comm_create_ipc) -- with the overlapped schedule, whose exchanges run on the auxiliary stream (the
    communicator's second lane) beside the main stream's all-reduces.  Two processes: the gathered U / p / phi_z carry the same SHA-256, the pressure solver
    takes the same iterations, the same collectives were issued (a sum of two is the same in either order; from three ranks on gloo's reduction order is its own
    and the peer-store transport folds in rank order, as the in-process group and the all-gather-and-fold of Comm::allreduce_ops do).  Three processes: the
    peer-store transport with 8 MB slots against itself with 16 KB slots -- every 5-plane group in chunks -- bit for bit, and both within 1e-5 of the single domain"""
    a = run(2, solver, 3, 3, 29741 + solver, FOAMYADE_TEST_COMM="host")
    b = run(2, solver, 3, 3, 29751 + solver, FOAMYADE_TEST_COMM="ipc", FOAMYADE_IPC_TIMEOUT_MS="8000")
    assert a["fields_sha"] == b["fields_sha"] and a["p_iters"] == b["p_iters"] and a["p_iters"] > 0, (a, b)
    assert a["comm"]["exchanges"] == b["comm"]["exchanges"] and a["comm"]["allreduces"] == b["comm"]["allreduces"] and a["comm"]["allgathers"] == b["comm"]["allgathers"], (a["comm"], b["comm"])
    c = run(3, solver, 3, 3, 29761 + solver, FOAMYADE_TEST_COMM="ipc", FOAMYADE_IPC_TIMEOUT_MS="8000")
    d = run(3, solver, 3, 3, 29771 + solver, FOAMYADE_TEST_COMM="ipc", FOAMYADE_IPC_SLOT_KB="16", FOAMYADE_IPC_TIMEOUT_MS="8000")
    assert c["fields_sha"] == d["fields_sha"] and c["p_iters"] == d["p_iters"] and c["comm"] == d["comm"], (c, d)
    for r in (b, c, d):
        for nm in ("U", "p"):
            assert r[f"{nm}_err"] <= 1e-5, r
    assert sum(v[1] for v in c["exchange_wait"].values()) > 0, c["exchange_wait"]        # the waits were sampled


def test_migration_and_a_parallel_yade_over_the_peer_store_transport():
    """particle migration (sized exchanges, an idle middle rank) and the parallel-Yade protocol with per-rank batch counts that differ, over fy_comm_create_ipc"""
    r = run(3, 1, 1, 1, 29791, FOAMYADE_TEST_COMM="ipc", FOAMYADE_IPC_TIMEOUT_MS="8000")
    assert r["crossed"] > 20 and r["migrated_total"] == r["n_records"] and r["everybody_on_its_owner"], r
    assert r["force_err_s1"] <= 1e-6 and r["p_iters_same_on_all_ranks"], r
    r = run(3, 1, 3, 2, 29781, FOAMYADE_TEST_COMM="ipc", FOAMYADE_IPC_TIMEOUT_MS="8000")
    for s in range(3):
        assert r[f"force_err_s{s}"] <= 1e-6 and r[f"found_same_s{s}"] and r[f"answers_s{s}"] == [1, 2], r
    assert r["p_iters_same_on_all_ranks"], r


def test_bench_goes_over_peer_stores_when_the_rccl_set_up_fails():
    """bench.py --gpus 2 with the RCCL slab set-up made to fail (FOAMYADE_BENCH_FAIL_RCCL): the ranks agree over the gloo group beside the nccl one, take the library's
    peer-store transport as the second choice -- the same two z-slabs, not N independent replicas -- and say so in the line's config.parallelism"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", FOAMYADE_BENCH_FAIL_RCCL="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--n", "32", "--particles", "60000", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, out.stdout[-2000:]
    j = json.loads(line[0])
    par = j["config"]["parallelism"]
    assert j["n_gpus"] == 2 and "2 z-slabs" in par and "fy_comm_create_ipc" in par and "SECOND CHOICE" in par and "FALLBACK" not in par, par
    assert j["value"] > 0 and j["config"]["global_cells"] == 2 * 32 ** 3
