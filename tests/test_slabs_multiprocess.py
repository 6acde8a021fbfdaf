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


@pytest.mark.parametrize("world,solver", [(2, 1), (3, 1), (2, 0)])
def test_slab_processes_match_the_single_domain(world, solver):
    r = run(world, solver, 3, 0, 29640 + 3 * world + solver)
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
