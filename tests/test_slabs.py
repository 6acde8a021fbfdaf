"""z-slab decomposition tested with N virtual slabs on ONE GPU: the solver code is the one each RCCL rank runs, only
the communicator back-end differs (device-to-device copies between threads instead of ncclSend/ncclRecv).  A decomposed run must
reproduce the single-domain run: bit-for-bit where no reduction order changes, to solver tolerance otherwise."""
import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu


def cavity(product, solver, n, nz, p_solver=1, **kw):
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (1.0, 0, 0)
    return product.make_case(solver, n, n, nz, 1.0 / n, 0.4 / n, 0.01, u_bc=[0] * 6, u_val=u_val, p_solver=p_solver, **kw)


def compare(a, b, names, rtol):
    for nm in names:
        x, y = a.get(nm), b.get(nm)
        sc = np.abs(y).max() + 1e-300
        assert x.shape == y.shape, nm
        assert np.abs(x - y).max() <= rtol * sc, (nm, np.abs(x - y).max() / sc)


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("n_slabs,p_solver", [(2, 0), (2, 1), (4, 1)])
def test_fv_slabs_match_single_domain(product, solver, n_slabs, p_solver):
    n, nz = 16, 40 if solver == 1 else 16            # pimple slabs must be at least 5 planes thick (particle halo), even
    if solver == 1:
        nz = 20 * (n_slabs // 2) * 2 if n_slabs > 2 else 24
    case = cavity(product, solver, n, nz, p_solver=p_solver)
    one = product.Solver(case)
    many = product.VirtualSlabs(case, n_slabs)
    rs = np.random.RandomState(5)
    U0 = rs.rand(n * n * nz, 3) * 0.1
    one.set("U", U0); many.set("U", U0)
    for _ in range(4):
        one.step(); many.step()
        so, sm = one.stats(), many.stats()
        for r in range(n_slabs):                     # every rank sees the same all-reduced scalars
            assert sm[r]["p_iters_total"] == sm[0]["p_iters_total"]
            assert sm[r]["courant_max"] == sm[0]["courant_max"]
        assert abs(so["p_iters_total"] - sm[0]["p_iters_total"]) <= 2
        assert np.isclose(so["courant_max"], sm[0]["courant_max"], rtol=1e-12)
    compare(many, one, ("U", "p", "phi_x", "phi_y", "phi_z"), 5e-6)
    many.close(); one.close()


def test_first_step_operators_are_identical(product):
    """before any iterative solve the decomposition must be exact: matrices and the SpMV do not depend on reduction order"""
    n, nz = 12, 16
    case = cavity(product, 0, n, nz, p_solver=0)
    one = product.Solver(case); many = product.VirtualSlabs(case, 2)
    rs = np.random.RandomState(7)
    U0 = rs.rand(n * n * nz, 3) * 0.2
    one.set("U", U0); many.set("U", U0)
    one.step(); many.step()
    for nm in ("p_diag", "p_ux", "p_uy", "p_uz", "mom_diag", "rAU"):
        np.testing.assert_array_equal(many.get(nm), one.get(nm), err_msg=nm)
    many.close(); one.close()


@pytest.mark.parametrize("solver,n_slabs,models", [(1, 2, 0), (1, 3, 0), (0, 2, 0), (1, 2, 3), (1, 2, -1), (1, 2, -2), (1, 2, -3)])
def test_coupled_slabs_match_single_domain(product, solver, n_slabs, models):
    """particles near slab interfaces: deposits/gathers reach up to 5 planes into the neighbours
    (models = 3: with the opt-in added-mass / Gaussian-torque models, whose vGrad / ddtU gathers need the same halos; -2 / -3: LES kEqn / RAS kEpsilon;
    models = -1: LES Smagorinsky, whose eddy viscosity is interpolated across the slab faces -- a moving lid on y+ makes it matter)"""
    n = 12
    nz = 12 * n_slabs
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else {}
    u_val = [(0, 0, 0)] * 6
    if solver == 0:
        u_val[3] = (1.0, 0, 0)
    if models < 0:                                   # -1 LES Smagorinsky, -2 LES kEqn (its k equation is solved across the slab faces)
        kw.update(turbulence_model=-models, les_ck=0.3, nut_initial=1e-5, nut_bc=[0, 0, 1, 0, 0, 1], nut_value=[0, 0, 0.0, 0, 0, 2e-5])
        if models <= -2:                             # (-3: RAS kEpsilon, two equations)
            kw.update(k_initial=4e-4, k_bc=[0, 0, 1, 0, 1, 0], k_value=[0, 0, 1e-4, 0, 1e-4, 0], k_convection_scheme=0, k_tol=1e-10)
        if models == -3:
            kw.update(eps_initial=3e-3, eps_bc=[0, 0, 0, 1, 0, 1], eps_value=[0, 0, 0, 2e-3, 0, 4e-3], eps_tol=1e-10)
        u_val[3] = (0.5, 0, 0)
        models = 0
    case = product.make_case(solver, n, n, nz, dx, 2e-4, 1e-5 if solver else 0.01, u_bc=[0] * 6, u_val=u_val, **kw)
    one = product.Solver(case); many = product.VirtualSlabs(case, n_slabs)
    if case.turbulence_model:
        assert np.all(many.get("nut") == 1e-5)
    if models:
        one.set_force_models(models)
        for sl in many.solvers:
            sl.set_force_models(models)
    gcase = gc.Case("s", n, n, nz, 0.1, gaussian=solver, np_=4000, seed=21, cluster=200, fast=20, vel_scale=0.05)
    for step in range(3):
        rec = gc.particle_records(gcase, step)
        rec = rec[(rec[:, 2] > 0) & (rec[:, 2] < nz * dx)]          # keep the probes inside the block in z
        one.set_particles(rec); many.set_particles(rec)
        one.step(); many.step()
        fo, fm = one.forces(), many.forces()
        for cols in ((slice(0, 3), slice(3, 6)) if models else (slice(0, 6),)):
            sc = np.abs(fo[:, cols]).max()
            assert sc > 0 and np.abs(fm[:, cols] - fo[:, cols]).max() <= 1e-6 * sc, (cols, np.abs(fm[:, cols] - fo[:, cols]).max() / sc)
    compare(many, one, ("U", "p", "alpha", "uSource") if solver else ("U", "p"), 1e-5)
    if case.turbulence_model:
        assert one.get("nut").max() > 1e-6
        compare(many, one, {1: ("nut",), 2: ("nut", "k"), 3: ("nut", "k", "epsilon")}[case.turbulence_model], 2e-5)
    many.close(); one.close()


@pytest.mark.parametrize("solver,n_slabs", [(1, 2), (1, 3), (0, 2)])
def test_particle_ownership_when_every_slab_gets_all_particles(product, solver, n_slabs):
    """owner = the slab that holds the particle's containing cell.  Every rank is handed the FULL record set (the
    reference's serial-Yade broadcast); exactly one locates each particle, the others report found = -1 and zero force, and the
    coupled result equals the single-domain run."""
    n = 12
    nz = 12 * n_slabs
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else {}
    u_val = [(0, 0, 0)] * 6
    if solver == 0:
        u_val[3] = (1.0, 0, 0)
    case = product.make_case(solver, n, n, nz, dx, 2e-4, 1e-5 if solver else 0.01, u_bc=[0] * 6, u_val=u_val, **kw)
    one = product.Solver(case); many = product.VirtualSlabs(case, n_slabs)
    gcase = gc.Case("s", n, n, nz, 0.1, gaussian=solver, np_=4000, seed=33, cluster=200, fast=20, outside=30, vel_scale=0.05)
    for step in range(2):
        rec = gc.particle_records(gcase, step)
        # particles exactly on a slab interface and just outside the block in z are part of the set
        rec[10:14, 2] = (nz // n_slabs) * dx
        rec[14, 2] = -0.4 * dx
        rec[15, 2] = nz * dx + 0.4 * dx
        one.set_particles(rec); many.set_particles_all(rec)
        one.step(); many.step()
        found = np.stack([s.found() for s in many.solvers])
        assert np.all((found == 1).sum(axis=0) <= 1)                                  # never two owners
        assert np.array_equal((found == 1).sum(axis=0) == 1, one.found() == 1)        # located somewhere <=> located in the single domain
        fo, fm = one.forces(), many.forces()
        sc = np.abs(fo).max()
        assert np.abs(fm - fo).max() <= 1e-6 * sc, np.abs(fm - fo).max() / sc
    compare(many, one, ("U", "p", "alpha", "uSource") if solver else ("U", "p"), 1e-5)
    many.close(); one.close()


@pytest.mark.parametrize("solver,n_slabs", [(1, 2), (1, 3), (0, 2)])
def test_particles_migrate_between_slabs(product, solver, n_slabs):
    """BASELINE north star: 'particles that cross slabs migrate via the same RCCL path'.  Each slab starts with its own particles; they
    random-walk (up to ~1.5 planes per step) across the interfaces; fy_migrate_particles hands the leavers to the neighbour that now
    owns them.  After every migration each slab holds exactly the particles inside its planes, no tag is lost or duplicated, and the
    coupled step equals the single-domain run on the same cloud."""
    n = 12
    nz = 12 * n_slabs
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else {}
    u_val = [(0, 0, 0)] * 6
    if solver == 0:
        u_val[3] = (1.0, 0, 0)
    case = product.make_case(solver, n, n, nz, dx, 2e-4, 1e-5 if solver else 0.01, u_bc=[0] * 6, u_val=u_val, **kw)
    one = product.Solver(case); many = product.VirtualSlabs(case, n_slabs)
    rs = np.random.RandomState(3)
    npart = 3000
    rec = np.zeros((npart, 10))
    rec[:, 0:2] = 0.1 * rs.random_sample((npart, 2))
    rec[:, 2] = nz * dx * (0.05 + 0.9 * rs.random_sample(npart))
    rec[:, 3:6] = 0.02 * rs.standard_normal((npart, 3)); rec[:, 9] = 0.2 * dx
    tags = np.arange(npart, dtype=np.int64) + 1000
    slab_h = (nz // n_slabs) * dx
    owner = np.clip(np.floor(rec[:, 2] / slab_h).astype(int), 0, n_slabs - 1)
    local_tags = [tags[owner == r] for r in range(n_slabs)]
    for r, s in enumerate(many.solvers):
        s.set_particles(rec[owner == r])
    moved_total = 0
    for step in range(4):
        # the DEM moves the particles: every slab updates ITS records in place (here on the host), some leave its planes
        pos = {}
        new_local = []
        for r, s in enumerate(many.solvers):
            mine = s.particles()
            assert mine.shape[0] == local_tags[r].size
            mine[:, 2] = np.clip(mine[:, 2] + 1.5 * dx * rs.standard_normal(mine.shape[0]), 0.01 * dx, nz * dx - 0.01 * dx)
            s.set_particles(mine)
            new_local.append(mine)
        local_tags = many.migrate(local_tags)
        allrec = np.zeros((npart, 10)); seen = np.zeros(npart, dtype=int)
        for r, s in enumerate(many.solvers):
            mine = s.particles()
            assert mine.shape[0] == local_tags[r].size
            kz = np.clip(np.floor(mine[:, 2] / dx).astype(int), 0, nz - 1)
            assert np.all(kz // (nz // n_slabs) == r)                   # everybody is where its particles are
            allrec[local_tags[r] - 1000] = mine
            seen[local_tags[r] - 1000] += 1
            moved_total += int((np.floor(np.concatenate(new_local)[:, 2] / slab_h).astype(int) != np.repeat(np.arange(n_slabs), [a.shape[0] for a in new_local])).sum()) if r == 0 else 0
        assert np.all(seen == 1)                                        # conservation of the population
        one.set_particles(allrec)
        one.step(); many.step()
        fo = one.forces()
        fm = np.zeros_like(fo)
        for r, s in enumerate(many.solvers):
            if local_tags[r].size:
                fm[local_tags[r] - 1000] = s.forces()
        sc = np.abs(fo).max()
        assert np.abs(fm - fo).max() <= 1e-6 * sc, np.abs(fm - fo).max() / sc
    assert moved_total > 20                                             # the walk really crossed interfaces
    compare(many, one, ("U", "p", "alpha", "uSource") if solver else ("U", "p"), 1e-5)
    many.close(); one.close()


@pytest.mark.parametrize("n", [1, 2, 3])
def test_comm_selftest_on_a_local_group(product, n):
    """fy_comm_selftest: the known-answer run of the slab solver's communication pattern (grouped two-field neighbour exchange, sum / max
    all-reduce, all-gather) that bench.py sends through RCCL in throw-away processes before a multi-GPU run -- here over the in-process
    communicators, one host thread per rank"""
    import threading
    comms = product.local_comm_group(n)
    errs = [None] * n

    def run(r):
        try:
            product.comm_selftest(comms[r], 0)
        except Exception as e:                                   # noqa: BLE001
            errs[r] = e
    ts = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    for t in ts: t.start()
    for t in ts: t.join(120)
    assert not any(t.is_alive() for t in ts)
    assert errs == [None] * n, errs


def test_rccl_selftest_child_mode_of_the_bench(product):
    """the child process bench.py's rccl_preflight spawns per rank, here as a world of one: RCCL communicator from a unique id, known-answer
    pattern, exit code 0"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rccl-selftest", product.rccl_unique_id().hex(), "--gpus", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]


def test_a_solver_on_an_rccl_communicator_of_one_rank_steps_like_the_single_domain(product):
    """every RCCL call the slab solver issues -- ncclCommInitRank, ncclCommSplit for the overlapped halo, grouped ncclSend / ncclRecv (none to post on one rank),
    ncclAllReduce, ncclAllGather behind the mixed sum / max diagnostics, the per-phase counters -- on the only world this box can form: one rank.  The slab code path
    (ghost planes, windows, collective diagnostics) then has to reproduce the single-domain run"""
    n, dx = 16, 0.1 / 16
    case = lambda: product.make_case(1, n, n, n, dx, 2e-4, 1e-5, u_bc=[0] * 6, u_val=[(0, 0, 0)] * 6, g=(0, 0, -9.81), p_bc=[2] * 6)
    comm = product.rccl_comm(0, 1, product.rccl_unique_id(), 0)
    product.comm_selftest(comm, 0)
    one = product.Solver(case())
    slab = product.Solver(case(), comm=comm)
    rs = np.random.RandomState(5)
    for _ in range(3):
        rec = np.zeros((3000, 10))
        rec[:, 0:3] = rs.random_sample((3000, 3)) * np.array([0.1, 0.1, 0.06]) + np.array([0.0, 0.0, 0.005])
        rec[:, 3:6] = 0.05 * rs.standard_normal((3000, 3)); rec[:, 9] = 0.2 * dx
        one.set_particles(rec); slab.set_particles(rec)
        one.step(); slab.step()
        sc = np.abs(one.forces()).max()
        assert np.abs(slab.forces() - one.forces()).max() <= 1e-9 * sc
    for nm in ("U", "p"):
        a, b = slab.get(nm), one.get(nm)
        assert np.abs(a - b).max() <= 1e-8 * (np.abs(b).max() + 1e-300), nm
    import ctypes
    buf = ctypes.create_string_buffer(1 << 14)
    assert product.lib().fy_comm_stats_by_tag(comm, buf, len(buf)) == 0
    counts = [tuple(int(x) for x in ln.split()[1:]) for ln in buf.value.decode().splitlines()]
    assert sum(c[1] + c[2] for c in counts) > 0                                # (all-reduces / all-gathers went through RCCL)
    slab.close(); one.close()


@pytest.mark.parametrize("n_slabs,workers", [(2, 2), (3, 4)])
def test_slabs_next_to_a_parallel_yade(product, n_slabs, workers):
    """z-slabs fed by a PARALLEL Yade (FoamYade.C:77-111, 114-155): every solver rank sends the bounding box of its own slab, every Yade
    worker sends each rank the particles inside that rank's box -- so the ranks receive different numbers of non-empty batches (a cluster of
    worker 1's particles sits in slab 0 only, the top slab gets nothing from some workers) and still walk the same W batches through their
    collectives.  What the workers get back equals the single domain's answer to the same workers: forces, found flags, fields, and the
    per-worker order of setCellVolFraction's assignments (the last worker that touched a cell wins, FoamYade.C:318-328)."""
    import threading
    from fake_parallel_yade import FakeParallelYade
    n = 12
    nz = 12 * n_slabs
    dx = 0.1 / n
    case = product.make_case(1, n, n, nz, dx, 2e-4, 1e-5, u_bc=[0] * 6, u_val=[(0, 0, 0)] * 6, g=(0, 0, -9.81), p_bc=[2] * 6)
    c = gc.Case("py", n, n, nz, 0.1, gaussian=1, np_=4000, seed=33, cluster=300, fast=20, vel_scale=0.05)
    bar = threading.Barrier(n_slabs)
    shared = {}

    def bcast_local(rank, view, root):
        if rank == root:
            shared["v"] = view.copy()
        bar.wait()
        if rank != root:
            view[:] = shared["v"]
        bar.wait()

    one_peer = FakeParallelYade(product, workers, 0, 1)
    peers = [FakeParallelYade(product, workers, r, n_slabs, bcast_local) for r in range(n_slabs)]
    one = product.Solver(case, transport=one_peer.T)
    many = product.VirtualSlabs(case, n_slabs, transports=[p.T for p in peers])
    # the boxes the ranks announced: the single domain the whole block, a slab its own planes (and the full cross-section)
    np.testing.assert_allclose(one_peer.bbox, [0, 0, 0, n * dx, n * dx, nz * dx], atol=1e-15)
    for r, p in enumerate(peers):
        np.testing.assert_allclose(p.bbox, [0, 0, r * 12 * dx, n * dx, n * dx, (r + 1) * 12 * dx], rtol=0, atol=1e-12)
    for step in range(3):
        rec = gc.particle_records(c, step)
        rec = rec[(rec[:, 2] >= 0) & (rec[:, 2] <= nz * dx) & np.all(rec[:, 0:2] >= 0, axis=1) & np.all(rec[:, 0:2] <= n * dx, axis=1)]
        order = np.argsort(rec[:, 2] + 0.3 * nz * dx * np.sin(7.0 * rec[:, 0] / dx), kind="stable")      # workers own bands that cross the slabs unevenly
        rec = np.ascontiguousarray(rec[order])
        if step == 1:                                                  # worker 1 only reaches slab 0 in this step
            lo, hi = 0, rec.shape[0] // workers
            rec[lo:hi, 2] = np.minimum(rec[lo:hi, 2], 10.5 * dx)
        rec[-5:, 2] = 12 * dx                                          # (the last worker's) exactly on the first interface: both neighbours are sent these
        for p in peers + [one_peer]:
            p.set_records(rec)
        one.step(); many.step()
        f1, a1, F1 = one_peer.gathered()
        assert np.all(a1 == 1)
        found = sum(p.gathered()[0] for p in peers)
        answers = sum(p.gathered()[1] for p in peers)
        F = sum(p.gathered()[2] for p in peers)
        assert np.all(answers >= 1) and np.all(answers[-5:] == 2) and answers.max() == 2
        assert np.array_equal(found, f1)                              # exactly one rank locates what the single domain locates
        sc = np.abs(F1).max()
        assert np.abs(F - F1).max() <= 1e-6 * sc, np.abs(F - F1).max() / sc
        if step == 1:
            assert len(peers[-1].sel[1]) == 0 and len(peers[0].sel[1]) > 0      # the top slab walked an empty batch for worker 1
        assert all(p.fluid_dt == [] for p in peers[1:]) and len(peers[0].fluid_dt) == step + 1      # the dt handshake is solver rank 0's
    compare(many, one, ("U", "p"), 1e-5)
    compare(many, one, ("alpha",), 1e-9)
    many.close(); one.close()


def _coupled_pimple_run(product, n_slabs, steps=3, n=12):
    nz = 12 * n_slabs
    dx = 0.1 / n
    case = product.make_case(1, n, n, nz, dx, 2e-4, 1e-5, u_bc=[0] * 6, u_val=[(0, 0, 0)] * 6, g=(0, 0, -9.81), p_bc=[2] * 6)
    many = product.VirtualSlabs(case, n_slabs)
    gcase = gc.Case("s", n, n, nz, 0.1, gaussian=1, np_=4000, seed=21, cluster=200, fast=20, vel_scale=0.05)
    per_step = []
    for step in range(steps):
        rec = gc.particle_records(gcase, step)
        rec = rec[(rec[:, 2] > 0) & (rec[:, 2] < nz * dx)]
        many.set_particles(rec)
        before = many.comm_stats_by_tag(0)
        many.step()
        after = many.comm_stats_by_tag(0)
        tot = [sum(after[k][q] - before.get(k, (0, 0, 0))[q] for k in after) for q in range(3)]
        per_step.append((tot, many.stats()[0]))
    return many, per_step


def test_collective_budget_per_step(product):
    """The slab solver's collectives are latency-bound RCCL calls on real hardware: their number per coupled step is a budget,
    asserted here so that it cannot creep back (round 3: ~112 per step at two slabs; round 4: 1 exchange + 1 all-gather + 2 all-reduces per
    PCG iteration, one collective per diagnostics group, no exchange for ghost planes that are still valid).
    pimpleFoamYade, nOuter 1, nCorr 2, momentum predictor with m Jacobi passes, I PCG iterations in the step's two solves:
      exchanges   <= 1 (step start) + 5 (particle phase) + 2 (stress row, rAU) + (m - 1) + 2 x 2 (HbyA, p per corrector) + 1 (U, second corrector)
                     + 1 (operator ghosts) + I
      all-reduces <= 1 (sum U) + m + 1 (reference term) + 2 (initial residuals) + 2 I + 2 (diagnostics groups)
      all-gathers <= 1 (coarse operators) + I"""
    many, per_step = _coupled_pimple_run(product, 2)
    for (ex, ar, ag), st in per_step[1:]:             # (the first step also carries one-off set-up exchanges)
        m, I = st["u_iters_total"] + 1, st["p_iters_total"]
        assert ex <= 14 + (m - 1) + I, (ex, m, I)
        assert ar <= 6 + m + 2 * I, (ar, m, I)
        assert ag <= 1 + I, (ag, I)
        if I <= 4 and m <= 3:
            assert ex + ar + ag <= 60
    many.close()


def test_deep_vcycle_equals_the_exchange_per_sweep_schedule(product, monkeypatch):
    """the communication-avoiding V-cycle recomputes the ghost rows the per-sweep schedule exchanges: same operands, same operations ->
    the pressure solver's iterates, and with them every field, are the same BITS (FOAMYADE_NO_DEEP_VCYCLE=1 is round 3's schedule).
    Fluid only: the particle phase's scatters are summed in an order that changes from run to run."""
    n, nz, n_slabs = 16, 36, 3

    def run():
        case = cavity(product, 1, n, nz, p_solver=1)
        vs = product.VirtualSlabs(case, n_slabs)
        vs.set("U", np.random.RandomState(5).rand(n * n * nz, 3) * 0.1)
        ex0 = vs.comm_stats(0)[0]
        its = []
        for _ in range(4):
            vs.step()
            its.append(vs.stats()[0]["p_iters_total"])
        return vs, vs.comm_stats(0)[0] - ex0, its

    deep, ex_deep, its_deep = run()
    monkeypatch.setenv("FOAMYADE_NO_DEEP_VCYCLE", "1")
    old, ex_old, its_old = run()
    monkeypatch.delenv("FOAMYADE_NO_DEEP_VCYCLE")
    assert ex_deep < ex_old and sum(its_deep) > 0                                     # fewer exchanges ...
    assert its_deep == its_old
    for nm in ("p", "U", "phi_z"):
        np.testing.assert_array_equal(deep.get(nm), old.get(nm), err_msg=nm)            # ... for the same answer
    deep.close(); old.close()


@pytest.mark.parametrize("solver", [0, 1])
def test_overlapped_exchanges_equal_the_serial_schedule(product, solver, monkeypatch):
    """round 5 (north_star: halo exchange overlapped with interior stencil work): every sweep that consumes a slab exchange runs its interior planes beside
    the exchange and its two end planes afterwards (plane windows of whole 256-cell blocks), the particle phase's exchanges run beside independent work;
    FOAMYADE_HALO_OVERLAP=0 is the exchange-then-consume schedule.  Fluid only (nothing is summed in an order that changes from run to run): the same
    BITS, the same iterations, the same number of collectives.  Then with particles: both schedules match the single domain."""
    n, nz, n_slabs = 16, 36, 3                       # planes of 256 cells; 12 planes per slab (> 2 x 5 particle-halo planes)
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else {}

    def run(particles):
        case = cavity(product, solver, n, nz, p_solver=1, **kw)
        vs = product.VirtualSlabs(case, n_slabs)
        vs.set("U", np.random.RandomState(5).rand(n * n * nz, 3) * 0.05)
        ex0 = vs.comm_stats(0)
        gcase = gc.Case("s", n, n, nz, n / 16 * 1.0, gaussian=solver, np_=3000, seed=21, cluster=200, fast=20, vel_scale=0.05)
        its = []
        for step in range(3):
            if particles:
                rec = gc.particle_records(gcase, step)
                vs.set_particles(rec[(rec[:, 2] > 0) & (rec[:, 2] < nz / n)])
            vs.step()
            its.append(vs.stats()[0]["p_iters_total"])
        ex1 = vs.comm_stats(0)
        return vs, [ex1[q] - ex0[q] for q in range(3)], its

    on, ex_on, its_on = run(False)
    monkeypatch.setenv("FOAMYADE_HALO_OVERLAP", "0")
    off, ex_off, its_off = run(False)
    assert its_on == its_off and sum(its_on) > 0 and ex_on == ex_off, (its_on, its_off, ex_on, ex_off)
    for nm in ("U", "p", "phi_x", "phi_y", "phi_z"):
        np.testing.assert_array_equal(on.get(nm), off.get(nm), err_msg=nm)
    on.close(); off.close()
    # with particles: either schedule against the single domain
    serial, _, _ = run(True)
    monkeypatch.delenv("FOAMYADE_HALO_OVERLAP")
    ovl, _, _ = run(True)
    compare(ovl, serial, ("U", "p", "alpha", "uSource") if solver else ("U", "p"), 1e-8)
    serial.close(); ovl.close()


@pytest.mark.parametrize("solver", [0, 1])
def test_stream_ordered_local_group_equals_the_host_synchronous_one(product, solver, monkeypatch):
    """the in-process communicator's host barriers serialise the virtual slabs, which hides a dependency the schedule forgets; with
    FOAMYADE_LOCALCOMM_STREAM=1 its collectives are event waits on the ranks' streams (nothing waits on the host, as under RCCL) and the slabs' kernels
    interleave freely.  The overlapped schedule must give the same BITS either way (fluid only), and the single domain's answer with particles."""
    n, nz, n_slabs = 16, 36, 3
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else {}

    def run(particles):
        case = cavity(product, solver, n, nz, p_solver=1, **kw)
        vs = product.VirtualSlabs(case, n_slabs)
        vs.set("U", np.random.RandomState(5).rand(n * n * nz, 3) * 0.05)
        gcase = gc.Case("s", n, n, nz, n / 16 * 1.0, gaussian=solver, np_=3000, seed=21, cluster=200, fast=20, vel_scale=0.05)
        its = []
        for step in range(4):
            if particles:
                rec = gc.particle_records(gcase, step)
                vs.set_particles(rec[(rec[:, 2] > 0) & (rec[:, 2] < nz / n)])
            vs.step()
            its.append(vs.stats()[0]["p_iters_total"])
        return vs, its

    sync, its_sync = run(False)
    sync_p, _ = run(True)
    monkeypatch.setenv("FOAMYADE_LOCALCOMM_STREAM", "1")
    free, its_free = run(False)
    free_p, _ = run(True)
    monkeypatch.delenv("FOAMYADE_LOCALCOMM_STREAM")
    assert its_sync == its_free and sum(its_sync) > 0
    for nm in ("U", "p", "phi_x", "phi_y", "phi_z"):
        np.testing.assert_array_equal(free.get(nm), sync.get(nm), err_msg=nm)
    compare(free_p, sync_p, ("U", "p", "alpha", "uSource") if solver else ("U", "p"), 1e-8)
    for v in (sync, sync_p, free, free_p):
        v.close()
