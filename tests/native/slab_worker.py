"""One rank of a multi-PROCESS z-slab run (launched by tests/test_slabs_multiprocess.py under torch.distributed.run, gloo): every rank is
its own OS process with its own solver object on the SAME GPU, the planes travel through the host-staged communicator
(fy_comm_create_host over gloo).  Unlike the in-process virtual slabs, nothing serialises the ranks here but the collectives themselves,
so a rank that issues them in another order or number than its neighbours hangs or fails the way it would under RCCL.
Rank 0 also runs the single-domain case and prints one JSON line with the differences."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import conftest  # noqa: E402
import golden_cases as gc  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    solver_kind, steps, migrate = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    prod = conftest.load_product()
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # FOAMYADE_TEST_COMM=ipc: the peer-store transport (fy_comm_create_ipc) instead of the host-staged one; gloo then carries its bootstrap only
    comm = prod.GlooIpcComm(dist, 0) if os.environ.get("FOAMYADE_TEST_COMM", "host") == "ipc" else prod.GlooHostComm(dist)
    n = 16 if migrate == 3 else 12                    # (mode 3: planes of 256 cells = whole blocks, so that the sweeps' plane windows are active)
    nz = 12 * world
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver_kind == 1 else {}
    u_val = [(0, 0, 0)] * 6
    if solver_kind == 0:
        u_val[3] = (1.0, 0, 0)
    case = prod.make_case(solver_kind, n, n, nz, dx, 2e-4, 1e-5 if solver_kind else 0.01, u_bc=[0] * 6, u_val=u_val, **kw)
    mine = prod.Solver(case, device=0, comm=comm.handle) if migrate != 2 else None        # (mode 2 creates its solvers with a transport)
    one = prod.Solver(case, device=0) if rank == 0 and migrate != 2 else None
    gcase = gc.Case("s", n, n, nz, 0.1, gaussian=solver_kind, np_=3000, seed=21, cluster=200, fast=20, vel_scale=0.05)
    out = {}
    import torch
    if migrate == 2:
        # ---- a PARALLEL Yade next to the slab processes (FoamYade.C:77-111, 114-155): two workers; each rank sends the box of its own slab and
        # gets, per worker, the particles inside it -- worker 1's particles reach slab 0 only in step 1, so the other ranks walk an empty
        # batch through their collectives.  The workers are emulated per process (same deterministic records everywhere); the dt handshake's
        # bcast_local goes over torch.distributed
        from fake_parallel_yade import FakeParallelYade

        def bcast_local(rk, view, root):
            t = torch.from_numpy(view.copy())
            dist.broadcast(t, root)
            view[:] = t.numpy()

        W = 2
        peer = FakeParallelYade(prod, W, rank, world, bcast_local)
        mine = prod.Solver(case, transport=peer.T, device=0, comm=comm.handle)
        one_peer = FakeParallelYade(prod, W, 0, 1) if rank == 0 else None
        one = prod.Solver(case, transport=one_peer.T, device=0) if rank == 0 else None
        for step in range(steps):
            rec = gc.particle_records(gcase, step)
            rec = rec[(rec[:, 2] >= 0) & (rec[:, 2] <= nz * dx) & np.all(rec[:, 0:2] >= 0, axis=1) & np.all(rec[:, 0:2] <= n * dx, axis=1)]
            rec = np.ascontiguousarray(rec[np.argsort(rec[:, 2] + 0.3 * nz * dx * np.sin(7.0 * rec[:, 0] / dx), kind="stable")])
            if step == 1:
                rec[:rec.shape[0] // W, 2] = np.minimum(rec[:rec.shape[0] // W, 2], 10.5 * dx)
            rec[-5:, 2] = n * dx                                   # on the first interface: sent to both neighbours, located by one
            peer.set_records(rec)
            mine.step()
            f, a, F = peer.gathered()
            tf, ta, tF = torch.from_numpy(f.astype(np.float64)), torch.from_numpy(a.astype(np.float64)), torch.from_numpy(np.ascontiguousarray(F))
            dist.all_reduce(tf); dist.all_reduce(ta); dist.all_reduce(tF)
            empty = torch.tensor([float(len(peer.sel.get(1, [])) == 0)], dtype=torch.float64)
            dist.all_reduce(empty)
            if one is not None:
                one_peer.set_records(rec)
                one.step()
                f1, a1, F1 = one_peer.gathered()
                sc = np.abs(F1).max()
                out[f"force_err_s{step}"] = float(np.abs(tF.numpy() - F1).max() / sc)
                out[f"found_same_s{step}"] = bool(np.array_equal(tf.numpy(), f1.astype(np.float64)))
                out[f"answers_s{step}"] = [int(ta.min().item()), int(ta.max().item())]
                out[f"ranks_with_an_empty_batch_s{step}"] = int(empty.item())
        migrate = 0
        steps = 0
    if migrate == 3:
        # ---- fluid only (no particles: nothing is summed in an order that changes from run to run), a lid moving on y+ / a bed's gravity: the fields'
        # SHA-256 after `steps` steps, for the A/B of FOAMYADE_HALO_OVERLAP (tests/test_slabs_multiprocess.py) -- the schedules must give the same bits
        import hashlib
        U0 = np.random.RandomState(5).rand(n * n * nz, 3) * 0.05
        per = n * n * (nz // world)
        mine.set("U", U0[rank * per:(rank + 1) * per])
        mine.enable_exchange_timing(True)
        for _ in range(steps):
            mine.step()
        h = hashlib.sha256()
        for nm in ("U", "p", "phi_z"):
            part = torch.from_numpy(np.ascontiguousarray(mine.get(nm)))
            parts = [torch.zeros_like(part) for _ in range(world)]
            dist.all_gather(parts, part)
            for q in parts:
                h.update(q.numpy().tobytes())
        out["fields_sha"] = h.hexdigest()
        out["p_iters"] = int(mine.stats()["p_iters_total"])
        out["exchange_wait"] = {k: list(v) for k, v in mine.exchange_wait().items()}
        migrate = 0
        steps = 0
        if one is not None:
            one.set("U", U0)
            for _ in range(int(sys.argv[2])):
                one.step()
    if migrate:
        # each rank holds the particles of its own slab; those near the top of slab 0 have moved up across the interface.  Rank 1 receives
        # them, rank 2 (and every rank above) neither sends nor receives anything -- and must take part in the migration's collectives anyway
        rec = gc.particle_records(gcase, 0)
        rec = rec[(rec[:, 2] > 0) & (rec[:, 2] < nz * dx)]
        slab_h = n * dx
        home = np.minimum((rec[:, 2] / slab_h).astype(int), world - 1)
        moved = rec.copy()
        up = (home == 0) & (rec[:, 2] > slab_h - 2.0 * dx)
        moved[up, 2] += 2.5 * dx
        ids = np.arange(rec.shape[0], dtype=np.int64)
        mine.set_particles(moved[home == rank])
        n_local, tags = mine.migrate(ids[home == rank])
        cnt = torch.tensor([float(n_local), float(up.sum()) if rank == 0 else 0.0], dtype=torch.float64)
        dist.all_reduce(cnt)
        out["migrated_total"], out["n_records"], out["crossed"] = cnt[0].item(), int(rec.shape[0]), cnt[1].item()
        new_home = np.minimum((moved[tags, 2] / slab_h).astype(int), world - 1)
        out_ok = torch.tensor([float(np.all(new_home == rank))], dtype=torch.float64)
        dist.all_reduce(out_ok, op=dist.ReduceOp.MIN)
        out["everybody_on_its_owner"] = bool(out_ok.item())
        mine.step()
        f = np.zeros((rec.shape[0], 6))
        f[tags] = mine.forces()
        f = torch.from_numpy(f)
        dist.all_reduce(f)
        if one is not None:
            one.set_particles(moved); one.step()
            fo = one.forces()
            out["force_err_s1"] = float(np.abs(f.numpy() - fo).max() / np.abs(fo).max())
    for step in range(0 if migrate else steps):
        rec = gc.particle_records(gcase, step)
        rec = rec[(rec[:, 2] > 0) & (rec[:, 2] < nz * dx)]
        # every rank is handed the full set (the serial-Yade broadcast): the ownership rule leaves each particle to one rank
        mine.set_particles(rec)
        mine.step()
        f = torch.from_numpy(np.ascontiguousarray(mine.forces()))
        dist.all_reduce(f)                                   # the protocol's all-reduce: non-owners contribute zeros
        if one is not None:
            one.set_particles(rec); one.step()
            fo = one.forces()
            sc = np.abs(fo).max()
            out[f"force_err_s{step}"] = float(np.abs(f.numpy() - fo).max() / sc)
    st = mine.stats()
    iters = torch.tensor([float(st["p_iters_total"])], dtype=torch.float64)
    lo, hi = iters.clone(), iters.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    out["p_iters_same_on_all_ranks"] = bool(lo.item() == hi.item())
    if not migrate:
        for nm in (("U", "p", "alpha") if solver_kind else ("U", "p")):
            part = torch.from_numpy(np.ascontiguousarray(mine.get(nm)))
            parts = [torch.zeros_like(part) for _ in range(world)]
            dist.all_gather(parts, part)
            if one is not None:
                full = np.concatenate([q.numpy() for q in parts])
                ref = one.get(nm)
                out[f"{nm}_err"] = float(np.abs(full - ref).max() / (np.abs(ref).max() + 1e-300))
    out["comm"] = comm.stats()
    mine.close()
    if one is not None:
        one.close()
        print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
