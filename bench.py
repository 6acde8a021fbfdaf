#!/usr/bin/env python3
"""Headline benchmark: coupled CFD-DEM steps/s at BASELINE.json configs[2] ("C3"):
pimpleFoamYade 4-way coupling, 10 M particles, 160^3 = 4 096 000 cells, one MI355X.

A "step" is one pass of pimpleFoamYade's time-loop body (pimpleFoamYade.C:60-114): Courant number, pre-coupling fields,
FoamYade::setParticleAction (k-d locate, Gaussian weights, void-fraction deposit, drag + Archimedes, momentum-source
back-scatter for every particle), UcEqn assembly + momentum predictor, nCorrectors pressure correctors (PCG + multigrid),
continuity errors, setSourceZero.  Particle records and all fields are resident in HBM when the timed region starts
(the reference receives particles over MPI on the host; the PCIe-inclusive figure is discussed in DESIGN.md, never here).

Contract: python bench.py --gpus N --steps K --warmup W ; for N>1 launched by torch.distributed.run, one rank per GPU.
N > 1 is ONE coupled simulation cut into N z-slabs (SURVEY.md 8e, config C4 weak series): a 160 x 160 x (160 N) box, each rank owns a
160^3 slab and the 10 M particles inside it; FV halos (1 plane), particle halos (5 planes), reverse-halo sums, Krylov scalars and the
coarse multigrid level travel over RCCL/xGMI (csrc/comm.cpp).  Weak scaling: per-GPU work is fixed as N grows.
`value` counts C3-sized slab-steps per second over the whole job (= N x coupled steps/s of the N-slab box; identical to coupled
steps/s at N = 1).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s copy-achievable)


def c3_case(prod, n, dt, p_solver, n_slabs=1, strong=False):
    """closed box, no-slip walls, g = (0,0,-9.81), nu = 1e-6, rho_p = 2650, rho_f = 1000, PIMPLE nOuter 1 nCorr 2,
    fixedFluxPressure walls (what a DPMFoam case with gravity uses).  n_slabs > 1: the weak-scaling box n x n x (n * n_slabs), or with
    strong=True the C3 box itself (BASELINE configs[3]); the solver cuts either into n_slabs z-slabs."""
    dx = 1.0 / n
    return prod.make_case(prod.FY_SOLVER_PIMPLE, n, n, n if strong else n * n_slabs, dx, dt, 1e-6, rho_f=1000.0, rho_p=2650.0, g=(0.0, 0.0, -9.81),
                          u_bc=[prod.FY_BC_U_FIXED_VALUE] * 6, u_val=[(0, 0, 0)] * 6, p_bc=[prod.FY_BC_P_FIXED_FLUX] * 6,
                          n_outer_correctors=1, n_correctors=2, p_solver=p_solver)


def c3_particles(torch, n_part, n, seed, device, slab=0):
    """Np particles uniform in the lower 60 % of this rank's unit-cube slab, r = 0.2 dx, at rest."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    dx = 1.0 / n
    rec = torch.zeros(n_part, 10, dtype=torch.float64)
    rec[:, 0:3] = torch.rand(n_part, 3, dtype=torch.float64, generator=g)
    rec[:, 2] *= 0.6
    rec[:, 2] += float(slab)
    rec[:, 9] = 0.2 * dx
    return rec.to(device).contiguous()


def c3_particles_strong(torch, n_part, n, device, rank, world):
    """BASELINE configs[3]: the ONE C3 cloud (seed 3, lower 60 % of the unit box); this rank keeps the particles inside its z-slab.  The upper
    slabs hold few or none -- that imbalance is the configuration's, not an artefact."""
    g = torch.Generator(device="cpu").manual_seed(3)
    dx = 1.0 / n
    pos = torch.rand(n_part, 3, dtype=torch.float64, generator=g)
    pos[:, 2] *= 0.6
    h = 1.0 / world
    mine = (pos[:, 2] >= rank * h) & (pos[:, 2] < (rank + 1) * h)
    rec = torch.zeros(int(mine.sum()), 10, dtype=torch.float64)
    rec[:, 0:3] = pos[mine]
    rec[:, 9] = 0.2 * dx
    return rec.to(device).contiguous()


def cpu_baseline(n_sample, ppc, dt, threads):
    """the CPU oracle (a faithful port of the reference's path, kind = "port") on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    dx = 1.0 / n_sample
    case = orc.fv_case(1, n_sample, n_sample, n_sample, dx, dt, 1e-6, g=(0, 0, -9.81), p_bc=[orc.P_FIXEDFLUX] * 6, n_outer=1, n_corr=2, p_solver=1)
    nc = n_sample ** 3
    n_part = int(round(ppc * nc))
    rs = np.random.RandomState(3)
    rec = np.zeros((n_part, 10))
    rec[:, 0:3] = rs.random_sample((n_part, 3))
    rec[:, 2] *= 0.6
    rec[:, 9] = 0.2 * dx
    out = {}
    # a 64^3 sample does not feed a hundred threads: time 1 thread and a few team sizes up to the core count, report the best
    for th in sorted(set([1] + [t for t in (8, 16, 32) if t <= threads] + ([threads] if threads <= 64 else []))):
        s = orc.FvSolver(case, threads=th)
        s.mesh = orc.Mesh(n_sample, n_sample, n_sample, dx)     # tree build is construction-time work, not timed (as on the GPU side)
        s.step(rec)                                              # warm-up step
        t0 = time.time()
        k = 0
        while k < 2 or (time.time() - t0 < 4.0 and k < 8):
            s.step(rec)
            k += 1
        out[th] = (time.time() - t0) / k
        s.close()
    return out, nc, n_part


def cpu_reference_as_written(n_sample=32, n_part=80000):
    """The reference's OWN particle path (FoamYade.C + meshTree.C compiled unmodified into oracle/_ref/ref_driver by `make -C oracle ref`)
    under mpiexec with a fake serial Yade: as written, i.e. quadratic buildCellPartList and one MPI_Allreduce per particle and force
    component.  Particle half only -- the reference's FV half is OpenFOAM library code and cannot be built here.  Per-step time =
    (wall of a 3-step run - wall of a 1-step run) / 2, which cancels tree build, file I/O and MPI start-up.  Returns None when the
    prebuilt driver or the MPI launcher is not there."""
    import shutil
    import subprocess
    import tempfile
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(drv) and os.path.exists(mpiexec)):
        return None
    dx = 1.0 / n_sample
    nc = n_sample ** 3
    rs = np.random.RandomState(5)
    rec = np.zeros((n_part, 10))
    rec[:, 0:3] = rs.random_sample((n_part, 3))
    rec[:, 2] *= 0.6
    rec[:, 9] = 0.2 * dx
    walls = {}
    try:
        for nsteps in (1, 3):
            d = tempfile.mkdtemp(prefix="fy_ref_")
            try:
                meta = [n_sample, n_sample, n_sample, repr(dx), 0.0, 0.0, 0.0, 1, 1, nsteps, 2650.0, 1000.0, 1e-6, 1e-4, 0.0, 0.0, -9.81]
                open(os.path.join(d, "meta.txt"), "w").write(" ".join(str(m) for m in meta) + "\n")
                for nm, comps in (("U", 3), ("gradP", 3), ("divT", 3), ("ddtU", 3), ("vGrad", 9)):
                    a = np.zeros((nc, comps))
                    if nm == "U":
                        a[:, 0] = 0.1
                    if nm == "gradP":
                        a[:, 2] = -9810.0
                    a.tofile(os.path.join(d, nm + ".bin"))
                for st in range(nsteps):
                    rec.tofile(os.path.join(d, f"records_s{st}.bin"))
                t0 = time.time()
                subprocess.run([mpiexec, "-n", "1", drv, d, ":", "-n", "1", drv, d], check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                walls[nsteps] = time.time() - t0
            finally:
                shutil.rmtree(d, ignore_errors=True)
    except Exception as e:                                        # noqa: BLE001  (a missing/failed reference run is reported, not fatal)
        return {"error": f"{type(e).__name__}: {e}"}
    per_step = max((walls[3] - walls[1]) / 2.0, 1e-9)
    return {"kind": "reference", "what": "FoamYade::setParticleAction AS WRITTEN (FoamYade.C + meshTree.C compiled unmodified; quadratic "
            "buildCellPartList, per-particle MPI_Allreduce to a fake serial Yade); particle half only, 1 core",
            "sample": f"{n_part} particles in the lower 60 % of a {n_sample}^3 box ({n_part / (0.6 * n_sample ** 3):.2f} per cell there; C3 has 4.07)",
            "s_per_step": round(per_step, 4), "particle_steps_per_sec": round(n_part / per_step, 1),
            "note": "not extrapolated to C3: the deposit is quadratic in the number of touched cells, so the as-written code does not reach that size"}


def max_over_ranks(elapsed, dist, device):
    """the contract's timing rule: the slowest rank defines the step time (works with nccl on GPUs and gloo on CPU)"""
    if dist is None:
        return elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_value(world, steps, elapsed):
    """whole-job throughput: every rank ran `steps` coupled steps in `elapsed` (max over ranks) seconds"""
    return world * steps / elapsed


def rccl_selftest_child(args):
    """child of rccl_preflight: join the throw-away communicator, run the library's known-answer pattern, exit 0 / 3"""
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    prod = ge.load_product()
    try:
        comm = prod.rccl_comm(rank, world, bytes.fromhex(args.rccl_selftest), local_rank)
        prod.comm_selftest(comm, local_rank)
    except Exception as e:                                                   # noqa: BLE001
        print(f"[rccl self-test rank {rank}] {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    raise SystemExit(0)


def rccl_preflight(prod, torch, dist, dev, rank, world, timeout=180):
    """Before a multi-GPU run commits to the library's RCCL communicator: the same operations the slab solver issues (grouped neighbour
    send/recv of two fields, sum / max all-reduce, all-gather), with known answers, in THROW-AWAY child processes -- one per rank, on a
    communicator of their own -- so that a fabric or library that cannot carry the pattern shows up here: a TIME-OUT makes the run fall back (labelled) instead
    of hanging; an error is only reported (the real set-up catches its own).  Returns "" or why this rank wants the fallback."""
    import subprocess
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt = torch.tensor(list(prod.rccl_unique_id()), dtype=torch.uint8, device=dev)
    dist.broadcast(idt, 0)
    cmd = [sys.executable, os.path.abspath(__file__), "--rccl-selftest", bytes(idt.cpu().tolist()).hex(), "--gpus", str(world)]
    try:
        r = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:      # an ERROR is not decisive (the real set-up below reports its own errors and falls back by itself): say so, go on
            print(f"[bench rank {rank}] RCCL self-test failed (rc {r.returncode}): {(r.stderr or r.stdout).strip()[-300:]}", file=sys.stderr, flush=True)
        return ""
    except subprocess.TimeoutExpired:
        return f"RCCL self-test did not finish within {timeout} s"          # a HANG is: the real run would hang the same way


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=160, help="cells per edge (C3 = 160)")
    ap.add_argument("--particles", type=int, default=10_000_000)
    ap.add_argument("--dt", type=float, default=1e-4)
    ap.add_argument("--p-solver", type=int, default=1, help="0 PCG+Jacobi, 1 PCG+multigrid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=64)
    ap.add_argument("--strong", action="store_true", help="N > 1: cut the ONE C3 box into N slabs (BASELINE configs[3]) instead of growing it (weak, the default)")
    ap.add_argument("--force-rccl", action="store_true", help="use the RCCL communicator even with one rank (smoke test of the RCCL path)")
    ap.add_argument("--rccl-selftest", default="", help=argparse.SUPPRESS)     # child mode: hex of the 128-byte RCCL id (see rccl_preflight)
    args = ap.parse_args()
    if args.rccl_selftest:
        rccl_selftest_child(args)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    prod = ge.load_product()
    strong = bool(args.strong and world > 1)
    case = c3_case(prod, args.n, args.dt, args.p_solver, world, strong)
    comm, solver, setup_err = None, None, ""
    try:
        if world > 1 and not os.environ.get("FOAMYADE_BENCH_NO_PREFLIGHT"):
            why = rccl_preflight(prod, torch, dist, dev, rank, world)
            t = torch.tensor([0.0 if why else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)                  # every rank must take the same path
            if float(t.item()) < 1.0:
                raise RuntimeError(why or "RCCL self-test failed on another rank")
        if world > 1 or args.force_rccl:
            # one RCCL communicator for the slab exchanges; the 128-byte unique id travels over torch.distributed
            os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                import glob
                for f in glob.glob(os.path.join(os.environ["FOAMYADE_TREE_CACHE_DIR"], "fy_tree_*.lock")):      # stale locks of a crashed run
                    os.remove(f)
                idt = torch.tensor(list(prod.rccl_unique_id()), dtype=torch.uint8, device=dev)
            if dist is not None:
                dist.broadcast(idt, 0)
            comm = prod.rccl_comm(rank, world, bytes(idt.cpu().tolist()), local_rank)
        solver = prod.Solver(case, device=local_rank, comm=comm)
    except Exception as e:                                       # noqa: BLE001  (reported below, never swallowed)
        setup_err = f"{type(e).__name__}: {e}"
    # every rank must take the same path: agree on whether the slab set-up worked everywhere
    slabs_ok = 0.0 if setup_err else 1.0
    if dist is not None:
        t = torch.tensor([slabs_ok], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        slabs_ok = float(t.item())
    parallelism = "single GPU" if world == 1 else f"{world} z-slabs of one {args.n}x{args.n}x{args.n if strong else args.n * world} box, RCCL halos + all-reduces over xGMI"
    slab_of_rank = rank
    if slabs_ok < 1.0:
        if world == 1:
            raise SystemExit(f"bench.py: solver set-up failed: {setup_err}")
        # LOUD fallback, labelled in the JSON line: N independent C3 boxes (no inter-GPU exchange), so that the run still reports
        # a per-GPU number instead of nothing.  This is NOT the sharded path.
        print(f"[bench rank {rank}] WARNING: z-slab/RCCL set-up failed on at least one rank ({setup_err or 'another rank'}); "
              f"falling back to {world} independent single-GPU replicas", file=sys.stderr, flush=True)
        if solver is not None:
            solver.close()
        case = c3_case(prod, args.n, args.dt, args.p_solver, 1)
        solver = prod.Solver(case, device=local_rank)
        parallelism = f"FALLBACK: {world} independent replicas of the single-GPU case, no exchange (z-slab/RCCL set-up failed: {setup_err or 'on another rank'})"
        slab_of_rank = 0
    if strong and slabs_ok >= 1.0:
        rec = c3_particles_strong(torch, args.particles, args.n, dev, rank, world)
    else:
        strong = False
        rec = c3_particles(torch, args.particles, args.n, 3 + rank, dev, slab=slab_of_rank)
    solver.set_particles_device(rec)
    solver.enable_particle_timing(True)
    nc = args.n ** 3 // (world if strong else 1)      # cells per rank

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        solver.step()
    solver.enable_kernel_timing(True)
    acc = dict(particle=0.0, locate_deposit=0.0, force=0.0, bin=0.0, finalize=0.0, momentum=0.0, pressure=0.0, other=0.0, p_iters=0, u_iters=0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        solver.step()
        st = solver.stats(); ct = solver.coupling_timings()
        acc["particle"] += st["ms_particle"]; acc["momentum"] += st["ms_momentum"]; acc["pressure"] += st["ms_pressure"]; acc["other"] += st["ms_other"]
        acc["locate_deposit"] += ct["locate_deposit"]; acc["force"] += ct["force"]; acc["bin"] += ct["bin"]; acc["finalize"] += ct["finalize"]
        acc["p_iters"] += st["p_iters_total"]; acc["u_iters"] += st["u_iters_total"]
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dist, dev)

    K = args.steps
    steps_per_s = K / elapsed if strong else aggregate_value(world, K, elapsed)      # strong: steps of the one box; weak: slab-steps
    # ---- per-kernel clocks (HIP events on the launch stream, collected inside the timed region).  The FV kernel clocks SAMPLE: the first
    # launch of a category in each step is timed (every launch of a category does the same work; an event record between two kernels
    # idles the stream for 5 - 10 us, so timing all ~15 launches per step would slow the step it measures by ~2 %): `launches` below =
    # timed launches
    kern = {}
    smooth_ms, smooth_n = solver.kernel_timing("mg_smooth_l0")
    apply_ms, apply_n = solver.kernel_timing("p_apply_dot")
    mom_ms, mom_n = solver.kernel_timing("mom_pass")
    np_part = int(rec.shape[0]) if strong else args.particles      # particles of this rank
    dep_ms = max(acc["finalize"], 0.0)
    cand = {
        # name: (total ms over the timed steps, launches, algorithmic bytes per launch, description) -- DESIGN.md section 3;
        # kbar = 5.46 stencil cells per particle, 12 B per (id, weight) pair
        # one pass since the candidate lists: positions 24 + velocity 24 + radius 8 + (cell, octant) list 2 B x 7.6 codes in,
        # chain length 4 + 12 B/pair out, 65 B per touched cell (accumulators + alpha/uParticle in k_finalize_cells)
        "k_locate_deposit+k_finalize_cells": (acc["locate_deposit"] + dep_ms, K, (24.0 + 24.0 + 8.0 + 15.2 + 4.0 + 12.0 * 5.46) * np_part + 65.0 * nc,
                     "k-d 'range' locate through per-(cell, octant) candidate lists + Gaussian weights + void-fraction deposit (LDS-aggregated "
                     "atomics) in one pass, then alpha/uParticle finalize"),
        "k_force_gaussian": (acc["force"], K, (64.0 + 12.0 * 5.46 + 52.0) * np_part + 176.0 * nc,
                             "drag + Archimedes + back-scatter: particle 64 B + stencil 12 B/pair in, force 52 B out, cell fields 112 B read + 64 B RMW"),
        "k_mg_smooth(level 0)": (smooth_ms, smooth_n, 56.0 * nc, "pEqn Laplacian apply fused with the damped-Jacobi update: 48 B/cell (diag, 3 upper, x, y) + b 8"),
        "k_p_apply_dot": (apply_ms, apply_n, 48.0 * nc, "pEqn Laplacian apply y = A p (+ p.Ap) inside PCG: 48 B/cell"),
        "k_mom_pass": (mom_ms, mom_n, (7 * 8 + 24 * 3) * nc, "fused momentum Jacobi pass: 7 coeffs + b,x,xn (3 comps)"),
    }
    # HBM traffic per launch from the committed PMC passes of this same command (tools/pmc_traffic.py; null if absent)
    traffic = {}
    for cand_file in sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1:]:
        try:
            traffic = {k: v.get("hbm_bytes_per_launch") for k, v in json.load(open(cand_file))["kernels"].items()}
        except Exception:
            traffic = {}
    if not (args.n == 160 and args.particles == 10_000_000 and world == 1):
        traffic = {}            # the PMC passes were taken on the default single-GPU workload only
    for nm, (ms, nl, bytes_per, desc) in cand.items():
        if nl:
            avg = ms / nl
            kern[nm] = dict(total_ms=ms, launches=int(nl), avg_ms=avg, achieved_GBps=bytes_per / (avg * 1e-3) / 1e9, alg_bytes=bytes_per, what=desc)
    dominant = max(kern, key=lambda k: kern[k]["total_ms"]) if kern else None
    lap = "k_mg_smooth(level 0)" if "k_mg_smooth(level 0)" in kern else "k_p_apply_dot"

    def roof(name):
        k = kern[name]
        return {"kernel": name, "bound": "hbm", "achieved": round(k["achieved_GBps"], 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(k["achieved_GBps"] / HBM_PEAK_GBPS, 4), "traffic": traffic.get(name.split("+")[0]), "avg_launch_ms": round(k["avg_ms"], 4),
                "launches": k["launches"], "algorithmic_bytes_per_launch": k["alg_bytes"], "what": k["what"]}

    out = {
        "metric": "coupled_steps_per_sec (pimpleFoamYade 4-way, 10M particles / 4M cells per GPU)" if (args.n == 160 and args.particles == 10_000_000)
        else f"coupled_steps_per_sec (pimpleFoamYade 4-way, {args.particles} particles / {nc} cells)",
        "value": round(steps_per_s, 4), "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / K, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "particle_steps_per_sec": round(steps_per_s * (args.particles if strong else np_part), 1),
        "coupled_steps_per_sec_of_the_whole_box": round(steps_per_s if strong else steps_per_s / world, 4),
        "config": {"workload": "C3: pimpleFoamYade Gaussian 4-way coupling, 160^3 = 4,096,000-cell closed box, 10,000,000 particles in the lower 60 %"
                   if (args.n == 160 and args.particles == 10_000_000) else f"non-default C3-like case {args.n}^3 cells / {args.particles} particles",
                   "cells": nc, "particles": np_part, "dt": args.dt, "pimple": {"nOuterCorrectors": 1, "nCorrectors": 2},
                   "p_solver": "PCG+MG V(2,2) damped Jacobi" if args.p_solver == 1 else "PCG+Jacobi",
                   "p_tol": [case.p_tol, case.p_rel_tol, case.p_final_tol, case.p_final_rel_tol],
                   "parallelism": parallelism,
                   "global_cells": nc * world, "global_particles": args.particles if strong else np_part * world},
        "per_step_ms": {k: round(acc[k] / K, 3) for k in ("particle", "bin", "locate_deposit", "finalize", "force", "momentum", "pressure", "other")},
        "p_iters_per_step": acc["p_iters"] / K, "u_iters_per_step": acc["u_iters"] / K,
        "roofline": roof(dominant) if dominant else None,
        "roofline_pEqn_laplacian": roof(lap) if lap in kern else None,
        "kernels": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"], "GBps": round(v["achieved_GBps"], 1)} for k, v in kern.items()},
    }
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        th = os.cpu_count() or 1
        ppc = args.particles / nc
        per, snc, snp = cpu_baseline(args.cpu_sample_n, ppc, args.dt, th)
        best_th = min(per, key=lambda k: per[k])
        scale = snc / nc                                   # linear-in-size extrapolation to the bench workload
        out["cpu_baseline"] = {
            "value": round((1.0 / per[best_th]) * scale, 6), "unit": "steps/s", "cores": int(best_th), "kind": "port",
            "single_thread_value": round((1.0 / per[1]) * scale, 6),
            "sample": f"same workload at {args.cpu_sample_n}^3 cells / {snp} particles ({snc / nc:.4f} of the bench size), CPU oracle (port of the "
                      f"reference path, de-quadraticised deposit), measured {per[best_th]:.2f} s/step on {best_th} threads ({per[1]:.2f} s/step on 1); "
                      f"value = measured steps/s x {scale:.5f} (linear-in-size extrapolation)"}
        ref = cpu_reference_as_written()
        if ref is not None:
            out["cpu_reference_as_written"] = ref
            pp = args.particles * steps_per_s
            if "particle_steps_per_sec" in ref:
                ref["gpu_particle_steps_per_sec_whole_coupled_step"] = round(pp, 1)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # RCCL prints a version banner through C stdio: get it out BEFORE the JSON line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
