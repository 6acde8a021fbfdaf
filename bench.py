#!/usr/bin/env python3
"""Headline benchmark: coupled CFD-DEM steps/s at BASELINE.json configs[2] ("C3"):
pimpleFoamYade 4-way coupling, 10 M particles, 160^3 = 4 096 000 cells, one MI355X.
`--config c2` is BASELINE configs[1] at full size instead (icoFoamYade point force, 200 x 100 x 50 = 1 M-cell channel, 1 M particles).
After the timed, HBM-resident region the same workload runs a few steps through the DROP-IN path (`--wire`, rank 0 of a single-GPU run):
an in-process fake Yade (parallel-Yade protocol, 4 workers) hands the records over as host buffers through the fy_transport callbacks
and takes the forces back, so that the PCIe copies and the wire-side host time are measured beside the headline number (never in it).

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
KBAR = 5.46                     # stencil cells per particle at 160^3
# algorithmic (compulsory) HBM bytes per particle of the two big particle kernels -- DESIGN.md section 3
LOCATE_BYTES_PER_PARTICLE = 80.0 + 4.0 + 56.0 + 15.2 + 4.0 + 12.0 * KBAR
LOCATE_WHAT = ("k-d 'range' locate through per-(cell, octant) candidate lists + Gaussian weights + void-fraction deposit in one pass, the wire record fetched "
               "through the binned placement by the kernel itself (round 3: no separate gather pass): record 80 + placement 4 + list 15 B in, binned SoA copy 56 + "
               "chain length 4 + 12 B/pair out per particle; 32 B of accumulators read-modify-written per cell")
FORCE_BYTES_PER_PARTICLE = 64.0 + 12.0 * KBAR + 52.0
# SURVEY.md 8(d) COMPULSORY bytes (every array once) of the same two kernels: the locate + deposit reads the 80-byte wire record and read-modify-writes
# the deposit accumulators (alpha 8 + uParticle 24 = 32 B per cell, twice); the force pass writes 48 B of force per particle, reads U, gradP, divT, C, V
# (104 B per cell) and read-modify-writes uSourceDrag + uSource (32 B per cell, twice).  Everything else in *_BYTES_PER_PARTICLE is traffic the
# design chose (binned SoA copy, stencil rows, placement)
LOCATE_COMPULSORY = (80.0, 64.0)            # per particle, per cell
FORCE_COMPULSORY = (48.0, 168.0)
FORCE_WHAT = ("drag + Archimedes + back-scatter: particle 64 B + stencil 12 B/pair in, force record 52 B out; per cell the 64-byte gather record read and "
              "32 B of momentum-source accumulators read-modify-written")


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


def c5_case(prod, dt, p_solver, n=320, u_in=0.05):
    """320^3 = 32 768 000-cell box, fluidized bed -- bottom (z-) inlet fixedValue U = (0,0,Uin), top (z+) outlet p = 0 with
    zeroGradient U, no-slip side walls (fixedFluxPressure there and at the inlet), g = (0,0,-9.81), nu = 1e-6; PIMPLE nOuter 1 nCorr 2"""
    U, ZG = prod.FY_BC_U_FIXED_VALUE, prod.FY_BC_U_ZERO_GRADIENT
    PX, PF = prod.FY_BC_P_FIXED_FLUX, prod.FY_BC_P_FIXED_VALUE
    return prod.make_case(prod.FY_SOLVER_PIMPLE, n, n, n, 1.0 / n, dt, 1e-6, rho_f=1000.0, rho_p=2650.0, g=(0.0, 0.0, -9.81),
                          u_bc=[U, U, U, U, U, ZG], u_val=[(0, 0, 0)] * 4 + [(0, 0, u_in), (0, 0, 0)], p_bc=[PX, PX, PX, PX, PX, PF], p_val=[0.0] * 6,
                          n_outer_correctors=1, n_correctors=2, p_solver=p_solver)


def c5_particles(torch, n_part, n, device, rank, world):
    """100 M particles uniform in the lower third of the unit box, seed 5, r = 0.2 dx, at rest; a rank keeps those inside its z-slab.  Generated
    in chunks: the host never holds more than one chunk beside the kept records"""
    g = torch.Generator(device="cpu").manual_seed(5)
    dx = 1.0 / n
    h = 1.0 / world
    keep = []
    left = n_part
    while left > 0:
        m = min(left, 20_000_000)
        pos = torch.rand(m, 3, dtype=torch.float64, generator=g)
        pos[:, 2] *= 1.0 / 3.0
        if world > 1:
            pos = pos[(pos[:, 2] >= rank * h) & (pos[:, 2] < (rank + 1) * h)]
        rec = torch.zeros(pos.shape[0], 10, dtype=torch.float64)
        rec[:, 0:3] = pos
        rec[:, 9] = 0.2 * dx
        keep.append(rec.to(device))
        left -= m
    return torch.cat(keep).contiguous()


def c2_case(prod, dt, p_solver):
    """200 x 100 x 50 channel, L = (2, 1, 0.5), inlet fixedValue U = (1,0,0) at x-, outlet p = 0 at x+, no-slip walls,
    nu = 1e-3, icoFoamYade (PISO nCorr 2), point force"""
    U, ZG = prod.FY_BC_U_FIXED_VALUE, prod.FY_BC_U_ZERO_GRADIENT
    PZ, PF = prod.FY_BC_P_ZERO_GRADIENT, prod.FY_BC_P_FIXED_VALUE
    return prod.make_case(prod.FY_SOLVER_ICO, 200, 100, 50, 0.01, dt, 1e-3, rho_f=1000.0, rho_p=2650.0, g=(0.0, 0.0, 0.0),
                          u_bc=[U, ZG, U, U, U, U], u_val=[(1, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0), (0, 0, 0)],
                          p_bc=[PZ, PF, PZ, PZ, PZ, PZ], n_correctors=2, p_solver=p_solver)


def c2_particles(torch, n_part, device):
    """1 M particles uniform in the channel, seed 2, r = 0.15 dx, at rest"""
    g = torch.Generator(device="cpu").manual_seed(2)
    rec = torch.zeros(n_part, 10, dtype=torch.float64)
    rec[:, 0:3] = torch.rand(n_part, 3, dtype=torch.float64, generator=g)
    rec[:, 0] *= 2.0
    rec[:, 2] *= 0.5
    rec[:, 9] = 0.15 * 0.01
    return rec.to(device).contiguous()


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                   # a container's CPU quota (cgroup v2): more threads than that only queue up
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return model, n


class FakeYadePeer:
    """In-process stand-in for Yade's FoamCoupling engine in PARALLEL-Yade mode (the only mode whose message count does not grow with
    the particle count): a master (world rank 0) and W workers (ranks 1..W); this process is the single solver rank W + 1.
    Every call the solver side issues (FoamYade.C:114-155, 239-243, 504-507, 537-553) is answered from host memory the way the MPI library
    would answer it -- a copy into / out of the caller's buffer."""

    def __init__(self, prod, records, workers):
        import ctypes as C
        self.C, self.rec, self.W = C, records, workers
        n = records.shape[0]
        self.bounds = [(w * n) // workers for w in range(workers + 1)]
        self.bytes_recv = 0
        self.force_checksum = 0.0
        self.fbuf = None
        T = prod.Transport()
        T.world_size, T.world_rank, T.local_rank, T.local_size = workers + 2, workers + 1, 0, 1
        self._cb = [prod._SEND(self.send), prod._RECV(self.recv), prod._BCAST(self.bcast), prod._BCAST(self.bcast), prod._ALLRED(self.allreduce)]
        T.send, T.recv, T.bcast_world, T.bcast_local, T.allreduce_world = self._cb
        self.T = T

    def _view(self, ptr, count, dtype):
        ct = self.C.c_int32 if dtype == 0 else self.C.c_double
        return np.ctypeslib.as_array((ct * count).from_address(ptr))

    def send(self, user, buf, count, dtype, dest, tag):
        if tag == 1005 and count:                      # forces: the MPI library would copy them out of the caller's buffer
            if self.fbuf is None or self.fbuf.size < count:
                self.fbuf = np.empty(count)
            self.fbuf[:count] = self._view(buf, count, dtype)
            self.force_checksum += float(self.fbuf[:count:max(count // 4096, 1)].sum())
            self.bytes_recv += 8 * count
        elif tag == 1004:
            self.bytes_recv += 4 * count
        return 0

    def recv(self, user, buf, count, dtype, src, tag):
        out = self._view(buf, count, dtype)
        if tag == 1003:
            out[:] = self.bounds[src] - self.bounds[src - 1]
        elif tag == 1002:
            out[:] = self.rec[self.bounds[src - 1]:self.bounds[src]].reshape(-1)
        elif tag == 1060:
            out[:] = 1e-6
        else:
            return 1
        return 0

    def bcast(self, user, buf, count, dtype, root):
        return 0

    def allreduce(self, user, inp, out, count, dtype, op):
        return 1                                       # serial-Yade only


def wire_leg(prod, torch, case, rec_host, steps, workers, device):
    """the drop-in path: same case, the particles arrive as host buffers from a fake Yade and the forces go back; returns per-step averages"""
    yade = FakeYadePeer(prod, rec_host, workers)
    solver = prod.Solver(case, transport=yade.T, device=device)
    solver.enable_particle_timing(True)
    acc = dict(step=0.0, copy_in=0.0, copy_out=0.0, wire_recv=0.0, wire_send=0.0, particle=0.0, bytes_in=0, bytes_out=0)
    for s in range(steps + 1):                         # one warm-up step (first-touch of the pinned staging buffers, tile capacities)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.step()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        ct = solver.coupling_timings(); st = solver.stats()
        if s == 0:
            continue
        acc["step"] += 1e3 * dt_
        acc["particle"] += st["ms_particle"]
        for k in ("copy_in", "copy_out", "wire_recv", "wire_send"):
            acc[k] += ct[k]
        acc["bytes_in"] += ct["bytes_in"]; acc["bytes_out"] += ct["bytes_out"]
    solver.close()
    K = float(steps)
    gbps = lambda b, ms: round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None
    return {"what": f"drop-in path: in-process fake Yade, parallel-Yade protocol with {workers} workers; records and forces cross PCIe through pinned "
                    "staging on a copy stream, pipelined with the batches' kernels; wire_* = host time inside the transport callbacks (a host copy per "
                    "message here, where MPI would do its own)",
            "steps": steps, "ms_per_step": round(acc["step"] / K, 3), "steps_per_sec": round(1e3 * K / acc["step"], 3),
            "per_step_ms": {"h2d": round(acc["copy_in"] / K, 3), "d2h": round(acc["copy_out"] / K, 3), "wire_recv": round(acc["wire_recv"] / K, 3),
                            "wire_send": round(acc["wire_send"] / K, 3), "particle_phase_incl_transfers": round(acc["particle"] / K, 3)},
            "bytes_per_step": {"h2d": int(acc["bytes_in"] / K), "d2h": int(acc["bytes_out"] / K)},
            "pcie_GBps": {"h2d": gbps(acc["bytes_in"], acc["copy_in"]), "d2h": gbps(acc["bytes_out"], acc["copy_out"])}}


def wire_leg_mpi(n, n_part, steps, workers, dt, c5, solver_ranks=1):
    """the drop-in path over REAL MPI: tools/native/wire_bench.cpp under mpiexec MPMD -- a Yade master, `workers` Yade worker processes that own
    the particles, and the solver side (fy_solver + the MPI transport of libfoamyade_mpi): ONE rank (solver_ranks = 1), or a computing rank and
    solver_ranks - 1 wire helpers (include/foamyade_mpi.h) -- parallel-Yade protocol, every record and every force crossing a process boundary
    through MPI_Send / MPI_Recv.  Returns the computing rank's JSON record, or None when the launcher or the binary is not there"""
    import subprocess
    exe = os.path.join(ROOT, "tools", "native", "wire_bench")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        return None
    a = [exe, str(n), str(n_part), str(steps), repr(dt), "c5" if c5 else "-", str(solver_ranks)]
    cmd = [mpiexec, "-n", "1"] + a + [":", "-n", str(workers)] + a + [":", "-n", str(solver_ranks)] + a
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"wire_bench rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
        j = json.loads(line[-1])
    except Exception as e:                                            # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    gbps = lambda b, ms: round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None
    side = ("one solver rank runs fy_solver with the MPI transport" if solver_ranks == 1 else
            f"the solver side is a computing rank (fy_solver) + {solver_ranks - 1} wire helpers that receive the records into a shared-memory arena and send the "
            "answers out of it in parallel (to Yade a {0}-rank solver; the library sees views of the arena, fy_transport::recv_view / send_reserve)".format(solver_ranks))
    return {"what": f"drop-in path over real MPI (MPICH, shared-memory transport): a Yade master + {workers} Yade worker PROCESSES own the particles, {side}; "
                    "parallel-Yade protocol (FoamYade.C:114-155, 239-243, 504-507, 537-549); wire_* = host time of the computing rank inside the transport's "
                    "data calls (one rank: MPI_Recv / MPI_Send of the records and results; helpers: waiting for their reports), h2d / d2h = the PCIe copies "
                    "on their own streams, overlapped with the other batches' receive and kernels; pcie floor = bytes / 56 GB/s",
            "solver_side_ranks": solver_ranks,
            "steps": j["steps"], "ms_per_step": j["ms_per_step"], "steps_per_sec": round(1e3 / j["ms_per_step"], 3),
            "per_step_ms": {k: j[k] for k in ("h2d", "d2h", "wire_recv", "wire_send", "particle_phase_incl_transfers")},
            "bytes_per_step": {"h2d": j["bytes_in"], "d2h": j["bytes_out"]},
            "pcie_GBps": {"h2d": gbps(j["bytes_in"], j["h2d"]), "d2h": gbps(j["bytes_out"], j["d2h"])},
            "mpi_GBps": {"recv": gbps(j["bytes_in"], j["wire_recv"]), "send": gbps(j["bytes_out"], j["wire_send"])},
            "located_at_the_workers": int(j["found_at_the_workers"]), "located_by_two_solver_ranks": int(j.get("found_by_two_ranks", 0))}


def oracle_case(orc, config, n_s, dt):
    """the bench workload as an oracle case (n_s: cells per edge; c2: n_s = 100 is the full 200 x 100 x 50 channel) -> (case, nx, ny, nz, dx)"""
    if config == "c2":
        scale = n_s / 100.0
        nx, ny, nz, dx = int(200 * scale), int(100 * scale), int(50 * scale), 0.01
        case = orc.fv_case(0, nx, ny, nz, dx, dt, 1e-3, u_bc=[0, 1, 0, 0, 0, 0], u_val=[(1.0, 0, 0)] + [(0, 0, 0)] * 5, p_bc=[0, 1, 0, 0, 0, 0],
                           p_val=[0.0] * 6, n_corr=2, p_solver=1)
    elif config == "c5":
        nx = ny = nz = n_s
        dx = 1.0 / n_s
        case = orc.fv_case(1, nx, ny, nz, dx, dt, 1e-6, g=(0, 0, -9.81), u_bc=[0, 0, 0, 0, 0, 1], u_val=[(0, 0, 0)] * 4 + [(0, 0, 0.05), (0, 0, 0)],
                           p_bc=[orc.P_FIXEDFLUX] * 5 + [1], p_val=[0.0] * 6, n_outer=1, n_corr=2, p_solver=1)
    else:
        nx = ny = nz = n_s
        dx = 1.0 / n_s
        case = orc.fv_case(1, nx, ny, nz, dx, dt, 1e-6, g=(0, 0, -9.81), p_bc=[orc.P_FIXEDFLUX] * 6, n_outer=1, n_corr=2, p_solver=1)
    return case, nx, ny, nz, dx


def bench_records_host(torch, config, n, n_part, velocities=False):
    """the timed region's particle records as a host array: the SAME generator calls that fill the device records (c3_particles / c2_particles), so
    the oracle and the HIP path can be compared on the bench's own cloud.  velocities: +-0.05 m/s per component (seed 77) -- not the BASELINE
    cloud, which is at rest; used by the parity test so that the drag terms are not identically zero"""
    rec = (c2_particles(torch, n_part, "cpu") if config == "c2" else c3_particles(torch, n_part, n, 3, "cpu")).numpy()
    if velocities:
        rs = np.random.Generator(np.random.PCG64(77))
        rec[:, 3:6] = (rs.random((n_part, 3)) - 0.5) * 0.1
    return rec


PARITY_FIELDS = ("U", "p", "phi_x", "phi_y", "phi_z")
PARITY_SOURCES = ("alpha", "uSource", "uSourceDrag")


def oracle_steps(orc, config, n_s, dt, th, rec, collect=False, steps=2):
    """`steps` coupled steps of the CPU oracle from the case's initial state on the records `rec`; the LAST one is timed (the first touches every
    array).  collect: also return what the last step left -- fields, the particle action's outputs, the sources as setParticleAction left them"""
    case, nx, ny, nz, _ = oracle_case(orc, config, n_s, dt)
    s = orc.FvSolver(case, threads=th)                       # tree build is construction-time work, not timed (as on the GPU side)
    el, ref, first = 0.0, None, None
    oracle_steps.seconds = []                                # every step's wall time (cpu_baseline reports them beside the timed one)
    for it in range(steps):
        last = it == steps - 1
        cap = {} if (collect and (last or it == 0)) else None
        t0 = time.time()
        out = s.step(rec, capture=cap)
        el = time.time() - t0
        oracle_steps.seconds.append(round(el, 3))
        if collect and it == 0 and steps > 1:
            first = dict(cap, force=out["force"])            # step 1 starts from exact inputs (the initial fields): its particle side carries no FV tolerance
        if collect and last:
            ref = {k: s.get(k) for k in PARITY_FIELDS}
            ref.update(cap)
            ref.update(force=out["force"], k=out["k"], ids=out["ids"], found=out["found"], chain_len=out["chain_len"], stats=s.stats(), first=first)
    s.close()
    return el, nx * ny * nz, ref


def hip_vs_oracle(prod, case, rec, ref, device=0, steps=2):
    """the same `steps` coupled steps on the HIP path (a fresh fy_solver, host records through fy_set_particles_host), compared with what
    oracle_steps collected.  Index work (k, stencil ids, found flags): exact.  Floating point: max |a - b| / max |b| per array."""
    s = prod.Solver(case, device=device)
    s.hold_sources(True)                                     # the step's closing setSourceZero waits: alpha / uSource / uSourceDrag stay readable
    s.set_particles(rec)
    gaussian = case.solver == prod.FY_SOLVER_PIMPLE

    def rel(a, b):
        return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))
    res = {"steps_compared": steps, "particles": int(rec.shape[0]), "cells": int(s.n_cells)}
    for it in range(steps):
        s.step()
        if it == 0 and ref.get("first") is not None:
            f1 = ref["first"]
            res["step1"] = {"force": rel(s.forces(), f1["force"]), "force_scale": float(np.abs(f1["force"]).max())}
            for nm in (PARITY_SOURCES if gaussian else ("uSource",)):
                res["step1"][nm] = rel(s.get(nm), f1[nm])
    st = s.stats()
    if gaussian:
        k, ids, _, chain = s.stencils()
        res["k_equal"] = bool(np.array_equal(k, ref["k"]))
        res["ids_equal"] = bool(np.array_equal(ids, ref["ids"]))
        res["chain_equal"] = bool(np.array_equal(chain, ref["chain_len"]))
        res["pairs"] = int(k.sum())
        del k, ids, chain
    res["found_equal"] = bool(np.array_equal(s.found(), ref["found"]))
    res["force"] = rel(s.forces(), ref["force"])
    res["force_scale"] = float(np.abs(ref["force"]).max())
    for nm in (PARITY_SOURCES if gaussian else ("uSource",)) + PARITY_FIELDS:
        res[nm] = rel(s.get(nm), ref[nm])
    res["p_iters"] = [int(st["p_iters_total"]), int(ref["stats"]["p_iters_total"])]
    res["u_iters"] = [int(st["u_iters_total"]), int(ref["stats"]["u_iters_total"])]
    res["what"] = ("HIP path vs the CPU oracle after %d coupled steps from the case's initial state on the bench's own records: *_equal = bit-exact index work; "
                   "the other entries are max |hip - oracle| / max |oracle| per array (step1: the particle side of the first step, whose inputs are exact; the "
                   "rest after the last step, where forces and sources gather fields that carry the linear solvers' tolerance); p_iters / u_iters = [hip, oracle] "
                   "of the last step" % steps)
    s.close()
    return res


def cpu_baseline(config, n_sample, n_part, dt, threads, full, torch=None, prod=None, p_solver=1, device=0):
    """the CPU oracle (a faithful port of the reference's path, kind = "port") on the GPU box's host cores, bounded to some tens of seconds:
    all usable threads on the bench's own workload (full: at its own size and on its own records, one warm-up + one timed step; else an n_sample^3
    sample with the same particles per cell), and ONE thread on a half-edge sample of that (1/8 of the cells and particles).
    full and prod given: what the two oracle steps left is compared with the HIP path's two steps (hip_vs_oracle) -- the parity of the headline
    configuration at its own size, paid for by the baseline's own oracle run.
    Returns ({threads: (s/step, cells)}, cells, native build?, parity dict or None)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    native = orc.use_native_build()

    def sample_records(n_s, npart):
        rs = np.random.RandomState(3)
        rec = np.zeros((npart, 10))
        rec[:, 0:3] = rs.random_sample((npart, 3))
        _, nx, ny, nz, dx = oracle_case(orc, config, n_s, dt)
        if config == "c2":
            rec[:, 0:3] *= np.array([nx * dx, ny * dx, nz * dx])
            rec[:, 9] = 0.15 * dx
        else:
            rec[:, 2] *= (1.0 / 3.0) if config == "c5" else 0.6
            rec[:, 9] = 0.2 * dx
        return rec

    parity = None
    if full and torch is not None and prod is not None and config in ("c3", "c2"):
        rec = bench_records_host(torch, config, n_sample, n_part)
        el, cells, ref = oracle_steps(orc, config, n_sample, dt, threads, rec, collect=True)
        cpu_baseline.step_seconds = list(oracle_steps.seconds)
        try:
            case = c2_case(prod, dt, p_solver) if config == "c2" else c3_case(prod, n_sample, dt, p_solver)
            parity = hip_vs_oracle(prod, case, rec, ref, device)
        except Exception as e:                                        # noqa: BLE001  (reported in the line)
            parity = {"error": f"{type(e).__name__}: {e}"}
        del ref, rec
        out = {threads: (el, cells)}
    else:
        el, cells, _ = oracle_steps(orc, config, n_sample, dt, threads, sample_records(n_sample, n_part))
        cpu_baseline.step_seconds = list(oracle_steps.seconds)
        out = {threads: (el, cells)}
    if threads != 1:
        el1, cells1, _ = oracle_steps(orc, config, n_sample // 2, dt, 1, sample_records(n_sample // 2, n_part // 8))
        out[1] = (el1, cells1)
    return out, out[threads][1], native, parity


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


def c2_subrecord(prod, torch, dev, device_index, steps=20, warmup=5):
    """BASELINE configs[1] at full size beside the headline (driver-observed since round 4): icoFoamYade point force, 200 x 100 x 50 cells, 1 M
    particles, dt = 2e-3 -- `steps` timed steps after `warmup`, same barriers as the headline's region"""
    case = c2_case(prod, 2e-3, 1)
    s = prod.Solver(case, device=device_index)
    rec = c2_particles(torch, 1_000_000, dev)
    s.set_particles_device(rec)
    s.enable_particle_timing(True)
    for _ in range(warmup):
        s.step()
    acc = dict(particle=0.0, momentum=0.0, pressure=0.0, other=0.0, force=0.0, p_iters=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.step()
        st = s.stats(); ct = s.coupling_timings()
        for k in ("particle", "momentum", "pressure", "other"):
            acc[k] += st["ms_" + k]
        acc["force"] += ct["force"]; acc["p_iters"] += st["p_iters_total"]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    s.close()
    f_ms = acc["force"] / steps
    return {"what": "BASELINE configs[1] at full size in the same run: icoFoamYade point-force coupling, 200 x 100 x 50 = 1,000,000-cell channel (inlet U = (1,0,0), "
                    "outlet p = 0, no-slip walls), 1,000,000 particles, dt = 2e-3, PISO nCorr 2; NOT the configuration `value` is quoted on",
            "value": round(steps / el, 3), "unit": "steps/s", "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * el / steps, 3),
            "particle_steps_per_sec": round(steps / el * 1_000_000, 1), "p_iters_per_step": acc["p_iters"] / steps,
            "per_step_ms": {k: round(acc[k] / steps, 3) for k in ("particle", "force", "momentum", "pressure", "other")},
            "k_point_force": {"avg_ms": round(f_ms, 4), "GBps": round(128.0e6 / (f_ms * 1e-3) / 1e9, 1) if f_ms > 0 else None,
                              "what": "128 B per particle (80 B record in, 48 B force out); bound by its three global FP64 atomics per particle"}}


def general_mesh_subrecord(timeout=180):
    """the general-mesh solver beside the headline (fy_ldu_solver): pimpleFoamYade, Gaussian 4-way coupling, on 128^3 WAVY hexahedra handed over
    as a polyhedral mesh (non-orthogonal, skewed; one non-orthogonal corrector; agglomeration-multigrid PCG) with 2.5 M particles -- tools/ldu_bench.py in a
    child process (its own context; the mesh is built with numpy there)"""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "ldu_bench.py"), "128", "10", "wavy", "2500000", "mg", "1e-6", "pimple"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    line = (r.stdout.strip().splitlines() or [""])[-1]
    d = json.loads(line)
    return {"what": "NOT the configuration `value` is quoted on: pimpleFoamYade on a general polyhedral mesh (fy_ldu_solver), 128^3 = 2,097,152 wavy hexahedra in owner / neighbour "
                    "addressing, 2,500,000 particles (Gaussian coupling on the explicit k-d tree), closed box under gravity, 1 non-orthogonal corrector, MG-PCG",
            "value": round(d["steps_per_s"], 3), "unit": "steps/s", "steps": d["steps"], "ms_per_step": round(d["ms_per_step_wall"], 3), "ms_particle": round(d["ms_particle"], 3),
            "p_iters_per_step": d["pcg_iters_per_step"], "cells": d["cells"], "particles": d["particles"]}


def laplacian_probe_child(n, reps):
    """child of laplacian_past_cache (also run under rocprofv3 --pmc): the pEqn Laplacian apply y = A p (k_p_apply, 48 B per cell: diag, three upper
    coefficients, x, y) on an n^3 operator -- 1.57 GB per launch at 320^3, six times the 256 MiB Infinity Cache -- timed by HIP events on its stream"""
    import torch  # noqa: F401
    prod = ge.load_product()
    case = prod.make_case(prod.FY_SOLVER_ICO, n, n, n, 1.0 / n, 1e-3, 1e-2, u_bc=[prod.FY_BC_U_FIXED_VALUE] * 6, u_val=[(0, 0, 0)] * 6, p_solver=1)
    s = prod.Solver(case)
    s.step()                                   # assembles the pressure matrix
    ms = s.time_p_apply(reps)
    s.close()
    print(json.dumps({"n": n, "reps": reps, "avg_ms": ms}), flush=True)


def laplacian_past_cache(n=320, reps=40, timeout=300):
    """roofline of the pEqn Laplacian where the Infinity Cache cannot hold it: two child passes of `bench.py --laplacian-probe n` under rocprofv3
    (--pmc FETCH_SIZE, then --pmc WRITE_SIZE; --kernel-trace only), each timing k_p_apply itself with HIP events; bytes from the counters as in
    live_pmc_traffic"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    nc = n ** 3
    out = {"kernel": "k_p_apply", "what": f"pEqn Laplacian apply y = A p on a {n}^3 operator: 48 B/cell (diag, 3 upper, x, y) = {48.0 * nc / 1e9:.2f} GB per launch, "
                                          "several times the 256 MiB Infinity Cache; HIP-event average over the launches of a child process without counters, bytes from two rocprofv3 --pmc passes of the same child",
           "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "algorithmic_bytes_per_launch": 48.0 * nc, "cells": nc}
    d = tempfile.mkdtemp(prefix="fy_lap_", dir="/tmp")
    try:
        cnt, times, plain = {}, [], None
        # the time comes from a pass of its own WITHOUT counters (as the main line's does: counter collection slows the kernels it watches, 0.49 against
        # 0.52 - 0.53 of the peak here); the two counter passes give the bytes
        for tag in ("", "FETCH_SIZE", "WRITE_SIZE"):
            od = os.path.join(d, tag or "plain")
            base = [sys.executable, os.path.abspath(__file__), "--laplacian-probe", str(n), "--laplacian-reps", str(reps)]
            cmd = ([rp, "--pmc", tag, "--kernel-trace", "--output-format", "csv", "-d", od, "--"] if (tag and os.path.exists(rp)) else []) + base
            env = dict(os.environ, TMPDIR="/tmp", FOAMYADE_TREE_CACHE_DIR="/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            line = [q for q in r.stdout.splitlines() if q.startswith("{")]
            if r.returncode != 0 or not line:
                out["error"] = f"probe failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
                return out
            if not tag:
                plain = json.loads(line[-1])["avg_ms"]
                continue
            times.append(json.loads(line[-1])["avg_ms"])
            files = glob.glob(os.path.join(od, "**", "*counter_collection.csv"), recursive=True)
            if files:
                v = [float(q["Counter_Value"]) for q in csv.DictReader(open(files[0])) if q["Counter_Name"] == tag and "k_p_apply" in q["Kernel_Name"] and "dot" not in q["Kernel_Name"]]
                if v:
                    cnt[tag] = 1024.0 * sum(v) / len(v)
        avg = plain if plain is not None else min(times)
        out.update({"avg_launch_ms": round(avg, 4), "avg_launch_ms_under_counters": round(min(times), 4) if times else None, "launches": reps, "achieved": round(48.0 * nc / (avg * 1e-3) / 1e9, 1),
                    "frac": round(48.0 * nc / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)})
        if len(cnt) == 2:
            t2, t1 = traffic_of(cnt)
            out.update({"traffic": round(t2), "traffic_reads_undoubled": round(t1), "traffic_GBps": round(t2 / (avg * 1e-3) / 1e9, 1)})
        else:
            out["traffic"] = None
    except Exception as e:                                            # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def live_pmc_traffic(extra_args, nc, steps=5, warmup=2, timeout=420):
    """HBM bytes per launch, measured NOW: two rocprofv3 child passes of this very command (--pmc FETCH_SIZE, then --pmc WRITE_SIZE, each with
    --kernel-trace only: the guide's HBM recipe, separate passes), `steps` timed + `warmup` steps each; the launches of the warm-up steps (the first
    step of a particle population still flushes its tables with global atomics) are dropped.  Both counters are KiB; reads = 2 x FETCH_SIZE on
    gfx950 (wide coalesced streams are counted at half their size: an UPPER bound for gather-dominated kernels).  Returns ({kernel: bytes}, note)"""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_traffic import classify
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    res = {}
    d = tempfile.mkdtemp(prefix="fy_pmc_", dir="/tmp")
    try:
        for tag in ("FETCH_SIZE", "WRITE_SIZE"):
            od = os.path.join(d, tag)
            cmd = [rp, "--pmc", tag, "--kernel-trace", "--output-format", "csv", "-d", od, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--wire", "0", "--no-moving", "--no-extras", "--pmc", "0"] + extra_args
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(od, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {tag} pass failed (rc {r.returncode}): {(r.stderr or '')[-200:]}"
            rows = [q for q in csv.DictReader(open(files[0])) if q["Counter_Name"] == tag]
            rows.sort(key=lambda q: int(q.get("Dispatch_Id", 0)))
            per = {}
            for q in rows:
                k = classify(q["Kernel_Name"], int(q.get("Grid_Size", q.get("Grid_Size_X", 0))), nc)
                if k:
                    per.setdefault(k, []).append(float(q["Counter_Value"]))
            for k, v in per.items():
                v = v[(len(v) * warmup) // (steps + warmup):]
                res.setdefault(k, {})[tag] = 1024.0 * sum(v) / max(len(v), 1)
    except Exception as e:                                            # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res, ""


def traffic_of(v):
    """HBM bytes per launch from the two counters: reads doubled as the guide's gfx950 correction prescribes (calibrated on wide coalesced streams;
    an UPPER bound for gather-dominated kernels), and as counted (a LOWER bound there)"""
    f, w = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    return 2.0 * f + w, f + w


def max_over_ranks(elapsed, dist, device, group=None):
    """the contract's timing rule: the slowest rank defines the step time (works with nccl on GPUs and gloo on CPU)"""
    if dist is None:
        return elapsed
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def aggregate_value(world, steps, elapsed):
    """whole-job throughput: every rank ran `steps` coupled steps in `elapsed` (max over ranks) seconds"""
    return world * steps / elapsed


def rccl_selftest_child(args):
    """child of rccl_preflight: join the throw-away communicator, run the library's known-answer pattern, exit 0 / 3"""
    rank, world, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    import torch  # noqa: F401  -- first, as in the parent: the library then dlopens the SAME librccl (the wheel's) that produced the unique id
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


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this very command under torch.distributed.run, one rank per GPU of this node
    (what the contract's N > 1 command line does), and hand its exit code back"""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    # (the arguments travel in the environment: torch.distributed.run's own parser takes a bare `--n 32` behind the script name for an abbreviation of ITS options)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)]
    env = dict(os.environ)
    env["FOAMYADE_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed steps (default 64: the placement rebuild of every 32nd coupling step falls inside the region)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=160, help="cells per edge (C3 = 160)")
    ap.add_argument("--particles", type=int, default=10_000_000)
    ap.add_argument("--dt", type=float, default=1e-4)
    ap.add_argument("--p-solver", type=int, default=1, help="0 PCG+Jacobi, 1 PCG+multigrid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-n", type=int, default=0, help="CPU baseline on an n^3 sample of the workload instead of the full size (0 = full size)")
    ap.add_argument("--config", default="c3", choices=["c3", "c2", "c5"], help="c3: BASELINE configs[2] (the metric's configuration); c2: configs[1] at full size; "
                    "c5: configs[4] at full size (320^3 cells, 100 M particles, fluidized bed: bottom inlet, top outlet) -- on one GPU, or with --gpus N cut into N z-slabs")
    ap.add_argument("--wire", type=int, default=2, help="steps of the drop-in (host-buffer / fake-Yade) leg after the timed region, 0 = skip")
    ap.add_argument("--wire-workers", type=int, default=-1, help="Yade worker processes of the drop-in leg (-1: as many as the host's usable cores carry beside the solver side, at most 6)")
    ap.add_argument("--laplacian-probe", type=int, default=0, help=argparse.SUPPRESS)       # child mode of laplacian_past_cache
    ap.add_argument("--laplacian-reps", type=int, default=40, help=argparse.SUPPRESS)
    ap.add_argument("--no-extras", action="store_true", help="skip the C2 and general-mesh sub-records and the past-the-Infinity-Cache Laplacian probe of the default run")
    ap.add_argument("--wire-helpers", type=int, default=-1, help="(-1: as many as --wire-workers) " "wire-helper ranks beside the computing rank in the drop-in leg over MPI (0: one solver rank receives everything)")
    ap.add_argument("--moving", action="store_true", help="not the BASELINE configuration: particles carry random velocities (+-0.05 m/s) and are displaced by "
                    "~0.1 dx per step (alternating random offsets, applied between the steps inside the timed region), so that the momentum deposit, "
                    "the re-bin amortisation and the pressure solver see a cloud that changes from step to step")
    ap.add_argument("--strong", action="store_true", help="N > 1: cut the ONE C3 box into N slabs (BASELINE configs[3]) instead of growing it (weak, the default)")
    ap.add_argument("--force-rccl", action="store_true", help="use the RCCL communicator even with one rank (smoke test of the RCCL path)")
    ap.add_argument("--rccl-selftest", default="", help=argparse.SUPPRESS)     # child mode: hex of the 128-byte RCCL id (see rccl_preflight)
    ap.add_argument("--ref-as-written", action="store_true", help="also time oracle/_ref/ref_driver (FoamYade.C + meshTree.C as written, built in the development "
                    "container against a stand-in OpenFOAM header) on a small sample; off by default")
    ap.add_argument("--no-moving", action="store_true", help="skip the moving-bed sub-record that follows the timed region (N = 1, C3 only)")
    ap.add_argument("--pmc", type=int, default=-1, help="1: measure roofline.traffic live (two rocprofv3 --pmc child passes of this command, FETCH_SIZE and "
                    "WRITE_SIZE, after everything else); 0: print null + the committed profile's path; -1 (default): live when rocprofv3 is there, N = 1, default C3")
    handed = os.environ.pop("FOAMYADE_BENCH_ARGV", None)           # (self_launch's ranks: the command line of the process that launched them)
    args = ap.parse_args(json.loads(handed) if (handed and len(sys.argv) == 1) else None)
    if args.rccl_selftest:
        rccl_selftest_child(args)
    if args.laplacian_probe:
        laplacian_probe_child(args.laplacian_probe, args.laplacian_reps)
        raise SystemExit(0)
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if os.environ.get("FOAMYADE_BENCH_LAUNCH_PROBE"):             # tests/test_bench_multirank.py: what did the launcher hand this rank?
        # (one write call: two ranks share the pipe, and print() hands the text and the newline over separately)
        os.write(1, (json.dumps({"rank": int(os.environ.get("RANK", "-1")), "world": int(os.environ.get("WORLD_SIZE", "-1")), "gpus": args.gpus, "steps": args.steps}) + "\n").encode())
        raise SystemExit(0)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    fb_group = None                                            # the gloo group beside an nccl default group (see below)
    ctl_group = None                                           # the group this script's own collectives run in (None: the default group)
    # FOAMYADE_COMM=rccl (default) | ipc: the slab transport.  ipc = the library's peer-store communicator (fy_comm_create_ipc: kernels that
    # store into the neighbours' hipIpc-mapped device windows); torch.distributed then only carries its bootstrap and this script's own timing collectives, over
    # gloo.  It is also what runs when the ranks outnumber the GPUs (N processes on one GPU: RCCL refuses duplicate devices), loudly labelled.
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    comm_kind = os.environ.get("FOAMYADE_COMM", "rccl").lower()
    if comm_kind not in ("rccl", "ipc"):
        raise SystemExit(f"bench.py: FOAMYADE_COMM={comm_kind}: rccl or ipc")
    fail_rccl = bool(os.environ.get("FOAMYADE_BENCH_FAIL_RCCL"))      # (test hook: the RCCL set-up "fails", the run must go on over the second-choice transport)
    if world > 1 and n_dev and world > n_dev and comm_kind == "rccl" and not fail_rccl:
        comm_kind = "ipc"
        if rank == 0:
            print(f"[bench] WARNING: {world} ranks on {n_dev} GPU(s): RCCL cannot run with ranks sharing a device; using the peer-store transport (FOAMYADE_COMM=ipc), "
                  f"the ranks TIME-SHARE the GPU(s) -- this is not a scaling measurement", file=sys.stderr, flush=True)
    local_dev = local_rank % max(n_dev, 1)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_dev)
        if comm_kind == "ipc":
            dist.init_process_group(backend="gloo")
        else:
            if fail_rccl:
                dist.init_process_group(backend="nccl")          # (lazy: the test hook may run with the ranks on ONE GPU, where RCCL itself could not connect)
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_dev))
            # a gloo group beside it: where the ranks agree on the path they take (an agreement that does not depend on RCCL's health) and the bootstrap of the
            # second-chance transport -- if the RCCL slab set-up fails the run goes over the library's own peer stores (fy_comm_create_ipc) before it gives up on slabs
            fb_group = dist.new_group(backend="gloo")
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product has no CPU path")
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    ddev = torch.device("cpu") if (dist is not None and comm_kind == "ipc") else dev      # where this script's own torch.distributed tensors live
    local_rank = local_dev

    prod = ge.load_product()
    c2, c5 = args.config == "c2", args.config == "c5"
    strong = bool((args.strong or c5) and world > 1)           # C5 on N GPUs is the ONE 320^3 box cut into N slabs (BASELINE configs[4])
    if c2:
        if world > 1:
            raise SystemExit("bench.py --config c2 is the single-GPU configuration (BASELINE configs[1])")
        if args.particles == 10_000_000:
            args.particles = 1_000_000
        if args.dt == 1e-4:
            args.dt = 2e-3
    if c5:
        if args.n == 160:
            args.n = 320
        if args.particles == 10_000_000:
            args.particles = 100_000_000
        if args.n % world or (args.n // world) % 2 or args.n // world < 10:
            raise SystemExit(f"bench.py --config c5: {args.n} planes do not cut into {world} slabs of an even number >= 10 of planes")
    case = c2_case(prod, args.dt, args.p_solver) if c2 else c5_case(prod, args.dt, args.p_solver, args.n) if c5 else c3_case(prod, args.n, args.dt, args.p_solver, world, strong)
    comm, solver, setup_err = None, None, ""
    ipc_comm = None
    try:
        if world > 1 and comm_kind == "rccl" and fail_rccl:
            raise RuntimeError("FOAMYADE_BENCH_FAIL_RCCL is set (a test of the fallback chain)")
        if world > 1 and comm_kind == "rccl" and not os.environ.get("FOAMYADE_BENCH_NO_PREFLIGHT"):
            why = rccl_preflight(prod, torch, dist, dev, rank, world)
            t = torch.tensor([0.0 if why else 1.0], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=fb_group)  # every rank must take the same path (agreed over gloo)
            if float(t.item()) < 1.0:
                raise RuntimeError(why or "RCCL self-test failed on another rank")
        if world > 1 and comm_kind == "ipc":
            os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
            ipc_comm = prod.GlooIpcComm(dist, local_rank)
            comm = ipc_comm.handle
        elif world > 1 or args.force_rccl:
            # one RCCL communicator for the slab exchanges; the 128-byte unique id travels over torch.distributed
            os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
            idt = torch.zeros(128, dtype=torch.uint8, device=dev)
            if rank == 0:
                idt = torch.tensor(list(prod.rccl_unique_id()), dtype=torch.uint8, device=dev)
            if dist is not None:
                dist.broadcast(idt, 0)
            comm = prod.rccl_comm(rank, world, bytes(idt.cpu().tolist()), local_rank)
        solver = prod.Solver(case, device=local_rank, comm=comm)
    except Exception as e:                                       # noqa: BLE001  (reported below, never swallowed)
        setup_err = f"{type(e).__name__}: {e}"
    # every rank must take the same path: agree on whether the slab set-up worked everywhere
    def agree(ok):
        if dist is None:
            return ok
        t = torch.tensor([ok], dtype=torch.float64, device=torch.device("cpu") if fb_group is not None else ddev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=fb_group)
        return float(t.item())
    slabs_ok = agree(0.0 if setup_err else 1.0)
    rccl_err = ""
    if slabs_ok < 1.0 and world > 1 and comm_kind == "rccl" and fb_group is not None:
        # second chance, LOUD: the same slabs over the library's peer-store transport, bootstrapped over the gloo group; this script's own collectives move there too
        rccl_err = setup_err or "on another rank"
        print(f"[bench rank {rank}] WARNING: the RCCL slab set-up failed ({rccl_err}); trying the peer-store transport (fy_comm_create_ipc)", file=sys.stderr, flush=True)
        setup_err = ""
        try:
            if solver is not None:
                solver.close(); solver = None
            os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
            ipc_comm = prod.GlooIpcComm(dist, local_rank, group=fb_group)
            comm = ipc_comm.handle
            solver = prod.Solver(case, device=local_rank, comm=comm)
        except Exception as e:                                   # noqa: BLE001
            setup_err = f"{type(e).__name__}: {e}"
        slabs_ok = agree(0.0 if setup_err else 1.0)
        if slabs_ok >= 1.0:
            comm_kind, ctl_group, ddev = "ipc", fb_group, torch.device("cpu")
        else:
            setup_err = f"RCCL: {rccl_err}; peer stores: {setup_err or 'on another rank'}"
    if fb_group is not None and ctl_group is None and slabs_ok < 1.0:
        ctl_group, ddev = fb_group, torch.device("cpu")         # (replicas: nothing below may depend on RCCL either)
    via = "RCCL halos + all-reduces over xGMI" if comm_kind == "rccl" else ("peer stores into hipIpc-mapped device windows (fy_comm_create_ipc)" +
                                                                            (f"; {world} ranks TIME-SHARE {n_dev} GPU(s): not a scaling measurement" if world > n_dev else "") +
                                                                            (f"; SECOND CHOICE: the RCCL set-up failed ({rccl_err})" if rccl_err else ""))
    parallelism = "single GPU" if world == 1 else f"{world} z-slabs of one {args.n}x{args.n}x{args.n if strong else args.n * world} box, {via}"
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
        c5 = False
        if args.config == "c5":
            args.n, args.particles = 160, 10_000_000
        case = c3_case(prod, args.n, args.dt, args.p_solver, 1)
        solver = prod.Solver(case, device=local_rank)
        parallelism = f"FALLBACK: {world} independent replicas of the single-GPU C3 case, no exchange (z-slab/RCCL set-up failed: {setup_err or 'on another rank'})"
        slab_of_rank = 0
    if c2:
        rec = c2_particles(torch, args.particles, dev)
    elif c5:
        rec = c5_particles(torch, args.particles, args.n, dev, rank, world)
    elif strong and slabs_ok >= 1.0:
        rec = c3_particles_strong(torch, args.particles, args.n, dev, rank, world)
    else:
        strong = False
        rec = c3_particles(torch, args.particles, args.n, 3 + rank, dev, slab=slab_of_rank)
    solver.set_particles_device(rec)
    solver.enable_particle_timing(True)
    nc = 200 * 100 * 50 if c2 else args.n ** 3 // (world if strong else 1)      # cells per rank

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=ctl_group)
        torch.cuda.synchronize()

    rebuild_ms = 0.0                        # the placement is rebuilt in the first two steps (binning; then once more, ordered by the chain lengths)
    for _ in range(args.warmup):
        solver.step()
        rebuild_ms = max(rebuild_ms, solver.coupling_timings()["bin"])
    solver.enable_kernel_timing(True)
    dxm = 0.01 if c2 else 1.0 / args.n
    phase_keys = ("particle", "locate_deposit", "force", "bin", "finalize", "fold", "momentum", "pressure", "other")

    def timed_region(steps, moving):
        """EXACTLY `steps` steps between two barriers (+ device synchronisations); moving: random velocities (+-0.05 m/s) and alternating random
        offsets of ~0.1 dx per step applied between the steps, inside the region"""
        acc = dict.fromkeys(phase_keys, 0.0)
        acc.update(p_iters=0, u_iters=0)
        jitter = None
        if moving:
            gj = torch.Generator(device="cpu").manual_seed(77)
            jitter = ((torch.rand(rec.shape[0], 3, dtype=torch.float64, generator=gj) - 0.5) * 0.2 * dxm).to(dev)
            rec[:, 3:6] = ((torch.rand(rec.shape[0], 3, dtype=torch.float64, generator=gj) - 0.5) * 0.1).to(dev)
            for it_ in range(2):                            # the displacement kernels load on first use: not in the timed region
                rec[:, 0:3] += jitter if it_ % 2 == 0 else -jitter
                solver.step()
        barrier()
        t0 = time.perf_counter()
        for it_ in range(steps):
            if jitter is not None:
                rec[:, 0:3] += jitter if it_ % 2 == 0 else -jitter
            solver.step()
            st = solver.stats(); ct = solver.coupling_timings()
            acc["particle"] += st["ms_particle"]; acc["momentum"] += st["ms_momentum"]; acc["pressure"] += st["ms_pressure"]; acc["other"] += st["ms_other"]
            acc["locate_deposit"] += ct["locate_deposit"]; acc["force"] += ct["force"]; acc["bin"] += ct["bin"]; acc["finalize"] += ct["finalize"]; acc["fold"] += ct["fold"]
            if rebuild_ms > 0 and ct["bin"] > 0.3 * rebuild_ms:
                acc["rebuilds"] = acc.get("rebuilds", 0) + 1
            acc["p_iters"] += st["p_iters_total"]; acc["u_iters"] += st["u_iters_total"]
        barrier()
        return acc, max_over_ranks(time.perf_counter() - t0, dist, ddev, ctl_group)

    acc, elapsed = timed_region(args.steps, args.moving)

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
    np_part = int(rec.shape[0])                                    # particles of this rank
    np_global = args.particles if (strong or world == 1) else np_part * world
    cand = {
        # name: (total ms over the timed steps, launches, algorithmic bytes per launch, description) -- DESIGN.md section 3.  The particle
        # kernels' durations are those of the kernel ALONE, bracketed by HIP events on its stream (fy_particle_timings)
        "k_locate_deposit": (acc["locate_deposit"], K, LOCATE_BYTES_PER_PARTICLE * np_part + 64.0 * nc, LOCATE_WHAT),
        "k_force_gaussian": (acc["force"], K, FORCE_BYTES_PER_PARTICLE * np_part + 128.0 * nc, FORCE_WHAT),
        "k_point_force": (acc["force"], K, 128.0 * np_part, "findCell + Stokes drag / torque + uSource scatter: 80 B record in, 48 B force out per particle"),
        "k_mg_smooth(level 0)": (smooth_ms, smooth_n, 56.0 * nc, "k_mg_smooth / k_mg_smooth_dot at level 0: pEqn Laplacian apply fused with the Jacobi update (and the z.r partials): 48 B/cell (diag, 3 upper, x, y) + b 8"),
        "k_p_apply_dot": (apply_ms, apply_n, 48.0 * nc, "pEqn Laplacian apply y = A p (+ p.Ap) inside PCG: 48 B/cell"),
        "k_mom_pass": (mom_ms, mom_n, (7 * 8 + 24 * 3) * nc, "fused momentum Jacobi pass: 7 coeffs + b,x,xn (3 comps)"),
    }
    for gone in (("k_locate_deposit", "k_force_gaussian") if c2 else ("k_point_force",)):
        cand.pop(gone)
    compulsory = {"k_locate_deposit": LOCATE_COMPULSORY[0] * np_part + LOCATE_COMPULSORY[1] * nc, "k_force_gaussian": FORCE_COMPULSORY[0] * np_part + FORCE_COMPULSORY[1] * nc}
    for nm, (ms, nl, bytes_per, desc) in cand.items():
        if nl:
            avg = ms / nl
            kern[nm] = dict(total_ms=ms, launches=int(nl), avg_ms=avg, achieved_GBps=bytes_per / (avg * 1e-3) / 1e9, alg_bytes=bytes_per, what=desc,
                            compulsory=compulsory.get(nm, bytes_per))
    dominant = max(kern, key=lambda k: kern[k]["total_ms"]) if kern else None
    lap = "k_mg_smooth(level 0)" if "k_mg_smooth(level 0)" in kern else "k_p_apply_dot"

    def roof(name):
        k = kern[name]
        # achieved / frac count SURVEY.md 8(d)'s ALGORITHMIC bytes -- every array of the path once (for the particle kernels: 80 Np + 64 Nc and
        # 48 Np + 168 Nc) -- divided by the live HIP-event duration; what the design itself moves on top of that (binned SoA copy, stencil rows,
        # placement) is reported beside it as design_*, never as `frac`
        comp_gbps = k["compulsory"] / (k["avg_ms"] * 1e-3) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": round(comp_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(comp_gbps / HBM_PEAK_GBPS, 4), "traffic": None, "avg_launch_ms": round(k["avg_ms"], 4),
                "launches": k["launches"], "algorithmic_bytes_per_launch": k["compulsory"], "what": k["what"],
                "design_bytes_per_launch": k["alg_bytes"], "design_GBps": round(k["achieved_GBps"], 1), "design_frac": round(k["achieved_GBps"] / HBM_PEAK_GBPS, 4)}

    default_c3 = (not c2) and (not c5) and args.n == 160 and args.particles == 10_000_000
    default_c5 = c5 and args.n == 320 and args.particles == 100_000_000
    p_iters, u_iters = acc["p_iters"] / K, acc["u_iters"] / K
    n_corr = 2
    # compulsory bytes of one coupled step -- particle phase 128 Np + 232 Nc; FV passes (280 + nCorr x 504) Nc; 128 Nc per
    # Krylov iteration (pressure and momentum sweeps alike)
    step_bytes = 128.0 * np_part + 232.0 * nc + (280.0 + n_corr * 504.0) * nc + 128.0 * nc * (p_iters + u_iters + 1)
    ms_step = 1e3 * elapsed / K
    if c2:
        metric = "coupled_steps_per_sec (icoFoamYade point force, 1M particles / 1M cells)" if args.particles == 1_000_000 else f"coupled_steps_per_sec (icoFoamYade point force, {args.particles} particles / {nc} cells)"
        workload = "C2: icoFoamYade point-force coupling, 200 x 100 x 50 = 1,000,000-cell channel (inlet U = (1,0,0), outlet p = 0, no-slip walls), " + f"{args.particles:,} particles uniform in the channel"
    elif c5:
        metric = ("coupled_steps_per_sec (pimpleFoamYade 4-way dense fluidized bed, 100M particles / 32M cells)" if default_c5
                  else f"coupled_steps_per_sec (pimpleFoamYade 4-way fluidized bed, {args.particles} particles / {args.n ** 3} cells)")
        workload = (f"C5: pimpleFoamYade Gaussian 4-way coupling, {args.n}^3 = {args.n ** 3:,}-cell fluidized bed (bottom inlet U = (0,0,0.05), top outlet p = 0, "
                    f"no-slip side walls), {args.particles:,} particles in the lower third" + (f", cut into {world} z-slabs" if world > 1 else ", on ONE GPU"))
    else:
        metric = "coupled_steps_per_sec (pimpleFoamYade 4-way, 10M particles / 4M cells per GPU)" if default_c3 else f"coupled_steps_per_sec (pimpleFoamYade 4-way, {args.particles} particles / {nc} cells)"
        workload = ("C3: pimpleFoamYade Gaussian 4-way coupling, 160^3 = 4,096,000-cell closed box, 10,000,000 particles in the lower 60 %" if default_c3
                    else f"non-default C3-like case {args.n}^3 cells / {args.particles} particles")

    def per_step(a, k):
        return {q: round(a[q] / k, 3) for q in ("particle", "bin", "locate_deposit", "finalize", "force", "fold", "momentum", "pressure", "other")}

    out = {
        "metric": metric,
        "value": round(steps_per_s, 4), "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" + (" (MOVING cloud: not the BASELINE configuration, see --moving)" if args.moving else ""),
        "particle_steps_per_sec": round((K / elapsed) * np_global, 1),
        "coupled_steps_per_sec_of_the_whole_box": round(steps_per_s if strong else steps_per_s / world, 4),
        "config": {"workload": workload,
                   "cells": nc, "particles": np_part, "dt": args.dt, ("piso" if c2 else "pimple"): ({"nCorrectors": 2} if c2 else {"nOuterCorrectors": 1, "nCorrectors": 2}),
                   "p_solver": "PCG+MG V(2,2), Chebyshev-weighted Jacobi pairs" if args.p_solver == 1 else "PCG+Jacobi",
                   "p_tol": [case.p_tol, case.p_rel_tol, case.p_final_tol, case.p_final_rel_tol],
                   "parallelism": parallelism,
                   "global_cells": nc * world, "global_particles": np_global},
        "per_step_ms": per_step(acc, K),
        "placement_rebuild": {"every_n_steps": int(os.environ.get("FOAMYADE_REBIN_INTERVAL", "32")), "ms": round(rebuild_ms, 3),
                              "what": "the binned placement of the particles (counting sort + chain-length ordering) is rebuilt every n-th coupling step; a timed region "
                                      "shorter than n steps may hold none: `ms` is the rebuild measured in the warm-up, ms / n its share of a step "
                                      "(per_step_ms.bin holds what fell inside the region)"},
        # how many placement rebuilds fell inside the timed region (a 20-step region holds none; its share of a step would be placement_rebuild.ms / every_n_steps)
        "rebuilds_in_region": int(acc.get("rebuilds", 0)),
        "ms_per_step_with_rebuild_share": round(ms_step + (0.0 if acc.get("rebuilds", 0) else rebuild_ms / max(int(os.environ.get("FOAMYADE_REBIN_INTERVAL", "32")), 1)), 4),
        "p_iters_per_step": p_iters, "u_iters_per_step": u_iters,
        "roofline": roof(dominant) if dominant else None,
        "roofline_pEqn_laplacian": roof(lap) if lap in kern else None,
        "whole_step": {"compulsory_bytes": step_bytes, "achieved_GBps": round(step_bytes / (ms_step * 1e-3) / 1e9, 1),
                       "frac_of_hbm_peak": round(step_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                       "what": "SURVEY.md 8(d) compulsory bytes of one coupled step (particle phase 128 Np + 232 Nc; FV passes (280 + nCorr 504) Nc; 128 Nc per "
                               "Krylov iteration) / ms_per_step"},
        "kernels": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"], "GBps": round(v["compulsory"] / (v["avg_ms"] * 1e-3) / 1e9, 1),
                        "design_GBps": round(v["achieved_GBps"], 1)} for k, v in kern.items()},
    }
    if world > 1 and slabs_ok >= 1.0:
        # how long each rank's stream sat waiting for slab exchanges, by phase -- sampled in a few EXTRA steps after the timed region (the clock costs an
        # event pair per exchange, which would slow the steps it measures).  FOAMYADE_HALO_OVERLAP=0 runs the exchange-then-consume schedule: the A/B
        try:
            solver.enable_exchange_timing(True)
            n_x = 5
            for _ in range(n_x):
                solver.step()
            torch.cuda.synchronize()
            w = solver.exchange_wait()
            mine = torch.tensor([w[k][0] / n_x for k in ("step_start", "particle", "momentum", "corrector")] +
                                [float(w[k][1]) / n_x for k in ("step_start", "particle", "momentum", "corrector")], dtype=torch.float64, device=ddev)
            allw = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allw, mine, group=ctl_group)
            solver.enable_exchange_timing(False)
            out["exchange_wait"] = {
                "what": "ms per step each rank's stream waited for slab exchanges (sampled over %d extra steps after the timed region): with the exchanges overlapped "
                        "(default) from the end of the interior planes' sweep to the ghost planes' arrival, with FOAMYADE_HALO_OVERLAP=0 the exchange itself; "
                        "`waits` = exchanges sampled per step; the particle phase's exchanges run inside the coupling object and are not sampled" % n_x,
                "overlap": os.environ.get("FOAMYADE_HALO_OVERLAP", "1") != "0",
                "per_rank": [{"rank": r, "ms_per_step": {k: round(float(t[q]), 4) for q, k in enumerate(("step_start", "particle", "momentum", "corrector"))},
                              "waits_per_step": {k: round(float(t[4 + q]), 1) for q, k in enumerate(("step_start", "particle", "momentum", "corrector"))}}
                             for r, t in enumerate(allw)]}
        except Exception as e:                                        # noqa: BLE001  (reported in the line)
            out["exchange_wait"] = {"error": f"{type(e).__name__}: {e}"}
    if default_c3 and world == 1 and not args.moving and not args.no_moving:
        # the BASELINE cloud is at rest and identical every step -- the workload's best case (1 PCG iteration per solve, zero momentum deposits
        # skipped, re-bin amortised).  The same run with a cloud that MOVES, for the same number of steps, so that the spread is in this record
        rec0 = rec.clone()
        macc, mel = timed_region(K, True)
        out["moving"] = {"what": "same case, same step count, particles with random velocities (+-0.05 m/s) displaced by ~0.1 dx per step between the steps "
                                 "(inside its timed region); NOT the BASELINE configuration -- reported beside `value`, never as it",
                         "value": round(K / mel, 4), "unit": "steps/s", "ms_per_step": round(1e3 * mel / K, 3), "p_iters_per_step": macc["p_iters"] / K,
                         "per_step_ms": per_step(macc, K)}
        rec.copy_(rec0)
        del rec0
    if rank == 0 and world == 1 and default_c3 and not args.moving and not args.no_extras:
        # driver-observed since round 4: BASELINE configs[1] at full size, and the pEqn Laplacian where no cache holds it
        try:
            out["c2"] = c2_subrecord(prod, torch, dev, local_rank)
        except Exception as e:                                        # noqa: BLE001  (reported in the line, never fatal for the headline)
            out["c2"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            out["general_mesh"] = general_mesh_subrecord()
        except Exception as e:                                        # noqa: BLE001
            out["general_mesh"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if args.wire_workers < 0:                                       # a process per core: master + workers + computing rank + helpers (round 5 sweep: 6 + 6 on 16 cores)
        args.wire_workers = max(2, min(6, (cpu_info()[1] - 2) // 2))
    if args.wire_helpers < 0:
        args.wire_helpers = args.wire_workers
    if rank == 0 and world == 1 and args.wire > 0:
        # the drop-in leg needs the device memory: the HBM-resident solver is done
        rec_host = rec.cpu().numpy()
        solver.close(); solver = None
        del rec
        torch.cuda.empty_cache()
        try:
            inproc = wire_leg(prod, torch, case, rec_host, args.wire, args.wire_workers, local_rank)
        except Exception as e:                                        # noqa: BLE001  (reported in the line, never fatal for the headline)
            inproc = {"error": f"{type(e).__name__}: {e}"}
        del rec_host
        mpi = None if c2 else wire_leg_mpi(args.n, args.particles, args.wire, args.wire_workers, args.dt, c5, args.wire_helpers + 1)
        mpi1 = None if (c2 or args.wire_helpers == 0) else wire_leg_mpi(args.n, args.particles, args.wire, args.wire_workers, args.dt, c5, 1)
        # the leg of record is the one over real MPI (what a Yade next to this library sees); the in-process peer (host copies at memcpy
        # speed, no process boundary) is the floor any copying transport has
        out["drop_in_path"] = mpi if (mpi and "error" not in mpi) else inproc
        if mpi is not None:
            out["drop_in_path_in_process_peer"] = inproc
        if mpi1 is not None:
            out["drop_in_path_one_receiving_rank"] = mpi1
        dp = out["drop_in_path"]
        if "per_step_ms" in dp:
            out["per_step_ms"].update({"h2d": dp["per_step_ms"]["h2d"], "d2h": dp["per_step_ms"]["d2h"],
                                       "wire": round(dp["per_step_ms"]["wire_recv"] + dp["per_step_ms"]["wire_send"], 3)})
            out["per_step_ms_note"] = "h2d / d2h / wire are the drop-in leg's (host buffers through the transport), measured after the timed region; every other entry and `value` are the HBM-resident run"
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        model, ncores = cpu_info()
        full = args.cpu_sample_n == 0 and not c5                  # C5 at full size would be minutes per CPU step: a 96^3 sample of it by default
        n_s = (100 if c2 else args.n) if full else (args.cpu_sample_n or 96)
        nc_s = (n_s ** 3 if not c2 else 1_000_000)
        n_part_cpu = args.particles if full else int(round(args.particles / nc * nc_s))
        if solver is not None:                                    # (the parity leg builds a fresh solver of the same size)
            solver.close(); solver = None
            torch.cuda.empty_cache()
        per, snc, native, parity = cpu_baseline(args.config, n_s, n_part_cpu, args.dt, ncores, full, torch, prod, args.p_solver, local_rank)
        (t_all, c_all), (t_one, c_one) = per[ncores], per[1]
        v_all, v_one = (c_all / nc) / t_all, (c_one / nc) / t_one      # steps/s of the bench workload (linear-in-size scaling where a sample was timed)
        best_th = ncores if v_all >= v_one else 1
        out["cpu_baseline"] = {
            "value": round(max(v_all, v_one), 6), "unit": "steps/s", "cores": int(best_th), "kind": "port",
            "single_thread_value": round(v_one, 6), "cpu_model": model, "host_cores_usable": ncores, "host_cores_present": os.cpu_count(),
            "sample": (f"CPU oracle (port of the reference path, de-quadraticised deposit; g++ -O3{' -march=native, compiled on this host' if native else ''}, no FMA contraction): on {ncores} threads "
                       + ("the bench workload itself at full size" if full else f"a {c_all / nc:.4f} sample of the bench workload with the same particles per cell")
                       + f" ({c_all} cells / {n_part_cpu} particles), one warm-up + 1 timed step = {t_all:.2f} s/step; on 1 thread a half-edge sample of that "
                       f"({c_one} cells / {n_part_cpu // 8} particles) = {t_one:.2f} s/step; samples scaled to the bench size linearly in the cell count")}
        # both steps of the multi-threaded leg as they were clocked (the first also touches every array for the first time; `value` is the second)
        out["cpu_baseline"]["step_seconds"] = getattr(cpu_baseline, "step_seconds", None)
        if parity is not None:
            out["cpu_baseline"]["parity_at_bench_size"] = parity
        # (the reference's own particle path built in the development container, oracle/_ref, is a CHECKER input -- the golden fixtures come from it; it is
        # not run on the GPU box unless asked for: BASELINE.md section 3 has the CPU restatement as the baseline of record)
        ref = cpu_reference_as_written() if (args.ref_as_written and not (c2 or c5)) else None
        if ref is not None:
            out["cpu_reference_as_written"] = ref
            pp = args.particles * steps_per_s
            if "particle_steps_per_sec" in ref:
                ref["gpu_particle_steps_per_sec_whole_coupled_step"] = round(pp, 1)
    if rank == 0 and world == 1:
        # roofline.traffic: measured live or null (never a committed figure passed off as this run's)
        committed = sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        note = "not measured in this run" + (f"; the builder's committed passes of the same command: {os.path.relpath(committed[-1], ROOT)}" if committed else "")
        want = args.pmc == 1 or (args.pmc < 0 and default_c3 and not args.moving)
        if want:
            solver_closed = solver is None
            if not solver_closed:
                solver.close(); solver = None
            torch.cuda.empty_cache()
            tr, why = live_pmc_traffic((["--config", args.config] if args.config != "c3" else []), nc)
            if tr:
                note = ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this command (5 timed steps each), bytes per launch = "
                        "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 (`traffic`; the doubled reads are an upper bound for the gather-dominated particle kernels) and "
                        "(FETCH_SIZE + WRITE_SIZE) x 1024 (`traffic_reads_undoubled`: the lower bound)")
                for rf in ("roofline", "roofline_pEqn_laplacian"):
                    if out.get(rf) and out[rf]["kernel"] in tr:
                        out[rf]["traffic"], out[rf]["traffic_reads_undoubled"] = (round(x) for x in traffic_of(tr[out[rf]["kernel"]]))
                out["traffic_per_launch"] = {k: round(traffic_of(v)[0]) for k, v in tr.items()}
                out["traffic_per_launch_reads_undoubled"] = {k: round(traffic_of(v)[1]) for k, v in tr.items()}
            else:
                note = f"live PMC pass failed ({why}); " + note
        out["traffic_note"] = note
        if default_c3 and not args.moving and not args.no_extras and args.pmc != 0:
            if solver is not None:
                solver.close(); solver = None
            torch.cuda.empty_cache()
            out["roofline_pEqn_laplacian_past_infinity_cache"] = laplacian_past_cache()
    if dist is not None:
        dist.barrier(group=ctl_group)
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # RCCL prints a version banner through C stdio: get it out BEFORE the JSON line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        if solver is not None:
            solver.close(); solver = None
        if ipc_comm is not None:
            ipc_comm.close()                    # (collective: the windows are unmapped behind a barrier of the bootstrap group)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
