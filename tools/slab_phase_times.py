#!/usr/bin/env python3
"""Where a slab rank spends its step compared with the single domain: phase times (fy_step_stats / fy_particle_timings) of rank 0 of S
virtual slabs (160^3 cells + 10 M particles each, all on one GPU, run ONE SLAB AT A TIME is not possible -- the slabs step together, so the
kernels of the S slabs share the GPU and every phase is inflated by about S; compare the RATIOS between phases, and the kernel counts).
usage: slab_phase_times.py [S] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
import torch  # noqa: E402

prod = ge.load_product()
n, npart = 160, 10_000_000


def run(slabs):
    case = bench.c3_case(prod, n, 1e-4, 1, slabs)
    if slabs == 1:
        solvers = [prod.Solver(case)]
        step = solvers[0].step
    else:
        vs = prod.VirtualSlabs(case, slabs)
        solvers, step = vs.solvers, vs.step
    recs = [bench.c3_particles(torch, npart, n, 3 + r, torch.device("cuda", 0), slab=r) for r in range(slabs)]
    for s, r in zip(solvers, recs):
        s.set_particles_device(r)
        s.enable_particle_timing(True)
    for _ in range(2):
        step()
    acc = {}
    for _ in range(steps):
        step()
        st = solvers[0].stats()
        for k in ("ms_particle", "ms_momentum", "ms_pressure", "ms_other", "ms_total", "p_iters_total"):
            acc[k] = acc.get(k, 0.0) + st[k] / steps
    (vs.close() if slabs > 1 else solvers[0].close())
    return acc


a = run(1)
b = run(S)
print("single domain :", {k: round(v, 3) for k, v in a.items()})
print(f"rank 0 of {S}   :", {k: round(v, 3) for k, v in b.items()}, "(all slabs share the GPU: divide by ~%d)" % S)
print("ratio / S     :", {k: round(b[k] / a[k] / (S if k != "p_iters_total" else 1), 3) for k in a})
