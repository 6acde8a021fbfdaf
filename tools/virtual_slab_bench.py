#!/usr/bin/env python3
"""Cost of the slab logic itself, without a network: S virtual slabs (threads + device-to-device copies) of 160^3 cells and 10 M
particles each on ONE GPU, against S times the single-slab step.  usage: virtual_slab_bench.py [S] [steps]"""
import os
import sys
import time

import numpy as np

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
        stats = solvers[0].comm_stats if hasattr(solvers[0], "comm_stats") else None
    else:
        vs = prod.VirtualSlabs(case, slabs)
        solvers, step = vs.solvers, vs.step
    recs = [bench.c3_particles(torch, npart, n, 3 + r, torch.device("cuda", 0), slab=r) for r in range(slabs)]
    for s, r in zip(solvers, recs):
        s.set_particles_device(r)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    extra = ""
    if slabs > 1:
        c = vs.comm_stats(0)
        extra = f" | rank 0 so far: {c[0]} exchanges, {c[1]} all-reduces, {c[2]} all-gathers, {c[3] / 1e6:.1f} MB sent"
        vs.close()
    else:
        solvers[0].close()
    return ms, extra


one, _ = run(1)
many, extra = run(S)
print(f"single slab {one:.2f} ms/step; {S} virtual slabs on one GPU {many:.2f} ms/step = {many / (S * one):.3f} x ({S} x single){extra}")
