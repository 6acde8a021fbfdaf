#!/usr/bin/env python3
"""BASELINE configs[3] (the ONE C3 box, 160^3 cells / 10 M particles, cut into S z-slabs) on S VIRTUAL slabs of one GPU: what the slab logic costs when the
work per slab shrinks.  All S slabs share one GPU, so S x less work per slab does not show as S x less time: the figure to read is the growth over the
single-domain step.  usage: virtual_strong_bench.py [S=8] [steps=5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
import torch  # noqa: E402

prod = ge.load_product()
n, npart = 160, 10_000_000
dev = torch.device("cuda", 0)


def run(slabs):
    case = bench.c3_case(prod, n, 1e-4, 1, slabs, strong=True)
    if slabs == 1:
        solvers = [prod.Solver(case)]
        step = solvers[0].step
    else:
        vs = prod.VirtualSlabs(case, slabs)
        solvers, step = vs.solvers, vs.step
    for r, s in enumerate(solvers):
        s.set_particles_device(bench.c3_particles_strong(torch, npart, n, dev, r, slabs))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = 0
    for _ in range(steps):
        step()
        its += solvers[0].stats()["p_iters_total"]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    (vs.close() if slabs > 1 else solvers[0].close())
    return ms, its / steps


one, i1 = run(1)
many, iS = run(S)
print(f"configs[3] strong: single domain {one:.2f} ms/step ({i1:.1f} PCG it.); {S} virtual slabs on one GPU {many:.2f} ms/step ({iS:.1f} it.)")
