#!/usr/bin/env python3
"""Collective calls per coupled step of the slab solver, by phase (fy_comm_stats_by_tag), on S virtual slabs of one GPU.
usage: comm_count.py [n=160] [S=2] [particles per slab=10_000_000] [steps=4] [moving=0]   (the pimpleFoamYade C3 slab per rank)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
npart = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
moving = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
import torch  # noqa: E402

prod = ge.load_product()
case = bench.c3_case(prod, n, 1e-4, 1, S)
vs = prod.VirtualSlabs(case, S)
recs = [bench.c3_particles(torch, npart, n, 3 + r, torch.device("cuda", 0), slab=r) for r in range(S)]
if moving:
    for r in recs:
        r[:, 3:6] = (torch.rand(r.shape[0], 3, dtype=torch.float64, device=r.device) - 0.5) * 2 * moving
for s, r in zip(vs.solvers, recs):
    s.set_particles_device(r)
for _ in range(2):
    vs.step()
a = vs.comm_stats_by_tag(0)
it0 = vs.stats()[0]["p_iters_total"]
its = 0
for _ in range(steps):
    vs.step()
    its += vs.stats()[0]["p_iters_total"]
b = vs.comm_stats_by_tag(0)
tot = [0.0, 0.0, 0.0]
print(f"{S} virtual slabs of {n}^3 cells, {npart} particles each; per step over {steps} steps ({its / steps:.2f} PCG iterations per step):")
print(f"{'phase':18s} {'exchanges':>10s} {'all-reduces':>12s} {'all-gathers':>12s}")
for k in sorted(b):
    d = [(b[k][q] - a.get(k, (0, 0, 0))[q]) / steps for q in range(3)]
    tot = [tot[q] + d[q] for q in range(3)]
    print(f"{k:18s} {d[0]:10.2f} {d[1]:12.2f} {d[2]:12.2f}")
print(f"{'TOTAL':18s} {tot[0]:10.2f} {tot[1]:12.2f} {tot[2]:12.2f}   = {sum(tot):.1f} collectives per step")
vs.close()
