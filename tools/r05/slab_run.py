#!/usr/bin/env python3
"""S slabs of 160^3 cells + 10 M particles each on one GPU (S = 1: the single domain), a fixed number of steps -- the command
tools/r05/slab_kernels.sh profiles.  usage: slab_run.py S steps"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

S, steps = int(sys.argv[1]), int(sys.argv[2])
os.environ.setdefault("FOAMYADE_TREE_CACHE_DIR", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp")
import torch  # noqa: E402

prod = ge.load_product()
n, npart = 160, 10_000_000
case = bench.c3_case(prod, n, 1e-4, 1, S)
if S == 1:
    solvers = [prod.Solver(case)]
    step = solvers[0].step
else:
    vs = prod.VirtualSlabs(case, S)
    solvers, step = vs.solvers, vs.step
recs = [bench.c3_particles(torch, npart, n, 3 + r, torch.device("cuda", 0), slab=r) for r in range(S)]
for s, r in zip(solvers, recs):
    s.set_particles_device(r)
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"S={S}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
