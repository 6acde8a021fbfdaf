#!/usr/bin/env python3
"""How much of a C3 step is the harness: the same 40 steps (a) as bench.py's timed region runs them (timing on, stats + coupling timings read every step),
(b) timing on, nothing read in the loop, (c) timing off, nothing read.  Development tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, __graft_entry__ as ge
import torch
prod = ge.load_product()
dev = torch.device("cuda", 0)
case = bench.c3_case(prod, 160, 1e-4, 1, 1, False)
s = prod.Solver(case, device=0)
rec = bench.c3_particles(torch, 10_000_000, 160, 3, dev, slab=0)
s.set_particles_device(rec)
for _ in range(5): s.step()
def run(timing, read, K=40):
    s.enable_particle_timing(timing); s.enable_kernel_timing(timing)
    for _ in range(2): s.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        s.step()
        if read: s.stats(); s.coupling_timings()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for rep in range(2):
    print("timing+read %.3f ms | timing, no read %.3f ms | no timing, no read %.3f ms" % (run(True, True), run(True, False), run(False, False)), flush=True)
