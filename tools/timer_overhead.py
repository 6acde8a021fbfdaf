#!/usr/bin/env python3
"""What the bench's own instrumentation costs: wall time of the C3 step with the HIP-event timers on and off (same process, alternating)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
import torch
prod = ge.load_product()
n, npart = 160, 10_000_000
case = bench.c3_case(prod, n, 1e-4, 1, 1)
s = prod.Solver(case)
s.set_particles_device(bench.c3_particles(torch, npart, n, 3, torch.device("cuda", 0), slab=0))
for _ in range(3): s.step()
def run(k=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): s.step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
for rep in range(3):
    s.enable_particle_timing(False); s.enable_kernel_timing(False); off = run()
    s.enable_particle_timing(True); s.enable_kernel_timing(False); part = run()
    s.enable_particle_timing(True); s.enable_kernel_timing(True); on = run()
    print(f"timers off {off:.3f} ms/step | phase timers {part:.3f} | phase timers + sampled kernel clocks {on:.3f}")
