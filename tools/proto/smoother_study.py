#!/usr/bin/env python3
"""Study (CPU oracle, no GPU): PCG iterations of the multigrid-preconditioned pressure solve under other smoothers.
Run once per smoother -- the oracle reads ORACLE_MG_SMOOTHER at first use (a study-only switch: apply tools/proto/smoother_study.patch to oracle/fv_oracle.cpp first, `git apply`):
    0 two-sweep Chebyshev-Jacobi pairs (the shipped smoother)   1 symmetric red-black Gauss-Seidel, one sweep each way   2 two RB-GS sweeps each way
    3..6 Chebyshev-Jacobi of degree 3..6
Memory passes over a level per V-cycle (what a GPU pays; the residual and the transfers add 2 whatever the smoother): a Jacobi sweep is one pass, a RED-BLACK
half sweep is one pass too with interleaved storage (it touches every cache line), so RB-GS(1,1) = 4 passes = Chebyshev-2 pairs, RB-GS(2,2) = 8, Chebyshev-n = 2 n."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
import golden_cases as gc  # noqa: E402

mode = os.environ.get("ORACLE_MG_SMOOTHER", "0")
orc.build()
out = [f"smoother {mode}:"]
# (a) manufactured Poisson problem, 64^3 and 96^3, tolerance 1e-10
for n in (64, 96):
    s = orc.FvSolver(orc.fv_case(0, n, n, n, 1.0 / n, 1e-3, 1e-2, u_bc=[orc.U_FIXED] * 6, u_val=[(0, 0, 0)] * 6, p_solver=1, p_tol=1e-10, p_rel_tol=0.0, p_final_tol=1e-10, p_final_rel_tol=0.0), threads=8)
    s.step()
    rAU = s.get("rAU")
    c = (np.arange(n) + 0.5) / n
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    ps = (np.cos(np.pi * X) * np.cos(np.pi * Y) * np.cos(np.pi * Z)).ravel()
    x, it = s.solve_p(rAU[0] * (1.0 / n) ** 3 * 3 * np.pi ** 2 * ps)
    out.append(f"poisson {n}^3: {it} it")
    s.close()
# (b) lid-driven cavity 48^3, 12 steps
n = 48
u_val = [(0, 0, 0)] * 6; u_val[orc.YMAX] = (1.0, 0, 0)
s = orc.FvSolver(orc.fv_case(0, n, n, n, 1.0 / n, 0.4 / n, 0.01, u_bc=[orc.U_FIXED] * 6, u_val=u_val, p_solver=1), threads=8)
its = 0
for k in range(12):
    s.step(); its += s.stats()["p_iters_total"]
out.append(f"cavity 48^3: {its / 12:.2f} it/step")
s.close()
# (c) a coupled moving bed: pimple 40^3, 250 k particles with random velocities, displaced between the steps (a scaled-down C3 'moving')
n = 40
dx = 1.0 / n
case = orc.fv_case(1, n, n, n, dx, 1e-4, 1e-6, rho_f=1000.0, rho_p=2650.0, g=(0, 0, -9.81), u_bc=[orc.U_FIXED] * 6, u_val=[(0, 0, 0)] * 6, p_bc=[orc.P_FIXEDFLUX] * 6, p_solver=1)
s = orc.FvSolver(case, threads=8)
rs = np.random.RandomState(3)
npart = 250_000
rec = np.zeros((npart, 10)); rec[:, 0:3] = rs.random_sample((npart, 3)); rec[:, 2] *= 0.6; rec[:, 9] = 0.2 * dx
rec[:, 3:6] = (rs.random_sample((npart, 3)) - 0.5) * 0.1
its = 0
t0 = time.time()
for k in range(8):
    s.step(rec)
    if k >= 2: its += s.stats()["p_iters_total"]
    rec[:, 0:3] += rec[:, 3:6] * (0.1 * dx / 0.05)
    rec[:, 0:3] = np.clip(rec[:, 0:3], 0.02 * dx, 1 - 0.02 * dx)
out.append(f"moving bed 40^3: {its / 6:.2f} it/step")
s.close()
print("  ".join(out), flush=True)

# Result (round 4, this container, 8 threads) -- PCG iterations: Poisson 64^3 / 96^3 (tolerance 1e-10), cavity 48^3 and moving bed 40^3 per step
#   0 Chebyshev-Jacobi pairs (shipped; 4 smoothing passes per level and cycle)   13 / 13 /  9.42 / 5.67
#   1 symmetric RB-GS, one sweep each way (4 passes)                              20 / 22 / 14.75 / 8.50     <- loses at equal bytes
#   2 RB-GS, two sweeps each way (8 passes)                                       10 / 10 /  6.83 / 4.67
#   3 Chebyshev degree 3 (6 passes)                                               12 / 12 /  9.17 / 5.00
#   4 Chebyshev degree 4 (8 passes)                                                9 /  9 /  6.50 / 4.00     <- beats RB-GS(2,2) at equal bytes
# With ~190 us of level-0 smoothing passes in a ~420 us PCG iteration at C3, degree 4 costs 1.43 x per iteration for 0.71 x the iterations: neutral
# (and measured slower on the GPU in round 2); degree 3: 1.21 x for 0.88 x.  The shipped pairs stay.
