#!/usr/bin/env python3
"""fy_ldu_solver (icoFoamYade on a general polyhedral mesh, LDU addressing) at scale: a lid-driven cavity on n^3 hexahedra written as a polyhedral mesh,
plain and wavy (non-orthogonal + skewed, one non-orthogonal corrector), against the structured fy_solver on the same lattice.
usage: ldu_bench.py [n=128] [steps=10] [kind=wavy|lattice|prisms] [particles=0] [p_solver=mg|diag] [nu=1e-4] [solver=ico|pimple]     -> one JSON line
pimple: a closed box under gravity (fixedFluxPressure walls), the cloud at rest in its lower 60 %, Gaussian 4-way coupling"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import poly_meshes as pm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kind = sys.argv[3] if len(sys.argv) > 3 else "wavy"
npart = int(sys.argv[4]) if len(sys.argv) > 4 else 0
psolver = sys.argv[5] if len(sys.argv) > 5 else "mg"
nu = float(sys.argv[6]) if len(sys.argv) > 6 else 1e-4
solver = sys.argv[7] if len(sys.argv) > 7 else "ico"
pimple = solver == "pimple"
prod = ge.load_product()
L = 0.1
t0 = time.time()
vm = None if kind == "lattice" else pm.wavy(0.15 * L / n * 4, (L, L, L))
mesh = pm.prism_block(n, n, max(n // 2, 1), (L, L, L), vm) if kind == "prisms" else pm.hex_block_fast(n, n, n, (L, L, L), vm)
t_mesh = time.time() - t0
lid = [(0, 0, 0)] * 6
lid[3] = (1.0, 0, 0)
dt = 0.2 * (L / n) / 1.0
kw = dict(n_non_orth=0 if kind == "lattice" else 1, p_tol=1e-6, p_rel_tol=0.05, p_final_tol=1e-6, u_tol=1e-5, p_max_iter=2000,
          p_solver=prod.FY_PSOLVER_PCG_MG if psolver == "mg" else prod.FY_PSOLVER_PCG_JACOBI)
t0 = time.time()
if pimple:
    dt = 1e-4
    kw.update(solver=1, g=(0.0, 0.0, -9.81), n_outer_correctors=1, u_relax=1.0)
    s = prod.LduSolver(mesh, dt, 1e-6, [0] * 6, [(0, 0, 0)] * 6, [2] * 6, **kw)
else:
    s = prod.LduSolver(mesh, dt, nu, [0] * 6, lid, [0] * 6, **kw)
t_create = time.time() - t0
if npart:
    rs = np.random.RandomState(5)
    rec = np.zeros((npart, 10))
    rec[:, 0:3] = L * (0.1 + 0.8 * rs.rand(npart, 3)); rec[:, 9] = 1e-5 + 0.5e-5 * rs.rand(npart)          # (radius)
    if pimple:
        rec[:, 0:3] = L * rs.rand(npart, 3) * np.array([1.0, 1.0, 0.6]); rec[:, 9] = 0.2 * L / n
    s.set_particles(rec)
for _ in range(3):
    s.step()
its = uits = 0
ms = mp = 0.0
t0 = time.time()
for _ in range(steps):
    s.step()
    st = s.stats()
    its += st["p_iters_total"]; uits += st["u_iters_total"]; ms += st["ms_total"]; mp += st["ms_particle"]
wall = time.time() - t0
out = dict(tool="ldu_bench", kind=kind, cells=int(mesh["n_cells"]), faces=int(len(mesh["owner"])), steps=steps, steps_per_s=steps / wall, ms_per_step_wall=1e3 * wall / steps,
           ms_per_step_stream=ms / steps, ms_particle=mp / steps, solver=solver, pcg_iters_per_step=its / steps, p_solver=psolver, nu=nu, u_sweeps_per_step=uits / steps, us_per_pcg_iter=1e3 * ms / max(its, 1), courant_max=st["courant_max"],
           cont_err=st["cont_err_sum_local"], particles=npart, mesh_build_s=t_mesh, create_s=t_create, non_orth=kw["n_non_orth"])
s.close()
if kind == "lattice" and n % 8 == 0:
    # the structured solver on the same block with the same controls (PCG + diagonal preconditioner)
    c = prod.make_case(prod.FY_SOLVER_ICO, n, n, n, L / n, dt, nu, u_val=lid, p_tol=1e-6, p_rel_tol=0.05, p_final_tol=1e-6, u_tol=1e-5, p_max_iter=2000,
                       p_solver=prod.FY_PSOLVER_PCG_MG if psolver == "mg" else prod.FY_PSOLVER_PCG_JACOBI)
    h = prod.Solver(c)
    for _ in range(3):
        h.step()
    t0 = time.time(); its2 = 0
    for _ in range(steps):
        h.step(); its2 += h.stats()["p_iters_total"]
    w2 = time.time() - t0
    out.update(structured_ms_per_step=1e3 * w2 / steps, structured_pcg_iters_per_step=its2 / steps)
    h.close()
print(json.dumps(out))
