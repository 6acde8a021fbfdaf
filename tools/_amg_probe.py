import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
import poly_meshes as pm
prod = ge.load_product()
for n in (16, 32, 64):
    L = 0.1
    mesh = pm.hex_block_fast(n, n, n, (L, L, L))
    lid = [(0, 0, 0)] * 6; lid[3] = (1.0, 0, 0)
    dt = 0.2 * (L / n)
    kw = dict(p_tol=1e-9, p_rel_tol=0.0, p_final_tol=1e-9, u_tol=1e-9, p_max_iter=2000)
    s = prod.LduSolver(mesh, dt, 1e-4, [0] * 6, lid, [0] * 6, p_solver=prod.FY_PSOLVER_PCG_MG, **kw)
    c = prod.make_case(prod.FY_SOLVER_ICO, n, n, n, L / n, dt, 1e-4, u_val=lid, p_solver=prod.FY_PSOLVER_PCG_MG, **kw)
    h = prod.Solver(c)
    a = []; b = []
    for _ in range(3):
        s.step(); h.step(); a.append(s.stats()["p_iters_total"]); b.append(h.stats()["p_iters_total"])
    # contraction of the LDU cycle: power iteration
    rs = np.random.RandomState(1)
    e = rs.standard_normal(n ** 3)
    for it in range(30):
        e = e - s.apply("p_precondition", s.apply("p_matrix", e))
        nr = np.sqrt(e @ s.apply("p_matrix", e)); e /= nr
    print(n, "ldu its", a, "structured its", b, "ldu asymptotic contraction", nr, "U diff", np.abs(s.get("U").reshape(-1,3) - h.get("U").reshape(-1,3)).max())
    s.close(); h.close()
