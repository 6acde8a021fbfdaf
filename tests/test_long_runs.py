"""Endurance of the product on the GPU: what a production run does that the parity tests (a handful of steps each) do not -- hundreds of coupled
steps with a cloud that moves, re-bins and changes size, and objects created and destroyed in a loop.  No oracle here: boundedness,
conservation and the device memory returning to where it was."""
import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_three_hundred_coupled_steps_of_a_settling_cloud(product):
    """pimpleFoamYade 4-way under gravity, 40 000 particles that fall, spread and are partly replaced every 50 steps (so the placement is rebuilt
    with and without chain lengths to order it by): every field stays finite and bounded, alpha within [0.1, 1], the continuity error and the
    pressure iterations small at every checked step"""
    n = 24
    L = 0.1
    dx = L / n
    case = product.make_case(product.FY_SOLVER_PIMPLE, n, n, 2 * n, dx, 2e-4, 1e-6, g=(0, 0, -9.81), u_bc=[0] * 6, p_bc=[2] * 6, p_solver=1)
    s = product.Solver(case)
    rs = np.random.RandomState(3)
    npart = 40000
    rec = np.zeros((npart, 10))
    rec[:, 0:2] = L * (0.05 + 0.9 * rs.random_sample((npart, 2)))
    rec[:, 2] = 2 * L * (0.3 + 0.6 * rs.random_sample(npart))
    rec[:, 9] = 0.15 * dx
    vel = np.zeros((npart, 3))
    for step in range(300):
        vel[:, 2] -= 2e-4 * 9.81 * (1 - 1000.0 / 2650.0)
        vel += 0.002 * rs.standard_normal((npart, 3))
        rec[:, 0:3] += 2e-4 * 50 * vel                               # (a coarse DEM stand-in: 50 sub-steps of free fall with jitter)
        rec[:, 0:2] = np.clip(rec[:, 0:2], 0.01 * L, 0.99 * L)
        low = rec[:, 2] < 0.02 * L
        rec[low, 2] = 0.02 * L; vel[low, 2] = 0.0
        rec[:, 3:6] = vel
        if step % 50 == 49:                                           # part of the population leaves, another arrives (the count changes)
            keep = rs.random_sample(rec.shape[0]) > 0.1
            rec, vel = rec[keep], vel[keep]
            m = 3000 + 500 * (step // 50)
            new = np.zeros((m, 10)); new[:, 0:2] = L * (0.05 + 0.9 * rs.random_sample((m, 2))); new[:, 2] = 2 * L * (0.8 + 0.15 * rs.random_sample(m)); new[:, 9] = 0.15 * dx
            rec = np.vstack([rec, new]); vel = np.vstack([vel, np.zeros((m, 3))])
            npart = rec.shape[0]
        s.hold_sources(True)
        s.set_particles(rec)
        s.step()
        if step % 25 == 24 or step == 299:
            U, p, a = s.get("U"), s.get("p"), s.get("alpha")
            F = s.forces()
            assert np.isfinite(U).all() and np.isfinite(p).all() and np.isfinite(F).all(), step
            assert np.abs(U).max() < 5.0 and a.min() >= 0.1 - 1e-12 and a.max() <= 1.0 + 1e-12 and a.min() < 1.0, (step, np.abs(U).max(), a.min())
            st = s.stats()
            assert st["cont_err_sum_local"] < 1e-4 and st["p_iters_total"] < 60, (step, st)
    s.close()


def test_objects_come_and_go_without_leaking_device_memory(product):
    n = 32
    rs = np.random.RandomState(0)
    rec = np.zeros((20000, 10)); rec[:, 0:3] = rs.random_sample((20000, 3)); rec[:, 9] = 0.2 / n

    def cycle():
        case = product.make_case(product.FY_SOLVER_PIMPLE, n, n, n, 1.0 / n, 1e-3, 1e-5, g=(0, 0, -9.81), u_bc=[0] * 6, p_bc=[2] * 6, p_solver=1,
                                 turbulence_model=product.TURBULENCE_KEPSILON, k_initial=1e-3, eps_initial=1e-2, nut_initial=1e-5)
        s = product.Solver(case)
        s.set_particles(rec)
        for _ in range(3):
            s.step()
        s.close()
        vs = product.VirtualSlabs(product.make_case(product.FY_SOLVER_ICO, n, n, n, 1.0 / n, 1e-3, 1e-2), 2)
        vs.step()
        vs.close()

    cycle()                                   # (first use: the runtime's own pools, kernel images, attribute caches)
    cycle()
    before = _free_bytes()
    for _ in range(10):
        cycle()
    after = _free_bytes()
    assert before - after < 8 << 20, (before, after)      # ten more cycles hold no more than noise (one cycle allocates ~150 MB)


def test_general_mesh_solver_two_hundred_coupled_steps_and_no_leak(product):
    """fy_ldu_solver, pimpleFoamYade with LES Smagorinsky and the multigrid-preconditioned PCG on wavy renumbered hexahedra: a cloud that settles over 200 coupled steps
    (fields finite and bounded, the void fraction within [floor, 1], continuity closed, PCG iterations small), then solvers created and destroyed in a loop"""
    import poly_meshes as pm
    n, L = 16, 0.1
    dx = L / n
    mesh = pm.hex_block(n, n, 2 * n, (L, L, 2 * L), pm.wavy(0.25 * dx, (L, L, 2 * L)), renumber_seed=2)
    mk = lambda: product.LduSolver(mesh, 2e-4, 1e-6, [0] * 6, [(0, 0, 0)] * 6, [2] * 6, solver=1, g=(0, 0, -9.81), n_non_orth=1, n_correctors=2, u_relax=1.0,
                                   p_solver=product.FY_PSOLVER_PCG_MG, turbulence_model=product.TURBULENCE_SMAGORINSKY, nut_initial=1e-6)
    s = mk()
    s.hold_sources(True)
    rs = np.random.RandomState(3)
    npart = 12000
    rec = np.zeros((npart, 10))
    rec[:, 0:2] = L * (0.05 + 0.9 * rs.random_sample((npart, 2)))
    rec[:, 2] = 2 * L * (0.3 + 0.6 * rs.random_sample(npart))
    rec[:, 9] = 0.15 * dx
    vel = np.zeros((npart, 3))
    for step in range(200):
        vel[:, 2] -= 2e-4 * 9.81 * (1 - 1000.0 / 2650.0)
        vel += 0.002 * rs.standard_normal((npart, 3))
        rec[:, 0:3] += 2e-4 * 50 * vel
        rec[:, 0:2] = np.clip(rec[:, 0:2], 0.01 * L, 0.99 * L)
        low = rec[:, 2] < 0.02 * L
        rec[low, 2] = 0.02 * L; vel[low, 2] = 0.0
        rec[:, 3:6] = vel
        s.set_particles(rec)
        s.step()
        if step % 40 == 39:
            U, p, a, nut = s.get("U"), s.get("p"), s.get("alpha"), s.get("nut")
            assert np.isfinite(U).all() and np.isfinite(p).all() and np.isfinite(s.forces()).all() and np.isfinite(nut).all(), step
            assert np.abs(U).max() < 5.0 and a.min() >= 0.1 - 1e-12 and a.max() <= 1.0 + 1e-12 and a.min() < 1.0 and nut.min() >= 0.0, (step, np.abs(U).max(), a.min())
            st = s.stats()
            assert st["cont_err_sum_local"] < 1e-4 and st["p_iters_total"] < 80, (step, st)
    s.close()

    def cycle():
        t = mk()
        t.set_particles(rec[:2000])
        t.step()
        t.close()
    cycle(); cycle()
    before = _free_bytes()
    for _ in range(8):
        cycle()
    assert before - _free_bytes() < 8 << 20
