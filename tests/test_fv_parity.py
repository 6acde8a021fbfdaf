"""GPU parity of the FV half: fy_solver (HIP, through the C-ABI) vs the CPU oracle (oracle/fv_oracle.cpp) on the same cases.

FV PARITY IS UNPINNED against the reference (its FV arithmetic is OpenFOAM-6 library code, absent here); the oracle is validated by
known-answer flows in tests/test_fv_oracle.py and this file checks that the HIP path reproduces the oracle.  Tolerances: both sides
run the same algorithm in FP64, differing in reduction order (and therefore, rarely, by one solver iteration), so fields agree to
the linear-solver tolerance, not to the bit."""
import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

XMIN, XMAX, YMIN, YMAX, ZMIN, ZMAX = range(6)


def both(product, oracle, solver, nx, ny, nz, dx, dt, nu, **kw):
    u_bc = kw.pop("u_bc", None); u_val = kw.pop("u_val", None); p_bc = kw.pop("p_bc", None); p_val = kw.pop("p_val", None)
    g = kw.pop("g", (0, 0, 0)); p_solver = kw.pop("p_solver", 1)
    oc = oracle.fv_case(solver, nx, ny, nz, dx, dt, nu, g=g, u_bc=u_bc, u_val=u_val, p_bc=p_bc, p_val=p_val, p_solver=p_solver, **kw)
    pk = dict(kw)
    if "n_outer" in pk: pk["n_outer_correctors"] = pk.pop("n_outer")
    if "n_corr" in pk: pk["n_correctors"] = pk.pop("n_corr")
    if "n_non_orth" in pk: pk["n_non_orth_correctors"] = pk.pop("n_non_orth")
    if "limiter_k" in pk: pk["convection_limiter_k"] = pk.pop("limiter_k")
    pc = product.make_case(solver, nx, ny, nz, dx, dt, nu, g=g, u_bc=u_bc, u_val=u_val, p_bc=p_bc, p_val=p_val, p_solver=p_solver, **pk)
    return oracle.FvSolver(oc), product.Solver(pc)


def cavity_bcs():
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (1.0, 0, 0)
    return dict(u_bc=[0] * 6, u_val=u_val)


def compare(o, s, names=("U", "p", "phi_x", "phi_y", "phi_z"), rtol=2e-6):
    for nm in names:
        a, b = s.get(nm), o.get(nm)
        sc = np.abs(b).max() + 1e-300
        assert np.abs(a - b).max() <= rtol * sc, (nm, np.abs(a - b).max() / sc)


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("p_solver", [0, 1])
def test_cavity_steps_match_oracle(product, oracle, solver, p_solver):
    n = 20
    o, s = both(product, oracle, solver, n, n, n, 1.0 / n, 0.4 / n, 0.01, p_solver=p_solver, **cavity_bcs())
    for step in range(6):
        o.step(); s.step()
        so, ss = o.stats(), s.stats()
        assert abs(so["p_iters_total"] - ss["p_iters_total"]) <= 2
        assert abs(so["u_iters_total"] - ss["u_iters_total"]) <= 1
        assert np.isclose(so["courant_max"], ss["courant_max"], rtol=1e-6)
    compare(o, s)
    assert ss["cont_err_sum_local"] < 1e-5


def test_operator_level_parity(product, oracle):
    """one step from a non-trivial state; compare the assembled matrices and the SpMV, which do not depend on solver iterations"""
    nx, ny, nz = 14, 10, 6
    u_bc = [1, 1, 0, 0, 0, 0]
    p_bc = [1, 1, 0, 0, 0, 0]
    o, s = both(product, oracle, 0, nx, ny, nz, 0.05, 0.01, 0.02, u_bc=u_bc, p_bc=p_bc, p_val=[0.3, 0.0, 0, 0, 0, 0])
    rs = np.random.RandomState(3)
    U0 = rs.rand(nx * ny * nz, 3) * 0.2
    o.set("U", U0); s.set("U", U0)
    o.step(); s.step()
    for nm in ("p_diag", "p_ux", "p_uy", "p_uz", "mom_diag", "rAU"):
        np.testing.assert_allclose(s.get(nm), o.get(nm), rtol=1e-9, atol=0, err_msg=nm)
    x = rs.rand(nx * ny * nz)
    np.testing.assert_allclose(s.apply_p(x), o.apply_p(x), rtol=1e-12, atol=1e-14 * np.abs(o.get("p_diag")).max())
    compare(o, s)


def test_odd_dims_multigrid(product, oracle):
    o, s = both(product, oracle, 1, 13, 9, 7, 0.02, 0.004, 0.005, **cavity_bcs())
    for _ in range(4):
        o.step(); s.step()
    compare(o, s)


def test_hydrostatic_fixed_flux(product, oracle):
    """a quiescent box under gravity is a fixed point of the discrete equations: what is left of U and of the error in grad p is the pressure
    solver's residual and nothing else -- so the bounds are the tight ones (1e-8 / 1e-5) at a tight solver tolerance, and at the default
    tolerance they scale with the residual the solver reports"""
    n = 12
    for tol, u_bound, g_rtol in ((1e-10, 1e-8, 1e-5), (1e-6, None, None)):
        o, s = both(product, oracle, 1, n, n, n, 0.1 / n, 1e-3, 1e-6, g=(0, 0, -9.81), p_bc=[2] * 6, p_tol=tol, p_final_tol=tol, p_rel_tol=0.0)
        for _ in range(3):
            o.step(); s.step()
        res = max(s.stats()["p_final_residual"], 1e-16)
        assert res <= tol
        p = s.get("p").reshape(n, n, n)
        gerr = np.abs((p[2:] - p[:-2]) / (2 * 0.1 / n) + 9.81).max() / 9.81
        umax = np.abs(s.get("U")).max()
        if u_bound is not None:
            assert umax < u_bound and gerr < g_rtol, (umax, gerr)
        else:                                            # default tolerance: bounded by the reported residual (same constants of proportionality)
            assert umax < 10.0 * res and gerr < 100.0 * res, (umax, gerr, res)
        compare(o, s, names=("p",), rtol=1e-5)
        o.close() if hasattr(o, "close") else None
        s.close()


@pytest.mark.parametrize("solver", [0, 1])
def test_coupled_steps_match_oracle(product, oracle, solver):
    """full loop body with particles: locate/interpolate/drag/back-scatter + PISO/PIMPLE, three coupled steps"""
    n = 16
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else cavity_bcs()
    nu = 1e-5 if solver == 1 else 0.01
    o, s = both(product, oracle, solver, n, n, n, dx, 2e-4, nu, **kw)
    case = gc.Case("cpl", n, n, n, 0.1, gaussian=solver, np_=3000, seed=9, cluster=150, fast=20, outside=20, vel_scale=0.05)
    for step in range(3):
        rec = gc.particle_records(case, step)
        fo = o.step(rec)["force"]
        s.set_particles(rec)
        s.step()
        fs = s.forces()
        sc = np.abs(fo).max()
        assert np.abs(fs - fo).max() <= 1e-6 * sc, np.abs(fs - fo).max() / sc
    compare(o, s, rtol=1e-5)
    assert np.abs(s.get("U")).max() > 0


def test_coupled_steps_with_optional_force_models(product, oracle):
    """pimple loop with fy_set_force_models(added mass | Gaussian torque): ddtU_f (pimpleFoamYade.C:73) and vGrad feed the models"""
    n = 16
    dx = 0.1 / n
    o, s = both(product, oracle, 1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), p_bc=[2] * 6)
    flags = product.FORCE_ADDED_MASS | product.FORCE_GAUSSIAN_TORQUE
    o.force_models = flags
    s.set_force_models(flags)
    case = gc.Case("cplx", n, n, n, 0.1, gaussian=1, np_=3000, seed=11, cluster=150, fast=20, outside=20, vel_scale=0.05)
    for step in range(3):
        rec = gc.particle_records(case, step)
        fo = o.step(rec)["force"]
        s.set_particles(rec)
        s.step()
        fs = s.forces()
        for cols in (slice(0, 3), slice(3, 6)):
            sc = np.abs(fo[:, cols]).max()
            assert sc > 0
            assert np.abs(fs[:, cols] - fo[:, cols]).max() <= 1e-6 * sc, (cols, np.abs(fs[:, cols] - fo[:, cols]).max() / sc)
        if step > 0:
            d_o, d_s = o.get("ddtU"), s.get("ddtU")
            assert np.abs(d_o).max() > 0 and np.abs(d_s - d_o).max() <= 1e-5 * np.abs(d_o).max()
    compare(o, s, rtol=1e-5)


@pytest.mark.parametrize("scheme", [1, 2])
@pytest.mark.parametrize("solver", [0, 1])
def test_upwind_convection_matches_oracle(product, oracle, solver, scheme):
    """divSchemes Gauss upwind / Gauss linearUpwind grad(U) (fy_case_desc.convection_scheme = 1 / 2): HIP path vs oracle, coupled steps
    at a cell Peclet number of ~30"""
    n = 16
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else cavity_bcs()
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.5, 0, 0)
    kw.setdefault("u_bc", [0] * 6); kw["u_val"] = u_val
    o, s = both(product, oracle, solver, n, n, n, dx, 2e-4, 1e-4, convection_scheme=scheme, **kw)
    case = gc.Case("cplu", n, n, n, 0.1, gaussian=solver, np_=2000, seed=19, cluster=100, fast=10, vel_scale=0.05)
    for step in range(4):
        rec = gc.particle_records(case, step)
        o.step(rec)
        s.set_particles(rec)
        s.step()
    compare(o, s, rtol=1e-5)
    # and it is a different discretisation from the default
    o2, s2 = both(product, oracle, solver, n, n, n, dx, 2e-4, 1e-4, **kw)
    for step in range(4):
        s2.set_particles(gc.particle_records(case, step))
        s2.step()
    assert np.abs(s2.get("U") - s.get("U")).max() > 1e-4 * np.abs(s.get("U")).max()
    o2.close(); s2.close()


def test_c2_channel_inlet_outlet_point_force(product, oracle):
    """BASELINE configs[1] in miniature: icoFoamYade point force in a channel -- inlet fixedValue U = (1,0,0), outlet zeroGradient U
    with p = 0, no-slip walls"""
    nx, ny, nz = 24, 12, 6
    dx = 2.0 / 200 * (200 / nx) / 1.0 * 0.1
    u_bc = [0, 1, 0, 0, 0, 0]
    u_val = [(1.0, 0, 0)] + [(0, 0, 0)] * 5
    p_bc = [0, 1, 0, 0, 0, 0]
    o, s = both(product, oracle, 0, nx, ny, nz, dx, 0.2 * dx, 1e-3, u_bc=u_bc, u_val=u_val, p_bc=p_bc, p_val=[0.0] * 6)
    case = gc.Case("c2", nx, ny, nz, nx * dx, gaussian=0, np_=2000, seed=2, radius_dx=0.15, vel_scale=0.2)
    for step in range(4):
        rec = gc.particle_records(case, step)
        fo = o.step(rec)["force"]
        s.set_particles(rec); s.step()
        sc = np.abs(fo).max()
        assert np.abs(s.forces() - fo).max() <= 1e-6 * sc
    compare(o, s, rtol=1e-5)
    U = s.get("U").reshape(nz, ny, nx, 3)
    assert U[nz // 2, ny // 2, :, 0].min() > 0.5            # the inflow has filled the channel


def test_c5_fluidized_bed_bcs(product, oracle):
    """BASELINE configs[4] in miniature: pimpleFoamYade 4-way, bottom inlet fixedValue U = (0,0,Uin), top outlet p = 0 with
    zeroGradient U, no-slip side walls, gravity, dense particle layer in the lower third"""
    n, nz = 10, 20
    dx = 0.05 / n
    u_bc = [0, 0, 0, 0, 0, 1]
    u_val = [(0, 0, 0)] * 4 + [(0, 0, 0.05)] + [(0, 0, 0)]
    p_bc = [2, 2, 2, 2, 2, 1]
    o, s = both(product, oracle, 1, n, n, nz, dx, 1e-4, 1e-5, g=(0, 0, -9.81), u_bc=u_bc, u_val=u_val, p_bc=p_bc, p_val=[0.0] * 6)
    rs = np.random.RandomState(5)
    npart = 4000
    rec = np.zeros((npart, 10))
    rec[:, 0:2] = rs.random_sample((npart, 2)) * n * dx
    rec[:, 2] = rs.random_sample(npart) * (nz * dx / 3.0)
    rec[:, 3:6] = (rs.random_sample((npart, 3)) - 0.5) * 0.02
    rec[:, 9] = 0.2 * dx
    for step in range(3):
        fo = o.step(rec)["force"]
        s.set_particles(rec); s.step()
        sc = np.abs(fo).max()
        assert np.abs(s.forces() - fo).max() <= 1e-6 * sc
    compare(o, s, rtol=1e-5)
    assert np.abs(fo).max() > 0 and np.abs(s.get("U")).max() > 0       # (alpha itself is reset by setSourceZero at the end of a step)


@pytest.mark.parametrize("solver,n_outer,n_corr,n_non_orth", [(1, 2, 1, 0), (1, 3, 2, 0), (1, 1, 3, 1), (0, 1, 1, 0), (0, 1, 3, 1)])
def test_corrector_counts_match_oracle(product, oracle, solver, n_outer, n_corr, n_non_orth):
    """pisoControl / pimpleControl loop counts other than the stock nOuter = 1, nCorr = 2 (icoFoamYade.C:97, pimpleFoamYade.C:91-105):
    the second outer iteration assembles its momentum matrix from the flux the first one corrected (the product exchanges phi and
    phiOld at the start of a step and reads the current flux from wherever it is), and one momentum assembly's correctors share rAUf
    and the coarse pressure operators."""
    n = 16
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else cavity_bcs()
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.3, 0, 0)
    kw.setdefault("u_bc", [0] * 6); kw["u_val"] = u_val
    o, s = both(product, oracle, solver, n, n, n, dx, 2e-4, 1e-4, n_outer=n_outer, n_corr=n_corr, n_non_orth=n_non_orth, **kw)
    case = gc.Case("cnt", n, n, n, 0.1, gaussian=solver, np_=1500, seed=23, cluster=100, fast=10, vel_scale=0.05)
    for step in range(4):
        rec = gc.particle_records(case, step)
        o.step(rec)
        s.set_particles(rec)
        s.step()
        so, ss = o.stats(), s.stats()
        assert so["p_solves"] == ss["p_solves"] == n_outer * n_corr * (n_non_orth + 1)
    compare(o, s, rtol=1e-5)


def test_adjustable_time_step_and_relaxation_match_oracle(product, oracle):
    """pimpleFoamYade with controlDict adjustTimeStep (readTimeControls.H / setDeltaT.H, pimpleFoamYade.C:62-64) and fvSolution
    relaxationFactors for UcEqn.relax() / p.relax() (UcEqn.H:12, pEqn.H:41), three outer correctors: same deltaT sequence, same fields"""
    n = 14
    kw = dict(n_outer=3, adjust_time_step=1, max_co=0.35, max_delta_t=0.1, u_relax=0.7, u_relax_final=1.0, p_relax=0.3, p_relax_final=1.0)
    o, s = both(product, oracle, 1, n, n, n, 1.0 / n, 2e-3, 0.01, **kw, **cavity_bcs())
    dts = []
    for step in range(25):
        o.step(); s.step()
        so, ss = o.stats(), s.stats()
        assert np.isclose(so["delta_t"], ss["delta_t"], rtol=1e-7), (step, so["delta_t"], ss["delta_t"])
        dts.append(ss["delta_t"])
    assert dts[0] == pytest.approx(2.4e-3) and max(dts) > 5 * dts[0]          # the step grows from rest until the Courant limit binds
    assert 0.2 < ss["courant_max"] * dts[-1] / dts[-2] < 0.45
    compare(o, s, rtol=1e-5)
    # relaxation really took part: the same run without it ends elsewhere
    o2, s2 = both(product, oracle, 1, n, n, n, 1.0 / n, 2e-3, 0.01, n_outer=3, adjust_time_step=1, max_co=0.35, max_delta_t=0.1, **cavity_bcs())
    for step in range(25):
        s2.step()
    assert np.abs(s2.get("p") - s.get("p")).max() > 1e-4 * np.abs(s.get("p")).max()
    for x in (o, s, o2, s2):
        x.close()


def test_adjustPhi_balances_a_free_outlet(product, oracle):
    """adjustPhi (icoFoamYade.C:108, pEqn.H:13-16): no patch fixes the pressure, U is prescribed at the inlet and free at the outlet -- the
    outflow of phiHbyA is scaled so that the pressure equation is solvable, and the corrected flux carries out exactly what comes in"""
    nx, ny, nz, dx = 16, 8, 6, 0.05
    u_bc = [0, 1, 0, 0, 0, 0]
    u_val = [(1.0, 0, 0)] + [(0, 0, 0)] * 5
    for solver in (0, 1):
        o, s = both(product, oracle, solver, nx, ny, nz, dx, 0.2 * dx, 0.01, u_bc=u_bc, u_val=u_val, p_bc=[0] * 6)
        # (from rest the outlet carries no flux that could be scaled and OpenFOAM stops with "Continuity error cannot be removed by
        # adjusting the outflow" -- so does fy_solver_step; start from a stream that leaves through the outlet, unevenly)
        U0 = np.zeros((nz, ny, nx, 3)); U0[..., 0] = 0.6 + 0.5 * np.linspace(0, 1, ny)[None, :, None]
        o.set("U", U0); s.set("U", U0)
        for step in range(5):
            o.step(); s.step()
        compare(o, s, rtol=1e-5)
        phix = s.get("phi_x").reshape(nz, ny, nx + 1)
        inflow, outflow = phix[:, :, 0].sum(), phix[:, :, nx].sum()
        assert inflow == pytest.approx(ny * nz * dx * dx) and outflow == pytest.approx(inflow, rel=1e-5)
        assert abs(s.stats()["cont_err_global"]) < 1e-9
        o.close(); s.close()


def test_cases_adjustPhi_cannot_balance_are_refused(product):
    """inflow through fixed-value patches that nothing can balance: OpenFOAM's adjustPhi ends such a run ("Continuity error cannot be removed
    by adjusting the outflow"); the library says so when the case is set up"""
    U = product.FY_BC_U_FIXED_VALUE
    PZ = product.FY_BC_P_ZERO_GRADIENT
    inflow = [(1, 0, 0)] + [(0, 0, 0)] * 5
    with pytest.raises(product.FoamYadeError) as e:
        product.Solver(product.make_case(product.FY_SOLVER_PIMPLE, 8, 8, 8, 0.1, 0.01, 0.01, u_bc=[U] * 6, u_val=inflow, p_bc=[PZ] * 6))
    assert "do not balance" in str(e.value)
    through = [(1, 0, 0), (1, 0, 0)] + [(0, 0, 0)] * 4        # what goes in comes out through fixed-value patches: adjustPhi is the identity
    s = product.Solver(product.make_case(product.FY_SOLVER_ICO, 8, 8, 8, 0.1, 0.01, 0.01, u_bc=[U] * 6, u_val=through, p_bc=[PZ] * 6))
    s.step()
    assert abs(s.stats()["cont_err_global"]) < 1e-10
    s.close()


@pytest.mark.parametrize("n_outer", [1, 2])
def test_smagorinsky_matches_oracle(product, oracle, n_outer):
    """pimpleFoamYade with LESModel Smagorinsky (DPMTurbulenceModels.C:73-74): nut from continuousPhaseTurbulence->correct() after the last
    corrector (pimpleFoamYade.C:101-104), nuEff = nu + nut in both parts of divDevRhoReff (UcEqn.H:7); coupled, so alpha nuEff varies too"""
    n = 16
    dx = 0.1 / n
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.5, 0, 0)
    nut_bc = [0, 0, 1, 0, 1, 1]
    nut_value = [0, 0, 0.0, 0, 2e-5, 0.0]
    o, s = both(product, oracle, 1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6,
                turbulence_model=1, les_ck=0.2, nut_bc=nut_bc, nut_value=nut_value, nut_initial=3e-5, n_outer=n_outer, u_relax=0.9 if n_outer > 1 else 1.0)
    np.testing.assert_array_equal(s.get("nut"), 3e-5)      # (n_outer = 2: correct() runs on the final outer iteration only, pimple.turbCorr())
    case = gc.Case("cpl", n, n, n, 0.1, gaussian=1, np_=2000, seed=5, cluster=100, fast=10, outside=10, vel_scale=0.05)
    for step in range(4):
        rec = gc.particle_records(case, step)
        o.step(rec)
        s.set_particles(rec)
        s.step()
        a, b = s.get("nut"), o.get("nut")
        assert b.max() > 1e-7
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9 * b.max())
    compare(o, s, rtol=1e-5)
    # and it is not the laminar answer
    ol, sl = both(product, oracle, 1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6)
    for step in range(4):
        rec = gc.particle_records(case, step)
        sl.set_particles(rec)
        sl.step()
    assert np.abs(sl.get("U") - s.get("U")).max() > 1e-4 * np.abs(sl.get("U")).max()


def test_turbulence_model_refused_where_the_reference_has_none(product):
    from importlib import import_module  # noqa: F401
    with pytest.raises(product.FoamYadeError):
        product.Solver(product.make_case(0, 8, 8, 8, 0.1, 0.01, 0.01, turbulence_model=1))           # icoFoamYade: laplacian(nu, U)
    with pytest.raises(product.FoamYadeError):
        product.Solver(product.make_case(1, 8, 8, 8, 0.1, 0.01, 0.01, turbulence_model=7))           # kEpsilon / kEqn: not built


@pytest.mark.parametrize("k_scheme", [0, 1])
def test_keqn_matches_oracle(product, oracle, k_scheme):
    """pimpleFoamYade with LESModel kEqn (DPMTurbulenceModels.C:76-77): a transport equation for the sub-grid kinetic energy, solved after
    the last corrector (pimpleFoamYade.C:101-104), nut = Ck sqrt(k) delta; coupled, so alpha weighs every term"""
    n = 16
    dx = 0.1 / n
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.5, 0, 0)
    kw = dict(turbulence_model=2, les_ck=0.2, nut_bc=[0, 3, 1, 3, 1, 1], nut_value=[0, 1e-5, 0.0, 4e-5, 2e-5, 0.0], nut_initial=3e-5,
              k_bc=[0, 1, 1, 1, 0, 1], k_value=[0, 1e-4, 0.0, 2e-4, 0, 0.0], k_initial=5e-4, k_convection_scheme=k_scheme, k_tol=1e-9, k_relax=0.9)
    o, s = both(product, oracle, 1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6, **kw)
    np.testing.assert_array_equal(s.get("k"), 5e-4)
    case = gc.Case("cpl", n, n, n, 0.1, gaussian=1, np_=2000, seed=5, cluster=100, fast=10, outside=10, vel_scale=0.05)
    for step in range(4):
        rec = gc.particle_records(case, step)
        o.step(rec)
        s.set_particles(rec)
        s.step()
        for nm in ("k", "nut"):
            a, b = s.get(nm), o.get(nm)
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9 * b.max(), err_msg=nm)
    assert not np.allclose(o.get("k"), 5e-4)
    compare(o, s, rtol=1e-5)


def test_kepsilon_matches_oracle(product, oracle):
    """pimpleFoamYade with RASModel kEpsilon (DPMTurbulenceModels.C:70-71), boundary types zeroGradient / fixedValue (no wall functions):
    epsilon equation, then k with the new epsilon, nut = Cmu k^2/eps; coupled"""
    n = 16
    dx = 0.1 / n
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.5, 0, 0)
    kw = dict(turbulence_model=3, nut_bc=[0, 0, 1, 0, 0, 1], nut_value=[0, 0, 0.0, 0, 0, 1e-5], nut_initial=2e-5,
              k_bc=[0, 1, 1, 1, 0, 1], k_value=[0, 1e-4, 2e-4, 2e-4, 0, 1e-4], k_initial=5e-4, k_convection_scheme=1, k_tol=1e-9, k_relax=0.9,
              eps_bc=[0, 1, 0, 1, 0, 0], eps_value=[0, 2e-3, 0, 3e-3, 0, 0], eps_initial=2e-3, eps_convection_scheme=0, eps_tol=1e-9, eps_relax=0.8,
              ras_c3=-0.33)
    o, s = both(product, oracle, 1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6, **kw)
    np.testing.assert_array_equal(s.get("epsilon"), 2e-3)
    case = gc.Case("cpl", n, n, n, 0.1, gaussian=1, np_=2000, seed=5, cluster=100, fast=10, outside=10, vel_scale=0.05)
    for step in range(4):
        rec = gc.particle_records(case, step)
        o.step(rec)
        s.set_particles(rec)
        s.step()
        for nm in ("epsilon", "k", "nut"):
            a, b = s.get(nm), o.get(nm)
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9 * b.max(), err_msg=nm)
    assert not np.allclose(o.get("k"), 5e-4) and not np.allclose(o.get("epsilon"), 2e-3)
    compare(o, s, rtol=1e-5)


def test_kepsilon_with_wall_functions_matches_oracle(product, oracle):
    """kEpsilon with nutkWallFunction / epsilonWallFunction (k: zeroGradient = kqRWallFunction) on the lid and two walls: imposed wall-cell
    epsilon, wall production, nut_w(y+) in the momentum and k equations -- from the file's value at the first step, from k afterwards"""
    n = 16
    dx = 0.1 / n
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (2.0, 0, 0)
    kw = dict(turbulence_model=3, nut_bc=[2, 3, 2, 2, 1, 3], nut_value=[0, 3e-5, 0, 0, 1e-5, 1e-5], nut_initial=2e-5,      # (x+ and z+: `calculated` nut)
              k_bc=[0, 0, 0, 0, 1, 0], k_value=[0, 0, 0, 0, 2e-3, 0], k_initial=4e-3, k_convection_scheme=1, k_tol=1e-9,
              eps_bc=[2, 0, 2, 2, 1, 0], eps_value=[0, 0, 0, 0, 0.05, 0], eps_initial=0.02, eps_convection_scheme=1, eps_tol=1e-9)
    o, s = both(product, oracle, 1, n, n, n, dx, 2e-4, 1e-6, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6, **kw)
    case = gc.Case("cpl", n, n, n, 0.1, gaussian=1, np_=1500, seed=6, cluster=80, fast=10, outside=10, vel_scale=0.05)
    for step in range(4):
        rec = gc.particle_records(case, step)
        o.step(rec)
        s.set_particles(rec)
        s.step()
        for nm in ("epsilon", "k", "nut"):
            a, b = s.get(nm), o.get(nm)
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9 * b.max(), err_msg=nm)
    compare(o, s, rtol=1e-5)
    eps = s.get("epsilon").reshape(n, n, n)
    assert eps[:, 0, :].min() > 1.3 * eps[n // 2, n // 2, n // 2]          # the wall cells sit on the log-law value (exact check: test_fv_oracle.py), above the core's
    # y+ of the lid-side cells is beyond the laminar sub-layer, so nut_w is live there: the run differs from the same case without wall functions
    kw2 = dict(kw, nut_bc=[0, 3, 0, 0, 1, 3], eps_bc=[0, 0, 0, 0, 1, 0])
    o2, s2 = both(product, oracle, 1, n, n, n, dx, 2e-4, 1e-6, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6, **kw2)
    for step in range(4):
        s2.set_particles(gc.particle_records(case, step))
        s2.step()
    assert np.abs(s2.get("k") - s.get("k")).max() > 1e-3 * np.abs(s.get("k")).max()


# ---- graded (rectilinear) single block: the kernels of namespace fy::gr (fv_kernels.hip compiled with the general geometry) vs the oracle ----
def geometric_sizes(n, ratio, length):
    r = ratio ** (1.0 / (n - 1)) if n > 1 else 1.0
    h = r ** np.arange(n)
    return h * (length / h.sum())


def wall_refined_sizes(n, ratio, length):
    half = geometric_sizes(n // 2, ratio, 0.5 * length)
    return np.concatenate([half, half[::-1]])


@pytest.mark.parametrize("solver,p_solver,scheme", [(0, 1, 0), (0, 0, 1), (1, 1, 0)])
def test_graded_cavity_steps_match_oracle(product, oracle, solver, p_solver, scheme):
    """lid-driven cavity on a block refined towards all walls (size ratio 4 in x and y, 2.5 in z): matrices, fluxes and fields of the graded
    kernels against the oracle's general operators, Gauss linear and Gauss upwind"""
    n = 14
    g = (wall_refined_sizes(n, 4.0, 1.0), wall_refined_sizes(n, 4.0, 1.0), wall_refined_sizes(n, 2.5, 0.8))
    o, s = both(product, oracle, solver, n, n, n, 1.0 / n, 0.01, 0.01, p_solver=p_solver, convection_scheme=scheme, grading=g, **cavity_bcs())
    for step in range(5):
        o.step(); s.step()
        so, ss = o.stats(), s.stats()
        assert abs(so["p_iters_total"] - ss["p_iters_total"]) <= 2
        assert np.isclose(so["courant_max"], ss["courant_max"], rtol=1e-6) and np.isclose(so["courant_mean"], ss["courant_mean"], rtol=1e-6)
        if step == 0:
            for nm in ("p_diag", "p_ux", "p_uy", "p_uz", "mom_diag", "rAU"):
                np.testing.assert_allclose(s.get(nm), o.get(nm), rtol=1e-10, err_msg=nm)
    compare(o, s)
    assert ss["cont_err_sum_local"] < 1e-5 and np.abs(s.get("U")).max() > 0.05


def test_graded_coupled_pimple_steps_match_oracle(product, oracle):
    """pimpleFoamYade 4-way on a graded box under gravity: the particle half runs on the explicit k-d tree of the graded block's centres with
    every cell's own volume, the FV half on the graded operators; forces and fields against the oracle over three coupled steps"""
    n = 12
    g = (geometric_sizes(n, 2.0, 0.1), wall_refined_sizes(n, 3.0, 0.1), geometric_sizes(n, 0.4, 0.12))
    o, s = both(product, oracle, 1, n, n, n, 0.1 / n, 2e-4, 1e-5, g=(0, 0, -9.81), p_bc=[2] * 6, grading=g)
    rs = np.random.RandomState(17)
    npart = 2500
    for step in range(3):
        rec = np.zeros((npart, 10))
        rec[:, 0:3] = rs.random_sample((npart, 3)) * np.array([0.1, 0.1, 0.07]) + np.array([0.0, 0.0, 0.005])
        rec[:, 3:6] = 0.05 * rs.standard_normal((npart, 3))
        rec[:, 9] = 0.2 * (0.1 / n)
        fo = o.step(rec)["force"]
        s.set_particles(rec); s.step()
        sc = np.abs(fo).max()
        assert sc > 0 and np.abs(s.forces() - fo).max() <= 1e-6 * sc, np.abs(s.forces() - fo).max() / sc
    compare(o, s, rtol=1e-5)


def test_graded_point_force_finds_cells_by_axis_search(product, oracle):
    """icoFoamYade on a graded channel with particles: findCell on the graded block is a search along each axis (particles on face planes and on
    the outer faces included); Stokes drag and the source field against closed forms on the host"""
    nx, ny, nz = 16, 10, 6
    g = (geometric_sizes(nx, 3.0, 0.4), wall_refined_sizes(ny, 4.0, 0.1), geometric_sizes(nz, 1.0, 0.06))
    U_, ZG = product.FY_BC_U_FIXED_VALUE, product.FY_BC_U_ZERO_GRADIENT
    PZ, PF = product.FY_BC_P_ZERO_GRADIENT, product.FY_BC_P_FIXED_VALUE
    case = product.make_case(0, nx, ny, nz, 0.01, 1e-3, 1e-3, u_bc=[U_, ZG, U_, U_, U_, U_], u_val=[(0.5, 0, 0)] + [(0, 0, 0)] * 5, p_bc=[PZ, PF, PZ, PZ, PZ, PZ], grading=g)
    s = product.Solver(case)
    s.hold_sources(True)
    faces = [np.concatenate([[0.0], np.cumsum(h)]) for h in g]
    rs = np.random.RandomState(3)
    npart = 4000
    rec = np.zeros((npart, 10))
    rec[:, 0:3] = rs.random_sample((npart, 3)) * np.array([0.4, 0.1, 0.06])
    rec[:50, 0] = faces[0][rs.randint(0, nx + 1, 50)]                # exactly on x face planes (the outer ones too)
    rec[50:100, 1] = faces[1][rs.randint(0, ny + 1, 50)]
    rec[100:130, 2] = 0.0601                                         # outside: not found
    rec[:, 3:6] = 0.1 * rs.standard_normal((npart, 3))
    rec[:, 9] = 5e-4
    for _ in range(2):
        Ub = s.get("U").reshape(-1, 3)
        s.set_particles(rec); s.step()
    F, found = s.forces(), s.found()
    hi = np.array([f[-1] for f in faces])                             # (the block ends where the summed sizes end, not at the nominal lengths)
    inside = np.all((rec[:, 0:3] >= 0) & (rec[:, 0:3] <= hi), axis=1)
    assert np.array_equal(found == 1, inside) and (~inside).sum() == 30
    idx = [np.minimum(np.searchsorted(f, rec[:, a], side="right") - 1, n - 1) for a, (f, n) in enumerate(zip(faces, (nx, ny, nz)))]
    cell = idx[0] + nx * (idx[1] + ny * idx[2])
    Fref = (3 * np.pi * 2 * rec[:, 9] * 1e-3 * 1000.0)[:, None] * (Ub[np.where(inside, cell, 0)] - rec[:, 3:6])
    Fref[~inside] = 0.0
    np.testing.assert_allclose(F[:, :3], Fref, rtol=1e-12, atol=1e-14 * np.abs(Fref).max())
    V = ((g[2][:, None, None] * g[1][None, :, None]) * g[0][None, None, :]).reshape(-1)
    uS = s.get("uSource").reshape(-1, 3)
    for a in range(3):
        ref = -np.bincount(cell[inside], weights=Fref[inside, a], minlength=V.size) / (V * 1000.0)
        np.testing.assert_allclose(uS[:, a], ref, rtol=1e-10, atol=1e-12 * np.abs(ref).max())
    s.close()


def test_graded_block_refusals(product):
    h = np.full(8, 0.125)
    with pytest.raises(product.FoamYadeError, match="z-slabs"):
        product.VirtualSlabs(product.make_case(0, 8, 8, 8, 0.125, 1e-3, 1e-3, grading=(h, h, h)), 2)


@pytest.mark.parametrize("what", ["smagorinsky", "kEqn", "kEpsilon_wall_functions", "linearUpwind"])
def test_graded_turbulence_closures_and_linear_upwind_match_oracle(product, oracle, what):
    """the closures of DPMTurbulenceModels.C:67-77 and Gauss linearUpwind on a block graded towards the walls, coupled with particles: per-cell
    LES delta, per-wall distances in the wall functions, linear weights in every face interpolate, |Sf|-weighted average in bound()"""
    from test_fv_oracle import TURB_CASES, TURB_FIELDS
    n = 12
    L = 0.1
    g = (wall_refined_sizes(n, 3.0, L), wall_refined_sizes(n, 4.0, L), geometric_sizes(n, 2.0, L))
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (2.0, 0, 0)
    o, s = both(product, oracle, 1, n, n, n, L / n, 2e-4, 1e-6, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6, grading=g, **TURB_CASES[what])
    rs = np.random.RandomState(23)
    npart = 1500
    for step in range(4):
        rec = np.zeros((npart, 10))
        rec[:, 0:3] = L * (0.05 + 0.9 * rs.random_sample((npart, 3)))
        rec[:, 3:6] = 0.05 * rs.standard_normal((npart, 3))
        rec[:, 9] = 0.1 * (L / n)
        fo = o.step(rec)["force"]
        s.set_particles(rec); s.step()
        sc = np.abs(fo).max()
        assert sc > 0 and np.abs(s.forces() - fo).max() <= 1e-6 * sc
        for nm in TURB_FIELDS[what]:
            a, b = s.get(nm), o.get(nm)
            np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9 * b.max(), err_msg=nm)
    compare(o, s, rtol=1e-5)
    assert np.abs(s.get("U")).max() > 1e-6
    o.close(); s.close()


# ---- symmetryPlane / slip sides ------------------------------------------------------------------------------------------------------
def vortex_field(nx, ny, nz, dx):
    """a swirling, divergence-free start that slides along every side of the box"""
    cs = [(np.arange(m) + 0.5) * dx for m in (nx, ny, nz)]
    L = [m * dx for m in (nx, ny, nz)]
    Z, Y, X = np.meshgrid(cs[2], cs[1], cs[0], indexing="ij")
    U = np.zeros((nz, ny, nx, 3))
    kx, ky, kz = np.pi / L[0], np.pi / L[1], np.pi / L[2]
    U[..., 0] = np.sin(kx * X) * np.cos(ky * Y) * np.cos(kz * Z) / kx
    U[..., 1] = -0.4 * np.cos(kx * X) * np.sin(ky * Y) * np.cos(kz * Z) / ky
    U[..., 2] = -0.6 * np.cos(kx * X) * np.cos(ky * Y) * np.sin(kz * Z) / kz
    return U.reshape(-1, 3)


@pytest.mark.parametrize("solver,kw", [(0, {}), (0, dict(convection_scheme=1)), (1, dict(u_relax=0.7, n_outer=2)), (1, dict(u_relax=0.0))])
def test_slip_sides_match_oracle(product, oracle, solver, kw):
    """four symmetry planes, a moving lid and a wall: the per-component boundary diagonal in the momentum solve, its component average in
    A() and the remainder in H(), with relax() on top in pimple"""
    nx, ny, nz = 14, 12, 10
    dx = 1.0 / 14
    u_bc = [2, 2, 0, 0, 2, 2]
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.5, 0, 0.2)
    o, s = both(product, oracle, solver, nx, ny, nz, dx, 0.02, 0.01, u_bc=u_bc, u_val=u_val, **kw)
    U0 = 0.3 * vortex_field(nx, ny, nz, dx)
    o.set("U", U0); s.set("U", U0)
    for step in range(5):
        o.step(); s.step()
        so, ss = o.stats(), s.stats()
        assert abs(so["p_iters_total"] - ss["p_iters_total"]) <= 2 and abs(so["u_iters_total"] - ss["u_iters_total"]) <= 1
        if step == 0:
            for nm in ("mom_diag", "rAU", "HbyA"):
                np.testing.assert_allclose(s.get(nm), o.get(nm), rtol=1e-9, atol=1e-12, err_msg=nm)
    compare(o, s)
    for side, nm in ((XMIN, "phi_x"), (ZMIN, "phi_z")):
        assert np.abs(s.get(nm)).max() > 1e-4
    px = s.get("phi_x").reshape(nz, ny, nx + 1)
    pz = s.get("phi_z").reshape(nz + 1, ny, nx)
    assert np.abs(px[:, :, 0]).max() == 0 and np.abs(px[:, :, -1]).max() == 0 and np.abs(pz[0]).max() == 0 and np.abs(pz[-1]).max() == 0
    o.close(); s.close()


def test_slip_sides_on_a_graded_block_match_oracle(product, oracle):
    n = 12
    g = (wall_refined_sizes(n, 3.0, 1.0), geometric_sizes(n, 2.0, 1.0), wall_refined_sizes(n, 2.0, 0.8))
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.5, 0, 0)
    o, s = both(product, oracle, 0, n, n, n, 1.0 / n, 0.01, 0.01, u_bc=[2, 2, 0, 0, 2, 2], u_val=u_val, grading=g)
    for step in range(5):
        o.step(); s.step()
    compare(o, s)
    assert np.abs(s.get("U")).max() > 0.02
    o.close(); s.close()


def test_slip_sides_in_virtual_slabs_match_single_domain(product):
    n, n_slabs = 12, 2
    nz = n * n_slabs
    dx = 1.0 / n
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (0.5, 0, 0.1)
    case = product.make_case(0, n, n, nz, dx, 0.02, 0.01, u_bc=[2, 2, 0, 0, 2, 2], u_val=u_val)
    one = product.Solver(case); many = product.VirtualSlabs(case, n_slabs)
    U0 = 0.3 * vortex_field(n, n, nz, dx)
    one.set("U", U0); many.set("U", U0)
    for step in range(4):
        one.step(); many.step()
    compare(one, many, ("U", "p"), 1e-5)
    one.close(); many.close()


def test_unknown_boundary_types_are_refused(product):
    case = product.make_case(0, 8, 8, 8, 0.125, 0.01, 0.01, u_bc=[0, 0, 0, 3, 0, 0])
    with pytest.raises(product.FoamYadeError):
        product.Solver(case)


@pytest.mark.parametrize("scheme,kw", [(3, dict(limiter_k=1.0)), (3, dict(limiter_k=0.3)), (4, {}), (5, {}), (6, {}), (7, {}), (8, {})])
def test_limited_convection_schemes_match_oracle(product, oracle, scheme, kw):
    """Gauss limitedLinear k | vanLeer | MUSCL | Minmod | SuperBee | QUICK for div(phi,U): the gradient ratio from grad(magSqr(U)), one limiter
    per face, implicit weights -- on a cavity at cell Peclet number 25, where the limiter is active over much of the box"""
    n = 16
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (1.0, 0, 0.3)
    o, s = both(product, oracle, 0, n, n, n, 1.0 / n, 0.4 / n, 2.5e-3, u_bc=[0] * 6, u_val=u_val, convection_scheme=scheme, **kw)
    for step in range(8):
        o.step(); s.step()
        if step == 0:
            for nm in ("mom_diag", "rAU"):
                np.testing.assert_allclose(s.get(nm), o.get(nm), rtol=1e-10, err_msg=nm)
    compare(o, s)
    # ... and it is neither the central nor the upwind answer
    for other in (0, 1):
        o2, s2 = both(product, oracle, 0, n, n, n, 1.0 / n, 0.4 / n, 2.5e-3, u_bc=[0] * 6, u_val=u_val, convection_scheme=other)
        for step in range(8):
            s2.step()
        assert np.abs(s2.get("U") - s.get("U")).max() > 1e-3
        o2.close(); s2.close()
    o.close(); s.close()


@pytest.mark.parametrize("variant", ["pimple_coupled", "graded", "slabs"])
def test_limited_schemes_in_pimple_on_graded_blocks_and_in_slabs(product, oracle, variant):
    n = 12
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (1.0, 0, 0.3)
    if variant == "pimple_coupled":
        L = 0.1
        o, s = both(product, oracle, 1, n, n, n, L / n, 2e-4, 1e-5, g=(0, 0, -9.81), u_bc=[0] * 6, u_val=u_val, p_bc=[2] * 6, convection_scheme=4)
        case = gc.Case("cpl", n, n, n, L, gaussian=1, np_=1500, seed=5, cluster=100, fast=10, outside=10, vel_scale=0.05)
        for step in range(4):
            rec = gc.particle_records(case, step)
            o.step(rec); s.set_particles(rec); s.step()
        compare(o, s, rtol=1e-5)
        o.close(); s.close()
    elif variant == "graded":
        g = (wall_refined_sizes(n, 3.0, 1.0), wall_refined_sizes(n, 4.0, 1.0), geometric_sizes(n, 2.0, 1.0))
        o, s = both(product, oracle, 0, n, n, n, 1.0 / n, 0.01, 2.5e-3, u_bc=[0] * 6, u_val=u_val, convection_scheme=3, limiter_k=1.0, grading=g)
        for step in range(6):
            o.step(); s.step()
        compare(o, s)
        o.close(); s.close()
    else:
        case = product.make_case(0, n, n, 2 * n, 1.0 / n, 0.4 / n, 2.5e-3, u_bc=[0] * 6, u_val=u_val, convection_scheme=8)
        one = product.Solver(case); many = product.VirtualSlabs(case, 2)
        for step in range(6):
            one.step(); many.step()
        compare(one, many, ("U", "p"), 1e-5)
        assert np.abs(one.get("U")).max() > 0.05
        one.close(); many.close()


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("switch", ["FOAMYADE_NO_FUSED_CORRECTOR", "FOAMYADE_FACES_FROM_ARRAYS", "FOAMYADE_STRIP_BLOCKS"])
def test_fused_corrector_sweeps_equal_the_separate_ones(product, solver, switch, monkeypatch):
    """round 5: the corrector's fused sweeps (k_corr_front / k_corr_back / k_bmom_faces, HbyA from the predictor's last pass, the ddtCorr coefficient from
    the step's opening sweep) evaluate the SAME per-face expressions as the separate sweeps of rounds 1 - 4, which FOAMYADE_NO_FUSED_CORRECTOR=1 brings
    back; FOAMYADE_FACES_FROM_ARRAYS=1 streams rAUf / alphacf from their face arrays instead of re-forming them; FOAMYADE_STRIP_BLOCKS=0 switches the
    strip order of the blocks off.  Same operands, same order: the fields agree to rounding of the reductions that steer the solvers' stopping."""
    n = 32                                       # a plane of 1024 cells = 4 blocks, nz % 8 == 0: the strip order is active
    dx = 0.1 / n
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else cavity_bcs()
    nu = 1e-5 if solver == 1 else 0.01
    case = gc.Case("cpl", n, n, n, 0.1, gaussian=solver, np_=20000, seed=9, cluster=150, fast=20, outside=20, vel_scale=0.05)
    out = []
    for on in (False, True):
        if on:
            monkeypatch.setenv(switch, "0" if switch == "FOAMYADE_STRIP_BLOCKS" else "1")
        pc = product.make_case(solver, n, n, n, dx, 2e-4, nu, g=kw.get("g", (0, 0, 0)), u_bc=kw.get("u_bc"), u_val=kw.get("u_val"), p_bc=kw.get("p_bc"), p_solver=1)
        s = product.Solver(pc)
        for step in range(3):
            s.set_particles(gc.particle_records(case, step))
            s.step()
        out.append({nm: s.get(nm) for nm in ("U", "p", "phi_x", "phi_y", "phi_z")})
        out[-1]["force"] = s.forces()
        out[-1]["stats"] = s.stats()
        s.close()
    a, b = out
    assert a["stats"]["p_iters_total"] == b["stats"]["p_iters_total"] and a["stats"]["u_iters_total"] == b["stats"]["u_iters_total"]
    for nm in ("U", "p", "phi_x", "phi_y", "phi_z", "force"):
        sc = np.abs(a[nm]).max() + 1e-300
        assert np.abs(a[nm] - b[nm]).max() <= 1e-11 * sc, (nm, np.abs(a[nm] - b[nm]).max() / sc)


@pytest.mark.parametrize("solver", [0, 1])
@pytest.mark.parametrize("n_slabs", [1, 3])
def test_two_cells_per_thread_sweeps_equal_the_one_cell_ones(product, solver, n_slabs, monkeypatch):
    """round 5: the pressure solver's scalar-field sweeps (Laplacian apply with its dot products, the multigrid smoothers, residual + restriction, the PCG
    vector update) run with two consecutive cells per thread and 16-byte loads wherever the level's rows are even; FOAMYADE_NO_PAIRS=1 brings the
    one-cell kernels back.  The rows are evaluated with the same operations in the same order and the block partials are folded in the same order, so a
    fluid-only run (nothing summed in an order that changes from run to run) gives the same BITS and the same iteration counts -- on one domain and on
    three virtual slabs (the communication-avoiding V-cycle's plane windows)."""
    n, nz = 16, 36
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else cavity_bcs()
    nu = 1e-5 if solver == 1 else 0.01
    out = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("FOAMYADE_NO_PAIRS", "1")
        pc = product.make_case(solver, n, n, nz, 0.1 / n, 2e-4, nu, g=kw.get("g", (0, 0, 0)), u_bc=kw.get("u_bc"), u_val=kw.get("u_val"), p_bc=kw.get("p_bc"), p_solver=1)
        s = product.Solver(pc) if n_slabs == 1 else product.VirtualSlabs(pc, n_slabs)
        s.set("U", np.random.RandomState(5).rand(n * n * nz, 3) * 0.05)
        its = []
        for step in range(4):
            s.step()
            st = s.stats() if n_slabs == 1 else s.stats()[0]
            its.append(st["p_iters_total"])
        out.append(({nm: s.get(nm) for nm in ("U", "p", "phi_x", "phi_y", "phi_z")}, its))
        s.close()
    monkeypatch.delenv("FOAMYADE_NO_PAIRS")
    product.Solver(product.make_case(0, 8, 8, 8, 0.1, 1e-3, 0.01, p_solver=1, **cavity_bcs())).close()      # (re-reads the switch for the tests that follow)
    (a, its_a), (b, its_b) = out
    assert its_a == its_b and sum(its_a) > 0, (its_a, its_b)
    for nm in a:
        np.testing.assert_array_equal(a[nm], b[nm], err_msg=nm)


@pytest.mark.parametrize("solver", [0, 1])
def test_tail_level_in_lds_equals_the_global_memory_tail(product, solver, monkeypatch):
    """round 5: the V-cycle tail keeps its first level (operator, right-hand side, both iterates) in LDS for the whole kernel; FOAMYADE_NO_TAIL_CACHE=1 makes
    every sweep read it from global memory again.  Same p_row on the same values: same bits, same iteration counts (fluid only)."""
    n = 40                                       # levels 40, 20, 10, 5: the tail is 10^3 + 5^3
    kw = dict(g=(0, 0, -9.81), p_bc=[2] * 6) if solver == 1 else cavity_bcs()
    nu = 1e-5 if solver == 1 else 0.01
    out = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("FOAMYADE_NO_TAIL_CACHE", "1")
        pc = product.make_case(solver, n, n, n, 0.1 / n, 2e-4, nu, g=kw.get("g", (0, 0, 0)), u_bc=kw.get("u_bc"), u_val=kw.get("u_val"), p_bc=kw.get("p_bc"), p_solver=1)
        s = product.Solver(pc)
        s.set("U", np.random.RandomState(7).rand(n * n * n, 3) * 0.05)
        its = []
        for step in range(3):
            s.step()
            its.append(s.stats()["p_iters_total"])
        out.append(({nm: s.get(nm) for nm in ("U", "p", "phi_x", "phi_y", "phi_z")}, its))
        s.close()
    (a, its_a), (b, its_b) = out
    assert its_a == its_b and sum(its_a) > 0, (its_a, its_b)
    for nm in a:
        np.testing.assert_array_equal(a[nm], b[nm], err_msg=nm)
