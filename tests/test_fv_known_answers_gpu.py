"""Known-answer flows on the PRODUCT (HIP path through the C-ABI), at resolutions the CPU oracle cannot afford in a test: the FV half has
no reference build to be pinned against (OpenFOAM-6 is absent), so besides 'HIP == oracle' (tests/test_fv_parity.py) the product itself
is held to physics -- Ghia, Ghia & Shin (1982) for the lid-driven cavity at Re = 100, the analytic Poiseuille profile, and second-order
convergence of the discretisation towards it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

XMIN, XMAX, YMIN, YMAX, ZMIN, ZMAX = range(6)
# Ghia, Ghia & Shin (1982), Table I, Re = 100: u along the vertical centre line x = 0.5
GHIA_Y = np.array([0.0547, 0.0625, 0.0703, 0.1016, 0.1719, 0.2813, 0.4531, 0.5000, 0.6172, 0.7344, 0.8516, 0.9531, 0.9609, 0.9688, 0.9766])
GHIA_U = np.array([-0.03717, -0.04192, -0.04775, -0.06434, -0.10150, -0.15662, -0.21090, -0.20581, -0.13641, 0.00332, 0.23151,
                   0.68717, 0.73722, 0.78871, 0.84123])


@pytest.mark.parametrize("solver", [0, 1])
def test_cavity_re100_matches_ghia_at_128(product, solver):
    n = 128
    dx = 1.0 / n
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (1.0, 0, 0)
    case = product.make_case(solver, n, n, 2, dx, 0.4 * dx, 0.01, u_bc=[0, 0, 0, 0, 1, 1], u_val=u_val, p_solver=1)   # z sides zero-gradient: 2-D flow
    s = product.Solver(case)
    prev = None
    for it in range(12000):
        s.step()
        if it % 250 == 249:
            U = s.get("U").reshape(2, n, n, 3)[0]
            if prev is not None and np.abs(U - prev).max() < 2e-6:
                break
            prev = U
    U = s.get("U").reshape(2, n, n, 3)
    assert np.abs(U[0] - U[1]).max() < 1e-6 and np.abs(U[..., 2]).max() < 1e-6          # it stayed two-dimensional (to solver tolerance)
    yc = (np.arange(n) + 0.5) / n
    uc = 0.5 * (U[0, :, n // 2 - 1, 0] + U[0, :, n // 2, 0])                             # x = 0.5 lies on a face
    err = np.abs(np.interp(GHIA_Y, yc, uc) - GHIA_U)
    assert err.max() < 0.006, err                                                       # 32^2 in the oracle test: 0.02
    st = s.stats()
    assert st["cont_err_sum_local"] < 1e-6 and st["courant_max"] < 1.0
    s.close()


def poiseuille_error(product, solver, ny):
    """pressure-driven plane channel: u(y) = G / (2 nu) y (H - y), H = 1, G = dp / L (kinematic)"""
    nx = 8
    dx = 1.0 / ny
    nu, G = 0.05, 0.4
    L = nx * dx
    u_bc = [1, 1, 0, 0, 1, 1]
    p_bc = [1, 1, 0, 0, 0, 0]
    case = product.make_case(solver, nx, ny, 2, dx, 0.2 * dx / 1.0, nu, u_bc=u_bc, p_bc=p_bc, p_val=[G * L, 0.0, 0, 0, 0, 0], p_solver=1)
    s = product.Solver(case)
    prev = None
    for it in range(60000):
        s.step()
        if it % 500 == 499:
            u = s.get("U").reshape(2, ny, nx, 3)[0, :, nx // 2, 0]
            if prev is not None and np.abs(u - prev).max() < 1e-9:
                break
            prev = u
    y = (np.arange(ny) + 0.5) * dx
    exact = G / (2 * nu) * y * (1.0 - y)
    u = s.get("U").reshape(2, ny, nx, 3)[0, :, nx // 2, 0]
    s.close()
    return np.abs(u - exact).max() / exact.max()


@pytest.mark.parametrize("solver", [0, 1])
def test_poiseuille_profile_and_second_order_convergence(product, solver):
    e1, e2 = poiseuille_error(product, solver, 16), poiseuille_error(product, solver, 32)
    assert e2 < 2e-3
    assert 3.0 < e1 / e2 < 5.0, (e1, e2)            # halving dx divides the error by ~4


# ---- closed-form checks shared with the CPU oracle's own known-answer tests (tests/test_fv_oracle.py holds the functions) ------------
def test_manufactured_poisson_solution_converges_at_second_order_hip(product):
    """PCG + multigrid on the device against -lap(p) = f with p* = cos cos cos and Neumann walls, up to 128^3: error falls 4x per halving of
    dx, the iteration count barely moves"""
    from test_fv_oracle import manufactured_poisson_errors
    mk = lambda n: product.Solver(product.make_case(product.FY_SOLVER_ICO, n, n, n, 1.0 / n, 0.01, 1e-14, p_solver=1, p_tol=1e-10, p_final_tol=1e-10, p_rel_tol=0.0))
    res = manufactured_poisson_errors(mk, (32, 64, 128))
    errs = [e for _, e, _ in res]
    its = [it for _, _, it in res]
    assert 3.8 < errs[0] / errs[1] < 4.2 and 3.8 < errs[1] / errs[2] < 4.2, errs
    assert its[2] <= 30 and its[2] - its[0] <= 8, res
    jac = manufactured_poisson_errors(lambda n: product.Solver(product.make_case(product.FY_SOLVER_ICO, n, n, n, 1.0 / n, 0.01, 1e-14, p_solver=0, p_tol=1e-10,
                                                                                 p_final_tol=1e-10, p_rel_tol=0.0)), (64,))
    assert jac[0][1] == pytest.approx(errs[1], rel=1e-3) and jac[0][2] > 4 * its[1]       # same answer, many more iterations without the multigrid


def test_decaying_shear_mode_has_the_discrete_and_the_continuum_rate_hip(product):
    from test_fv_oracle import decaying_shear_mode
    ZG, FV = product.FY_BC_U_ZERO_GRADIENT, product.FY_BC_U_FIXED_VALUE
    PZ, PF = product.FY_BC_P_ZERO_GRADIENT, product.FY_BC_P_FIXED_VALUE
    mk = lambda ny, dt, nu: product.Solver(product.make_case(product.FY_SOLVER_ICO, 4, ny, 1, 1.0 / ny, dt, nu, u_bc=[ZG, ZG, FV, FV, ZG, ZG],
                                                             p_bc=[PZ, PF, PZ, PZ, PZ, PZ], u_tol=1e-12, p_tol=1e-12, p_final_tol=1e-12, p_rel_tol=0.0))
    amp, disc, cont = decaying_shear_mode(mk, 64, 0.01, 0.05, 40)
    assert amp == pytest.approx(disc, rel=1e-7)
    e = []
    for ny, dt in ((16, 0.01), (32, 0.0025), (64, 0.000625)):
        amp, disc, cont = decaying_shear_mode(mk, ny, dt, 0.05, int(round(0.2 / dt)))
        e.append(abs(amp - cont))
    assert 3.3 < e[0] / e[1] < 4.7 and 3.3 < e[1] / e[2] < 4.7, e


def test_corrected_flux_is_divergence_free_to_solver_tolerance_hip(product):
    from test_fv_oracle import flux_identity
    n = 48
    u_val = [(0, 0, 0)] * 6
    u_val[YMAX] = (1.0, 0, 0)
    s = product.Solver(product.make_case(product.FY_SOLVER_ICO, n, n, n, 1.0 / n, 0.4 / n, 0.01, u_val=u_val))
    for _ in range(5):
        s.step()
    flux_identity(s, n, 0.4 / n)
    s.close()


def test_graded_poiseuille_on_a_wall_refined_mesh_hip(product):
    """the graded-block kernels held to physics on their own (no oracle in the loop): pressure-driven plane channel on meshes refined towards
    both walls (last / first size ratio 5) -- second-order approach to the parabola, and the discrete wall shear carries the driving force
    exactly (conservation)"""
    def sizes(n, ratio, length):
        r = ratio ** (1.0 / (n // 2 - 1))
        h = r ** np.arange(n // 2)
        h *= 0.5 * length / h.sum()
        return np.concatenate([h, h[::-1]])
    nu, G = 0.05, 0.4
    U_, ZG = product.FY_BC_U_FIXED_VALUE, product.FY_BC_U_ZERO_GRADIENT
    PZ, PF = product.FY_BC_P_ZERO_GRADIENT, product.FY_BC_P_FIXED_VALUE
    errs = {}
    for ny in (16, 32):
        nx, nz = 4, 2
        hy = sizes(ny, 5.0, 1.0)
        hx, hz = np.full(nx, 0.1), np.full(nz, 0.1)
        L = hx.sum()
        c = product.make_case(0, nx, ny, nz, 0.1, 0.02, nu, u_bc=[ZG, ZG, U_, U_, ZG, ZG], p_bc=[PF, PF, PZ, PZ, PZ, PZ], p_val=[G * L, 0.0, 0, 0, 0, 0],
                              u_tol=1e-10, p_tol=1e-10, p_final_tol=1e-10, grading=(hx, hy, hz))
        s = product.Solver(c)
        for _ in range(2500):
            s.step()
        U = s.get("U").reshape(nz, ny, nx, 3)
        yc = np.cumsum(hy) - 0.5 * hy
        exact = G / (2 * nu) * yc * (1.0 - yc)
        errs[ny] = np.abs(U[0, :, nx // 2, 0] - exact).max() / exact.max()
        assert np.abs(U[..., 1]).max() < 1e-7 and np.abs(U[..., 2]).max() < 1e-7
        assert abs(U[0, 0, nx // 2, 0] / yc[0] - G / (2 * nu)) < 1e-7 * G / (2 * nu)
        s.close()
    assert errs[32] < 0.3 * errs[16] and errs[32] < 5e-3, errs


# ---- Taylor-Green vortices between symmetry planes (the functions are shared with the oracle's CPU test) ---------------------------------
def _slip_solver(product, solver=0):
    def mk(shape, dx, dt, nu):
        return product.Solver(product.make_case(solver, shape[0], shape[1], shape[2], dx, dt, nu, u_bc=[product.FY_BC_U_SLIP] * 6, u_tol=1e-12, p_tol=1e-11,
                                                p_final_tol=1e-11, p_rel_tol=0.0))
    return mk


@pytest.mark.parametrize("plane", ["xy", "yz", "xz"])
def test_taylor_green_vortex_is_the_exact_navier_stokes_solution(product, plane):
    """two-dimensional Taylor-Green in each coordinate plane, exact for the full equations (Re = 31 and the Stokes regime), up to 64 x 64"""
    from test_fv_oracle import taylor_green_2d_checks
    taylor_green_2d_checks(_slip_solver(product, 0), plane, sizes=(8, 16, 32, 64))


def test_taylor_green_vortex_in_pimpleFoamYade_converges_at_first_order(product):
    """the same vortex through pimpleFoamYade's UcEqn (no particles, alphac = 1): its explicit stress term div(nuEff dev2(T(grad(Uc))))
    (divDevRhoReff; zero for a solenoidal field) is a Gauss gradient inside a Gauss divergence, one-sided in the cells along a boundary -- an
    O(1) residue in a layer of thickness dx, so the amplitude converges at first order where icoFoamYade's laplacian(nu, U) converges at second.
    (No relaxationFactors entry: relax(1) on a symmetry plane would lag the tangential components of the boundary cells by one iteration.)"""
    from test_fv_oracle import taylor_green
    nu, T = 0.1, 0.5
    errs = []
    for n in (16, 32, 64):
        dt = 0.1 * (8.0 / n) ** 2
        mk = lambda shape, dx, dt_, nu_: product.Solver(product.make_case(1, shape[0], shape[1], shape[2], dx, dt_, nu_, u_bc=[product.FY_BC_U_SLIP] * 6,
                                                                           u_tol=1e-12, p_tol=1e-11, p_final_tol=1e-11, p_rel_tol=0.0, u_relax=0.0))
        amp, dev = taylor_green(mk, (n, n, 1), nu, dt, int(round(T / dt)))
        errs.append(abs(amp - np.exp(-2 * nu * T)))
        assert dev < 0.01, (n, dev)
    assert errs[2] < 2e-3 and 1.6 < errs[0] / errs[1] < 2.6 and 1.6 < errs[1] / errs[2] < 2.6, errs


def test_three_dimensional_taylor_green_vortex(product):
    """Taylor & Green's three-dimensional vortex in one cell of its array (symmetry planes on all six sides), 16^3 .. 64^3: at Re = 0.03 the
    field decays in place at the Stokes rate exp(-3 nu t), second order in dx; at Re = 31 (no closed form) the run stays symmetric, loses
    kinetic energy at least as fast as the Stokes rate at t = 0 predicts, and conserves mass to solver tolerance"""
    from test_fv_oracle import taylor_green
    nu, T, amp0 = 1.0, 0.1, 0.01
    errs = []
    for n in (16, 32, 64):
        dt = 0.004 * (8.0 / n) ** 2
        amp, dev = taylor_green(_slip_solver(product), (n, n, n), nu, dt, int(round(T / dt)), amp0=amp0, three_d=True)
        errs.append(amp - np.exp(-3 * nu * T))
        assert dev < 5e-4, (n, dev)                       # (the secondary flow the convective term drives: O(Re), not a discretisation error)
    assert errs[0] > 0 and 3.7 < errs[0] / errs[1] < 4.3 and 3.7 < errs[1] / errs[2] < 4.3, errs
    assert errs[2] < 1e-4
    # Re = 31
    n, nu, dt = 32, 0.1, 0.01
    s = _slip_solver(product)((n, n, n), np.pi / n, dt, nu)
    c = (np.arange(n) + 0.5) * np.pi / n
    Z, Y, X = np.meshgrid(c, c, c, indexing="ij")
    U0 = np.zeros((n, n, n, 3))
    U0[..., 0] = np.sin(X) * np.cos(Y) * np.cos(Z)
    U0[..., 1] = -np.cos(X) * np.sin(Y) * np.cos(Z)
    s.set("U", U0.reshape(-1, 3))
    E0 = (U0 ** 2).sum()
    for _ in range(20):
        s.step()
    U = s.get("U").reshape(n, n, n, 3)
    E = (U ** 2).sum()
    t = 20 * dt
    assert 0.5 * np.exp(-6 * nu * t) < E / E0 < np.exp(-6 * nu * t) * 1.01, (E / E0, np.exp(-6 * nu * t))
    assert np.abs(U[..., 2]).max() > 1e-3                                  # the third component is born from the other two
    # the vortex array's symmetry: a quarter turn about the box's z axis maps the flow onto itself, u(x, y, z) = -v(y, pi - x, z), w(x, y, z) = w(y, pi - x, z)
    np.testing.assert_allclose(U[..., 0], -np.flip(np.swapaxes(U[..., 1], 1, 2), axis=2), atol=1e-8)
    np.testing.assert_allclose(U[..., 2], np.flip(np.swapaxes(U[..., 2], 1, 2), axis=2), atol=1e-8)
    assert s.stats()["cont_err_sum_local"] < 1e-8
    s.close()


def test_rayleigh_layer_over_an_impulsively_started_plate(product):
    """Stokes' first problem on the HIP solver, up to 512 cells across the layer and on a block graded towards the plate (shared with the oracle's test)"""
    from test_fv_oracle import rayleigh_layer, rayleigh_case, geometric_sizes

    def mk(ny, h, dt, nu, U0, solver):
        a, kw = rayleigh_case(ny, h, dt, nu, U0, solver)
        return product.Solver(product.make_case(*a, **kw))
    nu, U0, T = 0.01, 1.0, 0.5
    errs = []
    for ny, dt in ((64, 0.005), (128, 0.00125), (256, 0.0003125), (512, 0.000078125)):
        e, cross = rayleigh_layer(mk, np.full(ny, 1.0 / ny), nu, U0, dt, int(round(T / dt)))
        assert cross < 1e-9
        errs.append(e)
    assert errs[3] < 1.5e-4 and all(2.5 < a / b < 5.0 for a, b in zip(errs, errs[1:])), errs
    ep, cross = rayleigh_layer(mk, np.full(128, 1.0 / 128), nu, U0, 0.00125, 400, solver=1)
    assert abs(ep - errs[1]) < 3e-4 and cross < 1e-9
    eg, cross = rayleigh_layer(mk, geometric_sizes(48, 8.0, 1.0), nu, U0, 0.00125, 400)
    assert eg < 1.5 * errs[1] and cross < 1e-9, (eg, errs[1])
