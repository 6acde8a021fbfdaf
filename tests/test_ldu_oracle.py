"""Known answers for oracle/ldu_oracle.cpp -- icoFoamYade's loop body on a general polyhedral mesh in OpenFOAM's addressing, with the
non-orthogonal corrector loop (icoFoamYade.C:42, 114-131).  The FV arithmetic is OpenFOAM-6's, restated: PARITY UNPINNED; what pins the restatement is
(i) geometry identities, (ii) exactness properties of the corrected schemes on sheared meshes, (iii) the structured restatement (fv_oracle.cpp), which the
general one must reproduce on a lattice written as a polyhedral mesh -- also with its cells renumbered at random."""
import numpy as np
import pytest

import oracle as orc
import poly_meshes as pm


WALLS = dict(u_bc=[0] * 6, p_bc=[0] * 6)


def make(mesh, dt, nu, u_val=None, u_bc=None, p_bc=None, p_val=None, **kw):
    n = len(mesh["patch_start"])
    return orc.LduSolver(mesh, dt, nu, u_bc if u_bc is not None else [0] * n, u_val if u_val is not None else [(0, 0, 0)] * n, p_bc if p_bc is not None else [0] * n, p_val, **kw)


def test_geometry_identities(oracle):
    """closed cells (sum of outward face areas = 0), volumes that add up to the block's (a shear preserves it), exact centres of parallelepipeds, linear
    weights 1/2 and a correction vector orthogonal to nothing in particular but consistent: n = dcNO d + k"""
    mesh = pm.hex_block(5, 4, 3, (1.0, 0.8, 0.6), pm.shear(0.4, 0.2, -0.3), renumber_seed=2)
    s = make(mesh, 1e-3, 0.01)
    Sf, V, C, w, dc, k = (s.geometry(n) for n in ("Sf", "V", "C", "w", "dcNO", "kvec"))
    own, nei, ni = mesh["owner"], mesh["neighbour"], len(mesh["neighbour"])
    tot = np.zeros((mesh["n_cells"], 3))
    np.add.at(tot, own, Sf); np.subtract.at(tot, nei, Sf[:ni])
    assert np.abs(tot).max() < 1e-15
    assert V.sum() == pytest.approx(1.0 * 0.8 * 0.6, rel=1e-13) and np.allclose(V, V[0], rtol=1e-12)
    # lattice cell (i, j, k)'s centre = the map of the unsheared centre
    i, j, kk = np.meshgrid(np.arange(5), np.arange(4), np.arange(3), indexing="ij")
    c0 = np.stack([(i.ravel() + 0.5) / 5, (j.ravel() + 0.5) * 0.8 / 4, (kk.ravel() + 0.5) * 0.6 / 3], axis=1)
    want = pm.shear(0.4, 0.2, -0.3)(c0)
    got = C[mesh["perm"][(i + 5 * (j + 4 * kk)).ravel()]]
    np.testing.assert_allclose(got, want, atol=1e-14)
    np.testing.assert_allclose(w, 0.5, atol=1e-13)
    d = C[nei] - C[own[:ni]]
    n = Sf[:ni] / np.linalg.norm(Sf[:ni], axis=1)[:, None]
    np.testing.assert_allclose(dc[:ni, None] * d + k, n, atol=1e-13)
    assert np.abs(k).max() > 0.1                                        # (the mesh IS non-orthogonal)
    s.close()


def test_corrected_sngrad_is_exact_for_linear_fields_on_parallelepipeds(oracle):
    """a field linear in space on a block of parallelepipeds (non-orthogonal, no skewness): its Gauss-linear gradient is exact, and so is the corrected
    surface-normal gradient dcNO (s_N - s_P) + k.grad(s)_f = n.g on EVERY internal face -- weights, nonOrthDeltaCoeffs, correction vectors and gradient
    in one identity.  The uncorrected one is off by k.g, which is what the correctNonOrthogonal loop (icoFoamYade.C:114-131) and the corrected laplacian add back."""
    mesh = pm.hex_block(5, 6, 4, (1.0, 0.9, 0.7), pm.shear(0.35, -0.25, 0.4), renumber_seed=5)
    s = make(mesh, 1e-3, 0.01)
    C, Cf, Sf = s.geometry("C"), s.geometry("Cf"), s.geometry("Sf")
    ni = len(mesh["neighbour"])
    g = np.array([0.7, -1.3, 0.45])
    n = Sf[:ni] / np.linalg.norm(Sf[:ni], axis=1)[:, None]
    sn = s.sngrad(C @ g + 2.0, Cf[ni:] @ g + 2.0)
    np.testing.assert_allclose(sn, n @ g, atol=1e-12)
    un = s.sngrad(C @ g + 2.0, Cf[ni:] @ g + 2.0, corrected=False)
    assert np.abs(un - n @ g).max() > 0.2
    s.close()


@pytest.mark.parametrize("renumber", [None, 7])
def test_couette_is_steady_on_a_sheared_mesh(oracle, renumber):
    """u = U0 y / H between a fixed and a moving wall, on a block sheared ACROSS the flow's gradient (x += a z: the in- and outlet planes and the faces between
    z-neighbours are non-orthogonal, and no boundary condition contradicts the solution): steady to solver tolerance -- with the cells renumbered at
    random too (owner / neighbour addressing, nothing lattice-like left)"""
    nx, ny, nz, H = 5, 8, 4, 1.0
    mesh = pm.hex_block(nx, ny, nz, (1.0, H, 0.5), pm.shear(a_xz=0.4), patches=[("inout", [0, 1]), ("bottom", [2]), ("top", [3]), ("sides", [4, 5])], renumber_seed=renumber)
    s = make(mesh, 0.05, 0.1, u_bc=[1, 0, 0, 1], u_val=[(0, 0, 0), (0, 0, 0), (1.0, 0, 0), (0, 0, 0)], p_bc=[1, 0, 0, 0], p_val=[0.0, 0, 0, 0], n_non_orth=1,
             u_tol=1e-12, p_tol=1e-12, p_final_tol=1e-12, p_rel_tol=0.0)
    C = s.geometry("C")
    U0 = np.zeros((mesh["n_cells"], 3)); U0[:, 0] = C[:, 1] / H
    s.set("U", U0)
    for _ in range(5):
        s.step()
    U = s.get("U").reshape(-1, 3)
    assert np.abs(U - U0).max() < 1e-8
    assert np.abs(s.get("p")).max() < 1e-8
    s.close()


@pytest.mark.parametrize("scheme", [0, 1, 2, 3, 4, 7, 8])
def test_general_mesh_reproduces_the_structured_restatement_on_a_lattice(oracle, scheme):
    """a uniform block written as a polyhedral mesh (cells renumbered at random) against fv_oracle.cpp on the same block: the lid-driven cavity over five
    steps, with the structured side's Jacobi-preconditioned PCG (the same algorithm as the general side's)"""
    n = 8
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (1.0, 0, 0)
    mesh = pm.hex_block(n, n, n, renumber_seed=11)
    tol = dict(p_tol=1e-11, p_rel_tol=0.0, p_final_tol=1e-11, u_tol=1e-11)
    g = make(mesh, 0.4 / n, 0.01, u_val=u_val, convection_scheme=scheme, convection_limiter_k=0.6, **tol)          # 1 upwind, 2 linearUpwind, 3 limitedLinear 0.6, 4 vanLeer, 7 SuperBee, 8 QUICK
    f = orc.FvSolver(orc.fv_case(0, n, n, n, 1.0 / n, 0.4 / n, 0.01, u_bc=[0] * 6, u_val=u_val, p_solver=0, p_final_tol=1e-11, p_final_rel_tol=0.0, convection_scheme=scheme, limiter_k=0.6,
                                 **{k: v for k, v in tol.items() if k != "p_final_tol"}))
    rs = np.random.RandomState(3)
    U0 = rs.rand(n ** 3, 3) * 0.05
    f.set("U", U0)
    Ug = np.zeros_like(U0); Ug[mesh["perm"]] = U0
    g.set("U", Ug)
    for _ in range(5):
        f.step(); g.step()
    Uf, pf = f.get("U").reshape(-1, 3), f.get("p")
    Ugl, pgl = pm.to_lattice(mesh, g.get("U").reshape(-1, 3)), pm.to_lattice(mesh, g.get("p"))
    assert np.abs(Ugl - Uf).max() < 1e-8 * np.abs(Uf).max()
    assert np.abs((pgl - pgl.mean()) - (pf - pf.mean())).max() < 1e-7 * np.abs(pf).max()
    sg, sf = g.stats(), f.stats()
    assert sg["courant_max"] == pytest.approx(sf["courant_max"], rel=1e-9)
    f.close(); g.close()


@pytest.mark.parametrize("solver", ["ico", "pimple"])
def test_symmetry_sides_on_a_lattice_reproduce_the_structured_restatement(oracle, solver):
    """symmetryPlane / symmetry / slip sides (FY_BC_U_SLIP) of the general solver against fv_oracle.cpp's on the same block: the normal component's implicit
    coefficient as a per-component boundary diagonal (solve: per component; A(): their average; H(): the rest), the value U_P - n (n . U_P) wherever the
    boundary velocity is used; pimple with relaxation (fvMatrix::relax's cmptMax / cmptMin of a vector coefficient) and two outer correctors"""
    n = 8
    u_bc = [2, 2, 2, 0, 0, 0]
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (1.0, 0, 0.3)
    mesh = pm.hex_block(n, n, n, renumber_seed=4)
    tol = dict(p_tol=1e-12, p_rel_tol=0.0, p_final_tol=1e-12, u_tol=1e-12)
    rs = np.random.RandomState(8)
    U0 = rs.rand(n ** 3, 3) * 0.05
    Ug = np.zeros_like(U0); Ug[mesh["perm"]] = U0
    if solver == "ico":
        g = make(mesh, 0.4 / n, 0.01, u_val=u_val, u_bc=u_bc, **tol)
        f = orc.FvSolver(orc.fv_case(0, n, n, n, 1.0 / n, 0.4 / n, 0.01, u_bc=u_bc, u_val=u_val, p_solver=0, p_final_rel_tol=0.0, **tol))
    else:
        rel = dict(u_relax=0.7, u_relax_final=1.0, p_relax=0.6, p_relax_final=1.0)
        p_bc = [0, 0, 0, 2, 2, 2]
        f = orc.FvSolver(orc.fv_case(1, n, n, n, 1.0 / n, 0.4 / n, 0.01, g=(0, 0, -9.81), u_bc=u_bc, u_val=u_val, p_bc=p_bc, p_solver=0, n_outer=2, n_corr=2, p_final_rel_tol=0.0,
                                     p_max_iter=5000, **rel, **tol))
        g = orc.LduSolver(mesh, 0.4 / n, 0.01, u_bc, u_val, p_bc, solver=1, g=(0, 0, -9.81), n_outer=2, n_correctors=2, **rel, **tol)
    f.set("U", U0); g.set("U", Ug)
    for _ in range(4):
        f.step(); g.step()
    Uf, pf = f.get("U").reshape(-1, 3), f.get("p")
    Ugl, pgl = pm.to_lattice(mesh, g.get("U").reshape(-1, 3)), pm.to_lattice(mesh, g.get("p"))
    assert np.abs(Uf).max() > 0.1
    assert np.abs(Ugl - Uf).max() < 1e-8 * np.abs(Uf).max()
    assert np.abs((pgl - pgl.mean()) - (pf - pf.mean())).max() < 1e-7 * np.abs(pf).max()
    f.close(); g.close()


def test_symmetry_sides_with_oblique_normals(oracle):
    """the same cavity turned in space (no side's normal along an axis): a symmetry face's coefficient is then (|n_x|, |n_y|, |n_z|) deltaCoeffs per component with the
    rest of -n (n . U_P) deltaCoeffs explicit (lagging by one assembly), and fvMatrix::A() takes the AVERAGE of the three, (|n_x| + |n_y| + |n_z|) / 3, which is not
    invariant under rotation: 1 / A weighs the pressure term of the face fluxes in the cells along a symmetry side differently, so even the steady state agrees with the
    turned solution of the axis-aligned box to discretisation accuracy only (measured 3e-3 of the lid speed at 6^3 cells; a wrong sign or a missing term is O(1) or
    unstable).  Exact whatever the orientation: no flux through a symmetry face"""
    n = 6
    u_bc = [2, 2, 2, 0, 0, 0]
    a, b, c = 0.5, -0.35, 0.8
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    R = Rz @ Ry @ Rx
    lid_u = np.array([1.0, 0, 0.3])
    out = []
    for rot in (None, R):
        mesh = pm.hex_block(n, n, n, vertex_map=(None if rot is None else (lambda P: P @ rot.T)))
        u_val = [(0, 0, 0)] * 6
        u_val[3] = tuple(lid_u if rot is None else rot @ lid_u)
        s = make(mesh, 0.05, 0.05, u_val=u_val, u_bc=u_bc, p_tol=1e-12, p_rel_tol=0.0, p_final_tol=1e-12, u_tol=1e-12, n_non_orth=(0 if rot is None else 1))
        for _ in range(400):
            s.step()
        U1 = s.get("U").reshape(-1, 3).copy()
        s.step()
        U = s.get("U").reshape(-1, 3)
        assert np.abs(U - U1).max() < 1e-9                                  # steady
        out.append((U.copy(), s.get("p").copy()))
        if rot is not None:
            Sf = s.geometry("Sf")
            ni = len(mesh["neighbour"])
            phi = s.get("phi")
            slip_faces = np.concatenate([np.arange(mesh["patch_start"][q], mesh["patch_start"][q] + mesh["patch_size"][q]) for q in range(3)])
            assert np.abs(Sf[slip_faces] / np.linalg.norm(Sf[slip_faces], axis=1)[:, None]).max() < 0.95          # (oblique indeed)
            assert np.abs(phi[slip_faces]).max() < 1e-15 and ni <= slip_faces.min()
        s.close()
    (U0, p0), (Ur, pr) = out
    assert np.abs(U0).max() > 0.2
    assert np.abs(Ur - U0 @ R.T).max() < 1e-2 * np.abs(U0).max()
    assert np.abs((pr - pr.mean()) - (p0 - p0.mean())).max() < 5e-2 * np.abs(p0 - p0.mean()).max()


def test_cyclic_pairs_fold_into_internal_faces(oracle):
    """translational cyclic patches [OF-6 cyclicPolyPatch / cyclicFvPatch]: each pair of faces is one more internal face whose neighbour is seen at its image -- on a
    uniform block every folded face gets the weights, nonOrthDeltaCoeffs and (zero) correction vectors of the mesh's own faces, every cell its volume and centre; on a
    periodically distorted block the cells stay closed and a field linear in space has its exact Gauss gradient in every cell but for the jump the period adds"""
    n = 5
    mesh = pm.make_cyclic(pm.hex_block(n, n + 1, n + 2, (1.0, 1.2, 1.4), renumber_seed=3), [(0, 1), (4, 5)])
    s = make(mesh, 1e-3, 0.01, u_bc=[1, 1, 0, 0, 1, 1])
    nfold = n * (n + 1) + (n + 1) * (n + 2)
    assert s.ni == len(mesh["neighbour"]) + nfold and s.nf == len(mesh["owner"]) - nfold
    V, C, w, dc, kv, sep, Sf = (s.geometry(k) for k in ("V", "C", "w", "dcNO", "kvec", "sep", "Sf"))
    np.testing.assert_allclose(V, 1.0 * 1.2 * 1.4 / (n * (n + 1) * (n + 2)), rtol=1e-12)
    np.testing.assert_allclose(w, 0.5, atol=1e-12); assert np.abs(kv).max() < 1e-12
    fold = np.arange(len(mesh["neighbour"]), s.ni)
    assert np.allclose(np.abs(sep[fold]).sum(axis=1), np.where(np.abs(Sf[fold, 0]) > 0, 1.0, 1.4)) and not sep[:fold[0]].any()
    nrm = np.abs(Sf[:s.ni]) / np.linalg.norm(Sf[:s.ni], axis=1)[:, None]
    np.testing.assert_allclose(dc[:s.ni], nrm @ np.array([n / 1.0, (n + 1) / 1.2, (n + 2) / 1.4]), rtol=1e-12)
    s.close()
    L = (1.0, 1.2, 1.4)
    mesh = pm.make_cyclic(pm.hex_block(n, n + 1, n + 2, L, pm.wavy_periodic(0.03 * 12 / n, L), renumber_seed=3), [(0, 1), (2, 3), (4, 5)])
    s = make(mesh, 1e-3, 0.01)
    assert s.nf == s.ni                                                       # no boundary face left
    Sf, V, w = s.geometry("Sf"), s.geometry("V"), s.geometry("w")
    assert V.sum() == pytest.approx(L[0] * L[1] * L[2], rel=1e-12) and V.min() > 0 and np.abs(w - 0.5).max() > 1e-3
    s.close()


def test_cyclic_flow_moves_with_the_mesh(oracle):
    """a cavity periodic in x (lid along x: a Couette-like flow with a disturbance): the same initial field moved by two cells along the period, on a differently
    numbered mesh, gives the same flow moved by two cells -- nothing knows where the cyclic seam is"""
    n, m = 8, 2
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (1.0, 0, 0)
    tol = dict(p_tol=1e-12, p_rel_tol=0.0, p_final_tol=1e-12, u_tol=1e-12)
    rs = np.random.RandomState(5)
    U0 = rs.rand(n, n, n, 3) * 0.2                     # [k][j][i]
    out = []
    for seed, shift in ((1, 0), (2, m)):
        mesh = pm.make_cyclic(pm.hex_block(n, n, n, renumber_seed=seed), [(0, 1)])
        s = make(mesh, 0.4 / n, 0.01, u_val=u_val, convection_scheme=4, **tol)
        Ul = np.roll(U0, shift, axis=2).reshape(-1, 3)
        Ug = np.zeros_like(Ul); Ug[mesh["perm"]] = Ul
        s.set("U", Ug)
        for _ in range(4):
            s.step()
        assert s.stats()["cont_sum_local"] < 1e-12
        out.append((pm.to_lattice(mesh, s.get("U").reshape(-1, 3)).reshape(n, n, n, 3), pm.to_lattice(mesh, s.get("p")).reshape(n, n, n)))
        s.close()
    (Ua, pa), (Ub, pb) = out
    assert np.abs(Ua).max() > 0.3
    assert np.abs(np.roll(Ua, m, axis=2) - Ub).max() < 1e-9
    pa, pb = np.roll(pa, m, axis=2), pb
    assert np.abs((pa - pa.mean()) - (pb - pb.mean())).max() < 1e-8 * np.abs(pa - pa.mean()).max()


@pytest.mark.parametrize("wavy", [False, True])
def test_taylor_green_vortices_in_a_periodic_box(oracle, wavy):
    """the exact Navier-Stokes solution u = sin x cos y F, v = -cos x sin y F, p = (cos 2x + cos 2y) F^2 / 4, F = exp(-2 nu t) in a box of period 2 pi in x, y (and z):
    all three pairs of sides cyclic, no boundary face at all (the pressure level from the reference cell).  Second order: the error falls about fourfold from 12 to 24
    cells per period -- on the periodically distorted mesh too (non-orthogonal correctors across the folded faces)"""
    nu, T = 0.1, 0.5
    L = (2 * np.pi, 2 * np.pi, 0.75)
    errs = []
    for n, dt in ((12, 0.02), (24, 0.005)):
        vm = pm.wavy_periodic(0.03 * 12 / n, L) if wavy else None
        mesh = pm.make_cyclic(pm.hex_block(n, n, 3, L, vm, renumber_seed=7), [(0, 1), (2, 3), (4, 5)])
        s = make(mesh, dt, nu, n_non_orth=(2 if wavy else 0), p_tol=1e-11, p_rel_tol=0.0, p_final_tol=1e-11, u_tol=1e-11, p_max_iter=20000)
        C = s.geometry("C")
        ex = lambda t: np.stack([np.sin(C[:, 0]) * np.cos(C[:, 1]), -np.cos(C[:, 0]) * np.sin(C[:, 1]), 0 * C[:, 0]], axis=1) * np.exp(-2 * nu * t)
        s.set("U", ex(0.0))
        for _ in range(int(round(T / dt))):
            s.step()
        U = s.get("U").reshape(-1, 3)
        errs.append(np.abs(U - ex(T)).max())
        assert wavy or np.abs(U[:, 2]).max() < 1e-12
        assert s.stats()["cont_sum_local"] < 1e-11
        s.close()
    assert errs[0] < (0.06 if wavy else 0.03) and errs[1] < errs[0] / 3.0, errs


def test_non_orthogonal_correctors_converge_the_pressure_equation(oracle):
    """on a wavy (non-orthogonal, skewed) cavity the explicit part of the corrected laplacian, k.grad(p)_f, is formed from the pressure BEFORE each solve:
    phi = phiHbyA - pEqn.flux() is conservative whatever it was (the flux carries the same term: continuity errors at rounding with 0, 1 or 3 correctors),
    but the pressure only satisfies the CORRECTED equation -- sum_f (rAU_f |Sf| snGrad_corrected(p)) = div(phiHbyA) -- as far as the lag allows: each
    further pass of the correctNonOrthogonal loop (icoFoamYade.C:114-131) shrinks that residual"""
    n = 8
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (1.0, 0, 0)
    mesh = pm.hex_block(n, n, n, vertex_map=pm.wavy(0.03))
    own, nei = mesh["owner"], mesh["neighbour"]
    ni = len(nei)
    res = {}
    for no in (0, 1, 3):
        s = make(mesh, 0.02, 0.01, u_val=u_val, n_non_orth=no, p_tol=1e-12, p_rel_tol=0.0, p_final_tol=1e-12, p_max_iter=20000)
        for _ in range(4):
            s.step()
        assert s.stats()["cont_sum_local"] < 1e-12
        U = s.get("U").reshape(-1, 3)
        assert np.isfinite(U).all() and np.abs(U).max() < 1.5
        phi, p, coef, phiH, dc = s.get("phi"), s.get("p"), s.get("p_coef"), s.get("phiHbyA"), s.geometry("dcNO")
        assert np.abs(phi[ni:]).max() < 1e-14                              # a closed box
        flux = coef[:ni] / dc[:ni] * s.sngrad(p, p[own[ni:]])             # rAU_f |Sf| snGrad_corrected(p), zeroGradient walls
        r = np.zeros(mesh["n_cells"])
        np.add.at(r, own[:ni], flux - phiH[:ni]); np.subtract.at(r, nei, flux - phiH[:ni])
        np.subtract.at(r, own[ni:], phiH[ni:])
        r[0] = 0.0                                                         # (the reference cell's row carries setReference's term)
        res[no] = np.abs(r).max() / np.abs(phiH[:ni]).max()
        s.close()
    assert res[1] < 0.5 * res[0] and res[3] < 0.25 * res[1] and res[3] < 1e-3, res


def test_poiseuille_converges_at_second_order_on_wavy_meshes(oracle):
    """plane Poiseuille flow driven by a fixed pressure difference between the (plane) inlet and outlet, zeroGradient U there, walls at y = 0, H, on blocks
    whose INTERIOR vertices are displaced by a smooth map (non-orthogonal and skewed cells, all different): grad(u) and grad(p) both have components along the
    correction vectors, so the momentum equation's explicit laplacian correction and the pressure equation's both carry part of the answer.  The steady
    profile is u = G y (H - y) / (2 nu): the error falls ~4 x per halving (a smooth mapping keeps the schemes second order), and without the correctors'
    loop and the correction there would be an O(amplitude) cross flow"""
    H, L, nu, dp = 1.0, 1.0, 0.1, 0.8
    G = dp / L
    errs = []
    for n in (6, 12):
        mesh = pm.hex_block(n, n, max(n // 2, 3), (L, H, 0.5), pm.wavy(0.035, (L, H, 0.5)), patches=[("inlet", [0]), ("outlet", [1]), ("walls", [2, 3]), ("sides", [4, 5])])
        s = make(mesh, 0.25, nu, u_bc=[1, 1, 0, 1], p_bc=[1, 1, 0, 0], p_val=[dp, 0.0, 0, 0], n_non_orth=2, p_tol=1e-10, p_rel_tol=0.0, p_final_tol=1e-10, u_tol=1e-10, p_max_iter=20000)
        C = s.geometry("C")
        for _ in range(120):
            s.step()
        U = s.get("U").reshape(-1, 3)
        y = C[:, 1]
        umax = G * H * H / (8 * nu)
        errs.append(np.abs(U[:, 0] - G * y * (H - y) / (2 * nu)).max() / umax)
        assert np.abs(U[:, 1:]).max() < 0.02 * umax
        s.close()
    assert errs[0] < 0.08 and errs[1] < errs[0] / 2.8, errs


def test_prisms_geometry(oracle):
    """triangular prisms (triangular + quadrilateral faces, five-faced cells) under a shear: closed cells, equal volumes adding up to the box's, centres =
    the prisms' centroids (the triangle branch of the face decomposition, the pyramid sums over five faces)"""
    mesh = pm.prism_block(4, 4, 3, (1.0, 1.0, 0.9), pm.shear(0.3, 0.2, -0.1))
    assert sorted(set(np.diff(mesh["face_offsets"]))) == [3, 4]
    s = make(mesh, 1e-3, 0.01)
    Sf, V, C, Cf = (s.geometry(n) for n in ("Sf", "V", "C", "Cf"))
    own, nei, ni = mesh["owner"], mesh["neighbour"], len(mesh["neighbour"])
    tot = np.zeros((mesh["n_cells"], 3))
    np.add.at(tot, own, Sf); np.subtract.at(tot, nei, Sf[:ni])
    assert np.abs(tot).max() < 1e-15
    assert V.sum() == pytest.approx(0.9, rel=1e-13) and np.allclose(V, V[0], rtol=1e-12)
    # a prism's centroid = the mean of its six vertices (it is the affine image of a right prism)
    P, off, fp = mesh["points"], mesh["face_offsets"], mesh["face_points"]
    verts = [set() for _ in range(mesh["n_cells"])]
    for f in range(len(own)):
        verts[own[f]].update(fp[off[f]:off[f + 1]])
        if f < ni: verts[nei[f]].update(fp[off[f]:off[f + 1]])
    want = np.array([P[sorted(v)].mean(axis=0) for v in verts])
    np.testing.assert_allclose(C, want, atol=1e-14)
    s.close()


@pytest.mark.parametrize("n_outer,relax,adjust,les", [(1, 0.0, 0, 0), (2, 0.7, 0, 0), (1, 0.0, 1, 0), (2, 0.0, 0, 1), (1, 0.0, 0, 2), (2, 0.0, 0, 3), (1, 0.0, 0, 6)])
def test_pimple_on_a_lattice_reproduces_the_structured_restatement(oracle, n_outer, relax, adjust, les):
    """pimpleFoamYade's void-fraction-weighted equations (UcEqn.H, pEqn.H: gravity, fixedFluxPressure walls, PIMPLE outer correctors, relaxation) restated for a general
    mesh, against fv_oracle.cpp on the same block with a cloud: the general side is given the void fraction, the implicit drag coefficient and the explicit source the
    structured side's coupling left, step by step; the cells are renumbered at random"""
    n, box = 8, 0.1
    dx = box / n
    mesh = pm.hex_block(n, n, n, (box, box, box), renumber_seed=5)
    tol = dict(p_tol=1e-12, p_rel_tol=0.0, p_final_tol=1e-12, u_tol=1e-12)
    rel = dict(u_relax=relax, u_relax_final=1.0 if relax else 0.0, p_relax=0.6 if relax else 0.0, p_relax_final=1.0 if relax else 0.0,
               adjust_time_step=adjust, max_co=0.4, max_delta_t=3.2e-4)          # (adjustable: the step grows by 1.2 per step while the Courant number allows)
    if les == 1:                                         # LES Smagorinsky: nuEff in both parts of divDevRhoReff, nut renewed after the last outer corrector
        rel.update(turbulence_model=1, nut_initial=3e-5, les_delta_coeff=0.8)
    if les >= 2:                                         # (the last variants: Gauss upwind / linearUpwind / MUSCL for div(alphaPhic, Uc))
        rel.update(convection_scheme=les - 1)
    f = orc.FvSolver(orc.fv_case(1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), p_bc=[2] * 6, p_solver=0, n_outer=n_outer, n_corr=2, p_final_rel_tol=0.0, p_max_iter=5000, **rel, **tol))
    g = orc.LduSolver(mesh, 2e-4, 1e-5, [0] * 6, [(0, 0, 0)] * 6, [2] * 6, solver=1, g=(0, 0, -9.81), n_outer=n_outer, n_correctors=2, **rel, **tol)
    rs = np.random.RandomState(17)
    perm = mesh["perm"]
    for step in range(3):
        rec = np.zeros((1000, 10))
        rec[:, 0:3] = rs.random_sample((1000, 3)) * np.array([box, box, 0.6 * box]) + np.array([0.0, 0.0, 0.05 * box])
        rec[:, 3:6] = 0.05 * rs.standard_normal((1000, 3)); rec[:, 9] = 0.2 * dx
        cap = {}
        f.step(rec, capture=cap)
        assert cap["alpha"].min() < 0.9
        to_g = lambda a, nc: (lambda o: (o.__setitem__(perm, a.reshape(n ** 3, nc)), o)[1])(np.zeros((n ** 3, nc)))
        g.step(source=to_g(cap["uSource"], 3), alpha=to_g(cap["alpha"], 1), drag=to_g(cap["uSourceDrag"], 1))
        Uf, Ug = f.get("U").reshape(-1, 3), pm.to_lattice(mesh, g.get("U").reshape(-1, 3))
        assert np.abs(Ug - Uf).max() < 1e-8 * np.abs(Uf).max(), step
        assert g.stats()["delta_t"] == f.stats()["delta_t"] and (f.stats()["delta_t"] > 2e-4) == bool(adjust)
    pf, pg = f.get("p"), pm.to_lattice(mesh, g.get("p"))
    assert np.abs((pg - pg.mean()) - (pf - pf.mean())).max() < 1e-7 * np.abs(pf).max()
    assert np.abs(f.get("U")).max() > 1e-4
    if les == 1:
        nf, ng = f.get("nut"), pm.to_lattice(mesh, g.get("nut"))
        assert nf.max() > 0 and np.abs(ng - nf).max() < 1e-7 * nf.max() and not np.allclose(nf, 3e-5)
    f.close(); g.close()


@pytest.mark.parametrize("variant", ["plain", "upwind_relaxed_calculated"])
def test_les_kEqn_on_a_lattice_reproduces_the_structured_restatement(oracle, variant):
    """LESModel kEqn (DPMTurbulenceModels.C:76-77) on the general mesh: the k transport equation assembled over faces (alpha-weighted, SuSp / Sp sources, bound()), nut = Ck
    sqrt(k) delta, against fv_oracle.cpp's on the same block with a cloud and a moving wall -- with upwind convection of k, a relaxation factor, a fixed-value k patch and
    `calculated` nut patches too (the file's value before the first correctNut(), Ck sqrt(k_b) delta after)"""
    n, box = 8, 0.1
    dx = box / n
    mesh = pm.hex_block(n, n, n, (box, box, box), renumber_seed=5)
    tol = dict(p_tol=1e-12, p_rel_tol=0.0, p_final_tol=1e-12, u_tol=1e-12)
    les = dict(turbulence_model=2, nut_initial=3e-5, les_delta_coeff=0.8, k_initial=2e-4, k_tol=1e-13)
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (0.3, 0, 0.1)
    fkw, gkw = {}, {}
    if variant != "plain":
        les.update(k_convection_scheme=1, k_relax=0.8)
        fkw = dict(k_bc=[0, 0, 0, 1, 0, 0], k_value=[0, 0, 0, 5e-4, 0, 0], nut_bc=[3, 0, 3, 3, 0, 1], nut_value=[3e-5, 0, 3e-5, 6e-5, 0, 2e-5])
        gkw = dict(k_bc=[0, 0, 0, 1, 0, 0], k_val=[0, 0, 0, 5e-4, 0, 0], nut_bc=[3, 0, 3, 3, 0, 1], nut_val=[3e-5, 0, 3e-5, 6e-5, 0, 2e-5])
    else:
        les.update(k_convection_scheme=0)
    f = orc.FvSolver(orc.fv_case(1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), u_val=u_val, p_bc=[2] * 6, p_solver=0, n_outer=2, n_corr=2, p_final_rel_tol=0.0, p_max_iter=5000, **les, **fkw, **tol))
    g = orc.LduSolver(mesh, 2e-4, 1e-5, [0] * 6, u_val, [2] * 6, solver=1, g=(0, 0, -9.81), n_outer=2, n_correctors=2, **les, **gkw, **tol)
    rs = np.random.RandomState(17)
    perm = mesh["perm"]
    for step in range(3):
        rec = np.zeros((1000, 10))
        rec[:, 0:3] = rs.random_sample((1000, 3)) * np.array([box, box, 0.6 * box]) + np.array([0.0, 0.0, 0.05 * box])
        rec[:, 3:6] = 0.05 * rs.standard_normal((1000, 3)); rec[:, 9] = 0.2 * dx
        cap = {}
        f.step(rec, capture=cap)
        to_g = lambda a, nc: (lambda o: (o.__setitem__(perm, a.reshape(n ** 3, nc)), o)[1])(np.zeros((n ** 3, nc)))
        g.step(source=to_g(cap["uSource"], 3), alpha=to_g(cap["alpha"], 1), drag=to_g(cap["uSourceDrag"], 1))
        kf, kg = f.get("k"), pm.to_lattice(mesh, g.get("k"))
        assert np.abs(kg - kf).max() < 1e-8 * kf.max(), step
        nf, ng = f.get("nut"), pm.to_lattice(mesh, g.get("nut"))
        assert np.abs(ng - nf).max() < 1e-8 * nf.max(), step
        Uf, Ug = f.get("U").reshape(-1, 3), pm.to_lattice(mesh, g.get("U").reshape(-1, 3))
        assert np.abs(Ug - Uf).max() < 1e-8 * np.abs(Uf).max(), step
    assert not np.allclose(kf, 2e-4, rtol=1e-3) and kf.min() > 0
    f.close(); g.close()


def test_ras_kEpsilon_on_a_lattice_reproduces_the_structured_restatement(oracle):
    """RASModel kEpsilon (DPMTurbulenceModels.C:70-71) on the general mesh, without wall functions: the epsilon equation, then k with the new epsilon, nut = Cmu k^2 / epsilon,
    against fv_oracle.cpp's on the same block with a cloud and a moving wall -- non-default coefficients, upwind convection and relaxation of both, fixed-value patches of k
    and epsilon, `calculated` nut patches"""
    n, box = 8, 0.1
    dx = box / n
    mesh = pm.hex_block(n, n, n, (box, box, box), renumber_seed=5)
    tol = dict(p_tol=1e-12, p_rel_tol=0.0, p_final_tol=1e-12, u_tol=1e-12)
    ras = dict(turbulence_model=3, nut_initial=3e-5, k_initial=2e-4, eps_initial=1.2e-4, k_tol=1e-13, eps_tol=1e-13, ras_cmu=0.085, ras_c1=1.4, ras_c2=1.9, ras_c3=-0.33,
               ras_sigmak=1.1, ras_sigmaeps=1.25, k_convection_scheme=1, eps_convection_scheme=1, k_relax=0.8, eps_relax=0.7)
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (0.3, 0, 0.1)
    fkw = dict(k_bc=[0, 0, 0, 1, 0, 0], k_value=[0, 0, 0, 5e-4, 0, 0], eps_bc=[0, 0, 1, 1, 0, 0], eps_value=[0, 0, 2e-4, 3e-4, 0, 0], nut_bc=[3, 0, 3, 3, 0, 1], nut_value=[3e-5, 0, 3e-5, 6e-5, 0, 2e-5])
    gkw = dict(k_bc=fkw["k_bc"], k_val=fkw["k_value"], eps_bc=fkw["eps_bc"], eps_val=fkw["eps_value"], nut_bc=fkw["nut_bc"], nut_val=fkw["nut_value"])
    f = orc.FvSolver(orc.fv_case(1, n, n, n, dx, 2e-4, 1e-5, g=(0, 0, -9.81), u_val=u_val, p_bc=[2] * 6, p_solver=0, n_outer=2, n_corr=2, p_final_rel_tol=0.0, p_max_iter=5000, **ras, **fkw, **tol))
    g = orc.LduSolver(mesh, 2e-4, 1e-5, [0] * 6, u_val, [2] * 6, solver=1, g=(0, 0, -9.81), n_outer=2, n_correctors=2, **ras, **gkw, **tol)
    rs = np.random.RandomState(17)
    perm = mesh["perm"]
    for step in range(3):
        rec = np.zeros((1000, 10))
        rec[:, 0:3] = rs.random_sample((1000, 3)) * np.array([box, box, 0.6 * box]) + np.array([0.0, 0.0, 0.05 * box])
        rec[:, 3:6] = 0.05 * rs.standard_normal((1000, 3)); rec[:, 9] = 0.2 * dx
        cap = {}
        f.step(rec, capture=cap)
        to_g = lambda a, nc: (lambda o: (o.__setitem__(perm, a.reshape(n ** 3, nc)), o)[1])(np.zeros((n ** 3, nc)))
        g.step(source=to_g(cap["uSource"], 3), alpha=to_g(cap["alpha"], 1), drag=to_g(cap["uSourceDrag"], 1))
        for nm in ("epsilon", "k", "nut"):
            a, b = f.get(nm), pm.to_lattice(mesh, g.get(nm))
            assert np.abs(b - a).max() < 1e-8 * a.max(), (nm, step)
        Uf, Ug = f.get("U").reshape(-1, 3), pm.to_lattice(mesh, g.get("U").reshape(-1, 3))
        assert np.abs(Ug - Uf).max() < 1e-8 * np.abs(Uf).max(), step
    assert not np.allclose(f.get("k"), 2e-4, rtol=1e-3) and not np.allclose(f.get("epsilon"), 1.2e-4, rtol=1e-3)
    f.close(); g.close()


def test_two_to_one_refined_polyhedra(oracle):
    """a box whose upper layers are refined 2 x 2 (hexRef8-like): the cells under the interface are polyhedra with nine faces (four of them the fine cells' bottoms, the
    side faces five-point polygons with the hanging point) -- closed cells, volumes adding up, oblique centre-to-centre lines across the interface; the lid-driven cavity on it (distorted
    too) conserves mass to rounding"""
    L = (1.0, 1.0, 0.9)
    mesh = pm.refined_block(4, 4, 6, 3, L, pm.wavy(0.02, L))
    assert mesh["n_cells"] == 3 * 16 + 3 * 64 and set(np.diff(mesh["face_offsets"])) == {4, 5}
    u_val = [(0, 0, 0)] * 6
    u_val[5] = (1.0, 0, 0)                                                        # the lid over the fine layers
    s = make(mesh, 0.01, 0.01, u_val=u_val, n_non_orth=2, p_tol=1e-11, p_rel_tol=0.0, p_final_tol=1e-11, p_max_iter=20000)
    Sf, V, kv = s.geometry("Sf"), s.geometry("V"), s.geometry("kvec")
    own, nei, ni = mesh["owner"], mesh["neighbour"], len(mesh["neighbour"])
    tot = np.zeros((mesh["n_cells"], 3))
    np.add.at(tot, own, Sf); np.subtract.at(tot, nei, Sf[:ni])
    assert np.abs(tot).max() < 1e-15 and V.sum() == pytest.approx(0.9, rel=1e-13) and V.min() > 0
    assert np.bincount(np.concatenate([own, nei])).max() == 9 and np.abs(kv).max() > 0.3
    for _ in range(4):
        s.step()
    assert s.stats()["cont_sum_local"] < 1e-12 and np.abs(s.get("U")).max() > 0.01 and np.isfinite(s.get("p")).all()
    s.close()


def test_tetrahedra_geometry_and_a_cavity_on_them(oracle):
    """Kuhn tetrahedra (triangular faces only, four-faced cells, non-orthogonality around 50 degrees): closed cells, volumes adding up to the box's with every
    tetrahedron a sixth of its hexahedron, centres = the vertex means; the lid-driven cavity runs on them and conserves mass to rounding with two non-orthogonal passes"""
    mesh = pm.tet_block(4, 4, 3, (1.0, 1.0, 0.9), pm.wavy(0.02, (1.0, 1.0, 0.9)))
    assert set(np.diff(mesh["face_offsets"])) == {3} and mesh["n_cells"] == 4 * 4 * 3 * 6
    u_val = [(0, 0, 0)] * 6
    u_val[3] = (1.0, 0, 0)
    s = make(mesh, 0.01, 0.01, u_val=u_val, n_non_orth=2, p_tol=1e-11, p_rel_tol=0.0, p_final_tol=1e-11, p_max_iter=20000)
    Sf, V, C, kv = (s.geometry(n) for n in ("Sf", "V", "C", "kvec"))
    own, nei, ni = mesh["owner"], mesh["neighbour"], len(mesh["neighbour"])
    tot = np.zeros((mesh["n_cells"], 3))
    np.add.at(tot, own, Sf); np.subtract.at(tot, nei, Sf[:ni])
    assert np.abs(tot).max() < 1e-15 and V.sum() == pytest.approx(0.9, rel=1e-13) and V.min() > 0
    P, off, fp = mesh["points"], mesh["face_offsets"], mesh["face_points"]
    verts = [set() for _ in range(mesh["n_cells"])]
    for f in range(len(own)):
        verts[own[f]].update(fp[off[f]:off[f + 1]])
        if f < ni: verts[nei[f]].update(fp[off[f]:off[f + 1]])
    assert all(len(v) == 4 for v in verts)
    np.testing.assert_allclose(C, np.array([P[sorted(v)].mean(axis=0) for v in verts]), atol=1e-14)
    assert np.abs(kv).max() > 0.5                                    # (strongly non-orthogonal)
    for _ in range(3):
        s.step()
    assert s.stats()["cont_sum_local"] < 1e-12 and np.abs(s.get("U")).max() > 0.02 and np.isfinite(s.get("p")).all()
    s.close()


def test_rayleigh_layer_on_a_distorted_mesh(oracle):
    """Stokes' first problem (tests/test_fv_oracle.py: u = U0 erfc(y / 2 sqrt(nu t)) over a moving plate, exact for the full equations: the flow is parallel) on hexahedra
    whose interior vertices are displaced by a fixed smooth map -- non-orthogonal and skewed cells that do not become orthogonal under refinement.  A transient with a closed
    form for the corrected laplacian, the non-orthogonal corrector and the zero-gradient sides of the general restatement, started from the profile at t0 = 0.5 (a resolved
    layer: the impulsive start itself is first order on any mesh) and advanced to t = 1 with dt ~ h^2: the lattice converges at second order (0.0058, 0.0015, 0.00037); the
    distorted mesh between first and second (uncorrected skewness, one-sided boundary cells), the cross-flow and pressure the distortion excites stay under 0.1 % of U0.
    The pressure is fixed at BOTH open ends: with one end left zero-gradient for U and p alike (harmless on a lattice) the distorted boundary cells leave a pressure level
    of the order of the distortion uncontrolled and the error stops falling at 0.7 % (measured; also with the decaying shear mode sin(pi y) exp(-nu pi^2 t))"""
    from math import erfc, sqrt
    nu, U0, t0, T = 0.01, 1.0, 0.5, 1.0
    L = (0.25, 1.0, 0.25)
    res = {}
    for amp in (0.0, 0.01):
        errs, cross = [], []
        for ny, dt in ((16, 0.02), (32, 0.005), (64, 0.00125)) if amp else ((16, 0.02), (32, 0.005)):
            k = ny // 16
            mesh = pm.hex_block(4 * k, ny, 4 * k, L, pm.wavy(amp, L))
            u_val = [(0, 0, 0)] * 6
            u_val[2] = (U0, 0, 0)                          # ymin: the plate
            s = make(mesh, dt, nu, u_val=u_val, u_bc=[1, 1, 0, 1, 1, 1], p_bc=[1, 1, 0, 0, 0, 0], n_non_orth=1, p_tol=1e-10, p_rel_tol=0.0, p_final_tol=1e-10, u_tol=1e-10)
            C = s.geometry("C")
            prof = lambda t: np.array([U0 * erfc(y / (2 * sqrt(nu * t))) for y in C[:, 1]])
            Ui = np.zeros((len(C), 3)); Ui[:, 0] = prof(t0)
            s.set("U", Ui)
            for _ in range(int(round((T - t0) / dt))):
                s.step()
            U = s.get("U").reshape(-1, 3)
            errs.append(np.abs(U[:, 0] - prof(T)).max() / U0)
            cross.append((np.abs(U[:, 1]) + np.abs(U[:, 2])).max() / U0)
            s.close()
        res[amp] = (errs, cross)
    lat, dis = res[0.0], res[0.01]
    assert lat[0][0] < 0.007 and 3.5 < lat[0][0] / lat[0][1] < 4.5 and max(lat[1]) < 1e-9, lat
    assert dis[0][0] < 0.009 and dis[0][2] < 1.5e-3 and dis[0][0] / dis[0][1] > 2.0 and dis[0][1] / dis[0][2] > 1.8, dis
    assert max(dis[1]) < 1.2e-3 and dis[1][2] < dis[1][0], dis
