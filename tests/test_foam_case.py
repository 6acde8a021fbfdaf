"""OpenFOAM case I/O: the case reader behind fy_foam_case_* against two small case directories kept under
tests/golden/cases (authored for this repo in OpenFOAM's file format), its refusals, and -- on the GPU -- a run started from a case
directory, written with the reference's runTime.write() semantics and read back."""
import os
import shutil

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = os.path.join(HERE, "golden", "cases")
XMIN, XMAX, YMIN, YMAX, ZMIN, ZMAX = range(6)


def load(product_module, name, solver):
    return product_module.FoamCase(os.path.join(CASES, name), solver)


@pytest.fixture
def prod():
    from conftest import load_product
    return load_product()


def test_cavity_case_is_read_like_icoFoamYade_would(prod):
    fc = load(prod, "cavity_ico", prod.FY_SOLVER_ICO)
    c = fc.case
    assert (c.nx, c.ny, c.nz) == (16, 16, 16) and abs(c.dx - 0.1 / 16) < 1e-15 and list(c.origin) == [0.0, 0.0, 0.0]
    assert c.dt == 0.005 and fc.end_time == 0.05 and fc.write_interval_steps == 5 and fc.start_name == "0"
    assert c.nu == 0.01 and c.rho_particle == 2650 and c.rho_fluid == 1000 and list(c.g) == [0, 0, 0]
    assert fc.u_name == "U" and fc.phase == ""
    assert fc.patch_of_side[YMAX] == "movingWall" and all(fc.patch_of_side[s] == "fixedWalls" for s in (XMIN, XMAX, YMIN, ZMIN, ZMAX))
    assert all(c.u_bc[s] == prod.FY_BC_U_FIXED_VALUE for s in range(6))
    assert list(c.u_value[YMAX]) == [1.0, 0.0, 0.0] and all(list(c.u_value[s]) == [0, 0, 0] for s in (XMIN, XMAX, YMIN, ZMIN, ZMAX))
    assert all(c.p_bc[s] == prod.FY_BC_P_ZERO_GRADIENT for s in range(6))
    assert c.n_correctors == 2 and c.n_non_orth_correctors == 0 and c.p_ref_cell == 0 and c.p_ref_value == 0
    assert c.p_solver == prod.FY_PSOLVER_PCG_JACOBI and c.p_tol == 1e-7 and c.p_rel_tol == 0.05 and c.p_final_rel_tol == 0 and c.u_tol == 1e-6
    U, p = fc.initial_fields()
    assert U.shape == (4096, 3) and not U.any() and not p.any()
    fc.close()


def test_bed_case_is_read_like_pimpleFoamYade_would(prod):
    fc = load(prod, "bed_pimple", prod.FY_SOLVER_PIMPLE)
    c = fc.case
    assert (c.nx, c.ny, c.nz) == (12, 12, 24) and abs(c.dx - 0.005) < 1e-15 and np.allclose(list(c.origin), [-0.03, -0.03, 0.0])
    assert fc.u_name == "U.water" and fc.phase == "water" and c.rho_fluid == 1000 and c.nu == 1e-6 and list(c.g) == [0, 0, -9.81]
    assert fc.write_interval_steps == 5 and c.dt == 0.0002
    assert fc.patch_of_side == ["walls", "walls", "walls", "walls", "bottom", "top"]
    assert c.u_bc[ZMIN] == prod.FY_BC_U_FIXED_VALUE and list(c.u_value[ZMIN]) == [0, 0, 0.02] and c.u_bc[ZMAX] == prod.FY_BC_U_ZERO_GRADIENT
    assert c.p_bc[ZMAX] == prod.FY_BC_P_FIXED_VALUE and c.p_bc[ZMIN] == prod.FY_BC_P_FIXED_FLUX and c.p_bc[XMIN] == prod.FY_BC_P_FIXED_FLUX
    assert c.n_outer_correctors == 1 and c.n_correctors == 2 and c.momentum_predictor == 1
    assert c.p_solver == prod.FY_PSOLVER_PCG_MG and c.u_tol == 1e-5 and c.u_rel_tol == 0.1            # "(U.water|k|epsilon)" key
    fc.close()


@pytest.mark.parametrize("ty", ["slip", "symmetryPlane", "symmetry"])
def test_symmetry_sides_are_read_as_slip(prod, tmp_path, ty):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    text = (dst / "0/U").read_text()
    assert "noSlip" in text
    (dst / "0/U").write_text(text.replace("noSlip", ty, 1))
    if ty != "slip":                                   # (a symmetry patch carries its type in every field file)
        ptext = (dst / "0/p").read_text()
        i = ptext.index("fixedWalls")
        (dst / "0/p").write_text(ptext[:i] + ptext[i:].replace("zeroGradient", ty, 1))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    c = fc.case
    assert c.u_bc[YMAX] == prod.FY_BC_U_FIXED_VALUE and all(c.u_bc[s] == prod.FY_BC_U_SLIP for s in (XMIN, XMAX, YMIN, ZMIN, ZMAX))
    assert all(c.p_bc[s] == prod.FY_BC_P_ZERO_GRADIENT for s in range(6))
    fc.close()


@pytest.mark.parametrize("edit,needle", [
    (("system/blockMeshDict", "edges ( );", "edges ( arc 0 1 (0.5 -0.1 0) );"), "edges"),
    (("system/blockMeshDict", "simpleGrading (1 1 1)", "edgeGrading (1 1 1 1 1 1 1 1 1 1 1 1)"), "edgeGrading"),
    (("system/blockMeshDict", "(3 7 6 2)", "(3 7 6 1)"), "not a side"),
    (("0/U", "noSlip", "partialSlip"), "not supported"),
    (("0/p", "type            zeroGradient;", "type            totalPressure;"), "not supported"),
    (("system/controlDict", "startFrom       startTime;", "startFrom       someTime;"), "startFrom"),
    (("constant/transportProperties", "fluidDensity", "fluidDensityX"), "fluidDensity"),
    (("system/fvSolution", "PISO", "SIMPLE"), "PISO"),
    (("0/U", "value           uniform (1 0 0);", "#include \"lid\""), "lid"),
    (("0/U", "value           uniform (1 0 0);", "#calc \"1+1\";"), "#calc"),
    (("system/fvSchemes", "div(phi,U)       Gauss linear;", "div(phi,U)       Gauss limitedLinearV 1;"), "div(phi,U)"),
    (("system/fvSchemes", "div(phi,U)       Gauss linear;", "div(phi,U)       Gauss limitedLinear;"), "coefficient"),
    (("system/fvSchemes", "default Euler;", "default CrankNicolson 0.9;"), "ddtSchemes"),
])
def test_what_is_outside_the_supported_subset_is_refused_by_name(prod, tmp_path, edit, needle):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    rel, old, new = edit
    text = (dst / rel).read_text()
    assert old in text
    (dst / rel).write_text(text.replace(old, new, 1))
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert needle in str(e.value) and rel.split("/")[-1] in str(e.value)


def test_simple_grading_is_read_as_a_geometric_progression(prod, tmp_path):
    """blockMeshDict simpleGrading (ex ey ez): last / first cell size per direction, geometric in between [OF-6 blockMesh lineDivide]; cells that
    are uniform but not cubes come out as a (trivially) graded block as well"""
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    text = (dst / "system/blockMeshDict").read_text()
    (dst / "system/blockMeshDict").write_text(text.replace("simpleGrading (1 1 1)", "simpleGrading (4 0.25 1)", 1).replace("(16 16 16)", "(16 16 8)", 1))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    c = fc.case
    assert (c.nx, c.ny, c.nz) == (16, 16, 8) and bool(c.hx) and bool(c.hy) and bool(c.hz)
    hx = np.array([c.hx[q] for q in range(16)]); hy = np.array([c.hy[q] for q in range(16)]); hz = np.array([c.hz[q] for q in range(8)])
    L = 16 * 0.1 / 16                                                      # the cavity fixture's edge
    np.testing.assert_allclose([hx.sum(), hy.sum(), hz.sum()], [L, L, L], rtol=1e-13)
    np.testing.assert_allclose(hx[-1] / hx[0], 4.0, rtol=1e-12); np.testing.assert_allclose(hy[-1] / hy[0], 0.25, rtol=1e-12)
    np.testing.assert_allclose(hx[1:] / hx[:-1], 4.0 ** (1 / 15), rtol=1e-12)
    np.testing.assert_allclose(hz, L / 8, rtol=1e-13)                       # uniform, twice as thick as a cube would be
    fc.close()


@pytest.mark.gpu
def test_graded_case_directory_runs_like_the_hand_built_graded_case(prod, tmp_path):
    """a cavity case whose blockMeshDict grades the block towards two walls: the run started from the directory equals the run of the same
    fy_case_desc built by hand with the same cell sizes"""
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    text = (dst / "system/blockMeshDict").read_text()
    (dst / "system/blockMeshDict").write_text(text.replace("simpleGrading (1 1 1)", "simpleGrading (3 0.5 1)", 1))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    c = fc.case
    g = tuple(np.array([h[q] for q in range(n)]) for h, n in ((c.hx, c.nx), (c.hy, c.ny), (c.hz, c.nz)))
    s = prod.Solver(c)
    U0, p0 = fc.initial_fields()
    s.set("U", U0); s.set("p", p0)
    ref = prod.Solver(prod.make_case(0, c.nx, c.ny, c.nz, c.dx, c.dt, c.nu, rho_f=c.rho_fluid, rho_p=c.rho_particle, g=tuple(c.g),
                                     u_bc=list(c.u_bc), u_val=[tuple(c.u_value[q]) for q in range(6)], p_bc=list(c.p_bc), p_val=list(c.p_value),
                                     origin=tuple(c.origin), n_correctors=c.n_correctors, p_solver=c.p_solver, p_tol=c.p_tol, p_rel_tol=c.p_rel_tol,
                                     p_final_tol=c.p_final_tol, p_final_rel_tol=c.p_final_rel_tol, u_tol=c.u_tol, u_rel_tol=c.u_rel_tol, grading=g))
    for _ in range(5):
        s.step(); ref.step()
    for nm in ("U", "p"):
        np.testing.assert_array_equal(s.get(nm), ref.get(nm))
    assert np.abs(s.get("U")).max() > 0.05
    s.close(); ref.close(); fc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,solver", [("cavity_ico", 0), ("bed_pimple", 1)])
def test_run_from_case_directory_write_and_read_back(prod, tmp_path, name, solver):
    dst = tmp_path / name
    shutil.copytree(os.path.join(CASES, name), dst)
    fc = prod.FoamCase(dst, solver)
    s = prod.Solver(fc.case)
    U0, p0 = fc.initial_fields()
    s.set("U", U0); s.set("p", p0)
    s.hold_sources(True)
    # the same case built by hand
    c = fc.case
    ref = prod.Solver(prod.make_case(solver, c.nx, c.ny, c.nz, c.dx, c.dt, c.nu, rho_f=c.rho_fluid, rho_p=c.rho_particle, g=tuple(c.g),
                                     u_bc=list(c.u_bc), u_val=[tuple(c.u_value[q]) for q in range(6)], p_bc=list(c.p_bc), p_val=list(c.p_value),
                                     origin=tuple(c.origin), n_outer_correctors=c.n_outer_correctors, n_correctors=c.n_correctors,
                                     p_solver=c.p_solver, p_tol=c.p_tol, p_rel_tol=c.p_rel_tol, p_final_tol=c.p_final_tol,
                                     p_final_rel_tol=c.p_final_rel_tol, u_tol=c.u_tol, u_rel_tol=c.u_rel_tol))
    rec = None
    if solver == 1:
        rs = np.random.RandomState(4)
        rec = np.zeros((3000, 10))
        rec[:, 0:2] = -0.03 + 0.06 * rs.random_sample((3000, 2)); rec[:, 2] = 0.05 * rs.random_sample(3000); rec[:, 9] = 0.2 * c.dx
    for _ in range(fc.write_interval_steps):
        for sv in (s, ref):
            if rec is not None:
                sv.set_particles(rec)
            sv.step()
    # two runs of the coupled case differ at rounding level (atomic accumulation order in the particle phase)
    np.testing.assert_allclose(s.get("U"), ref.get("U"), rtol=1e-9, atol=1e-12)
    tname = "%g" % (fc.start_time + fc.write_interval_steps * fc.delta_t)
    alpha_held = s.get("alpha") if solver == 1 else None
    fc.write(s, tname)
    if solver == 1:
        assert alpha_held.min() < 1.0                                   # this step's void fraction, not the reset field
        assert np.all(ref.get("alpha") == 1.0)                          # (without hold_sources it is gone by now)
    # read the written time directory back as a start time
    text = (dst / "system/controlDict").read_text().replace("startTime       0;", "startTime       %s;" % tname)
    (dst / "system/controlDict").write_text(text)
    fc2 = prod.FoamCase(dst, solver)
    U1, p1 = fc2.initial_fields()
    np.testing.assert_array_equal(U1, s.get("U").reshape(-1, 3))        # %.17g round-trips every double
    np.testing.assert_array_equal(p1, s.get("p"))
    assert fc2.start_name == tname and list(fc2.case.u_bc) == list(fc.case.u_bc) and list(fc2.case.p_bc) == list(fc.case.p_bc)
    assert os.path.exists(dst / tname / fc.u_name)
    if solver == 1:
        a = (dst / tname / "alpha.water").read_text()
        assert "nonuniform List<scalar> %d" % fc.n_cells in a
    for o in (fc, fc2):
        o.close()
    s.close(); ref.close()


@pytest.mark.gpu
def test_foamYadeHip_executable_runs_the_cavity_case(prod, tmp_path):
    """the reference's main() as a thin executable over the C-ABI: same numbers as driving the library from Python"""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "yade-openfoam-coupling_amd", "bin", "foamYadeHip")
    if not os.path.exists(exe):
        pytest.fail("foamYadeHip has not been built: run __graft_entry__.build()")
    dst = tmp_path / "cavity"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    out = subprocess.run([exe, "-solver", "ico", "-case", str(dst)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Courant Number mean" in out.stdout and out.stdout.rstrip().endswith("End")
    assert sorted(d for d in os.listdir(dst) if d[0].isdigit()) == ["0", "0.025", "0.05"]
    fc = prod.FoamCase(dst, 0)
    s = prod.Solver(fc.case)
    for _ in range(10):
        s.step()
    (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startTime       0;", "startTime       0.05;"))
    fc2 = prod.FoamCase(dst, 0)
    U, p = fc2.initial_fields()
    np.testing.assert_array_equal(U, s.get("U").reshape(-1, 3))
    np.testing.assert_array_equal(p, s.get("p"))
    assert np.abs(U).max() > 0.1
    for o in (fc, fc2):
        o.close()
    s.close()


def test_dictionary_syntax_variants_are_understood(prod, tmp_path):
    """counted and compact lists, block comments inside lists, dimensioned scalars with and without the repeated name, a nonuniform
    start field, runTime write control -- the same case must come out"""
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    ref = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    bm = (dst / "system/blockMeshDict").read_text()
    bm = bm.replace("vertices\n(", "vertices 8\n(   /* eight corners,\n   counted list */").replace("(0 0 0) (1 0 0)", "(0 0 0)(1 0 0) /* no blank between tuples */")
    (dst / "system/blockMeshDict").write_text(bm)
    tp = (dst / "constant/transportProperties").read_text().replace("nu              nu [0 2 -1 0 0 0 0] 0.01;", "nu              [0 2 -1 0 0 0 0] 1e-2;")
    tp = tp.replace("partDensity     partDensity [1 -3 0 0 0 0 0] 2650;", "partDensity     2650.0;")
    (dst / "constant/transportProperties").write_text(tp)
    cd = (dst / "system/controlDict").read_text().replace("writeControl    timeStep;", "writeControl    runTime;").replace("writeInterval   5;", "writeInterval   0.025;")
    (dst / "system/controlDict").write_text(cd)
    n = ref.n_cells
    vals = np.arange(n, dtype=np.float64) * 1e-3
    ptxt = (dst / "0/p").read_text().replace("internalField   uniform 0;", "internalField   nonuniform List<scalar> %d\n(\n%s\n)\n;" % (n, "\n".join(repr(float(v)) for v in vals)))
    (dst / "0/p").write_text(ptxt)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    a, b = fc.case, ref.case
    for f in ("nx", "ny", "nz", "dx", "dt", "nu", "rho_particle", "rho_fluid", "n_correctors", "p_solver", "p_tol", "p_rel_tol"):
        assert getattr(a, f) == getattr(b, f), f
    assert fc.write_interval_steps == 5 and fc.patch_of_side == ref.patch_of_side
    U, p = fc.initial_fields()
    np.testing.assert_array_equal(p, vals)
    # a nonuniform list of the wrong length is an error, not a truncation
    (dst / "0/p").write_text(ptxt.replace("List<scalar> %d" % n, "List<scalar> %d" % (n - 1)))
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert "internalField" in str(e.value)
    for o in (fc, ref):
        o.close()


def test_gauss_upwind_is_read_as_the_upwind_scheme(prod, tmp_path):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "bed_pimple"), dst)
    f = dst / "system/fvSchemes"
    f.write_text(f.read_text().replace("div(phi,U)       Gauss linear;", "div(phi,U)       Gauss upwind;").replace("div(alphaPhic,Uc) Gauss linear;", "div(alphaPhic,Uc) Gauss upwind;"))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert fc.case.convection_scheme == prod.FY_CONVECTION_UPWIND
    fc.close()
    f.write_text(f.read_text().replace("div(phi,U)       Gauss upwind;", "div(phi,U)       Gauss linear;"))        # each executable reads its own term
    assert prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE).case.convection_scheme == prod.FY_CONVECTION_UPWIND
    assert prod.FoamCase(os.path.join(CASES, "bed_pimple"), prod.FY_SOLVER_PIMPLE).case.convection_scheme == prod.FY_CONVECTION_LINEAR
    f.write_text(f.read_text().replace("div(phi,U)       Gauss linear;", "div(phi,U)       Gauss linearUpwind grad(U);").replace("div(alphaPhic,Uc) Gauss upwind;", "div(alphaPhic,Uc) Gauss linearUpwind grad(Uc);"))
    assert prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE).case.convection_scheme == prod.FY_CONVECTION_LINEAR_UPWIND
    f.write_text(f.read_text().replace("Gauss linearUpwind grad(U);", "Gauss linearUpwindV grad(U);"))                    # a different limiter: refused
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert "divSchemes" in str(e.value)


def test_upwind_convection_next_to_the_linear_stress_term_is_read(prod, tmp_path):
    """a realistic pimpleFoamYade fvSchemes: upwinded convection plus the (always Gauss linear) explicit stress term of divDevRhoReff;
    only the convection keys select the scheme, and an upwinded non-convection entry is refused by name"""
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "bed_pimple"), dst)
    f = dst / "system/fvSchemes"
    stress = "    div(((alpha*nuEff)*dev2(T(grad(U))))) Gauss linear;\n"
    txt = f.read_text().replace("div(phi,U)       Gauss linear;", "div(phi,U)       Gauss upwind;").replace(
        "div(alphaPhic,Uc) Gauss linear;", "div(alphaPhic,Uc) Gauss upwind;\n" + stress)
    f.write_text(txt)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert fc.case.convection_scheme == prod.FY_CONVECTION_UPWIND
    fc.close()
    f.write_text(txt.replace(stress, stress.replace("Gauss linear", "Gauss upwind")))
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert "must be Gauss linear" in str(e.value)


def test_time_step_control_and_relaxation_factors_are_read(prod, tmp_path):
    """controlDict adjustTimeStep / maxCo / maxDeltaT (readTimeControls.H; only pimpleFoamYade's loop includes setDeltaT.H, pimpleFoamYade.C:62-64)
    and fvSolution relaxationFactors (UcEqn.relax() UcEqn.H:12, p.relax() pEqn.H:41)"""
    dst = tmp_path / "bed"
    shutil.copytree(os.path.join(CASES, "bed_pimple"), dst)
    base = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert base.case.adjust_time_step == 0 and base.case.u_relax == 0.0 and base.case.p_relax == 0.0      # no entries: relax() does nothing
    base.close()
    cd = dst / "system/controlDict"
    cd.write_text(cd.read_text().rstrip() + "\nadjustTimeStep  yes;\nmaxCo           0.4;\nmaxDeltaT       0.001;\n")
    with pytest.raises(prod.FoamYadeError) as e:          # output times that would cut the time step short are not implemented: said so
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert "writeControl" in str(e.value)
    cd.write_text(cd.read_text().replace("adjustableRunTime", "timeStep").replace("writeInterval   0.001;", "writeInterval   5;"))
    fs = dst / "system/fvSolution"
    fs.write_text(fs.read_text().rstrip() + '\nrelaxationFactors\n{\n    equations { "U.water" 0.7; "U.waterFinal" 1; }\n    fields { p 0.3; pFinal 1; }\n}\n')
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    c = fc.case
    assert c.adjust_time_step == 1 and c.max_co == 0.4 and c.max_delta_t == 0.001
    assert (c.u_relax, c.u_relax_final, c.p_relax, c.p_relax_final) == (0.7, 1.0, 0.3, 1.0)
    fc.close()
    fs.write_text(fs.read_text().replace('equations { "U.water" 0.7; "U.waterFinal" 1; }', 'equations { ".*" 1; }'))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert fc.case.u_relax == 1.0 and fc.case.u_relax_final == 1.0
    fc.close()
    # icoFoamYade's loop has no setDeltaT.H (icoFoamYade.C:65-70): the switch is ignored there, as the reference ignores it
    dst2 = tmp_path / "cav"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst2)
    cd2 = dst2 / "system/controlDict"
    cd2.write_text(cd2.read_text().rstrip() + "\nadjustTimeStep  yes;\nmaxCo           0.4;\n")
    fc = prod.FoamCase(dst2, prod.FY_SOLVER_ICO)
    assert fc.case.adjust_time_step == 0
    fc.close()


LES_PROPS = """FoamFile { version 2.0; format ascii; class dictionary; object turbulenceProperties.water; }
simulationType  LES;
LES
{
    LESModel        Smagorinsky;
    turbulence      on;
    printCoeffs     on;
    delta           cubeRootVol;
    SmagorinskyCoeffs { Ck 0.1; Ce 1.0; }
    cubeRootVolCoeffs { deltaCoeff 1.5; }
}
"""
NUT_FILE = """FoamFile { version 2.0; format ascii; class volScalarField; object nut.water; }
dimensions      [0 2 -1 0 0 0 0];
internalField   uniform 2e-6;
boundaryField
{
    bottom { type fixedValue; value uniform 1e-6; }
    top    { type zeroGradient; }
    walls  { type zeroGradient; }
}
"""


def les_case(tmp_path):
    dst = tmp_path / "bed_les"
    shutil.copytree(os.path.join(CASES, "bed_pimple"), dst)
    (dst / "constant/turbulenceProperties.water").write_text(LES_PROPS)
    (dst / "0/nut.water").write_text(NUT_FILE)
    return dst


def test_les_smagorinsky_case_is_read(prod, tmp_path):
    """constant/turbulenceProperties.<phase> + <start>/nut.<phase> (continuousPhaseTurbulence, pimpleFoamYade/createFields.H;
    DPMTurbulenceModels.C:67-77): laminar and LES Smagorinsky are read, the other two models are refused by name"""
    lam = prod.FoamCase(os.path.join(CASES, "bed_pimple"), prod.FY_SOLVER_PIMPLE)
    assert lam.case.turbulence_model == prod.TURBULENCE_LAMINAR            # no file: the laminar model
    lam.close()
    dst = les_case(tmp_path)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    c = fc.case
    assert c.turbulence_model == prod.TURBULENCE_SMAGORINSKY and (c.les_ck, c.les_ce, c.les_delta_coeff) == (0.1, 1.0, 1.5)
    assert c.nut_initial == 2e-6 and np.all(fc.initial_nut() == 2e-6)
    assert list(c.nut_bc) == [0, 0, 0, 0, 1, 0] and c.nut_value[ZMIN] == 1e-6
    fc.close()
    for old, new, needle in (("LESModel        Smagorinsky;", "LESModel        dynamicKEqn;", "dynamicKEqn"),
                             ("simulationType  LES;", "simulationType  DES;", "DES"),
                             ("delta           cubeRootVol;", "delta           vanDriest;", "cubeRootVol"),
                             ("turbulence      on;", "turbulence      off;", "turbulence off")):
        (dst / "constant/turbulenceProperties.water").write_text(LES_PROPS.replace(old, new))
        with pytest.raises(prod.FoamYadeError) as e:
            prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
        assert needle in str(e.value) and "turbulenceProperties" in str(e.value)
    (dst / "constant/turbulenceProperties.water").write_text(LES_PROPS)
    (dst / "0/nut.water").write_text(NUT_FILE.replace("top    { type zeroGradient; }", "top    { type nutkWallFunction; value uniform 0; }"))
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert "nutkWallFunction" in str(e.value) and "nut.water" in str(e.value)
    os.remove(dst / "0/nut.water")
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert "nut.water" in str(e.value)
    # a laminar file says what the missing file means
    (dst / "constant/turbulenceProperties.water").write_text("simulationType laminar;\n")
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert fc.case.turbulence_model == prod.TURBULENCE_LAMINAR
    fc.close()


@pytest.mark.gpu
def test_les_case_runs_and_writes_nut(prod, tmp_path):
    dst = les_case(tmp_path)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    s = prod.Solver(fc.case)
    U0, p0 = fc.initial_fields()
    s.set("U", U0); s.set("p", p0); s.set("nut", fc.initial_nut())
    for _ in range(3):
        s.step()
    nut = s.get("nut")
    assert nut.max() > 0 and not np.all(nut == 2e-6)                         # correct() replaced the file's values
    fc.write(s, "0.0006")
    text = (dst / "system/controlDict").read_text().replace("startTime       0;", "startTime       0.0006;")
    (dst / "system/controlDict").write_text(text)
    fc2 = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    np.testing.assert_array_equal(fc2.initial_nut(), nut)
    assert list(fc2.case.nut_bc) == list(fc.case.nut_bc) and fc2.case.nut_value[ZMIN] == 1e-6
    fc.close(); fc2.close(); s.close()


def test_les_keqn_case_is_read(prod, tmp_path):
    """LESModel kEqn: kEqnCoeffs, <start>/k.<phase>, the k convection scheme, solvers.k.<phase>, relaxationFactors equations k.<phase>"""
    dst = les_case(tmp_path)
    (dst / "constant/turbulenceProperties.water").write_text(LES_PROPS.replace("LESModel        Smagorinsky;", "LESModel        kEqn;").replace("SmagorinskyCoeffs", "kEqnCoeffs"))
    (dst / "0/k.water").write_text(NUT_FILE.replace("object nut.water", "object k.water").replace("[0 2 -1 0 0 0 0]", "[0 2 -2 0 0 0 0]")
                                   .replace("uniform 2e-6", "uniform 3e-4").replace("top    { type zeroGradient; }", "top    { type kqRWallFunction; value uniform 0; }"))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)                         # the case's "(U.water|k|epsilon)" pattern names k
    assert (fc.case.k_tol, fc.case.k_rel_tol) == (1e-5, 0.1) and fc.case.k_convection_scheme == 1      # no div entry for k: `default` is none -> upwind
    fc.close()
    sol = (dst / "system/fvSolution").read_text()
    (dst / "system/fvSolution").write_text(sol.replace('"(U.water|k|epsilon)"', "U.water"))
    with pytest.raises(prod.FoamYadeError) as e:                           # now nothing names k
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert "k.water" in str(e.value) and "fvSolution" in str(e.value)
    sol = (dst / "system/fvSolution").read_text()
    assert "solvers" in sol
    sol = sol.replace("solvers\n{", "solvers\n{\n    k.water { solver smoothSolver; smoother symGaussSeidel; tolerance 1e-7; relTol 0.01; maxIter 50; }", 1)
    assert "k.water {" in sol
    (dst / "system/fvSolution").write_text(sol)
    sch = (dst / "system/fvSchemes").read_text().replace("div(alphaPhic,Uc) Gauss linear;", "div(alphaPhic,Uc) Gauss linear;\n    div(alphaPhi.water,k.water) Gauss upwind;")
    (dst / "system/fvSchemes").write_text(sch)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    c = fc.case
    assert c.turbulence_model == prod.TURBULENCE_KEQN and (c.les_ck, c.les_ce) == (0.1, 1.0)
    assert c.k_initial == 3e-4 and np.all(fc.initial_k() == 3e-4) and list(c.k_bc) == [0, 0, 0, 0, 1, 0] and c.k_value[ZMIN] == 1e-6
    assert c.k_convection_scheme == 1 and (c.k_tol, c.k_rel_tol, c.k_max_iter) == (1e-7, 0.01, 50)
    assert c.convection_scheme == 0                                        # the momentum convection stays Gauss linear
    fc.close()


def test_ras_kepsilon_case_is_read(prod, tmp_path):
    """simulationType RAS, RASModel kEpsilon (DPMTurbulenceModels.C:70-71): coefficients, k / epsilon / nut files, schemes, solver entries;
    wall functions are refused by name"""
    dst = les_case(tmp_path)
    (dst / "constant/turbulenceProperties.water").write_text(
        "simulationType RAS;\nRAS { RASModel kEpsilon; turbulence on; kEpsilonCoeffs { Cmu 0.085; C1 1.4; C2 1.9; C3 -0.33; sigmak 1.1; sigmaEps 1.25; } }\n")
    kfile = NUT_FILE.replace("object nut.water", "object k.water").replace("[0 2 -1 0 0 0 0]", "[0 2 -2 0 0 0 0]").replace("uniform 2e-6", "uniform 3e-4")
    (dst / "0/k.water").write_text(kfile)
    efile = NUT_FILE.replace("object nut.water", "object epsilon.water").replace("[0 2 -1 0 0 0 0]", "[0 2 -3 0 0 0 0]").replace("uniform 2e-6", "uniform 5e-3").replace("uniform 1e-6", "uniform 4e-3")
    (dst / "0/epsilon.water").write_text(efile)
    sch = (dst / "system/fvSchemes").read_text().replace("div(alphaPhic,Uc) Gauss linear;", "div(alphaPhic,Uc) Gauss linear;\n    div(alphaPhi.water,k.water) Gauss upwind;\n    div(alphaPhi.water,epsilon.water) Gauss linear;")
    (dst / "system/fvSchemes").write_text(sch)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    c = fc.case
    assert c.turbulence_model == prod.TURBULENCE_KEPSILON
    assert (c.ras_cmu, c.ras_c1, c.ras_c2, c.ras_c3, c.ras_sigmak, c.ras_sigmaeps) == (0.085, 1.4, 1.9, -0.33, 1.1, 1.25)
    assert c.k_initial == 3e-4 and c.eps_initial == 5e-3 and np.all(fc.initial_epsilon() == 5e-3) and c.eps_value[ZMIN] == 4e-3 and list(c.eps_bc) == [0, 0, 0, 0, 1, 0]
    assert c.k_convection_scheme == 1 and c.eps_convection_scheme == 0
    assert (c.eps_tol, c.eps_rel_tol) == (1e-5, 0.1)                       # the "(U.water|k|epsilon)" entry
    fc.close()
    # wall functions on the walls: epsilonWallFunction, nutkWallFunction (kappa / E from the patch), kqRWallFunction
    (dst / "0/epsilon.water").write_text(efile.replace("walls  { type zeroGradient; }", "walls  { type epsilonWallFunction; value uniform 5e-3; }"))
    (dst / "0/nut.water").write_text(NUT_FILE.replace("walls  { type zeroGradient; }", "walls  { type nutkWallFunction; kappa 0.4; E 9.0; value uniform 0; }")
                                     .replace("top    { type zeroGradient; }", "top    { type calculated; value uniform 3e-6; }"))
    (dst / "0/k.water").write_text(kfile.replace("walls  { type zeroGradient; }", "walls  { type kqRWallFunction; value uniform 3e-4; }"))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    c = fc.case
    assert list(c.eps_bc) == [2, 2, 2, 2, 1, 0] and list(c.nut_bc) == [2, 2, 2, 2, 1, 3] and list(c.k_bc) == [0, 0, 0, 0, 1, 0] and c.nut_value[ZMAX] == 3e-6
    assert (c.wf_kappa, c.wf_E) == (0.4, 9.0)
    fc.close()
    (dst / "0/epsilon.water").write_text(efile.replace("walls  { type zeroGradient; }", "walls  { type epsilonLowReWallFunction; value uniform 5e-3; }"))
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert "epsilonLowReWallFunction" in str(e.value) and "epsilon.water" in str(e.value)


@pytest.mark.gpu
def test_foamYadeHip_executable_runs_a_kepsilon_case(prod, tmp_path):
    """the executable on the bed case with RAS kEpsilon and wall functions on the walls: start-time nut / k / epsilon handed to the solver, the three
    fields written with the time directories, same numbers as driving the library from Python"""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "yade-openfoam-coupling_amd", "bin", "foamYadeHip")
    if not os.path.exists(exe):
        pytest.fail("foamYadeHip has not been built: run __graft_entry__.build()")
    dst = les_case(tmp_path)
    (dst / "constant/turbulenceProperties.water").write_text("simulationType RAS;\nRAS { RASModel kEpsilon; turbulence on; }\n")
    base = NUT_FILE.replace("walls  { type zeroGradient; }", "walls  { type WALLTYPE; value uniform VAL; }")
    (dst / "0/nut.water").write_text(base.replace("WALLTYPE", "nutkWallFunction").replace("VAL", "0"))
    (dst / "0/k.water").write_text(base.replace("object nut.water", "object k.water").replace("[0 2 -1 0 0 0 0]", "[0 2 -2 0 0 0 0]").replace("uniform 2e-6", "uniform 3e-4")
                                   .replace("uniform 1e-6", "uniform 2e-4").replace("WALLTYPE", "kqRWallFunction").replace("VAL", "3e-4"))
    (dst / "0/epsilon.water").write_text(base.replace("object nut.water", "object epsilon.water").replace("[0 2 -1 0 0 0 0]", "[0 2 -3 0 0 0 0]").replace("uniform 2e-6", "uniform 5e-3")
                                         .replace("uniform 1e-6", "uniform 4e-3").replace("WALLTYPE", "epsilonWallFunction").replace("VAL", "5e-3"))
    out = subprocess.run([exe, "-solver", "pimple", "-case", str(dst)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    times = sorted((d for d in os.listdir(dst) if d[0].isdigit()), key=float)
    assert len(times) >= 2
    last = times[-1]
    for f in ("U.water", "p", "alpha.water", "nut.water", "k.water", "epsilon.water"):
        assert os.path.exists(dst / last / f), f
    fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    s = prod.Solver(fc.case)
    U0, p0 = fc.initial_fields()
    s.set("U", U0); s.set("p", p0); s.set("nut", fc.initial_nut()); s.set("k", fc.initial_k()); s.set("epsilon", fc.initial_epsilon())
    nsteps = int(round((float(last) - fc.start_time) / fc.delta_t))
    for _ in range(nsteps):
        s.step()
    (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startTime       0;", "startTime       %s;" % last))
    fc2 = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    np.testing.assert_array_equal(fc2.initial_k(), s.get("k"))
    np.testing.assert_array_equal(fc2.initial_epsilon(), s.get("epsilon"))
    np.testing.assert_array_equal(fc2.initial_nut(), s.get("nut"))
    assert not np.all(s.get("k") == 3e-4)
    for o in (fc, fc2):
        o.close()
    s.close()


def test_include_files_and_macros_are_expanded(prod, tmp_path):
    """#include "file" (relative to the including file), $name / ${name} in value position, `$dict;` merged in keyword position
    [OF-6 functionEntries::includeEntry, primitiveEntry::expandVariable] -- what the tutorial cases' 0/ files use"""
    dst = tmp_path / "cavity"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    (dst / "0/include").mkdir()
    (dst / "0/include/initialConditions").write_text("lidVelocity (0.75 0 0);\npressure 0;\nwall { type noSlip; }\n")
    U = (dst / "0/U").read_text()
    assert "internalField   uniform (0 0 0);" in U and "value           uniform (1 0 0);" in U
    U = U.replace("internalField   uniform (0 0 0);", '#include "include/initialConditions"\ninternalField   uniform (0 0 0);')
    U = U.replace("value           uniform (1 0 0);", "value           uniform $lidVelocity;")
    U = U.replace("type            noSlip;", "$wall;")
    (dst / "0/U").write_text(U)
    p = (dst / "0/p").read_text().replace("internalField   uniform 0;", '#include "include/initialConditions"\ninternalField   uniform ${pressure};')
    (dst / "0/p").write_text(p)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    c = fc.case
    assert list(c.u_value[YMAX]) == [0.75, 0.0, 0.0] and all(c.u_bc[s] == prod.FY_BC_U_FIXED_VALUE for s in range(6))
    U0, p0 = fc.initial_fields()
    assert not U0.any() and not p0.any()
    fc.close()
    (dst / "0/p").write_text(p.replace("${pressure}", "$nowhere"))
    with pytest.raises(prod.FoamYadeError) as e:
        prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert "$nowhere" in str(e.value)


# ---- blockMeshDict beyond one simply graded block ---------------------------------------------------------------------------------------
def _edit_block_mesh(tmp_path, old, new):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    f = dst / "system/blockMeshDict"
    text = f.read_text()
    assert old in text
    f.write_text(text.replace(old, new, 1))
    return dst


def test_multi_grading_sections_follow_lineDivide(prod, tmp_path):
    """simpleGrading with a list of (length fraction, cell fraction, expansion ratio) sections per direction [OF-6 blockMesh lineDivide]:
    x refined towards both walls, y towards the lid through unnormalised fractions and a negative ratio (= 1 / 3), z uniform"""
    dst = _edit_block_mesh(tmp_path, "simpleGrading (1 1 1)", "simpleGrading ( ((0.5 0.5 4) (0.5 0.5 0.25))  ((2 3 1) (2 1 -3))  1 )")
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    c = fc.case
    hx = np.ctypeslib.as_array(c.hx, (16,)).copy(); hy = np.ctypeslib.as_array(c.hy, (16,)).copy(); hz = np.ctypeslib.as_array(c.hz, (16,)).copy()
    np.testing.assert_allclose([hx.sum(), hy.sum(), hz.sum()], 0.1, rtol=1e-13)
    np.testing.assert_allclose(hx[:8].sum(), 0.05, rtol=1e-13)
    np.testing.assert_allclose(hx[7] / hx[0], 4.0, rtol=1e-12); np.testing.assert_allclose(hx[15] / hx[8], 0.25, rtol=1e-12)
    np.testing.assert_allclose(hx[1:8] / hx[:7], 4.0 ** (1 / 7), rtol=1e-12)
    np.testing.assert_allclose(hx, hx[::-1], rtol=1e-12)
    # y: half the length in 12 uniform cells, the other half in 4 cells shrinking by 3 overall
    np.testing.assert_allclose(hy[:12], 0.05 / 12, rtol=1e-12)
    np.testing.assert_allclose(hy[12:].sum(), 0.05, rtol=1e-13); np.testing.assert_allclose(hy[15] / hy[12], 1.0 / 3.0, rtol=1e-12)
    np.testing.assert_allclose(hz, 0.1 / 16, rtol=1e-13)
    fc.close()


MULTI_BLOCK = """
vertices
(
    (0 0 0) (0.4 0 0) (1 0 0)   (0 1 0) (0.4 1 0) (1 1 0)
    (0 0 0.5) (0.4 0 0.5) (1 0 0.5)   (0 1 0.5) (0.4 1 0.5) (1 1 0.5)
    (0 0 1) (0.4 0 1) (1 0 1)   (0 1 1) (0.4 1 1) (1 1 1)
);

blocks
(
    hex (0 1 4 3 6 7 10 9)     (6 16 8)  simpleGrading (2 1 1)
    hex (1 2 5 4 7 8 11 10)    (10 16 8) simpleGrading (1 1 1)
    hex (6 7 10 9 12 13 16 15) lower (6 16 8)  simpleGrading (2 1 0.5)
    hex (7 8 11 10 13 14 17 16) (10 16 8) simpleGrading (1 1 0.5)
);

edges ( );

boundary
(
    movingWall
    {
        type wall;
        faces ( (3 9 10 4) (4 10 11 5) (9 15 16 10) (10 16 17 11) );
    }
    fixedWalls
    {
        type wall;
        faces
        (
            (0 6 9 3) (6 12 15 9)
            (2 5 11 8) (8 11 17 14)
            (0 1 7 6) (1 2 8 7) (6 7 13 12) (7 8 14 13)
            (0 3 4 1) (1 4 5 2)
            (12 13 16 15) (13 14 17 16)
        );
    }
);
"""


def _multi_block_case(tmp_path, text=MULTI_BLOCK):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    f = dst / "system/blockMeshDict"
    head = f.read_text()
    f.write_text(head[:head.index("vertices")] + text + "\nmergePatchPairs ( );\n")
    return dst


def test_blocks_that_tile_a_box_are_merged_into_one_rectilinear_block(prod, tmp_path):
    """2 x 1 x 2 hex blocks with different cell counts and gradings, conforming along the faces they share"""
    fc = prod.FoamCase(_multi_block_case(tmp_path), prod.FY_SOLVER_ICO)
    c = fc.case
    assert (c.nx, c.ny, c.nz) == (16, 16, 16) and list(c.origin) == [0.0, 0.0, 0.0]
    hx = np.ctypeslib.as_array(c.hx, (16,)).copy(); hy = np.ctypeslib.as_array(c.hy, (16,)).copy(); hz = np.ctypeslib.as_array(c.hz, (16,)).copy()
    np.testing.assert_allclose(hx[:6].sum(), 0.04, rtol=1e-13); np.testing.assert_allclose(hx[5] / hx[0], 2.0, rtol=1e-12)
    np.testing.assert_allclose(hx[6:], 0.06 / 10, rtol=1e-13)
    np.testing.assert_allclose(hy, 0.1 / 16, rtol=1e-13)
    np.testing.assert_allclose(hz[:8], 0.05 / 8, rtol=1e-13); np.testing.assert_allclose(hz[15] / hz[8], 0.5, rtol=1e-12)
    assert fc.patch_of_side[YMAX] == "movingWall" and all(fc.patch_of_side[s] == "fixedWalls" for s in (XMIN, XMAX, YMIN, ZMIN, ZMAX))
    fc.close()


@pytest.mark.parametrize("old,new,needle", [
    ("hex (7 8 11 10 13 14 17 16) (10 16 8) simpleGrading (1 1 0.5)", "", "tensor product"),
    ("(10 16 8) simpleGrading (1 1 0.5)", "(10 16 8) simpleGrading (1 1 0.25)", "disagree"),
    ("(10 16 8) simpleGrading (1 1 1)", "(12 16 8) simpleGrading (1 1 1)", "disagree"),
    ("(10 16 17 11) );\n    }", ");\n    }\n    inlet { type patch; faces ( (10 16 17 11) ); }", "shared by the patches"),
    ("(13 14 17 16)\n", "\n", "only in part"),
    ("(12 13 16 15) (13 14 17 16)", "(12 13 16 15) (13 14 17 16) (13 14 17 16)", "more than once"),
])
def test_block_arrangements_outside_the_tensor_product_are_refused(prod, tmp_path, old, new, needle):
    assert old in MULTI_BLOCK
    with pytest.raises(prod.FoamYadeError, match=needle):
        prod.FoamCase(_multi_block_case(tmp_path, MULTI_BLOCK.replace(old, new, 1)), prod.FY_SOLVER_ICO)


@pytest.mark.gpu
def test_a_box_cut_into_four_uniform_blocks_runs_like_the_one_block_cavity(prod, tmp_path):
    """the same 16^3 cavity described as 2 x 1 x 2 hex blocks of equal cubes (one with a cellZone name): read as the uniform block, the run is
    the one-block case's bit for bit; with graded blocks it is the hand-built graded case's"""
    uni = MULTI_BLOCK.replace("0.4", "0.5").replace("(6 16 8)", "(8 16 8)").replace("(10 16 8)", "(8 16 8)").replace("simpleGrading (2 1 1)", "simpleGrading (1 1 1)")
    uni = uni.replace("simpleGrading (2 1 0.5)", "simpleGrading (1 1 1)").replace("simpleGrading (1 1 0.5)", "simpleGrading (1 1 1)")
    runs = []
    for k, dst in enumerate((_multi_block_case(tmp_path / "a", uni), os.path.join(CASES, "cavity_ico"))):
        fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
        assert not fc.case.hx and (fc.case.nx, fc.case.ny, fc.case.nz) == (16, 16, 16)
        s = prod.Solver(fc.case)
        U0, p0 = fc.initial_fields()
        s.set("U", U0); s.set("p", p0)
        for _ in range(5):
            s.step()
        runs.append((s.get("U"), s.get("p")))
        s.close(); fc.close()
    np.testing.assert_array_equal(runs[0][0], runs[1][0]); np.testing.assert_array_equal(runs[0][1], runs[1][1])
    assert np.abs(runs[0][0]).max() > 0.05
    # graded blocks
    fc = prod.FoamCase(_multi_block_case(tmp_path / "b"), prod.FY_SOLVER_ICO)
    c = fc.case
    g = tuple(np.array([h[q] for q in range(n)]) for h, n in ((c.hx, c.nx), (c.hy, c.ny), (c.hz, c.nz)))
    s = prod.Solver(c)
    ref = prod.Solver(prod.make_case(0, c.nx, c.ny, c.nz, c.dx, c.dt, c.nu, rho_f=c.rho_fluid, rho_p=c.rho_particle, g=tuple(c.g),
                                     u_bc=list(c.u_bc), u_val=[tuple(c.u_value[q]) for q in range(6)], p_bc=list(c.p_bc), p_val=list(c.p_value),
                                     origin=tuple(c.origin), n_correctors=c.n_correctors, p_solver=c.p_solver, p_tol=c.p_tol, p_rel_tol=c.p_rel_tol,
                                     p_final_tol=c.p_final_tol, p_final_rel_tol=c.p_final_rel_tol, u_tol=c.u_tol, u_rel_tol=c.u_rel_tol, grading=g))
    for _ in range(5):
        s.step(); ref.step()
    np.testing.assert_array_equal(s.get("U"), ref.get("U"))
    assert np.abs(s.get("U")).max() > 0.05
    s.close(); ref.close(); fc.close()


# ---- controlDict's input / output settings -----------------------------------------------------------------------------------------------
def _binary_field_file(cls, obj, dims, values, patches):
    """an OpenFOAM field file in the binary stream format (what `writeFormat binary` produces): the list's doubles between the parentheses"""
    v = np.ascontiguousarray(values, dtype="<f8")
    ty = "vector" if v.ndim == 2 else "scalar"
    head = ("FoamFile\n{\n    version     2.0;\n    format      binary;\n    arch        \"LSB;label=32;scalar=64\";\n    class       %s;\n"
            "    location    \"0\";\n    object      %s;\n}\n\ndimensions      %s;\n\ninternalField   nonuniform List<%s> %d\n(" % (cls, obj, dims, ty, v.shape[0]))
    tail = ")\n;\n\nboundaryField\n{\n" + "".join("    %s\n    {\n%s    }\n" % (n, t) for n, t in patches) + "}\n"
    return head.encode() + v.tobytes() + tail.encode()


def test_binary_field_files_are_read(prod, tmp_path):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    rs = np.random.RandomState(2)
    U = rs.standard_normal((4096, 3)); p = rs.standard_normal(4096)
    U[7] = (np.frombuffer(b");\n}\n()(", dtype="<f8")[0], 1e-300, -0.0)            # bytes that would end the list, were they read as text
    (dst / "0/U").write_bytes(_binary_field_file("volVectorField", "U", "[0 1 -1 0 0 0 0]", U,
                                                 [("movingWall", "        type            fixedValue;\n        value           uniform (1 0 0);\n"),
                                                  ("fixedWalls", "        type            noSlip;\n")]))
    (dst / "0/p").write_bytes(_binary_field_file("volScalarField", "p", "[0 2 -2 0 0 0 0]", p,
                                                 [("movingWall", "        type            zeroGradient;\n"), ("fixedWalls", "        type            zeroGradient;\n")]))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    U0, p0 = fc.initial_fields()
    np.testing.assert_array_equal(U0, U); np.testing.assert_array_equal(p0, p)
    assert list(fc.case.u_value[YMAX]) == [1.0, 0.0, 0.0]
    fc.close()
    # a list that is shorter than it says
    data = (dst / "0/p").read_bytes()
    (dst / "0/p").write_bytes(data[:600])
    with pytest.raises(prod.FoamYadeError, match="past the end"):
        prod.FoamCase(dst, prod.FY_SOLVER_ICO)


@pytest.mark.parametrize("start_from,expect", [("latestTime", "0.02"), ("firstTime", "0"), ("startTime", "0")])
def test_start_from_picks_the_time_directory(prod, tmp_path, start_from, expect):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    for name, val in (("0.02", 0.25), ("0.01", 0.125)):
        shutil.copytree(dst / "0", dst / name)
        pf = dst / name / "p"
        pf.write_text(pf.read_text().replace("internalField   uniform 0;", "internalField   uniform %g;" % val))
    os.makedirs(dst / "postProcessing" / "7")                                           # (not a time directory of the case)
    cd = dst / "system/controlDict"
    cd.write_text(cd.read_text().replace("startFrom       startTime;", "startFrom       %s;" % start_from))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert fc.start_name == expect
    _, p0 = fc.initial_fields()
    assert np.all(p0 == {"0.02": 0.25, "0": 0.0}[expect])
    fc.close()


@pytest.mark.parametrize("key,val,needle", [("writeFormat", "xml", "writeFormat"), ("writeCompression", "on", "writeCompression"), ("writePrecision", "0", "writePrecision"),
                                            ("writeControl", "clockTime", "writeControl")])
def test_output_settings_outside_the_subset_are_refused(prod, tmp_path, key, val, needle):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    cd = dst / "system/controlDict"
    text = cd.read_text()
    import re
    text, k = re.subn(r"(?m)^%s\s+[^;]*;" % key, "%s %s;" % (key, val), text)
    if not k:
        text += "\n%s %s;\n" % (key, val)
    cd.write_text(text)
    with pytest.raises(prod.FoamYadeError, match=needle):
        prod.FoamCase(dst, prod.FY_SOLVER_ICO)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["binary", "ascii6"])
def test_write_format_precision_and_purge(prod, tmp_path, fmt):
    """writeFormat binary: the time directory restarts the run exactly (startFrom latestTime reads it back); writePrecision 6: six digits in
    the file; purgeWrite 2: only the two newest time directories this run wrote are kept"""
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    cd = dst / "system/controlDict"
    extra = "\nwriteFormat binary;\npurgeWrite 2;\n" if fmt == "binary" else "\nwriteFormat ascii;\nwritePrecision 6;\npurgeWrite 2;\n"
    import re
    text = cd.read_text()
    for key in ("writeFormat", "writePrecision", "purgeWrite"):
        text = re.sub(r"(?m)^%s\s+[^;]*;\n" % key, "", text)
    cd.write_text(text.replace("startFrom       startTime;", "startFrom       latestTime;") + extra)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert fc.start_name == "0"
    s = prod.Solver(fc.case)
    U0, p0 = fc.initial_fields()
    s.set("U", U0); s.set("p", p0)
    for k in range(1, 5):
        s.step()
        fc.write(s, "%g" % (0.005 * k))
    assert sorted(d for d in os.listdir(dst) if d[0].isdigit()) == ["0", "0.015", "0.02"]
    U, p = s.get("U"), s.get("p")
    raw = (dst / "0.02" / "U").read_bytes()
    fc2 = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert fc2.start_name == "0.02"
    U2, p2 = fc2.initial_fields()
    if fmt == "binary":
        assert b"format      binary;" in raw and U.astype("<f8").tobytes() in raw
        np.testing.assert_array_equal(U2.reshape(U.shape), U); np.testing.assert_array_equal(p2, p)
    else:
        assert b"format      ascii;" in raw
        np.testing.assert_allclose(U2.reshape(U.shape), U, rtol=5e-6, atol=1e-300)
        assert np.abs(U2.reshape(U.shape) - U).max() > 0 and max(len(w) for w in raw.split(b"boundaryField")[0].split()[-300:]) <= 14
    s.close(); fc.close(); fc2.close()


@pytest.mark.parametrize("text,scheme,k", [("Gauss limitedLinear 0.5", 3, 0.5), ("bounded Gauss vanLeer", 4, 1.0), ("Gauss MUSCL", 5, 1.0), ("Gauss Minmod", 6, 1.0),
                                           ("Gauss SuperBee", 7, 1.0), ("Gauss QUICK", 8, 1.0), ("bounded Gauss upwind", 1, 1.0), ("Gauss linearUpwind grad(U)", 2, 1.0)])
def test_convection_schemes_are_read_by_name(prod, tmp_path, text, scheme, k):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    f = dst / "system/fvSchemes"
    t = f.read_text()
    assert "div(phi,U)       Gauss linear;" in t
    f.write_text(t.replace("div(phi,U)       Gauss linear;", "div(phi,U)       %s;" % text))
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert fc.case.convection_scheme == scheme and fc.case.convection_limiter_k == k
    fc.close()


# ---- decomposed cases (processorN directories) ---------------------------------------------------------------------------------------------
def decompose_case(dst, n_proc, fields=("U.water", "p"), real_output=False):
    """what `decomposePar` with simpleCoeffs (1 1 n) leaves for a one-block case, as far as the field files go: processorR/<start>/<field> with the
    R-th z-slab's cells (global order) and the processor patches next to the case's own.  Test infrastructure (no OpenFOAM here to run the tool)"""
    import re
    for r in range(n_proc):
        pdir = dst / ("processor%d" % r) / "0"
        os.makedirs(pdir)
        for fname in os.listdir(dst / "0"):
            text = (dst / "0" / fname).read_text()
            m = re.search(r"internalField\s+nonuniform\s+List<(\w+)>\s+(\d+)\s*\(", text)
            if m:
                n = int(m.group(2)); per = n // n_proc
                body_start = m.end()
                body_end = text.index("\n)", body_start)
                rows = [ln for ln in text[body_start:body_end].split("\n") if ln.strip()]
                assert len(rows) == n
                text = text[:m.start()] + "internalField   nonuniform List<%s> %d\n(\n" % (m.group(1), per) + "\n".join(rows[r * per:(r + 1) * per]) + text[body_end:]
            # decomposePar keeps a patch that has no face on this processor -- the block's z sides on the ranks that do not touch them -- and writes
            # its `value` as an empty list [OF-6 behaviour, recalled]
            if real_output:
                for patch, touches in (("bottom", r == 0), ("top", r == n_proc - 1)):
                    if not touches:
                        text = re.sub(r"(%s\s*\{[^}]*?)value\s+uniform\s+(\([^)]*\)|[^;]+);" % patch,
                                      lambda mm: mm.group(1) + "value nonuniform List<%s> 0();" % ("vector" if mm.group(2).startswith("(") else "scalar"), text)
            procs = ""
            for nb in (r - 1, r + 1):
                if 0 <= nb < n_proc:
                    procs += "    procBoundary%dto%d\n    {\n        type            processor;\n        value           uniform %s;\n    }\n" % (
                        r, nb, "(0 0 0)" if "Vector" in text.split("class", 1)[1].split(";", 1)[0] else "0")
            k = text.rindex("}")
            (pdir / fname).write_text(text[:k] + procs + text[k:])


def test_processor_directories_are_read_and_written(prod, tmp_path):
    """fy_foam_case_open_processor: mesh, controls and schemes from the case, the fields of processorR (its slab's cells; uniform or a nonuniform
    list of the slab's length), the processor patches kept and written back; a processor directory of the wrong size is refused"""
    dst = tmp_path / "bed"
    shutil.copytree(os.path.join(CASES, "bed_pimple"), dst)
    whole = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    n = whole.n_cells
    rs = np.random.RandomState(5)
    p_all = rs.standard_normal(n)
    ptext = (dst / "0/p").read_text()
    import re
    ptext = re.sub(r"internalField\s+uniform\s+[^;]+;", "internalField   nonuniform List<scalar> %d\n(\n%s\n)\n;" % (n, "\n".join(repr(float(v)) for v in p_all)), ptext)
    (dst / "0/p").write_text(ptext)
    decompose_case(dst, 3)
    for r in range(3):
        fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE, processor=(r, 3))
        assert fc.n_cells == n and fc.field_cells == n // 3 and fc.field_offset == r * (n // 3)
        assert list(fc.case.u_bc) == list(whole.case.u_bc) and fc.case.nz == whole.case.nz
        U, p = fc.initial_fields()
        np.testing.assert_array_equal(p, p_all[r * (n // 3):(r + 1) * (n // 3)])
        fc.close()
    with pytest.raises(prod.FoamYadeError, match="more than 2 parts"):
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE, processor=(0, 2))
    with pytest.raises(prod.FoamYadeError, match="internalField"):
        prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE, processor=(0, 6))      # (processor3 .. 5 are missing too, but rank 0's list has the wrong length first)
    whole.close()


def test_processor_directories_shaped_like_decomposePar_output(prod, tmp_path):
    """(advisor, round 3) a patch without faces on a processor carries `value nonuniform List<vector> 0()`: the ranks that do not touch the inlet read
    their directory all the same; and every processor patch this library writes has a `value` entry (processorFvPatchField reads one: a file
    without it would be refused by reconstructPar) -- also alpha's, whose start time had no file"""
    dst = tmp_path / "bed"
    shutil.copytree(os.path.join(CASES, "bed_pimple"), dst)
    decompose_case(dst, 3, real_output=True)
    assert "nonuniform List<vector> 0()" in (dst / "processor1/0/U.water").read_text()
    whole = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE)
    for r in range(3):
        fc = prod.FoamCase(dst, prod.FY_SOLVER_PIMPLE, processor=(r, 3))
        assert list(fc.case.u_bc) == list(whole.case.u_bc) and list(fc.case.p_bc) == list(whole.case.p_bc)
        if r == 0:
            assert tuple(fc.case.u_value[4]) == tuple(whole.case.u_value[4])          # (the rank that holds the inlet's faces knows its value)
        nloc = fc.field_cells
        fc.write_fields("0.5", np.zeros((nloc, 3)), np.zeros(nloc), alpha=np.ones(nloc))
        for fname in ("U.water", "p", "alpha.water"):
            text = (dst / ("processor%d" % r) / "0.5" / fname).read_text()
            import re
            for m in re.finditer(r"procBoundary\w+\s*\{([^}]*)\}", text):
                assert "type" in m.group(1) and "processor" in m.group(1) and "value" in m.group(1), (fname, m.group(0))
            assert text.count("procBoundary") == (1 if r in (0, 2) else 2), fname
        fc.close()
    whole.close()


# ---- constant/polyMesh (what the reference's solvers read; written here the way blockMesh numbers things) --------------------------------------
def write_poly_mesh(dst, xs, ys, zs, patches, split_x=None):
    """points / faces / owner / neighbour / boundary of a rectilinear box with node coordinates xs, ys, zs.  patches = [(name, type, [sides])], sides
    0..5 = x- x+ y- y+ z- z+.  split_x = m: the cells are numbered as blockMesh numbers TWO blocks joined at x index m (block by block), otherwise as
    one block (i fastest).  Test infrastructure: there is no blockMesh here to write the files"""
    nx, ny, nz = len(xs) - 1, len(ys) - 1, len(zs) - 1
    pid = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)

    def cid(i, j, k):
        if split_x is None:
            return i + nx * (j + ny * k)
        m = split_x
        return i + m * (j + ny * k) if i < m else m * ny * nz + (i - m) + (nx - m) * (j + ny * k)
    quad = {0: lambda i, j, k: (pid(i, j, k), pid(i, j, k + 1), pid(i, j + 1, k + 1), pid(i, j + 1, k)),
            1: lambda i, j, k: (pid(i, j, k), pid(i + 1, j, k), pid(i + 1, j, k + 1), pid(i, j, k + 1)),
            2: lambda i, j, k: (pid(i, j, k), pid(i, j + 1, k), pid(i + 1, j + 1, k), pid(i + 1, j, k))}
    internal = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                c = cid(i, j, k)
                if i + 1 < nx: internal.append((min(c, cid(i + 1, j, k)), max(c, cid(i + 1, j, k)), quad[0](i + 1, j, k)))
                if j + 1 < ny: internal.append((min(c, cid(i, j + 1, k)), max(c, cid(i, j + 1, k)), quad[1](i, j + 1, k)))
                if k + 1 < nz: internal.append((min(c, cid(i, j, k + 1)), max(c, cid(i, j, k + 1)), quad[2](i, j, k + 1)))
    internal.sort(key=lambda t: (t[0], t[1]))
    faces = [t[2] for t in internal]; owner = [t[0] for t in internal]; neigh = [t[1] for t in internal]
    btext = []
    for name, ty, sides in patches:
        start = len(faces)
        for sd in sides:
            a, hi = sd // 2, sd % 2
            rng = [range(nx), range(ny), range(nz)]
            rng[a] = [([nx, ny, nz][a] if hi else 0)]
            for k in rng[2]:
                for j in rng[1]:
                    for i in rng[0]:
                        faces.append(quad[a](i, j, k))
                        ci, cj, ck = (i - (a == 0 and hi), j - (a == 1 and hi), k - (a == 2 and hi))
                        owner.append(cid(ci, cj, ck))
        btext.append("    %s\n    {\n        type            %s;\n        nFaces          %d;\n        startFace       %d;\n    }\n" % (name, ty, len(faces) - start, start))
    pm = dst / "constant" / "polyMesh"
    os.makedirs(pm, exist_ok=True)
    head = lambda cls, obj: "FoamFile\n{\n    version     2.0;\n    format      ascii;\n    class       %s;\n    location    \"constant/polyMesh\";\n    object      %s;\n}\n\n" % (cls, obj)
    (pm / "points").write_text(head("vectorField", "points") + "%d\n(\n" % ((nx + 1) * (ny + 1) * (nz + 1)) +
                               "".join("(%r %r %r)\n" % (float(xs[i]), float(ys[j]), float(zs[k])) for k in range(nz + 1) for j in range(ny + 1) for i in range(nx + 1)) + ")\n")
    (pm / "faces").write_text(head("faceList", "faces") + "%d\n(\n" % len(faces) + "".join("4(%d %d %d %d)\n" % f for f in faces) + ")\n")
    (pm / "owner").write_text(head("labelList", "owner") + "%d\n(\n" % len(owner) + "".join("%d\n" % o for o in owner) + ")\n")
    (pm / "neighbour").write_text(head("labelList", "neighbour") + "%d\n(\n" % len(neigh) + "".join("%d\n" % o for o in neigh) + ")\n")
    (pm / "boundary").write_text(head("polyBoundaryMesh", "boundary") + "%d\n(\n" % len(patches) + "".join(btext) + ")\n")
    return cid


CAVITY_PATCHES = [("movingWall", "wall", [3]), ("fixedWalls", "wall", [0, 1, 2, 4, 5])]


def test_polyMesh_is_read_where_the_case_has_one(prod, tmp_path):
    """constant/polyMesh takes precedence over blockMeshDict (it is what icoFoamYade's createMesh.H reads): a graded one-block mesh gives the block the
    dictionary gives; with blockMeshDict removed the case still opens; a mesh that is no lattice is refused"""
    dst = _edit_block_mesh(tmp_path, "simpleGrading (1 1 1)", "simpleGrading (3 0.5 1)")
    ref = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    c = ref.case
    g = [np.array([h[q] for q in range(n)]) for h, n in ((c.hx, c.nx), (c.hy, c.ny), (c.hz, c.nz))]
    nodes = [np.concatenate([[0.0], np.cumsum(h)]) for h in g]
    write_poly_mesh(dst, nodes[0], nodes[1], nodes[2], CAVITY_PATCHES)
    os.remove(dst / "system/blockMeshDict")
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    d = fc.case
    assert (d.nx, d.ny, d.nz) == (c.nx, c.ny, c.nz) and fc.patch_of_side == ref.patch_of_side
    for a, hn in enumerate(("hx", "hy", "hz")):
        np.testing.assert_allclose([getattr(d, hn)[q] for q in range(16)], g[a], rtol=1e-12)
    assert list(d.u_bc) == list(c.u_bc) and list(d.u_value[YMAX]) == [1.0, 0.0, 0.0]
    fc.close(); ref.close()
    # one point moved off the lattice
    ptxt = (dst / "constant/polyMesh/points").read_text().split("\n")
    k = next(i for i, ln in enumerate(ptxt) if ln.startswith("(") and len(ln) > 2) + 40
    ptxt[k] = "(0.0123 0.0456 0.0789)"
    (dst / "constant/polyMesh/points").write_text("\n".join(ptxt))
    with pytest.raises(prod.FoamYadeError, match="not a rectilinear lattice"):
        prod.FoamCase(dst, prod.FY_SOLVER_ICO)


def test_polyMesh_numbered_block_by_block_is_mapped_to_the_lattice(prod, tmp_path):
    """two blocks joined in x: blockMesh numbers the cells of the first block, then those of the second; the field files follow the MESH's numbers and are
    read (and written) through the map"""
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    n = 16
    nodes = np.linspace(0.0, 0.1, n + 1)
    cid = write_poly_mesh(dst, nodes, nodes, nodes, CAVITY_PATCHES, split_x=6)
    vals = np.zeros(n ** 3)
    lattice = np.zeros(n ** 3)
    for k in range(n):
        for j in range(n):
            for i in range(n):
                vals[cid(i, j, k)] = 1000.0 * i + 10.0 * j + 0.1 * k           # (the value names the cell's place)
                lattice[i + n * (j + n * k)] = 1000.0 * i + 10.0 * j + 0.1 * k
    ptext = (dst / "0/p").read_text().replace("internalField   uniform 0;", "internalField   nonuniform List<scalar> %d\n(\n%s\n)\n;" % (n ** 3, "\n".join(repr(float(v)) for v in vals)))
    (dst / "0/p").write_text(ptext)
    fc = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    assert (fc.case.nx, fc.case.ny, fc.case.nz) == (n, n, n) and not fc.case.hx
    U, p = fc.initial_fields()
    np.testing.assert_array_equal(p, lattice)
    fc.close()


@pytest.mark.gpu
def test_case_with_a_block_by_block_polyMesh_runs_and_writes_in_the_meshs_numbering(prod, tmp_path):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    n = 16
    nodes = np.linspace(0.0, 0.1, n + 1)
    cid = write_poly_mesh(dst, nodes, nodes, nodes, CAVITY_PATCHES, split_x=6)
    perm = np.array([cid(i, j, k) for k in range(n) for j in range(n) for i in range(n)])      # lattice index -> the mesh's cell number
    runs = []
    for d in (dst, os.path.join(CASES, "cavity_ico")):
        fc = prod.FoamCase(d, prod.FY_SOLVER_ICO)
        s = prod.Solver(fc.case)
        U0, p0 = fc.initial_fields()
        s.set("U", U0); s.set("p", p0)
        for _ in range(5):
            s.step()
        runs.append((s.get("U").reshape(-1, 3), s.get("p")))
        if d is dst:
            fc.write(s, "0.025")
        s.close(); fc.close()
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    # the written file lists the cells in the mesh's order
    cd = dst / "system/controlDict"
    cd.write_text(cd.read_text().replace("startFrom       startTime;", "startFrom       latestTime;"))
    import re
    rows = re.search(r"internalField\s+nonuniform List<scalar> \d+\s*\(([^)]*)\)", (dst / "0.025" / "p").read_text()).group(1).split()
    np.testing.assert_array_equal(np.array([float(r) for r in rows])[perm], runs[0][1])
    fc2 = prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    U2, p2 = fc2.initial_fields()
    np.testing.assert_array_equal(U2, runs[0][0]); np.testing.assert_array_equal(p2, runs[0][1])
    fc2.close()
