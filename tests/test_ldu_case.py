"""An icoFoamYade case directory whose constant/polyMesh is NOT a block (fy_foam_case_open_general -> fy_ldu_solver): what createMesh.H / createFields.H read
(icoFoamYade.C:42-44) and runTime.write() writes back (icoFoamYade.C:142).  The meshes are written by tests/poly_meshes.py (there is no blockMesh here)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import poly_meshes as pm

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = os.path.join(HERE, "golden", "cases")
CAVITY = [("movingWall", [3]), ("fixedWalls", [0, 1, 2, 4, 5])]


@pytest.fixture
def prod():
    from conftest import load_product
    return load_product()


def general_cavity(tmp_path, mesh, p_file=None):
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    os.remove(dst / "system/blockMeshDict")
    pm.write_poly_mesh_files(dst, mesh)
    return dst


def test_general_polyMesh_case_is_read(prod, tmp_path):
    """a sheared, randomly renumbered block: the block reader refuses it by name, the general reader returns the files' arrays, the patches in the boundary
    file's order with their conditions, and the controls of the dictionaries"""
    mesh = pm.hex_block(6, 5, 4, (0.1, 0.1, 0.1), pm.shear(0.3, 0.1, 0.2), patches=CAVITY, renumber_seed=3)
    dst = general_cavity(tmp_path, mesh)
    with pytest.raises(prod.FoamYadeError, match="not a rectilinear lattice"):
        prod.FoamCase(dst, prod.FY_SOLVER_ICO)
    fc = prod.GeneralFoamCase(dst)
    for k in ("face_offsets", "face_points", "owner", "neighbour", "patch_start", "patch_size"):
        np.testing.assert_array_equal(fc.mesh[k], mesh[k], err_msg=k)
    np.testing.assert_array_equal(fc.mesh["points"], mesh["points"])                 # (%r round-trips doubles)
    assert fc.mesh["n_cells"] == 120 and fc.n_cells == 120 and fc.patch_names == ["movingWall", "fixedWalls"]
    assert fc.u_bc == [prod.FY_BC_U_FIXED_VALUE] * 2 and fc.p_bc == [prod.FY_BC_P_ZERO_GRADIENT] * 2
    np.testing.assert_array_equal(fc.u_value, [[1, 0, 0], [0, 0, 0]])
    ref = prod.FoamCase(os.path.join(CASES, "cavity_ico"), prod.FY_SOLVER_ICO)
    lc, d = fc.ldu_case, ref.case
    assert (lc.dt, lc.nu, lc.rho_fluid, lc.rho_particle) == (d.dt, d.nu, d.rho_fluid, d.rho_particle)
    assert (lc.n_correctors, lc.n_non_orth_correctors, lc.p_ref_cell, lc.p_ref_value) == (d.n_correctors, d.n_non_orth_correctors, d.p_ref_cell, d.p_ref_value)
    assert (lc.p_tol, lc.p_rel_tol, lc.p_final_tol, lc.u_tol) == (d.p_tol, d.p_rel_tol, d.p_final_tol, d.u_tol)
    assert (fc.start_time, fc.end_time, fc.delta_t, fc.write_interval_steps) == (ref.start_time, ref.end_time, ref.delta_t, ref.write_interval_steps)
    U, p = fc.initial_fields()
    assert U.shape == (120, 3) and not U.any() and not p.any()
    ref.close()
    # runTime.write() from host arrays: the case's own patch entries come back, the values in the mesh's numbering
    Uw = np.arange(360.0).reshape(120, 3); pw = np.arange(120.0) * 0.5
    fc.write_fields("0.5", Uw, pw)
    text = (dst / "0.5/U").read_text()
    assert "movingWall" in text and "fixedWalls" in text and "uniform ( 1 0 0 )" in text and "noSlip" in text
    (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startFrom       startTime;", "startFrom       latestTime;"))
    fc.close()
    fc = prod.GeneralFoamCase(dst)
    U, p = fc.initial_fields()
    np.testing.assert_array_equal(U, Uw); np.testing.assert_array_equal(p, pw)
    assert fc.start_name == "0.5"
    fc.close()


@pytest.mark.parametrize("label64", [False, True])
def test_binary_polyMesh_is_read(prod, tmp_path, label64):
    """writeFormat binary: points as raw doubles, owner / neighbour as raw labels (32 or 64 bit by the header's arch), faces as a faceCompactList -- prisms, so that the
    face sizes differ; the arrays equal the ASCII reader's.  A lattice written this way opens through the block reader too"""
    mesh = pm.prism_block(3, 3, 2, (0.1, 0.1, 0.1), pm.wavy(0.004, (0.1, 0.1, 0.1)))
    dst = general_cavity(tmp_path, mesh)
    for nm in ("U", "p"):
        t = (dst / "0" / nm).read_text()
        body = "".join("    %s { type %s; }\n" % (s_, ("noSlip" if nm == "U" else "zeroGradient")) for s_ in pm.SIDES)
        (dst / "0" / nm).write_text(t[:t.index("boundaryField")] + "boundaryField\n{\n" + body + "}\n")
    pm.write_poly_mesh_files(dst, mesh, binary=True, label64=label64)
    assert b"faceCompactList" in (dst / "constant/polyMesh/faces").read_bytes()
    fc = prod.GeneralFoamCase(dst)
    for k in ("face_offsets", "face_points", "owner", "neighbour", "patch_start", "patch_size"):
        np.testing.assert_array_equal(fc.mesh[k], mesh[k], err_msg=k)
    np.testing.assert_array_equal(fc.mesh["points"], mesh["points"])
    fc.close()
    blk = tmp_path / "blk"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), blk)
    os.remove(blk / "system/blockMeshDict")
    pm.write_poly_mesh_files(blk, pm.hex_block(16, 16, 16, (0.1, 0.1, 0.1), patches=CAVITY), binary=True, label64=label64)
    b = prod.FoamCase(blk, prod.FY_SOLVER_ICO)
    assert (b.case.nx, b.case.ny, b.case.nz) == (16, 16, 16) and abs(b.case.dx - 0.1 / 16) < 1e-15
    b.close()
    raw = (dst / "constant/polyMesh/owner").read_bytes()
    (dst / "constant/polyMesh/owner").write_bytes(raw[:-40])                 # a truncated list is an error, not a short mesh
    with pytest.raises(prod.FoamYadeError, match="runs past the end"):
        prod.GeneralFoamCase(dst)


def test_prism_polyMesh_case_is_read(prod, tmp_path):
    """triangular and quadrilateral faces in one faces file"""
    mesh = pm.prism_block(3, 3, 2, (0.1, 0.1, 0.1), pm.wavy(0.004, (0.1, 0.1, 0.1)))
    dst = general_cavity(tmp_path, mesh)
    for nm in ("U", "p"):       # six patches, one per side: the cavity's two entries are replaced
        t = (dst / "0" / nm).read_text()
        head = t[:t.index("boundaryField")]
        body = "".join("    %s { type %s; }\n" % (s, ("noSlip" if nm == "U" else "zeroGradient")) for s in pm.SIDES)
        if nm == "U":
            body = body.replace("ymax { type noSlip; }", "ymax { type fixedValue; value uniform (1 0 0); }")
        (dst / "0" / nm).write_text(head + "boundaryField\n{\n" + body + "}\n")
    fc = prod.GeneralFoamCase(dst)
    for k in ("face_offsets", "face_points", "owner", "neighbour", "patch_start", "patch_size"):
        np.testing.assert_array_equal(fc.mesh[k], mesh[k], err_msg=k)
    assert fc.patch_names == list(pm.SIDES) and fc.mesh["n_cells"] == 36
    np.testing.assert_array_equal(fc.u_value[3], [1, 0, 0])
    fc.close()


@pytest.mark.parametrize("entry,scheme,k", [("Gauss linear", 0, None), ("Gauss upwind", 1, None), ("Gauss linearUpwind grad(U)", 2, None),
                                            ("Gauss limitedLinear 0.4", 3, 0.4), ("Gauss vanLeer", 4, None), ("Gauss MUSCL", 5, None),
                                            ("Gauss Minmod", 6, None), ("Gauss SuperBee", 7, None), ("Gauss QUICK", 8, None)])
def test_general_case_convection_scheme_is_read(prod, tmp_path, entry, scheme, k):
    """divSchemes div(phi,U) of a general-mesh case: every scheme of the block solver's list arrives in fy_ldu_case (the limiter constant with limitedLinear)"""
    mesh = pm.hex_block(4, 4, 3, (0.1, 0.1, 0.1), pm.shear(0.2, 0.0, 0.1), patches=CAVITY)
    dst = general_cavity(tmp_path, mesh)
    f = dst / "system/fvSchemes"
    t = f.read_text()
    assert "div(phi,U)       Gauss linear" in t
    f.write_text(t.replace("div(phi,U)       Gauss linear", "div(phi,U)       " + entry))
    fc = prod.GeneralFoamCase(dst)
    assert fc.ldu_case.convection_scheme == scheme
    if k is not None:
        assert fc.ldu_case.convection_limiter_k == k
    fc.close()


def test_general_case_with_symmetry_patches_is_read(prod, tmp_path):
    """constant/polyMesh/boundary with a symmetryPlane and a symmetry patch: the fields' entries must carry the patch's own type (as fvPatchField::New insists);
    U arrives as FY_BC_U_SLIP, p as zeroGradient; an entry of another type on such a patch is refused by name"""
    mesh = pm.hex_block(5, 4, 3, (0.1, 0.1, 0.1), pm.shear(0.2, 0.0, 0.1), patches=[("movingWall", [3]), ("fixedWalls", [1, 2, 5]), ("mirror", [0]), ("mirror2", [4])])
    dst = general_cavity(tmp_path, mesh)
    b = dst / "constant/polyMesh/boundary"
    t = b.read_text()
    i, j = t.index("mirror"), t.index("mirror2")
    t = t[:i] + t[i:j].replace("type wall", "type symmetryPlane", 1).replace("type            wall", "type            symmetryPlane", 1) + t[j:].replace("type wall", "type symmetry", 1).replace("type            wall", "type            symmetry", 1)
    assert "symmetryPlane" in t and t.count("symmetry") == 2
    b.write_text(t)
    for nm in ("U", "p"):
        f = dst / "0" / nm
        ft = f.read_text()
        k = ft.rindex("}")
        f.write_text(ft[:k] + "    mirror { type symmetryPlane; }\n    mirror2 { type symmetry; }\n}\n")
    fc = prod.GeneralFoamCase(dst)
    assert fc.patch_names == ["movingWall", "fixedWalls", "mirror", "mirror2"]
    assert fc.u_bc == [prod.FY_BC_U_FIXED_VALUE, prod.FY_BC_U_FIXED_VALUE, prod.FY_BC_U_SLIP, prod.FY_BC_U_SLIP] and fc.p_bc == [prod.FY_BC_P_ZERO_GRADIENT] * 4
    fc.write_fields("0.5", np.zeros((60, 3)), np.zeros(60))
    wt = (dst / "0.5/U").read_text()
    assert "symmetryPlane" in wt and "mirror2" in wt
    fc.close()
    f = dst / "0/U"
    f.write_text(f.read_text().replace("mirror { type symmetryPlane; }", "mirror { type noSlip; }"))
    with pytest.raises(prod.FoamYadeError, match="is a symmetryPlane patch"):
        prod.GeneralFoamCase(dst)


def cyclic_case(tmp_path):
    mesh = pm.make_cyclic(pm.hex_block(6, 5, 4, (0.1, 0.1, 0.1), pm.wavy_periodic(0.002, (0.1, 0.1, 0.1)), patches=[("movingWall", [3]), ("fixedWalls", [2, 4, 5]), ("left", [0]), ("right", [1])]),
                          [(2, 3)])
    dst = general_cavity(tmp_path, mesh)
    b = dst / "constant/polyMesh/boundary"
    t = b.read_text()
    i, j = t.index("left"), t.index("right")
    cyc = lambda txt, nbr: txt.replace("type wall", "type cyclic; neighbourPatch %s; transform translational" % nbr, 1).replace("type            wall", "type            cyclic; neighbourPatch %s; transform translational" % nbr, 1)
    b.write_text(t[:i] + cyc(t[i:j], "right") + cyc(t[j:], "left"))
    assert b.read_text().count("cyclic") == 2
    for nm in ("U", "p"):
        f = dst / "0" / nm
        ft = f.read_text()
        f.write_text(ft[:ft.rindex("}")] + "    left { type cyclic; }\n    right { type cyclic; }\n}\n")
    return mesh, dst


def test_general_case_with_cyclic_patches_is_read(prod, tmp_path):
    """constant/polyMesh/boundary with a cyclic pair (neighbourPatch): the partner indices arrive in fy_poly_mesh.patch_neighbour, the fields' entries must be cyclic
    too and are written back as they were; a rotational transform or a partner that is no cyclic patch is refused by name"""
    mesh, dst = cyclic_case(tmp_path)
    fc = prod.GeneralFoamCase(dst)
    assert fc.patch_names == ["movingWall", "fixedWalls", "left", "right"]
    np.testing.assert_array_equal(fc.mesh["patch_neighbour"], [-1, -1, 3, 2])
    np.testing.assert_array_equal(fc.mesh["owner"], mesh["owner"])
    fc.write_fields("0.5", np.zeros((120, 3)), np.zeros(120))
    wt = (dst / "0.5/U").read_text()
    assert wt.count("cyclic") == 2
    fc.close()
    b = dst / "constant/polyMesh/boundary"
    t = b.read_text()
    b.write_text(t.replace("transform translational", "transform rotational"))
    with pytest.raises(prod.FoamYadeError, match="rotational"):
        prod.GeneralFoamCase(dst)
    b.write_text(t.replace("neighbourPatch right", "neighbourPatch fixedWalls"))
    with pytest.raises(prod.FoamYadeError, match="not another cyclic patch"):
        prod.GeneralFoamCase(dst)
    b.write_text(t)
    f = dst / "0/p"
    f.write_text(f.read_text().replace("left { type cyclic; }", "left { type zeroGradient; }"))
    with pytest.raises(prod.FoamYadeError, match="is a cyclic patch"):
        prod.GeneralFoamCase(dst)


@pytest.mark.gpu
def test_foamYadeHip_executable_runs_a_cyclic_case(prod, tmp_path):
    """foamYadeHip -solver ico on a case with a cyclic pair: the flow is carried round the period (the lid drags it along x; with walls there it would turn back)"""
    import re
    mesh, dst = cyclic_case(tmp_path)
    cd = (dst / "system/controlDict").read_text()
    cd = re.sub(r"endTime\s+[0-9.eE+-]+;", "endTime         0.05;", cd)
    cd = re.sub(r"writeInterval\s+[0-9.eE+-]+;", "writeInterval   10;", cd)
    (dst / "system/controlDict").write_text(cd)
    exe = os.path.join(os.path.dirname(prod.__file__), "bin", "foamYadeHip")
    out = subprocess.run([exe, "-solver", "ico", "-case", str(dst)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "general polyhedral mesh" in out.stdout and out.stdout.rstrip().endswith("End")
    (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startFrom       startTime;", "startFrom       latestTime;"))
    fc = prod.GeneralFoamCase(dst)
    U, p = fc.initial_fields()
    assert fc.start_time == pytest.approx(0.05) and U[:, 0].min() > 0 and U[:, 0].max() > 0.2          # every cell moves with the lid: nothing turns back at x = 0 / x = L
    assert "cyclic" in (dst / fc.start_name / "U").read_text()
    fc.close()


@pytest.mark.parametrize("edit,needle", [
    (("system/fvSchemes", "Gauss linear corrected", "Gauss linear uncorrected"), "must be 'corrected'"),
    (("system/fvSchemes", "default corrected", "default orthogonal"), "must be 'corrected'"),
    (("constant/polyMesh/boundary", "type            wall;", "type            cyclic;"), "cyclic"),
    (("0/U", "noSlip", "partialSlip"), "partialSlip"),
    (("0/p", "zeroGradient", "fixedFluxPressure"), "fixedFluxPressure"),
])
def test_what_the_general_solver_cannot_do_is_refused_by_name(prod, tmp_path, edit, needle):
    mesh = pm.hex_block(4, 4, 4, (0.1, 0.1, 0.1), pm.shear(0.2), patches=CAVITY)
    dst = general_cavity(tmp_path, mesh)
    f, old, new = edit
    t = (dst / f).read_text()
    assert old in t
    (dst / f).write_text(t.replace(old, new))
    with pytest.raises(prod.FoamYadeError, match=needle):
        prod.GeneralFoamCase(dst)


BED = [("bottom", [4]), ("top", [5]), ("walls", [0, 1, 2, 3])]


def general_bed(tmp_path, amp=0.15):
    """tests/golden/cases/bed_pimple with its block replaced by a wavy polyhedral mesh of the same box and patches"""
    dst = tmp_path / "bed"
    shutil.copytree(os.path.join(CASES, "bed_pimple"), dst)
    os.remove(dst / "system/blockMeshDict")
    L = (0.06, 0.06, 0.12)
    wav = pm.wavy(amp * 0.005, L)
    mesh = pm.hex_block(12, 12, 24, L, lambda P: wav(P) + np.array([-0.03, -0.03, 0.0]), patches=BED)
    pm.write_poly_mesh_files(dst, mesh, {"bottom": "patch", "top": "patch", "walls": "wall"})
    return dst, mesh


def test_general_pimple_case_is_read(prod, tmp_path):
    """pimpleFoamYade's dictionaries on a general mesh: PIMPLE controls, gravity, the phase's names, fixedFluxPressure patches; what the general solver does not carry
    (the transport-equation closures) is refused by name"""
    dst, mesh = general_bed(tmp_path)
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)
    ref = prod.FoamCase(os.path.join(CASES, "bed_pimple"), prod.FY_SOLVER_PIMPLE)
    lc, d = fc.ldu_case, ref.case
    assert lc.solver == prod.FY_SOLVER_PIMPLE and (lc.n_outer_correctors, lc.n_correctors, lc.n_non_orth_correctors) == (d.n_outer_correctors, d.n_correctors, d.n_non_orth_correctors)
    assert list(lc.g) == list(d.g) and (lc.u_relax, lc.u_relax_final, lc.p_relax, lc.p_relax_final) == (d.u_relax, d.u_relax_final, d.p_relax, d.p_relax_final)
    assert (lc.dt, lc.nu, lc.rho_fluid, lc.rho_particle, lc.p_solver) == (d.dt, d.nu, d.rho_fluid, d.rho_particle, d.p_solver)
    assert fc.patch_names == ["bottom", "top", "walls"] and fc.u_name == "U.water" and fc.phase == "water"
    assert fc.p_bc == [prod.FY_BC_P_FIXED_FLUX, prod.FY_BC_P_FIXED_VALUE, prod.FY_BC_P_FIXED_FLUX]
    assert fc.u_bc == [prod.FY_BC_U_FIXED_VALUE, prod.FY_BC_U_ZERO_GRADIENT, prod.FY_BC_U_FIXED_VALUE]
    np.testing.assert_array_equal(fc.u_value, [[0, 0, 0.02], [0, 0, 0], [0, 0, 0]])
    fc.close(); ref.close()
    tp = "FoamFile { version 2.0; format ascii; class dictionary; object turbulenceProperties.water; }\nsimulationType LES;\nLES { LESModel %s; delta cubeRootVol; turbulence on; cubeRootVolCoeffs { deltaCoeff 0.9; } }\n"
    (dst / "constant/turbulenceProperties.water").write_text(tp % "Smagorinsky")
    (dst / "0/nut.water").write_text((dst / "0/p").read_text().replace("object      p;", "object nut.water;").replace("[0 2 -2 0 0 0 0]", "[0 2 -1 0 0 0 0]").replace("internalField   uniform 0;", "internalField   uniform 2e-6;")
                                     .replace("bottom { type fixedFluxPressure; value uniform 0; }", "bottom { type fixedValue; value uniform 1e-6; }").replace("fixedFluxPressure; value uniform 0;", "zeroGradient;")
                                     .replace("top    { type fixedValue; value uniform 0; }", "top    { type calculated; value uniform 3e-6; }"))
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)             # LES Smagorinsky is carried: coefficients, delta, nut's file and patches
    lc = fc.ldu_case
    assert (lc.turbulence_model, lc.les_delta_coeff, lc.nut_initial) == (prod.TURBULENCE_SMAGORINSKY, 0.9, 2e-6)
    assert [lc.nut_bc[q] for q in range(3)] == [1, 1, 0] and [lc.nut_value[q] for q in range(3)] == [1e-6, 3e-6, 0.0]
    np.testing.assert_array_equal(fc.initial_nut(), 2e-6)
    fc.close()
    (dst / "constant/turbulenceProperties.water").write_text("simulationType RAS;\nRAS { RASModel kEpsilon; turbulence on; }\n")      # (kEqn: test_general_kEqn_case_runs_and_writes_k)
    (dst / "0/k.water").write_text((dst / "0/nut.water").read_text().replace("nut.water", "k.water"))
    nt = (dst / "0/nut.water").read_text()
    (dst / "0/nut.water").write_text(nt.replace("top    { type calculated; value uniform 3e-6; }", "top    { type nutkWallFunction; value uniform 0; }"))
    with pytest.raises(prod.FoamYadeError, match="nutkWallFunction"):      # (kEpsilon itself is carried: test_general_kEpsilon_case_runs; its wall functions need the block solver)
        prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)
    os.remove(dst / "0/k.water"); os.remove(dst / "0/nut.water")
    os.remove(dst / "constant/turbulenceProperties.water")
    cd = (dst / "system/controlDict").read_text()
    (dst / "system/controlDict").write_text(cd.replace("adjustTimeStep  no;", "adjustTimeStep  yes;\nmaxCo 0.5;\nmaxDeltaT 0.001;").replace("writeControl    adjustableRunTime;", "writeControl    timeStep;").replace("writeInterval   0.001;", "writeInterval   5;"))
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)             # setDeltaT.H's controls (pimpleFoamYade.C:62-64)
    assert (fc.ldu_case.adjust_time_step, fc.ldu_case.max_co, fc.ldu_case.max_delta_t) == (1, 0.5, 0.001)
    fc.close()


@pytest.mark.gpu
def test_foamYadeHip_executable_runs_a_general_pimple_case(prod, tmp_path):
    """foamYadeHip -solver pimple on the wavy bed: falls through to the general solver, runs controlDict's ten steps (inflow at the bottom, fixed pressure at the top),
    writes U.water, p and alpha.water with the case's patch entries; the library-driven run from the same directory gives the same fields"""
    dst, mesh = general_bed(tmp_path)
    exe = os.path.join(os.path.dirname(prod.__file__), "bin", "foamYadeHip")
    out = subprocess.run([exe, "-solver", "pimple", "-case", str(dst)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "general polyhedral mesh" in out.stdout and out.stdout.rstrip().endswith("End")
    for nm in ("U.water", "p", "alpha.water"):
        assert (dst / "0.002" / nm).exists(), os.listdir(dst / "0.002")
    t = (dst / "0.002/alpha.water").read_text()
    assert "bottom" in t and "zeroGradient" in t
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)
    s = prod.LduSolver.from_foam_case(fc)
    for _ in range(10):
        s.step()
    U = s.get("U").reshape(-1, 3)
    s.close(); fc.close()
    cd = (dst / "system/controlDict").read_text()
    (dst / "system/controlDict").write_text(cd.replace("startFrom       startTime;", "startFrom       latestTime;"))
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)
    assert fc.start_name == "0.002"
    U1, p1 = fc.initial_fields()
    np.testing.assert_allclose(U1, U, rtol=0, atol=1e-12 * np.abs(U).max())
    assert np.abs(U[:, 2]).max() > 0.01                              # the inflow has arrived
    fc.close()


@pytest.mark.gpu
def test_general_les_case_runs_and_writes_nut(prod, tmp_path):
    """LES Smagorinsky on the wavy bed from its case directory: nut.<phase> of the start time is read, renewed by the model after the last outer corrector, and written with
    the file's own patch entries"""
    dst, mesh = general_bed(tmp_path)
    (dst / "constant/turbulenceProperties.water").write_text("FoamFile { version 2.0; format ascii; class dictionary; object turbulenceProperties.water; }\nsimulationType LES;\n"
                                                             "LES { LESModel Smagorinsky; delta cubeRootVol; turbulence on; }\n")
    (dst / "0/nut.water").write_text((dst / "0/p").read_text().replace("object      p;", "object nut.water;").replace("[0 2 -2 0 0 0 0]", "[0 2 -1 0 0 0 0]").replace("internalField   uniform 0;", "internalField   uniform 2e-6;")
                                     .replace("fixedFluxPressure; value uniform 0;", "zeroGradient;").replace("top    { type fixedValue; value uniform 0; }", "top    { type calculated; value uniform 3e-6; }"))
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)
    s = prod.LduSolver.from_foam_case(fc)
    np.testing.assert_array_equal(s.get("nut"), 2e-6)
    for _ in range(5):
        s.step()
    nut = s.get("nut")
    assert nut.max() > 0 and not np.allclose(nut, 2e-6) and np.isfinite(nut).all()
    fc.write(s, "0.001")
    t = (dst / "0.001/nut.water").read_text()
    assert "calculated" in t and "uniform 3e-06" in t.replace("3e-6", "3e-06") and "nonuniform List<scalar>" in t
    s.close(); fc.close()


@pytest.mark.gpu
def test_general_kEqn_case_runs_and_writes_k(prod, tmp_path):
    """LES kEqn on the wavy bed from its case directory: k.<phase> and nut.<phase> of the start time are read (a `calculated` nut patch keeps the file's value until the first
    correctNut()), the controls of k come from fvSchemes / fvSolution, k and nut are written back with the files' own patch entries"""
    dst, mesh = general_bed(tmp_path)
    (dst / "constant/turbulenceProperties.water").write_text("FoamFile { version 2.0; format ascii; class dictionary; object turbulenceProperties.water; }\nsimulationType LES;\n"
                                                             "LES { LESModel kEqn; delta cubeRootVol; turbulence on; kEqnCoeffs { Ck 0.1; Ce 1.0; } }\n")
    nut = (dst / "0/p").read_text().replace("object      p;", "object nut.water;").replace("[0 2 -2 0 0 0 0]", "[0 2 -1 0 0 0 0]").replace("internalField   uniform 0;", "internalField   uniform 2e-6;") \
        .replace("fixedFluxPressure; value uniform 0;", "zeroGradient;").replace("top    { type fixedValue; value uniform 0; }", "top    { type calculated; value uniform 3e-6; }")
    (dst / "0/nut.water").write_text(nut)
    (dst / "0/k.water").write_text(nut.replace("object nut.water;", "object k.water;").replace("[0 2 -1 0 0 0 0]", "[0 2 -2 0 0 0 0]").replace("uniform 2e-6;", "uniform 3e-4;")
                                   .replace("top    { type calculated; value uniform 3e-6; }", "top    { type fixedValue; value uniform 4e-4; }"))
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)
    lc = fc.ldu_case
    assert lc.turbulence_model == prod.TURBULENCE_KEQN and (lc.les_ck, lc.les_ce) == (0.1, 1.0) and lc.k_initial == 3e-4 and (lc.k_tol, lc.k_rel_tol) == (1e-5, 0.1)
    top = fc.patch_names.index("top")
    assert lc.k_bc[top] == 1 and lc.k_value[top] == 4e-4 and lc.nut_bc[top] == 3 and lc.nut_value[top] == 3e-6
    s = prod.LduSolver.from_foam_case(fc)
    np.testing.assert_array_equal(s.get("k"), 3e-4); np.testing.assert_array_equal(s.get("nut"), 2e-6)
    for _ in range(5):
        s.step()
    k, nutv = s.get("k"), s.get("nut")
    assert k.min() > 0 and not np.allclose(k, 3e-4) and np.isfinite(k).all() and np.allclose(nutv, 0.1 * np.sqrt(k) * np.cbrt(s.geometry("V")), rtol=1e-12)
    fc.write(s, "0.001")
    t = (dst / "0.001/k.water").read_text()
    assert "fixedValue" in t and "nonuniform List<scalar>" in t and "calculated" in (dst / "0.001/nut.water").read_text()
    s.close(); fc.close()


@pytest.mark.gpu
def test_general_kEpsilon_case_runs(prod, tmp_path):
    """RAS kEpsilon (without wall functions) on the wavy bed from its case directory: coefficients, k / epsilon / nut files, the controls of both equations; k, epsilon and nut
    written back; nut = Cmu k^2 / epsilon after the first step"""
    dst, mesh = general_bed(tmp_path)
    (dst / "constant/turbulenceProperties.water").write_text("simulationType RAS;\nRAS { RASModel kEpsilon; turbulence on; kEpsilonCoeffs { Cmu 0.085; C1 1.4; C2 1.9; C3 -0.33; sigmak 1.1; sigmaEps 1.25; } }\n")
    nut = (dst / "0/p").read_text().replace("object      p;", "object nut.water;").replace("[0 2 -2 0 0 0 0]", "[0 2 -1 0 0 0 0]").replace("internalField   uniform 0;", "internalField   uniform 2e-6;") \
        .replace("fixedFluxPressure; value uniform 0;", "zeroGradient;").replace("top    { type fixedValue; value uniform 0; }", "top    { type calculated; value uniform 3e-6; }")
    (dst / "0/nut.water").write_text(nut)
    kf = nut.replace("object nut.water;", "object k.water;").replace("[0 2 -1 0 0 0 0]", "[0 2 -2 0 0 0 0]").replace("uniform 2e-6;", "uniform 3e-4;").replace("top    { type calculated; value uniform 3e-6; }", "top    { type fixedValue; value uniform 4e-4; }")
    (dst / "0/k.water").write_text(kf)
    (dst / "0/epsilon.water").write_text(kf.replace("object k.water;", "object epsilon.water;").replace("[0 2 -2 0 0 0 0]", "[0 2 -3 0 0 0 0]").replace("uniform 3e-4;", "uniform 2e-3;").replace("uniform 4e-4;", "uniform 3e-3;"))
    fc = prod.GeneralFoamCase(dst, prod.FY_SOLVER_PIMPLE)
    lc = fc.ldu_case
    assert lc.turbulence_model == prod.TURBULENCE_KEPSILON and (lc.ras_cmu, lc.ras_c1, lc.ras_c2, lc.ras_c3, lc.ras_sigmak, lc.ras_sigmaeps) == (0.085, 1.4, 1.9, -0.33, 1.1, 1.25)
    top = fc.patch_names.index("top")
    assert (lc.k_initial, lc.eps_initial) == (3e-4, 2e-3) and lc.eps_bc[top] == 1 and lc.eps_value[top] == 3e-3 and lc.nut_bc[top] == 3
    s = prod.LduSolver.from_foam_case(fc)
    np.testing.assert_array_equal(s.get("epsilon"), 2e-3)
    for _ in range(5):
        s.step()
    k, e, nutv = s.get("k"), s.get("epsilon"), s.get("nut")
    assert k.min() > 0 and e.min() > 0 and not np.allclose(e, 2e-3) and np.allclose(nutv, 0.085 * k * k / e, rtol=1e-12)
    fc.write(s, "0.001")
    assert "nonuniform List<scalar>" in (dst / "0.001/epsilon.water").read_text() and "fixedValue" in (dst / "0.001/epsilon.water").read_text()
    s.close(); fc.close()


@pytest.mark.gpu
def test_run_from_a_general_case_directory_equals_the_hand_built_solver(prod, tmp_path):
    """LduSolver.from_foam_case on a wavy renumbered block = the solver built from the same arrays by hand, bit for bit; write, reopen from latestTime"""
    n = 8
    mesh = pm.hex_block(n, n, n, (0.1, 0.1, 0.1), pm.wavy(0.003, (0.1, 0.1, 0.1)), patches=CAVITY, renumber_seed=7)
    dst = general_cavity(tmp_path, mesh)
    fc = prod.GeneralFoamCase(dst)
    s = prod.LduSolver.from_foam_case(fc)
    lc = fc.ldu_case
    h = prod.LduSolver(mesh, lc.dt, lc.nu, [0, 0], [(1, 0, 0), (0, 0, 0)], [0, 0], n_correctors=lc.n_correctors, n_non_orth_correctors=lc.n_non_orth_correctors,
                       p_tol=lc.p_tol, p_rel_tol=lc.p_rel_tol, p_final_tol=lc.p_final_tol, p_final_rel_tol=lc.p_final_rel_tol, p_max_iter=lc.p_max_iter,
                       u_tol=lc.u_tol, u_rel_tol=lc.u_rel_tol, u_max_iter=lc.u_max_iter, p_ref_cell=lc.p_ref_cell, p_ref_value=lc.p_ref_value,
                       rho_fluid=lc.rho_fluid, rho_particle=lc.rho_particle, momentum_predictor=lc.momentum_predictor)
    for _ in range(3):
        s.step(); h.step()
    np.testing.assert_array_equal(s.get("U"), h.get("U")); np.testing.assert_array_equal(s.get("p"), h.get("p"))
    assert np.abs(s.get("U")).max() > 0.05
    fc.write(s, "0.015")
    U, p = s.get("U").reshape(-1, 3), s.get("p")
    s.close(); h.close(); fc.close()
    (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startFrom       startTime;", "startFrom       latestTime;"))
    fc = prod.GeneralFoamCase(dst)
    U1, p1 = fc.initial_fields()
    np.testing.assert_array_equal(U1, U); np.testing.assert_array_equal(p1, p)
    fc.close()


@pytest.mark.gpu
def test_foamYadeHip_executable_runs_a_general_mesh_case(prod, tmp_path):
    """foamYadeHip -solver ico on a case the block reader refuses: it falls through to the general solver, prints icoFoamYade's log lines and writes the
    time directories of controlDict"""
    n = 8
    mesh = pm.prism_block(n, n, 4, (0.1, 0.1, 0.1), pm.wavy(0.002, (0.1, 0.1, 0.1)))
    dst = general_cavity(tmp_path, mesh)
    for nm in ("U", "p"):
        t = (dst / "0" / nm).read_text()
        head = t[:t.index("boundaryField")]
        body = "".join("    %s { type %s; }\n" % (s, ("noSlip" if nm == "U" else "zeroGradient")) for s in pm.SIDES)
        if nm == "U":
            body = body.replace("ymax { type noSlip; }", "ymax { type fixedValue; value uniform (1 0 0); }")
        (dst / "0" / nm).write_text(head + "boundaryField\n{\n" + body + "}\n")
    cd = (dst / "system/controlDict").read_text()
    import re
    cd = re.sub(r"endTime\s+[0-9.eE+-]+;", "endTime         0.02;", cd)
    cd = re.sub(r"writeInterval\s+[0-9.eE+-]+;", "writeInterval   2;", cd)
    (dst / "system/controlDict").write_text(cd)
    exe = os.path.join(os.path.dirname(prod.__file__), "bin", "foamYadeHip")
    out = subprocess.run([exe, "-solver", "ico", "-case", str(dst)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "general polyhedral mesh" in out.stdout and "Courant Number mean" in out.stdout and out.stdout.rstrip().endswith("End")
    fc = prod.GeneralFoamCase(dst)
    written = sorted(d for d in os.listdir(dst) if d[0].isdigit() and d != "0")
    assert written, os.listdir(dst)
    t = (dst / written[-1] / "U").read_text()
    assert "nonuniform List<vector>" in t and "ymax" in t
    fc.close()
