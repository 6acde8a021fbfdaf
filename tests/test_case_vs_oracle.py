"""Case directories against the ORACLE (round 5; VERDICT round 4, task 7): the product reads a case directory with its own reader
(csrc/foam_case.cpp), the tests read the same directory with a reader of their own (tests/foam_dict_reader.py, no shared code) and hand it to the CPU
oracle; `foamYadeHip` run on the directory must leave the oracle's U / p in the time directories it writes, and the library driven from the product's
reader the oracle's phi.  Until now every case-directory test compared the product with itself.

What the reference reads through OpenFOAM: icoFoamYade/createFields.H:29-45,166-169, pimpleFoamYade/createFields.H:3-15,83-86, createControl.H; what
it writes: runTime.write() (icoFoamYade.C:142, pimpleFoamYade.C:107).  FV parity itself stays UNPINNED (no OpenFOAM here): these tests pin the READER."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import foam_dict_reader as fr
import poly_meshes as pm

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = os.path.join(HERE, "golden", "cases")
EXE = os.path.join(os.path.dirname(HERE), "yade-openfoam-coupling_amd", "bin", "foamYadeHip")


@pytest.fixture
def prod():
    from conftest import load_product
    return load_product()


def close(a, b, rtol, what):
    sc = np.abs(b).max() + 1e-300
    assert np.abs(a - b).max() <= rtol * sc, (what, np.abs(a - b).max() / sc)


@pytest.mark.parametrize("name,solver", [("cavity_ico", 0), ("bed_pimple", 1)])
def test_the_two_readers_agree_on_the_golden_cases(prod, oracle, name, solver):
    """no GPU: the product's reader and the tests' reader extract the same case from the same files"""
    d = os.path.join(CASES, name)
    fc = prod.FoamCase(d, solver)
    oc, c = fr.block_case_for_oracle(oracle, d, solver)
    pc = fc.case
    assert (pc.nx, pc.ny, pc.nz) == (c["nx"], c["ny"], c["nz"]) and pc.dx == pytest.approx(c["dx"], rel=1e-14) and np.allclose(list(pc.origin), c["origin"], atol=1e-15)
    assert (pc.dt, pc.nu, pc.rho_fluid, pc.rho_particle) == (c["dt"], c["nu"], c["rho_f"], c["rho_p"]) and tuple(pc.g) == c["g"]
    assert list(pc.u_bc) == c["u_bc"] and list(pc.p_bc) == c["p_bc"] and [tuple(pc.u_value[q]) for q in range(6)] == c["u_val"] and list(pc.p_value) == c["p_val"]
    assert (pc.n_correctors, pc.n_non_orth_correctors, pc.momentum_predictor, pc.p_ref_cell, pc.p_ref_value) == (c["n_corr"], c["n_non_orth"], c["momentum_predictor"], c["p_ref_cell"], c["p_ref_value"])
    assert max(pc.n_outer_correctors, 1) == c["n_outer"] and pc.p_solver == c["p_solver"]
    assert (pc.p_tol, pc.p_rel_tol, pc.p_final_tol, pc.p_final_rel_tol, pc.u_tol, pc.u_rel_tol) == (c["p_tol"], c["p_rel_tol"], c["p_final_tol"], c["p_final_rel_tol"], c["u_tol"], c["u_rel_tol"])
    assert (fc.delta_t, fc.end_time, fc.write_interval_steps, fc.u_name, fc.patch_of_side) == (c["dt"], c["end_time"], c["write_interval_steps"], c["u_name"], c["side"])
    fc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,solver", [("cavity_ico", 0), ("bed_pimple", 1)])
def test_foamYadeHip_on_a_block_case_leaves_the_oracles_fields(prod, oracle, tmp_path, name, solver):
    if not os.path.exists(EXE):
        pytest.fail("foamYadeHip has not been built: run __graft_entry__.build()")
    dst = tmp_path / name
    shutil.copytree(os.path.join(CASES, name), dst)
    out = subprocess.run([EXE, "-solver", "ico" if solver == 0 else "pimple", "-case", str(dst)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    oc, c = fr.block_case_for_oracle(oracle, str(dst), solver)
    n_steps = int(round((c["end_time"] - c["start_time"]) / c["dt"]))
    o = oracle.FvSolver(oc)
    # the same loop on the product's own reader, for phi (the time directories hold U and p)
    fc = prod.FoamCase(str(dst), solver)
    s = prod.Solver(fc.case)
    for k in range(1, n_steps + 1):
        o.step(); s.step()
        if k % c["write_interval_steps"] == 0:
            tname = "%g" % (c["start_time"] + k * c["dt"])
            assert os.path.isdir(dst / tname), (tname, sorted(os.listdir(dst)))
            close(fr.written_field(str(dst), tname, c["u_name"], 3).ravel(), o.get("U"), 1e-5, "U written at " + tname)
            pw, po = fr.written_field(str(dst), tname, "p", 1), o.get("p")
            if all(b != fr.P_FIXED for b in c["p_bc"]):
                pw = pw - pw.mean(); po = po - po.mean()                 # (the level is set by the reference cell: compare the shape)
            close(pw, po, 1e-5, "p written at " + tname)
    assert np.abs(o.get("U")).max() > 0
    for nm in ("phi_x", "phi_y", "phi_z"):
        close(s.get(nm), o.get(nm), 1e-5, nm)
    so, ss = o.stats(), s.stats()
    assert abs(so["p_iters_total"] - ss["p_iters_total"]) <= 2
    fc.close(); s.close(); o.close()


@pytest.mark.gpu
def test_coupled_run_from_the_bed_case_matches_the_oracle(prod, oracle, tmp_path):
    """the pimple case with a particle cloud (what Yade would send), library driven from the product's reader vs the oracle from the tests' reader"""
    d = os.path.join(CASES, "bed_pimple")
    oc, c = fr.block_case_for_oracle(oracle, d, 1)
    o = oracle.FvSolver(oc)
    fc = prod.FoamCase(d, 1)
    s = prod.Solver(fc.case)
    rs = np.random.RandomState(4)
    rec = np.zeros((3000, 10))
    rec[:, 0:2] = -0.03 + 0.06 * rs.random_sample((3000, 2)); rec[:, 2] = 0.05 * rs.random_sample(3000); rec[:, 3:6] = 0.02 * (rs.random_sample((3000, 3)) - 0.5)
    rec[:, 9] = 0.2 * c["dx"]
    for _ in range(4):
        fo = o.step(rec)["force"]
        s.set_particles(rec); s.step()
        close(s.forces(), fo, 1e-6, "force")
    for nm in ("U", "p", "phi_x", "phi_y", "phi_z"):
        close(s.get(nm), o.get(nm), 1e-5, nm)
    fc.close(); s.close(); o.close()


@pytest.mark.gpu
def test_foamYadeHip_on_a_polyMesh_case_leaves_the_oracles_fields(prod, oracle, tmp_path):
    """a wavy, renumbered hexahedral mesh in constant/polyMesh: foamYadeHip (general-mesh solver) against oracle/ldu_oracle.cpp fed by the tests' own
    reader of points / faces / owner / neighbour / boundary"""
    if not os.path.exists(EXE):
        pytest.fail("foamYadeHip has not been built: run __graft_entry__.build()")
    mesh = pm.hex_block(10, 10, 10, (0.1, 0.1, 0.1), pm.wavy(0.003, (0.1, 0.1, 0.1)), patches=[("movingWall", [3]), ("fixedWalls", [0, 1, 2, 4, 5])], renumber_seed=5)
    dst = tmp_path / "case"
    shutil.copytree(os.path.join(CASES, "cavity_ico"), dst)
    os.remove(dst / "system/blockMeshDict")
    pm.write_poly_mesh_files(dst, mesh)
    out = subprocess.run([EXE, "-solver", "ico", "-case", str(dst)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "general polyhedral mesh" in out.stdout
    m2, args, kw, c = fr.poly_case_for_oracle(str(dst), 0)
    for k in ("face_offsets", "face_points", "owner", "neighbour", "patch_start", "patch_size"):
        np.testing.assert_array_equal(m2[k], mesh[k], err_msg=k)
    np.testing.assert_array_equal(m2["points"], mesh["points"])
    o = oracle.LduSolver(m2, *args, **kw)
    n_steps = int(round((c["end_time"] - c["start_time"]) / c["dt"]))
    for k in range(1, n_steps + 1):
        o.step()
        if k % c["write_interval_steps"] == 0:
            tname = "%g" % (c["start_time"] + k * c["dt"])
            close(fr.written_field(str(dst), tname, "U", 3).ravel(), o.get("U"), 1e-5, "U written at " + tname)
            pw, po = fr.written_field(str(dst), tname, "p", 1), o.get("p")
            close(pw - pw.mean(), po - po.mean(), 2e-5, "p written at " + tname)
    assert np.abs(o.get("U")).max() > 0.1
    # phi: the library driven from the product's reader of the same directory
    gc_ = prod.GeneralFoamCase(str(dst))
    h = prod.LduSolver.from_foam_case(gc_)
    for _ in range(n_steps):
        h.step()
    close(h.get("phi"), o.get("phi"), 2e-5, "phi")
    h.close(); gc_.close(); o.close()
