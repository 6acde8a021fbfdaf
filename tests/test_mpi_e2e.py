"""Drop-in boundary over real MPI: an unmodified-Yade-shaped peer (world rank 0) and the solver rank (C++ facade -> C-ABI -> HIP
kernels, MPI transport) under `mpiexec` MPMD, compared with what the reference's fake Yade received (golden wire_force).
Skipped when the image has no MPI launcher (it is off-PATH under /opt/conda here)."""
import os
import subprocess

import numpy as np
import pytest

import golden_cases as gc
import golden_util as gu
from conftest import ROOT

pytestmark = pytest.mark.gpu
MPIEXEC = "/opt/conda/bin/mpiexec"
EXE = os.path.join(ROOT, "tests", "native", "mpi_e2e")


@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(EXE)), reason="no MPI launcher / e2e binary (run __graft_entry__.build())")
@pytest.mark.parametrize("name", ["g16_serial_2step", "p32_serial_c1"])
def test_serial_yade_over_mpi(tmp_path, name):
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    f = gc.fluid_fields(c)
    d = str(tmp_path)
    meta = [c.nx, c.ny, c.nz, repr(c.dx), repr(c.origin[0]), repr(c.origin[1]), repr(c.origin[2]), c.gaussian, repr(c.rhoP), repr(c.rhoF),
            repr(c.nu), repr(c.dt), repr(c.g[0]), repr(c.g[1]), repr(c.g[2])]
    open(os.path.join(d, "meta.txt"), "w").write(" ".join(str(m) for m in meta) + "\n")
    for nm, arr in f.items():
        np.ascontiguousarray(arr).tofile(os.path.join(d, nm + ".bin"))
    rec = g["records_s0"]
    np.ascontiguousarray(rec).tofile(os.path.join(d, "records.bin"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    subprocess.run([MPIEXEC, "-n", "1", EXE, d, ":", "-n", "1", EXE, d], check=True, timeout=300, env=env)
    F = np.fromfile(os.path.join(d, "yade_force.bin")).reshape(-1, 6)
    fref = g["wire_force_s0"]
    sc = np.abs(fref).max()
    np.testing.assert_allclose(F, fref, rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * sc)
    np.testing.assert_array_equal(np.fromfile(os.path.join(d, "yade_owner.bin"), dtype=np.int32), g["wire_owner_s0"])
    assert np.fromfile(os.path.join(d, "yade_fluiddt.bin"))[0] == c.dt
    assert np.fromfile(os.path.join(d, "foam_yadedt.bin"))[0] == 1.25e-5
    a = np.fromfile(os.path.join(d, "foam_alpha.bin"))
    np.testing.assert_allclose(a, gu.dense(g, "alpha", 0, c.ncells, 1, 1.0), rtol=gu.RTOL_GPU)


FAKE_YADE = os.path.join(ROOT, "tests", "native", "fake_yade")
RUNNER = os.path.join(ROOT, "yade-openfoam-coupling_amd", "bin", "foamYadeHip_mpi")


@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(FAKE_YADE) and os.path.exists(RUNNER)), reason="no MPI launcher / binaries (run __graft_entry__.build())")
def test_foamYadeHip_mpi_next_to_a_serial_yade(product, tmp_path):
    """the whole stack the way the reference is launched (README.md:29): `mpiexec -n 1 <yade> : -n 1 <solver> -case ...` -- a serial-Yade
    peer over real MPI, the executable, the OpenFOAM case reader, the PIMPLE loop and the HIP kernels; the forces Yade receives in the
    last step equal those of the library driven directly with the same particles"""
    import shutil
    case_src = os.path.join(ROOT, "tests", "golden", "cases", "bed_pimple")
    dst = tmp_path / "bed"
    shutil.copytree(case_src, dst)
    fc = product.FoamCase(dst, 1)
    c = fc.case
    rs = np.random.RandomState(8)
    rec = np.zeros((2500, 10))
    rec[:, 0:2] = -0.03 + 0.06 * rs.random_sample((2500, 2)); rec[:, 2] = 0.06 * rs.random_sample(2500)
    rec[:, 3:6] = 0.01 * rs.standard_normal((2500, 3)); rec[:, 9] = 0.2 * c.dx
    rec.tofile(tmp_path / "records.bin")
    nsteps = int(round((fc.end_time - fc.start_time) / fc.delta_t))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([MPIEXEC, "-n", "1", FAKE_YADE, str(tmp_path / "records.bin"), "1", str(nsteps), str(tmp_path / "force.bin"), ":",
                          "-n", "1", RUNNER, "-solver", "pimple", "-case", str(dst)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "End" in out.stdout
    F = np.fromfile(tmp_path / "force.bin").reshape(-1, 6)
    s = product.Solver(c)
    for _ in range(nsteps):
        s.set_particles(rec)
        s.step()
    Fr = s.forces()
    sc = np.abs(Fr).max()
    assert sc > 0
    np.testing.assert_allclose(F, Fr, rtol=1e-8, atol=1e-10 * sc)
    assert sorted(d for d in os.listdir(dst) if d[0].isdigit()) == ["0", "0.001", "0.002"]
    s.close(); fc.close()


@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(FAKE_YADE) and os.path.exists(RUNNER)), reason="no MPI launcher / binaries (run __graft_entry__.build())")
@pytest.mark.parametrize("n_fluid,comm_flag", [(2, "-hostComm"), (3, "-hostComm"), (3, "-ipcComm")])
def test_foamYadeHip_mpi_parallel_next_to_a_serial_yade(product, tmp_path, n_fluid, comm_flag):
    """the reference's `-parallel` launch (README.md:29): `mpiexec -n 1 <yade> : -n N <solver> -parallel` -- N solver PROCESSES, each with its
    z-slab of the undecomposed case, halos / reductions / the coarse-level gather staged through the host and moved by MPI (`-hostComm`) or stored by the
    library's kernels straight into the other ranks' hipIpc-mapped device windows (`-ipcComm`, fy_comm_create_ipc, MPI carrying its bootstrap only) -- the ranks
    share the box's one GPU here; with a GPU per rank the same executable takes RCCL --, every rank answering the serial-Yade protocol for the particles of
    its slab; the forces Yade receives and the gathered, undecomposed time directory equal the one-rank run's"""
    import shutil
    case_src = os.path.join(ROOT, "tests", "golden", "cases", "bed_pimple")
    rs = np.random.RandomState(8)
    runs = {}
    for tag, nf in (("one", 1), ("many", n_fluid)):
        dst = tmp_path / tag / "bed"
        shutil.copytree(case_src, dst)
        fc = product.FoamCase(dst, 1)
        c = fc.case
        assert c.nz % (2 * n_fluid) == 0
        if tag == "one":
            rec = np.zeros((2500, 10))
            rec[:, 0:2] = -0.03 + 0.06 * rs.random_sample((2500, 2)); rec[:, 2] = 0.11 * rs.random_sample(2500)       # (the cloud spans every slab)
            rec[:, 3:6] = 0.01 * rs.standard_normal((2500, 3)); rec[:, 9] = 0.2 * c.dx
        rec.tofile(tmp_path / tag / "records.bin")
        nsteps = int(round((fc.end_time - fc.start_time) / fc.delta_t))
        tlast = "%g" % fc.end_time
        fc.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [MPIEXEC, "-n", "1", FAKE_YADE, str(tmp_path / tag / "records.bin"), "1", str(nsteps), str(tmp_path / tag / "force.bin"), ":",
               "-n", str(nf), RUNNER, "-solver", "pimple", "-case", str(dst)] + (["-parallel", "-nYade", "1", comm_flag] if nf > 1 else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2000:])
        assert out.stdout.count("End") == 1 and ("Decomposition: %d z-slabs" % nf in out.stdout) == (nf > 1)
        if nf > 1 and comm_flag == "-ipcComm":
            assert "peer stores into hipIpc-mapped device windows" in out.stdout
        (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startFrom       startTime;", "startFrom       latestTime;"))
        fc2 = product.FoamCase(dst, 1)
        assert fc2.start_name == tlast
        U, p = fc2.initial_fields()
        runs[tag] = (np.fromfile(tmp_path / tag / "force.bin").reshape(-1, 6), U, p, (dst / tlast / "alpha.water").read_text())
        fc2.close()
    Fo, Uo, po, ao = runs["one"]
    Fm, Um, pm, am = runs["many"]
    sc = np.abs(Fo).max()
    assert sc > 0 and np.abs(Fm - Fo).max() <= 1e-6 * sc
    assert np.abs(Um - Uo).max() <= 1e-5 * np.abs(Uo).max() and np.abs(pm - po).max() <= 1e-5 * np.abs(po).max()
    assert np.abs(Uo).max() > 0 and "nonuniform List<scalar>" in am


@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(FAKE_YADE) and os.path.exists(RUNNER)), reason="no MPI launcher / binaries (run __graft_entry__.build())")
def test_foamYadeHip_mpi_parallel_on_a_decomposed_case(product, tmp_path):
    """the reference's workflow to the letter: decomposePar, then `mpiexec -n 1 <yade> : -n N <solver> -parallel` -- every solver rank reads the fields
    of its processor directory and writes its time directories there (with the processor patches), nobody gathers; put together, the processors'
    fields are the one-rank run's"""
    import shutil
    from test_foam_case import decompose_case
    case_src = os.path.join(ROOT, "tests", "golden", "cases", "bed_pimple")
    rs = np.random.RandomState(8)
    rec = None
    out_fields = {}
    for tag, nf in (("one", 1), ("many", 2)):
        dst = tmp_path / tag / "bed"
        shutil.copytree(case_src, dst)
        fc = product.FoamCase(dst, 1)
        c = fc.case
        if rec is None:
            rec = np.zeros((2000, 10))
            rec[:, 0:2] = -0.03 + 0.06 * rs.random_sample((2000, 2)); rec[:, 2] = 0.11 * rs.random_sample(2000)
            rec[:, 3:6] = 0.01 * rs.standard_normal((2000, 3)); rec[:, 9] = 0.2 * c.dx
        rec.tofile(tmp_path / tag / "records.bin")
        nsteps = int(round((fc.end_time - fc.start_time) / fc.delta_t))
        tlast = "%g" % fc.end_time
        n = fc.n_cells
        fc.close()
        if nf > 1:
            decompose_case(dst, nf)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [MPIEXEC, "-n", "1", FAKE_YADE, str(tmp_path / tag / "records.bin"), "1", str(nsteps), str(tmp_path / tag / "force.bin"), ":",
               "-n", str(nf), RUNNER, "-solver", "pimple", "-case", str(dst)] + (["-parallel", "-nYade", "1", "-hostComm"] if nf > 1 else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2000:])
        if nf == 1:
            (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startFrom       startTime;", "startFrom       latestTime;"))
            f2 = product.FoamCase(dst, 1)
            out_fields[tag] = f2.initial_fields()
            f2.close()
        else:
            assert "decomposed (processor directories)" in out.stdout
            assert not os.path.exists(dst / tlast)                                  # nothing gathered into the case root
            (dst / "system/controlDict").write_text((dst / "system/controlDict").read_text().replace("startFrom       startTime;", "startFrom       latestTime;"))
            Us, ps = [], []
            for r in range(nf):
                text = (dst / ("processor%d" % r) / tlast / "U.water").read_text()
                import re
                assert re.search(r"procBoundary%dto%d\s*\{\s*type\s+processor;" % (r, 1 - r), text)
                assert "procBoundary%dto%d" % (r, 1 - r) in (dst / ("processor%d" % r) / tlast / "alpha.water").read_text()
                f2 = product.FoamCase(dst, 1, processor=(r, nf))
                assert f2.start_name == tlast and f2.field_cells == n // nf
                U, p = f2.initial_fields()
                Us.append(U); ps.append(p)
                f2.close()
            out_fields[tag] = (np.vstack(Us), np.concatenate(ps))
    Uo, po = out_fields["one"]
    Um, pm = out_fields["many"]
    assert np.abs(Uo).max() > 0
    assert np.abs(Um - Uo).max() <= 1e-5 * np.abs(Uo).max() and np.abs(pm - po).max() <= 1e-5 * np.abs(po).max()
    Fo = np.fromfile(tmp_path / "one" / "force.bin"); Fm = np.fromfile(tmp_path / "many" / "force.bin")
    assert np.abs(Fm - Fo).max() <= 1e-6 * np.abs(Fo).max()


WIRE_BENCH = os.path.join(ROOT, "tools", "native", "wire_bench")


@pytest.mark.skipif(not (os.path.exists(MPIEXEC) and os.path.exists(WIRE_BENCH)), reason="no MPI launcher / binary (run __graft_entry__.build())")
@pytest.mark.parametrize("solver_ranks,axis", [(3, None), (5, None), (4, "2")])
def test_wire_helpers_answer_like_one_solver_rank(solver_ranks, axis):
    """K solver-side ranks in front of the one GPU -- a computing rank and K - 1 wire helpers that receive a parallel Yade's records into a
    shared-memory arena and send the answers out of it (include/foamyade_mpi.h; FoamYade.C:77-155, 239-243, 504-507 with K solver ranks) -- give
    every Yade worker what ONE solver rank gives it: the same particles located, none by two ranks (a particle whose sphere reaches two helpers'
    boxes is sent to both and found by exactly one), the same forces.  tools/native/wire_bench.cpp is the peer: a master and three workers."""
    import json

    def run(k, env_extra):
        a = [WIRE_BENCH, "24", "30000", "2", "1e-4", "-", str(k)]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
        out = subprocess.run([MPIEXEC, "-n", "1"] + a + [":", "-n", "3"] + a + [":", "-n", str(k)] + a, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, (out.stderr or out.stdout)[-1500:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    one = run(1, {})
    many = run(solver_ranks, {"FOAMYADE_WIRE_CUT_AXIS": axis} if axis else {})
    assert many["solver_side_ranks"] == solver_ranks and one["solver_side_ranks"] == 1
    assert many["found_by_two_ranks"] == 0
    assert many["found_at_the_workers"] == one["found_at_the_workers"] > 29000
    assert many["bytes_in"] > one["bytes_in"]                       # (the particles at the cuts travel twice)
    assert abs(many["sum_fz_at_the_workers"] - one["sum_fz_at_the_workers"]) <= 1e-9 * abs(one["sum_fz_at_the_workers"])
