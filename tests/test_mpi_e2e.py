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
