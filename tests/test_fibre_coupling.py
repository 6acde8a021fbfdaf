"""Fibre coupling (FoamYade::fibreCpl, FoamYade.H:102): Yade sends 15 doubles per particle (FoamYade.C:131-136 parallel, :161-165
serial); the position is read with stride 15 (:194-198), velocity / spin / radius with stride 10 from the same buffer (:211-221).
The fixtures come from the reference itself, run with the flag set (oracle/ref_driver.cpp, tests/golden/gen_golden.py)."""
import numpy as np
import pytest

import golden_cases as gc
import golden_util as gu

NAMES = [c.name for c in gc.FIBRE_CASES]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_fibre_matches_reference(oracle, name):
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    fields = gu.check_inputs_reproducible(c, g)
    mesh = oracle.Mesh(c.nx, c.ny, c.nz, c.dx, c.origin)
    assert np.array_equal(mesh.pre, g["tree_preorder"])
    mut = oracle.fresh_mutable(mesh.Nc)
    for s in range(c.nsteps):
        rec = g[f"records_s{s}"]
        assert rec.shape[1] == 15
        off = gu.batch_offsets(c, rec.shape[0])
        narrow = oracle.fibre_narrow(rec, off)
        # the stride-10 read is NOT the first ten columns of the particle's own record
        assert (narrow[:, 3:10] != rec[:, 3:10]).any(axis=1).mean() > 0.9
        assert np.all(narrow[:, 9] > 0)                                   # fixture construction: every radius read is a radius
        out = oracle.particle_action(mesh, fields, mut, rec, off, c.gaussian, c.rhoP, c.rhoF, c.nu, fibre=True)
        kref = g[f"k_s{s}"].astype(np.int32)
        ok = out["chain_len"] <= 12
        assert np.array_equal(out["k"][ok], kref[ok])
        assert np.array_equal(out["ids"][ok], g[f"ids_s{s}"][ok])
        np.testing.assert_allclose(out["w"][ok], g[f"w_s{s}"][ok], rtol=gu.RTOL_ORACLE, atol=0)
        fref = g[f"force_s{s}"]
        scale = np.abs(fref).max() + 1e-300
        np.testing.assert_allclose(out["force"][ok], fref[ok], rtol=gu.RTOL_ORACLE, atol=1e-14 * scale)
        # reading the records the plain way gives different forces: the fixture does pin the quirk
        mut2 = oracle.fresh_mutable(mesh.Nc)
        plain = oracle.particle_action(mesh, fields, mut2, rec[:, :10], off, c.gaussian, c.rhoP, c.rhoF, c.nu)
        assert not np.allclose(plain["force"][ok], fref[ok], rtol=1e-3, atol=1e-6 * scale)
        if np.all(ok):
            for nm, comps, dflt in (("alpha", 1, 1.0), ("uSource", 3, 0.0)):
                ref = gu.dense(g, nm, s, mesh.Nc, comps, dflt)
                sc = np.abs(ref).max() + 1e-300
                np.testing.assert_allclose(mut[nm], ref, rtol=gu.RTOL_ORACLE, atol=1e-13 * sc)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("resident", [False, True])
def test_product_fibre_matches_reference(product, name, resident):
    import torch
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    fields = gu.check_inputs_reproducible(c, g)
    Nc = c.ncells
    mut = dict(uSourceDrag=np.full(Nc, 5.0), alpha=np.zeros(Nc), uSource=np.full((Nc, 3), 3.0), uParticle=np.full((Nc, 3), 4.0))
    mesh = product.BlockMesh(c.nx, c.ny, c.nz, c.dx, c.origin)
    fy = product.FoamYade(mesh, fields["U"], fields["gradP"], fields["vGrad"], fields["divT"], fields["ddtU"], c.g,
                          mut["uSourceDrag"], mut["alpha"], mut["uSource"], mut["uParticle"], bool(c.gaussian))
    fy.setScalarProperties(c.rhoP, c.rhoF, c.nu)
    if resident:
        fy.setFibreCoupling(True)
    else:
        fy.fibreCpl = True          # what the reference's callers do: assign the public flag (FoamYade.H:102); the mirror forwards it on next use
    for s in range(c.nsteps):
        rec = g[f"records_s{s}"]
        off = gu.batch_offsets(c, rec.shape[0])
        nb = len(off) - 1
        parts = [np.ascontiguousarray(rec[off[b]:off[b + 1]]) for b in range(nb)]
        if resident:
            fy.setParticlesDevice([torch.from_numpy(p).cuda() for p in parts])
        else:
            fy.setParticles(parts)
        fy.setParticleAction(c.dt)
        k = np.concatenate([fy.stencils(b)[0] for b in range(nb)])
        ids = np.concatenate([fy.stencils(b)[1] for b in range(nb)])
        chain = np.concatenate([fy.stencils(b)[3] for b in range(nb)])
        F = np.concatenate([fy.forces(b) for b in range(nb)])
        kref = g[f"k_s{s}"].astype(np.int32)
        ok = chain <= 12
        assert np.array_equal(k[ok], kref[ok])
        assert np.array_equal(ids[ok], g[f"ids_s{s}"][ok])
        fref = g[f"force_s{s}"]
        scale = np.abs(fref).max() + 1e-300
        np.testing.assert_allclose(F[ok], fref[ok], rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * scale)
        if np.all(ok):
            for nm, comps, dflt in (("alpha", 1, 1.0), ("uSource", 3, 0.0)):
                ref = gu.dense(g, nm, s, Nc, comps, dflt)
                sc = np.abs(ref).max() + 1e-300
                np.testing.assert_allclose(mut[nm], ref, rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * sc)
        fy.setSourceZero()
    fy.close()
