"""The particle half on a NON-UNIFORM mesh (a graded blockMesh block: geometric cell sizes): the k-d tree is built over the
cell centres as they are (explicit nodes, no lattice shortcuts), `interpRange` follows the reference in coming from V[0] alone (quirk Q7,
FoamYade.C:69), the void fraction uses each cell's own volume.  Fixtures: the reference itself run on the graded mesh
(oracle/ref_driver.cpp takes mesh.C(), mesh.V(), mesh.points() from files; tests/golden/gen_golden.py).  Gaussian mode only."""
import numpy as np
import pytest

import golden_cases as gc
import golden_util as gu

NAMES = [c.name for c in gc.GRADED_CASES]


def mesh_of(c, g):
    C, V, P = gc.cell_centres(c), gc.cell_volumes(c), gc.mesh_points(c)
    assert gc.sha(C) == str(g["centres_sha"][0]) and gc.sha(V) == str(g["volumes_sha"][0])
    assert V.max() / V.min() > 2.0                                       # it IS non-uniform
    return C, V, P.min(axis=0), P.max(axis=0)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_on_a_graded_mesh_matches_reference(oracle, name):
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    fields = gu.check_inputs_reproducible(c, g)
    C, V, lo, hi = mesh_of(c, g)
    mesh = oracle.Mesh(c.nx, c.ny, c.nz, c.dx, c.origin, centres=C, volumes=V, bbmin=lo, bbmax=hi)
    assert np.array_equal(mesh.pre, g["tree_preorder"])
    rng, sig = g["interp_scalars"]
    assert abs(rng - 4 * V[0] ** (1.0 / 3.0)) < 1e-15                    # Q7: from the FIRST cell's volume
    if c.n_yade > 1:
        np.testing.assert_array_equal(g["wire_bbox"], np.concatenate([lo, hi]))
    mut = oracle.fresh_mutable(mesh.Nc)
    for s in range(c.nsteps):
        rec = g[f"records_s{s}"]
        out = oracle.particle_action(mesh, fields, mut, rec, gu.batch_offsets(c, rec.shape[0]), 1, c.rhoP, c.rhoF, c.nu)
        kref = g[f"k_s{s}"].astype(np.int32)
        ok = out["chain_len"] <= 12
        assert ok.mean() > 0.95
        assert np.array_equal(out["k"][ok], kref[ok])
        assert np.array_equal(out["ids"][ok], g[f"ids_s{s}"][ok])
        np.testing.assert_allclose(out["w"][ok], g[f"w_s{s}"][ok], rtol=gu.RTOL_ORACLE, atol=0)
        fref = g[f"force_s{s}"]
        scale = np.abs(fref).max() + 1e-300
        np.testing.assert_allclose(out["force"][ok], fref[ok], rtol=gu.RTOL_ORACLE, atol=1e-14 * scale)
        if np.all(ok):
            for nm, comps, dflt in (("alpha", 1, 1.0), ("uSourceDrag", 1, 0.0), ("uParticle", 3, 0.0), ("uSource", 3, 0.0)):
                ref = gu.dense(g, nm, s, mesh.Nc, comps, dflt)
                sc = np.abs(ref).max() + 1e-300
                np.testing.assert_allclose(mut[nm], ref, rtol=gu.RTOL_ORACLE, atol=1e-13 * sc, err_msg=nm)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_product_on_a_graded_mesh_matches_reference(product, name):
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    fields = gu.check_inputs_reproducible(c, g)
    C, V, lo, hi = mesh_of(c, g)
    Nc = c.ncells
    mut = dict(uSourceDrag=np.full(Nc, 5.0), alpha=np.zeros(Nc), uSource=np.full((Nc, 3), 3.0), uParticle=np.full((Nc, 3), 4.0))
    mesh = product.GeneralMesh(C, V, lo, hi)                              # fy_mesh_desc.nx = 0: no lattice assumption anywhere
    fy = product.FoamYade(mesh, fields["U"], fields["gradP"], fields["vGrad"], fields["divT"], fields["ddtU"], c.g,
                          mut["uSourceDrag"], mut["alpha"], mut["uSource"], mut["uParticle"], True)
    fy.setScalarProperties(c.rhoP, c.rhoF, c.nu)
    assert np.array_equal(fy.tree_preorder(), g["tree_preorder"])
    assert fy.interpRange == g["interp_scalars"][0]
    for s in range(c.nsteps):
        rec = g[f"records_s{s}"]
        off = gu.batch_offsets(c, rec.shape[0])
        nb = len(off) - 1
        fy.setParticles([rec[off[b]:off[b + 1]] for b in range(nb)])
        fy.setParticleAction(c.dt)
        k = np.concatenate([fy.stencils(b)[0] for b in range(nb)])
        ids = np.concatenate([fy.stencils(b)[1] for b in range(nb)])
        w = np.concatenate([fy.stencils(b)[2] for b in range(nb)])
        chain = np.concatenate([fy.stencils(b)[3] for b in range(nb)])
        F = np.concatenate([fy.forces(b) for b in range(nb)])
        found = np.concatenate([fy.found(b) for b in range(nb)])
        kref = g[f"k_s{s}"].astype(np.int32)
        ok = chain <= 12
        assert np.array_equal(k[ok], kref[ok])
        assert np.array_equal(ids[ok], g[f"ids_s{s}"][ok])
        assert np.array_equal(found, np.where(kref > 0, 1, -1))
        np.testing.assert_allclose(w[ok], g[f"w_s{s}"][ok], rtol=gu.RTOL_GPU, atol=0)
        fref = g[f"force_s{s}"]
        scale = np.abs(fref).max() + 1e-300
        np.testing.assert_allclose(F[ok], fref[ok], rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * scale)
        if np.all(ok):
            for nm, comps, dflt in (("alpha", 1, 1.0), ("uSourceDrag", 1, 0.0), ("uParticle", 3, 0.0), ("uSource", 3, 0.0)):
                ref = gu.dense(g, nm, s, Nc, comps, dflt)
                sc = np.abs(ref).max() + 1e-300
                np.testing.assert_allclose(mut[nm], ref, rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * sc, err_msg=nm)
        fy.setSourceZero()
    fy.close()
