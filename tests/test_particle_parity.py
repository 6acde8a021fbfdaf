"""GPU parity of the particle half: HIP path (through the C-ABI) vs the golden vectors produced by the reference's own
code, and vs the CPU oracle on larger seeded inputs.  Index work (tree, k, stencil ids, found flags) is bit-exact;
floating point within RTOL_GPU (device exp/pow differ from glibc in the last ulps, atomics reorder per-cell sums)."""
import numpy as np
import pytest

import golden_cases as gc
import golden_util as gu

pytestmark = pytest.mark.gpu


def make_engine(product, c, fields, mut, transport=None):
    mesh = product.BlockMesh(c.nx, c.ny, c.nz, c.dx, c.origin)
    fy = product.FoamYade(mesh, fields["U"], fields["gradP"], fields["vGrad"], fields["divT"], fields["ddtU"], c.g,
                          mut["uSourceDrag"], mut["alpha"], mut["uSource"], mut["uParticle"], bool(c.gaussian), transport=transport)
    fy.setScalarProperties(c.rhoP, c.rhoF, c.nu)
    return mesh, fy


def seeded_mutable(Nc):
    # deliberately not the post-initFields state (same values the reference driver used)
    return dict(uSourceDrag=np.full(Nc, 5.0), alpha=np.zeros(Nc), uSource=np.full((Nc, 3), 3.0), uParticle=np.full((Nc, 3), 4.0))


def assert_close(a, b, rtol, what):
    scale = np.abs(b).max() + 1e-300
    np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol * 1e-3 * scale, err_msg=what)


@pytest.mark.parametrize("name", [c.name for c in gc.CASES])
def test_tree_matches_reference(product, name):
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    mesh, fy = make_engine(product, c, gc.fluid_fields(c), seeded_mutable(c.ncells))
    assert np.array_equal(fy.tree_preorder(), g["tree_preorder"])
    fy.close()


@pytest.mark.parametrize("name", [c.name for c in gc.CASES])
def test_set_particle_action_matches_reference(product, oracle, name):
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    fields = gu.check_inputs_reproducible(c, g)
    mut = seeded_mutable(c.ncells)
    mesh, fy = make_engine(product, c, fields, mut)
    # FoamYade::initFields (FoamYade.C:56-68)
    assert np.all(mut["alpha"] == 1.0) and np.all(mut["uSource"] == 0.0)
    if c.gaussian:
        assert np.all(mut["uSourceDrag"] == 0.0) and np.all(mut["uParticle"] == 0.0)
        assert fy.interpRange == g["interp_scalars"][0]
    else:
        assert np.all(mut["uSourceDrag"] == 5.0) and np.all(mut["uParticle"] == 4.0)     # untouched in point mode
    for s in range(c.nsteps):
        rec = g[f"records_s{s}"]
        n = rec.shape[0]
        off = gu.batch_offsets(c, n)
        fy.setParticles([rec[off[b]:off[b + 1]] for b in range(len(off) - 1)])
        fy.setParticleAction(c.dt)
        k = np.concatenate([fy.stencils(b)[0] for b in range(len(off) - 1)])
        ids = np.concatenate([fy.stencils(b)[1] for b in range(len(off) - 1)])
        w = np.concatenate([fy.stencils(b)[2] for b in range(len(off) - 1)])
        chain = np.concatenate([fy.stencils(b)[3] for b in range(len(off) - 1)])
        F = np.concatenate([fy.forces(b) for b in range(len(off) - 1)])
        found = np.concatenate([fy.found(b) for b in range(len(off) - 1)])
        kref = g[f"k_s{s}"].astype(np.int32)
        ok = chain <= 12                       # beyond: reference behaviour undefined (meshTree.H:66-68)
        # ---- index work, bit exact vs the reference
        assert np.array_equal(k[ok], kref[ok])
        assert np.array_equal(ids[ok], g[f"ids_s{s}"][ok])
        assert np.array_equal(found, np.where(kref > 0, 1, -1))
        # ---- floating point vs the reference
        assert_close(w[ok], g[f"w_s{s}"][ok], gu.RTOL_GPU, "weights")
        assert_close(F[ok], g[f"force_s{s}"][ok], gu.RTOL_GPU, "force/torque")
        if np.all(ok):
            for nm, comps, dflt in (("alpha", 1, 1.0), ("uSourceDrag", 1, 0.0), ("uParticle", 3, 0.0), ("uSource", 3, 0.0)):
                if not c.gaussian and nm in ("uSourceDrag", "uParticle"):
                    continue
                assert_close(mut[nm], gu.dense(g, nm, s, c.ncells, comps, dflt), gu.RTOL_GPU, nm)
            # alpha floor and touched-cell pattern are discrete: exact
            aref = gu.dense(g, "alpha", s, c.ncells, 1, 1.0)
            assert np.array_equal(mut["alpha"] == 0.1, aref == 0.1)
            assert np.array_equal(mut["alpha"] == 1.0, aref == 1.0)
        fy.setSourceZero()
        assert np.all(mut["uSource"] == 0.0) and np.all(mut["alpha"] == 1.0)
        if c.gaussian:
            assert np.all(mut["uSourceDrag"] == 0.0) and np.all(mut["uParticle"] == 0.0)
    fy.close()


@pytest.mark.parametrize("name", [c.name for c in gc.CASES if c.gaussian])
def test_optional_force_models_match_reference(product, name):
    """fy_set_force_models: Gaussian calcHydroTorque (FoamYade.C:465-479) + addedMassForce (FoamYade.C:392-413) against what the
    reference's own methods produced when oracle/ref_driver.cpp called them on top of each step."""
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    fields = gu.check_inputs_reproducible(c, g)
    mut = seeded_mutable(c.ncells)
    mesh, fy = make_engine(product, c, fields, mut)
    fy.setForceModels(product.FORCE_ADDED_MASS | product.FORCE_GAUSSIAN_TORQUE)
    for s in range(c.nsteps):
        rec = g[f"records_s{s}"]
        off = gu.batch_offsets(c, rec.shape[0])
        fy.setParticles([rec[off[b]:off[b + 1]] for b in range(len(off) - 1)])
        fy.setParticleAction(c.dt)
        chain = np.concatenate([fy.stencils(b)[3] for b in range(len(off) - 1)])
        F = np.concatenate([fy.forces(b) for b in range(len(off) - 1)])
        ok = chain <= 12
        fref = g[f"forcex_s{s}"]
        assert_close(F[ok][:, :3], fref[ok][:, :3], gu.RTOL_GPU, "force incl. added mass")
        assert_close(F[ok][:, 3:], fref[ok][:, 3:], gu.RTOL_GPU, "Gaussian torque")
        if np.all(ok):
            assert_close(mut["uSource"], gu.dense(g, "uSourcex", s, c.ncells, 3, 0.0), gu.RTOL_GPU, "uSource incl. added mass")
            # the models leave the other three fields alone
            for nm, comps, dflt in (("alpha", 1, 1.0), ("uSourceDrag", 1, 0.0), ("uParticle", 3, 0.0)):
                assert_close(mut[nm], gu.dense(g, nm, s, c.ncells, comps, dflt), gu.RTOL_GPU, nm)
        fy.setSourceZero()
    # switching them off again restores the shipped behaviour
    fy.setForceModels(0)
    rec = g["records_s0"]
    off = gu.batch_offsets(c, rec.shape[0])
    fy.setParticles([rec[off[b]:off[b + 1]] for b in range(len(off) - 1)])
    fy.setParticleAction(c.dt)
    F = np.concatenate([fy.forces(b) for b in range(len(off) - 1)])
    chain = np.concatenate([fy.stencils(b)[3] for b in range(len(off) - 1)])
    assert_close(F[chain <= 12], g["force_s0"][chain <= 12], gu.RTOL_GPU, "force, models off")
    fy.close()


def test_force_models_rejected_in_point_mode(product):
    c = gc.CASES_BY_NAME["p32_serial_c1"]
    mesh, fy = make_engine(product, c, gc.fluid_fields(c), seeded_mutable(c.ncells))
    with pytest.raises(product.FoamYadeError):
        fy.setForceModels(product.FORCE_ADDED_MASS)
    fy.close()


@pytest.mark.parametrize("dims,npart,gaussian", [((48, 40, 36), 60000, 1), ((64, 64, 64), 200000, 1), ((50, 30, 20), 50000, 0)])
def test_against_oracle_seeded(product, oracle, dims, npart, gaussian):
    """sizes the oracle finishes in seconds; inputs are seeded, nothing reads /root/reference"""
    nx, ny, nz = dims
    c = gc.Case("seeded", nx, ny, nz, 0.3, origin=(0.1, -0.2, 0.05), gaussian=gaussian, np_=npart, seed=77, cluster=2000,
                fast=500, outside=500, nu=1e-6 if gaussian else 1e-3)
    fields = gc.fluid_fields(c)
    rec = gc.particle_records(c, 0)
    mut = seeded_mutable(c.ncells)
    mesh, fy = make_engine(product, c, fields, mut)
    om = oracle.Mesh(nx, ny, nz, c.dx, c.origin)
    assert np.array_equal(fy.tree_preorder(), om.pre)
    omut = oracle.fresh_mutable(c.ncells)
    if not gaussian:
        omut["uSourceDrag"][:] = 5.0; omut["uParticle"][:] = 4.0
    ref = oracle.particle_action(om, fields, omut, rec, np.array([0, rec.shape[0]], np.int32), gaussian, c.rhoP, c.rhoF, c.nu, threads=8)
    fy.setParticles([rec])
    fy.setParticleAction(c.dt)
    k, ids, w, chain = fy.stencils(0)
    if gaussian:
        assert np.array_equal(chain, ref["chain_len"])
    assert np.array_equal(k, ref["k"])
    assert np.array_equal(ids, ref["ids"])
    assert np.array_equal(fy.found(0), ref["found"])
    assert_close(w, ref["w"], gu.RTOL_GPU, "weights")
    assert_close(fy.forces(0), ref["force"], gu.RTOL_GPU, "force")
    for nm in ("alpha", "uSource") + (("uSourceDrag", "uParticle") if gaussian else ()):
        assert_close(mut[nm], omut[nm], gu.RTOL_GPU, nm)
    assert np.array_equal(mut["alpha"] == 0.1, omut["alpha"] == 0.1)
    if gaussian:      # the candidate lists did the work: only the 500 particles outside the block (and the odd one on a face) were walked
        assert fy.locate_walk_count <= 600, fy.locate_walk_count                    # (-1: lists switched off, see test_locate_paths.py)
    fy.close()


@pytest.mark.parametrize("dims", [(96, 80, 64), (33, 47, 29)])
def test_locate_on_lattice_positions(product, oracle, dims):
    """k_locate starts every walk at a node precomputed per CELL (k_build_locate_start).  Queries that sit exactly on cell faces,
    edges, corners and centres (where the cell assignment is ambiguous and squared distances tie), next to the tree's top-level split
    planes, on and just outside the block's boundary: stencil ids and chain lengths must equal the oracle's, bit for bit."""
    nx, ny, nz = dims
    c = gc.Case("lattice", nx, ny, nz, 0.25, origin=(-0.3, 0.2, 0.0), gaussian=1, np_=10, seed=5)
    rng = np.random.default_rng(123)
    n = 60000
    ijk = np.stack([rng.integers(0, nx + 1, n), rng.integers(0, ny + 1, n), rng.integers(0, nz + 1, n)], axis=1).astype(np.float64)
    frac = rng.choice([0.0, 0.5, 1.0], size=(n, 3))                      # faces / centres
    jitter = rng.choice([0.0, 0.0, 1e-13, -1e-13, 1e-3, -1e-3], size=(n, 3))
    pos = np.asarray(c.origin) + (ijk + frac * (rng.random((n, 1)) < 0.8) + rng.random((n, 3)) * (frac == 0.5) * 0.0 + jitter) * c.dx
    # a band around the mid-planes (the root and its children split there)
    mid = np.asarray(c.origin) + (np.array([nx, ny, nz]) * 0.5 + rng.normal(0, 1.5, (n // 4, 3))) * c.dx
    pos[: n // 4] = mid
    rec = np.zeros((n, 10))
    rec[:, 0:3] = pos
    rec[:, 3:6] = rng.normal(0, 0.05, (n, 3))
    rec[:, 9] = 0.02 * c.dx
    fields = gc.fluid_fields(c)
    mut = seeded_mutable(c.ncells)
    mesh, fy = make_engine(product, c, fields, mut)
    om = oracle.Mesh(nx, ny, nz, c.dx, c.origin)
    omut = oracle.fresh_mutable(c.ncells)
    ref = oracle.particle_action(om, fields, omut, rec, np.array([0, n], np.int32), 1, c.rhoP, c.rhoF, c.nu, threads=8)
    fy.setParticles([rec])
    fy.setParticleAction(c.dt)
    k, ids, w, chain = fy.stencils(0)
    assert np.array_equal(chain, ref["chain_len"])
    assert np.array_equal(k, ref["k"])
    assert np.array_equal(ids, ref["ids"])
    assert (k > 0).sum() > 0.9 * n and (k == 0).sum() > 0
    assert fy.locate_walk_count > 0.3 * n or fy.locate_walk_count == -1          # on-face queries are the walk's (the list kernel hands them over)
    fy.close()


@pytest.mark.parametrize("dims,origin", [((40, 40, 40), (0.0, 0.0, 0.0)), ((37, 52, 21), (-3.3, 12.7, 0.4)), ((12, 9, 7), (0.1, 0.1, 0.1))])
def test_locate_lists_equal_the_walk(product, oracle, dims, origin):
    """k_locate_lists scans a per-(cell, octant) candidate list instead of walking the tree.  Queries spread over the cells, and bands
    that straddle what decides its path -- the 8e-6 dx hand-over distance from a face, the octant planes through the cell centre
    (+-1e-13 .. 1e-3 dx and exactly on them) -- must give the oracle's chains bit for bit, whichever kernel took them."""
    nx, ny, nz = dims
    c = gc.Case("lists", nx, ny, nz, 0.2, origin=origin, gaussian=1, np_=10, seed=5)
    rng = np.random.default_rng(321)
    n = 80000
    ijk = np.stack([rng.integers(0, nx, n), rng.integers(0, ny, n), rng.integers(0, nz, n)], axis=1).astype(np.float64)
    t = rng.random((n, 3))
    band = rng.choice([1e-13, 1e-9, 1e-6, 4e-6, 7.9e-6, 8.1e-6, 1.6e-5, 1e-4, 1e-3], size=(n, 3)) * rng.choice([-1.0, 1.0], size=(n, 3))
    kind = rng.integers(0, 4, size=(n, 3))                     # 0/1: anywhere, 2: next to a face, 3: next to the centre plane
    t = np.where(kind == 2, np.where(band > 0, band, 1.0 + band), t)
    t = np.where(kind == 3, 0.5 + band * (rng.random((n, 3)) < 0.9), t)
    pos = np.asarray(c.origin) + (ijk + t) * c.dx
    rec = np.zeros((n, 10))
    rec[:, 0:3] = pos
    rec[:, 3:6] = rng.normal(0, 0.05, (n, 3))
    rec[:, 9] = 0.02 * c.dx
    fields = gc.fluid_fields(c)
    mesh, fy = make_engine(product, c, fields, seeded_mutable(c.ncells))
    om = oracle.Mesh(nx, ny, nz, c.dx, c.origin)
    ref = oracle.particle_action(om, fields, oracle.fresh_mutable(c.ncells), rec, np.array([0, n], np.int32), 1, c.rhoP, c.rhoF, c.nu, threads=8)
    fy.setParticles([rec])
    fy.setParticleAction(c.dt)
    k, ids, w, chain = fy.stencils(0)
    assert np.array_equal(chain, ref["chain_len"])
    assert np.array_equal(k, ref["k"])
    assert np.array_equal(ids, ref["ids"])
    assert_close(w, ref["w"], gu.RTOL_GPU, "weights")
    walked = fy.locate_walk_count
    assert 0.05 * n < walked < 0.6 * n or walked == -1, walked            # both kernels took a share
    fy.close()


@pytest.mark.parametrize("name", [c.name for c in gc.CASES])
def test_nearest_cell_matches_reference(product, oracle, name):
    """fy_nearest_cells_host = meshTree::nearestCell (meshTree.C:66-135): bit exact against what the reference's own method returned for every
    record position of the golden cases (nn_<case>.npz), and against the oracle on lattice-aligned and far-outside queries"""
    import os
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    nn = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nn_" + name + ".npz"))
    mesh, fy = make_engine(product, c, gc.fluid_fields(c), seeded_mutable(c.ncells))
    for s in range(c.nsteps):
        rec = g[f"records_s{s}"]
        assert np.array_equal(fy.nearest_cells(rec[:, 0:3]), nn[f"nn_s{s}"])
    # adversarial queries: cell centres, face / edge / corner points (exact ties), points far outside the block
    C = gc.cell_centres(c)
    rs = np.random.RandomState(4)
    pick = C[rs.randint(0, c.ncells, 400)]
    q = np.concatenate([pick, pick + 0.5 * c.dx * rs.randint(-1, 2, (400, 3)), pick + c.dx * rs.uniform(-9, 9, (400, 3))])
    np.testing.assert_array_equal(fy.nearest_cells(q), oracle.nearest_cell(C, fy.tree_preorder(), q))
    # inside the block the nearest centre is the containing cell (uniform hex): the findCell stand-in of SURVEY.md 8a A6
    inner = np.asarray(c.origin) + rs.uniform(0.001, 0.999, (500, 3)) * np.array([c.nx, c.ny, c.nz]) * c.dx
    ijk = np.floor((inner - np.asarray(c.origin)) / c.dx).astype(int)
    np.testing.assert_array_equal(fy.nearest_cells(inner), ijk[:, 0] + c.nx * (ijk[:, 1] + c.ny * ijk[:, 2]))
    fy.close()


def test_tree_cache_is_shared_and_survives_a_stale_lock(product, tmp_path, monkeypatch):
    """FOAMYADE_TREE_CACHE_DIR: the first object publishes the k-d pre-order, later ones load it (same tree, bit for bit); a lock file left
    behind by a crashed builder (older than the staleness bound) neither stalls the next run nor stops it from publishing; a fresh lock of a
    live builder is waited for only while it is fresh.  A truncated cache file is ignored and replaced."""
    import glob
    import os
    import time
    monkeypatch.setenv("FOAMYADE_TREE_CACHE_DIR", str(tmp_path))
    c = gc.Case("cache", 20, 16, 12, 0.2, gaussian=1, np_=10, seed=5)
    fields = gc.fluid_fields(c)
    mesh, fy = make_engine(product, c, fields, seeded_mutable(c.ncells))
    pre0 = fy.tree_preorder().copy()
    fy.close()
    files = glob.glob(str(tmp_path / "fy_tree_20x16x12_*.bin"))
    assert len(files) == 1 and os.path.getsize(files[0]) == 4 * c.ncells and not glob.glob(str(tmp_path / "*.lock")) and not glob.glob(str(tmp_path / "*.tmp*"))
    mesh, fy = make_engine(product, c, fields, seeded_mutable(c.ncells))      # served from the cache
    assert np.array_equal(fy.tree_preorder(), pre0)
    fy.close()
    # a crashed builder: no cache file, a lock two minutes old
    os.remove(files[0])
    lock = files[0] + ".lock"
    open(lock, "w").close()
    old = time.time() - 120.0
    os.utime(lock, (old, old))
    t0 = time.time()
    mesh, fy = make_engine(product, c, fields, seeded_mutable(c.ncells))
    assert time.time() - t0 < 20.0                                           # no 60-second stall
    assert np.array_equal(fy.tree_preorder(), pre0)
    fy.close()
    assert os.path.exists(files[0]) and not os.path.exists(lock)             # published again, stale lock gone
    # a damaged cache file is not trusted
    with open(files[0], "r+b") as f:
        f.truncate(4 * c.ncells - 8)
    mesh, fy = make_engine(product, c, fields, seeded_mutable(c.ncells))
    assert np.array_equal(fy.tree_preorder(), pre0)
    fy.close()
    assert os.path.getsize(files[0]) == 4 * c.ncells
