"""The MPI wire contract with Yade's FoamCoupling engine, exercised through the transport callbacks of
the C-ABI with an in-process fake Yade peer.  What Yade would receive is compared with what the fake Yade ranks of the REFERENCE
run received (tests/golden/*.npz: wire_owner / wire_found / wire_force / wire_bbox / wire_fluiddt), and the sequence of calls
(kind, count, type, peer, tag) is compared with the reference's call sites FoamYade.C:99-108,122-125,149-153,176,181,228,239-243,
504-531,538-549."""
import ctypes as C

import numpy as np
import pytest

import golden_cases as gc
import golden_util as gu

pytestmark = pytest.mark.gpu

TAG_SZ, TAG_BBOX, TAG_DATA, TAG_FORCE, TAG_RES, TAG_FDT, TAG_YDT = 1003, 1001, 1002, 1005, 1004, 1050, 1060


class FakeYade:
    """answers the Foam side's calls the way Yade ranks would; records everything it is sent"""

    def __init__(self, prod, c, g, step_records):
        self.prod, self.c, self.g = prod, c, g
        self.records = step_records           # list per step of (n,10) arrays
        self.W = c.n_yade - 1                 # workers in parallel mode
        self.serial = c.n_yade == 1
        self.log = []
        self.step = 0
        self.sent = {}                        # (tag, dest) -> list of arrays
        self.allred_int, self.allred_dbl = [], []
        self.bcast_stage = 0
        self.recv_stage = {}
        T = prod.Transport()
        T.world_size = c.n_yade + 1
        T.world_rank = c.n_yade              # Yade ranks first, the single Foam rank last
        T.local_rank, T.local_size = 0, 1
        self._cb = [prod._SEND(self.send), prod._RECV(self.recv), prod._BCAST(self.bcast_world), prod._BCAST(self.bcast_local),
                    prod._ALLRED(self.allreduce)]
        T.send, T.recv, T.bcast_world, T.bcast_local, T.allreduce_world = self._cb
        self.T = T

    @staticmethod
    def _view(ptr, count, dtype):
        ct = C.c_int32 if dtype == 0 else C.c_double
        return np.ctypeslib.as_array((ct * count).from_address(ptr))

    def worker_slice(self, w):
        n = self.records[self.step].shape[0]
        lo, hi = gc.split_range(n, self.W, w)
        return lo, hi

    def send(self, user, buf, count, dtype, dest, tag):
        self.log.append(("send", count, dtype, dest, tag))
        self.sent.setdefault((tag, dest), []).append(self._view(buf, count, dtype).copy())
        return 0

    def recv(self, user, buf, count, dtype, src, tag):
        self.log.append(("recv", count, dtype, src, tag))
        out = self._view(buf, count, dtype)
        if tag == TAG_SZ:
            lo, hi = self.worker_slice(src - 1)
            out[:] = hi - lo
        elif tag == TAG_DATA:
            lo, hi = self.worker_slice(src - 1)
            out[:] = self.records[self.step][lo:hi].ravel()
        elif tag == TAG_YDT:
            out[:] = 1.25e-5 * (self.step + 1)
        else:
            return 1
        return 0

    def bcast_world(self, user, buf, count, dtype, root):
        self.log.append(("bcast_world", count, dtype, root, -1))
        out = self._view(buf, count, dtype)
        if self.bcast_stage == 0:
            out[:] = self.records[self.step].shape[0]
        elif self.bcast_stage == 1:
            out[:] = self.records[self.step].ravel()
        elif self.bcast_stage >= 99:
            out[:] = self.finalize_value               # finalizeRun
        else:
            out[:] = 1.25e-5 * (self.step + 1)          # yadeDT
        self.bcast_stage += 1
        return 0

    def bcast_local(self, user, buf, count, dtype, root):
        self.log.append(("bcast_local", count, dtype, root, -1))
        return 0

    def allreduce(self, user, inp, out, count, dtype, op):
        self.log.append(("allreduce", count, dtype, op, -1))
        a = self._view(inp, count, dtype)
        o = self._view(out, count, dtype)
        if dtype == 0:
            self.allred_int.append(int(a[0]))
            o[:] = np.maximum(a, -5)                   # Yade contributes a negative int
        else:
            self.allred_dbl.append(float(a[0]))
            o[:] = a + 0.0
        return 0

    def next_step(self):
        self.step += 1
        self.bcast_stage = 0
        self.log.clear(); self.sent.clear(); self.allred_int.clear(); self.allred_dbl.clear()


@pytest.mark.parametrize("name", ["g16_serial_2step", "p32_serial_c1", "g16_parallel3", "p16_parallel2", "g8_fibre_serial", "p16_fibre_parallel2"])
def test_wire_protocol_matches_reference(product, name):
    prod = product
    c = gc.CASES_BY_NAME[name]
    g = gu.load(name)
    fields = gu.check_inputs_reproducible(c, g)
    recs = [g[f"records_s{s}"] for s in range(c.nsteps)]
    yade = FakeYade(prod, c, g, recs)
    Nc = c.ncells
    mut = dict(uSourceDrag=np.full(Nc, 5.0), alpha=np.zeros(Nc), uSource=np.full((Nc, 3), 3.0), uParticle=np.full((Nc, 3), 4.0))
    mesh = prod.BlockMesh(c.nx, c.ny, c.nz, c.dx, c.origin)
    fy = prod.FoamYade(mesh, fields["U"], fields["gradP"], fields["vGrad"], fields["divT"], fields["ddtU"], c.g, mut["uSourceDrag"],
                       mut["alpha"], mut["uSource"], mut["uParticle"], bool(c.gaussian), transport=yade.T)
    fy.setScalarProperties(c.rhoP, c.rhoF, c.nu)
    L = 10
    if c.fibre:
        fy.setFibreCoupling(True)            # FoamYade::fibreCpl: 15 doubles per particle on the wire (FoamYade.C:131-136,161-165)
        L = 15
    foam = c.n_yade
    if not yade.serial:
        # sendMeshBbox: 6 doubles to every Yade rank including the master, tag 1001 (FoamYade.C:96-108)
        assert yade.log == [("send", 6, 1, r, TAG_BBOX) for r in range(c.n_yade)]
        for r in range(c.n_yade):
            np.testing.assert_array_equal(yade.sent[(TAG_BBOX, r)][0], g["wire_bbox"])
        yade.log.clear(); yade.sent.clear()
    else:
        assert yade.log == []
    for s in range(c.nsteps):
        n = recs[s].shape[0]
        fy.setParticleAction(c.dt)
        kref = g[f"k_s{s}"].astype(int)
        fref = g[f"wire_force_s{s}"]
        scale = np.abs(fref).max() + 1e-300
        if yade.serial:
            exp = [("bcast_world", 1, 0, 0, -1), ("bcast_world", L * n, 1, 0, -1)] + [("allreduce", 1, 0, 0, -1)] * n     # FoamYade.C:176,181,228
            owner = np.array(yade.allred_int)
            np.testing.assert_array_equal(np.maximum(owner, -5), g[f"wire_owner_s{s}"] * foam)     # found ? worldRank : 0
            if c.gaussian:
                exp += [("allreduce", 1, 1, 1, -1)] * (6 * n)                                       # FoamYade.C:510-516
                got = np.array(yade.allred_dbl).reshape(n, 6)
            else:
                nfound = int((kref > 0).sum())
                exp += [("send", 6, 1, 0, TAG_FORCE)] * nfound                                      # FoamYade.C:519-531
                got = np.zeros((n, 6))
                got[kref > 0] = np.array(yade.sent[(TAG_FORCE, 0)])
            exp += [("send", 1, 1, 0, TAG_FDT), ("bcast_world", 1, 1, 0, -1)]                       # FoamYade.C:538-540,549
            assert yade.log == exp
            np.testing.assert_allclose(got, fref, rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * scale)
        else:
            W = yade.W
            exp = [("recv", 1, 0, w + 1, TAG_SZ) for w in range(W)]                                 # FoamYade.C:122-125
            sl = [gc.split_range(n, W, w) for w in range(W)]
            live = [w for w in range(W) if sl[w][1] > sl[w][0]]
            exp += [("recv", L * (sl[w][1] - sl[w][0]), 1, w + 1, TAG_DATA) for w in live]         # FoamYade.C:149-153
            exp += [("send", sl[w][1] - sl[w][0], 0, w + 1, TAG_RES) for w in live]                 # FoamYade.C:239-243
            exp += [("send", 6 * (sl[w][1] - sl[w][0]), 1, w + 1, TAG_FORCE) for w in live]         # FoamYade.C:504-507
            exp += [("send", 1, 1, 0, TAG_FDT), ("recv", 1, 1, 0, TAG_YDT), ("bcast_local", 1, 1, 0, -1)]   # FoamYade.C:538-547
            assert yade.log == exp
            found = np.concatenate([yade.sent[(TAG_RES, w + 1)][0] for w in live])
            np.testing.assert_array_equal(found, g[f"wire_found_s{s}"])
            got = np.concatenate([yade.sent[(TAG_FORCE, w + 1)][0] for w in live]).reshape(n, 6)
            np.testing.assert_allclose(got, fref, rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * scale)
        # dt handshake: Yade receives the fluid dt, Foam keeps Yade's dt (FoamYade.C:537-553)
        np.testing.assert_array_equal(yade.sent[(TAG_FDT, 0)][0], g[f"wire_fluiddt_s{s}"])
        assert fy.yadeDT == g[f"foam_yadedt_s{s}"][0]
        # fields seen by the solver afterwards
        for nm, comps, dflt in (("alpha", 1, 1.0), ("uSource", 3, 0.0)):
            ref = gu.dense(g, nm, s, Nc, comps, dflt)
            sc = np.abs(ref).max() + 1e-300
            np.testing.assert_allclose(mut[nm], ref, rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * sc)
        fy.setSourceZero()
        yade.next_step()
    # FoamYade::finalizeRun (FoamYade.C:595-599): one int broadcast from Yade's rank 0 over the world communicator; 10 = finalize
    yade.log.clear(); yade.bcast_stage = 99; yade.finalize_value = 10
    assert fy.finalizeRun() == 10
    assert yade.log == [("bcast_world", 1, 0, 0, -1)]
    fy.close()


def _engine(prod, c, yade, fields, mut):
    mesh = prod.BlockMesh(c.nx, c.ny, c.nz, c.dx, c.origin)
    fy = prod.FoamYade(mesh, fields["U"], fields["gradP"], fields["vGrad"], fields["divT"], fields["ddtU"], c.g, mut["uSourceDrag"],
                       mut["alpha"], mut["uSource"], mut["uParticle"], bool(c.gaussian), transport=yade.T)
    fy.setScalarProperties(c.rhoP, c.rhoF, c.nu)
    return fy


def test_serial_yade_with_no_particles(product):
    """N = 0 from a serial Yade: the reference still issues both broadcasts (FoamYade.C:176,181 -- the second with count 0), no
    per-particle collectives, and the dt handshake; the fields stay at their reset values"""
    c = gc.CASES_BY_NAME["g16_serial_2step"]
    g = gu.load(c.name)
    fields = gc.fluid_fields(c)
    yade = FakeYade(product, c, g, [np.zeros((0, 10)), g["records_s0"]])
    Nc = c.ncells
    mut = dict(uSourceDrag=np.full(Nc, 5.0), alpha=np.zeros(Nc), uSource=np.full((Nc, 3), 3.0), uParticle=np.full((Nc, 3), 4.0))
    fy = _engine(product, c, yade, fields, mut)
    fy.setParticleAction(c.dt)
    assert yade.log == [("bcast_world", 1, 0, 0, -1), ("bcast_world", 0, 1, 0, -1), ("send", 1, 1, 0, TAG_FDT), ("bcast_world", 1, 1, 0, -1)]
    assert np.all(mut["alpha"] == 1.0) and np.all(mut["uSource"] == 0.0) and np.all(mut["uSourceDrag"] == 0.0)
    fy.setSourceZero()
    # ... and the next step, with particles, is the golden one
    yade.next_step()
    fy.setParticleAction(c.dt)
    n = g["records_s0"].shape[0]
    got = np.array(yade.allred_dbl).reshape(n, 6)
    fref = g["wire_force_s0"]
    np.testing.assert_allclose(got, fref, rtol=gu.RTOL_GPU, atol=gu.RTOL_GPU * 1e-3 * np.abs(fref).max())
    fy.close()


class LopsidedYade(FakeYade):
    """parallel Yade whose LAST worker has no particle for this solver rank (numParticlesProc = 0, FoamYade.C:127-128)"""

    def worker_slice(self, w):
        n = self.records[self.step].shape[0]
        if w == self.W - 1:
            return n, n
        lo, hi = gc.split_range(n, self.W - 1, w)
        return lo, hi


def test_parallel_yade_worker_without_particles_is_skipped(product):
    """a worker that reports zero particles is not in inCommProcs: no data receive from it, no found / force send to it
    (FoamYade.C:127-155, 239-243, 504-507); the others are served as usual"""
    c = gc.CASES_BY_NAME["g16_parallel3"]
    g = gu.load(c.name)
    fields = gc.fluid_fields(c)
    recs = [g["records_s0"]]
    yade = LopsidedYade(product, c, g, recs)
    Nc = c.ncells
    mut = dict(uSourceDrag=np.full(Nc, 5.0), alpha=np.zeros(Nc), uSource=np.full((Nc, 3), 3.0), uParticle=np.full((Nc, 3), 4.0))
    fy = _engine(product, c, yade, fields, mut)
    yade.log.clear(); yade.sent.clear()
    fy.setParticleAction(c.dt)
    W = yade.W
    idle = W                                    # world rank of the worker without particles (workers are ranks 1..W)
    kinds = [(e[0], e[3], e[4]) for e in yade.log]
    assert [k for k in kinds if k[0] == "recv" and k[2] == TAG_SZ] == [("recv", r, TAG_SZ) for r in range(1, W + 1)]      # every worker is asked
    assert ("recv", idle, TAG_DATA) not in kinds and ("send", idle, TAG_RES) not in kinds and ("send", idle, TAG_FORCE) not in kinds
    n = recs[0].shape[0]
    F = np.zeros((n, 6)); found = np.zeros(n, dtype=int)
    for w in range(W - 1):
        lo, hi = yade.worker_slice(w)
        F[lo:hi] = yade.sent[(TAG_FORCE, w + 1)][0].reshape(-1, 6)
        found[lo:hi] = yade.sent[(TAG_RES, w + 1)][0]
    # same particles, differently split over the workers: the per-particle forces are those of the golden run up to the
    # order in which the batches deposit (setCellVolFraction is an assignment per batch, FoamYade.C:318-328), so compare found flags
    # exactly and forces of the FIRST batch only against a direct run with the same batches
    np.testing.assert_array_equal(found, np.where(g["k_s0"].astype(int) > 0, 1, -1))
    mut2 = dict(uSourceDrag=np.zeros(Nc), alpha=np.ones(Nc), uSource=np.zeros((Nc, 3)), uParticle=np.zeros((Nc, 3)))
    mesh = product.BlockMesh(c.nx, c.ny, c.nz, c.dx, c.origin)
    ref = product.FoamYade(mesh, fields["U"], fields["gradP"], fields["vGrad"], fields["divT"], fields["ddtU"], c.g, mut2["uSourceDrag"],
                           mut2["alpha"], mut2["uSource"], mut2["uParticle"], True)
    ref.setScalarProperties(c.rhoP, c.rhoF, c.nu)
    batches = [recs[0][slice(*yade.worker_slice(w))] for w in range(W - 1)]
    ref.setParticles(batches)
    ref.setParticleAction(c.dt)
    Fref = np.concatenate([ref.forces(b) for b in range(W - 1)])
    np.testing.assert_allclose(F, Fref, rtol=1e-9, atol=1e-12 * np.abs(Fref).max())
    fy.close(); ref.close()


@pytest.mark.parametrize("solver", [0, 1])
def test_general_mesh_solver_answers_yade_over_the_wire(product, solver):
    """fy_ldu_solver next to a parallel Yade through the transport callbacks: the solver's own step receives the records (sizes, then data, per worker) and sends found
    flags and forces back -- the forces an identical solver computes when it is handed the records directly.  icoFoamYade (point force, face-walk locate) with two
    workers: two batches against one, the momentum source summed in another order, so from the second step on the fields agree to rounding, not to the bit;
    pimpleFoamYade (Gaussian) with one worker: setCellVolFraction ASSIGNS per worker (FoamYade.C:318-328, the last worker that touched a cell wins), so only the same
    split gives the same void fraction.  Wavy renumbered hexahedra"""
    import types
    import poly_meshes as pm
    n, L = 8, 0.1
    mesh = pm.hex_block(n, n, n, (L, L, L), pm.wavy(0.2 * L / n, (L, L, L)), renumber_seed=6)
    rs = np.random.RandomState(9)
    npart = 900
    rec = np.zeros((npart, 10))
    rec[:, 0:3] = L * (0.02 + 0.96 * rs.random_sample((npart, 3))); rec[:, 3:6] = 0.05 * rs.standard_normal((npart, 3)); rec[:, 9] = 0.15 * L / n
    rec[:7, 2] = 3.0 * L                                   # a few far outside the mesh: not found (the Gaussian locate reaches 4.47 cell sizes beyond it: quirk Q8)
    W = 1 if solver else 2
    c = types.SimpleNamespace(n_yade=W + 1)
    yade = FakeYade(product, c, None, [rec, rec])
    kw = dict(solver=1, g=(0, 0, -9.81), u_relax=1.0) if solver else {}
    pbc = [2] * 6 if solver else [0] * 6
    mk = lambda tr: product.LduSolver(mesh, 2e-4, 1e-5, [0] * 6, [(0.2, 0, 0) if q == 3 else (0, 0, 0) for q in range(6)], pbc, transport=tr, **kw)
    wired, direct = mk(yade.T), mk(None)
    assert [e[:1] + e[2:] for e in yade.log] == [("send", 1, r, TAG_BBOX) for r in range(W + 1)]      # the mesh's bounding box to every Yade rank (FoamYade.C:96-108)
    yade.log.clear(); yade.sent.clear()
    for step in range(2):
        wired.step()
        direct.set_particles(rec); direct.step()
        found = np.concatenate([yade.sent[(TAG_RES, w + 1)][0] for w in range(W)])
        got = np.concatenate([yade.sent[(TAG_FORCE, w + 1)][0] for w in range(W)]).reshape(npart, 6)
        np.testing.assert_array_equal(found, direct.found())
        assert (found[:7] != 1).all() and (found[7:] == 1).sum() > 0.95 * (npart - 7)
        ref = direct.forces()
        if step == 0 and not solver:                     # (point force: no sum in the way; the Gaussian deposits are summed in whatever order the lanes arrive)
            np.testing.assert_array_equal(got, ref)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9 * np.abs(ref).max())
        assert np.abs(got).max() > 0
        yade.next_step()
    np.testing.assert_allclose(wired.get("U"), direct.get("U"), rtol=0, atol=1e-9 * np.abs(direct.get("U")).max())
    wired.close(); direct.close()
