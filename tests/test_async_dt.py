"""The dt handshake on the asynchronous wire path (FoamYade.C:605-632, 537-553).

With a zero-copy transport (fy_transport::recv_view / send_reserve / send_commit) fy_solver lets the fluid solve run while the answers cross PCIe and
hands them over as they land.  The reference sends the forces AND the fluid's dt from inside setParticleAction, before the fluid solve (FoamYade.C:630-631):
Yade's master waits for that dt right after the forces and only then starts its DEM sub-steps, which are meant to run beside the fluid solve.  So the
fluid's dt must leave before the first answer is handed over (the hand-overs happen from the solver's host waits, i.e. during the solve), and only the
blocking half -- Yade's dt coming back -- may wait for the end of the step.  The answers themselves must be those of the copying transport."""
import ctypes as C
import types

import numpy as np
import pytest

from test_wire_protocol import FakeYade, TAG_DATA, TAG_FDT, TAG_FORCE, TAG_RES, TAG_YDT

pytestmark = pytest.mark.gpu

_RECV_VIEW = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_SEND_RESERVE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int)
_SEND_COMMIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)


class ViewYade(FakeYade):
    """FakeYade whose transport also owns staging memory and hands out views of it (include/foamyade_hip.h, "optional zero-copy wire")"""

    def __init__(self, prod, c, g, step_records):
        super().__init__(prod, c, g, step_records)
        self._held, self._out = {}, {}
        self._vcb = [_RECV_VIEW(self.recv_view), _SEND_RESERVE(self.send_reserve), _SEND_COMMIT(self.send_commit)]
        self.T.recv_view, self.T.send_reserve, self.T.send_commit = (C.cast(f, C.c_void_p) for f in self._vcb)

    def recv_view(self, user, bufpp, count, dtype, src, tag, pieces):
        self.log.append(("recv_view", count, dtype, src, tag))
        if tag != TAG_DATA:
            return 1
        lo, hi = self.worker_slice(src - 1)
        a = np.ascontiguousarray(self.records[self.step][lo:hi]).ravel().copy()
        assert a.size == count
        self._held[src] = a
        bufpp[0] = a.ctypes.data
        C.memset(pieces, 0, 8)                  # fy_wire_pieces: n = 0 (one message, no cuts), axis = 0
        return 0

    def send_reserve(self, user, bufpp, count, dtype, dest, tag):
        self.log.append(("send_reserve", count, dtype, dest, tag))
        a = np.zeros(count, dtype=np.int32 if dtype == 0 else np.float64)
        self._out[(tag, dest)] = a
        bufpp[0] = a.ctypes.data
        return 0

    def send_commit(self, user, buf, count, dtype, dest, tag):
        self.log.append(("send_commit", count, dtype, dest, tag))
        a = self._out[(tag, dest)]
        assert buf == a.ctypes.data and a.size == count
        self.sent.setdefault((tag, dest), []).append(a.copy())
        return 0


def _run(product, yade_cls, rec, steps=2):
    n, L = 16, 0.1
    c = types.SimpleNamespace(n_yade=3)         # a master and two workers
    yade = yade_cls(product, c, None, [rec] * steps)
    case = product.make_case(product.FY_SOLVER_PIMPLE, n, n, n, L / n, 2e-4, 1e-5, g=(0, 0, -9.81), p_solver=1)
    s = product.Solver(case, transport=yade.T)
    out = []
    for _ in range(steps):
        yade.log.clear(); yade.sent.clear()
        s.step()
        out.append(([(e[0], e[4], e[3]) for e in yade.log], {k: [a.copy() for a in v] for k, v in yade.sent.items()}))
        yade.step += 1
    s.close()
    return out


def test_fluid_dt_leaves_before_the_answers_and_the_fluid_solve(product):
    rs = np.random.RandomState(11)
    npart, L = 4000, 0.1
    rec = np.zeros((npart, 10))
    rec[:, 0:3] = L * (0.05 + 0.9 * rs.random_sample((npart, 3))); rec[:, 2] *= 0.6
    rec[:, 3:6] = 0.02 * rs.standard_normal((npart, 3)); rec[:, 9] = 0.2 * L / 16
    views = _run(product, ViewYade, rec)
    plain = _run(product, FakeYade, rec)
    for (kinds, sent), (pkinds, psent) in zip(views, plain):
        ops = [k[:2] for k in kinds]
        commits = [q for q, k in enumerate(ops) if k[0] == "send_commit"]
        assert len(commits) == 4 and {ops[q][1] for q in commits} == {TAG_RES, TAG_FORCE}      # found flags + forces, two workers: by view, none by copy
        assert ("send", TAG_FORCE) not in ops and ("send", TAG_RES) not in ops
        i_dt = ops.index(("send", TAG_FDT))
        # An earlier worker's answers may go out while the next worker's records still come in (Coupling::recv_yade_intrs); the LAST worker's can only be
        # handed over from the fluid solve's host waits or at its end.  The fluid's dt is on its way before that ...
        last_worker = [q for q in commits if kinds[q][2] == 2]
        assert len(last_worker) == 2 and i_dt < last_worker[0], kinds
        i_ydt = ops.index(("recv", TAG_YDT))
        assert i_ydt > commits[-1]                       # ... and only Yade's dt coming back waits for the end of the step
        assert ops.count(("send", TAG_FDT)) == 1 and ops.count(("recv", TAG_YDT)) == 1
        # the copying transport keeps the reference's order: answers, then the handshake (FoamYade.C:630-631)
        pops = [k[:2] for k in pkinds]
        assert pops.index(("send", TAG_FDT)) > max(q for q, k in enumerate(pops) if k == ("send", TAG_FORCE))
        for w in (1, 2):
            np.testing.assert_array_equal(sent[(TAG_RES, w)][0], psent[(TAG_RES, w)][0])
            a, b = sent[(TAG_FORCE, w)][0], psent[(TAG_FORCE, w)][0]
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9 * np.abs(b).max())       # (the Gaussian deposits are summed in whatever order the lanes arrive)
            assert np.abs(b).max() > 0
