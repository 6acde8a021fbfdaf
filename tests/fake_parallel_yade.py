"""Test double of a PARALLEL Yade (a master + W workers) next to one OR SEVERAL solver ranks, through the fy_transport callbacks
(FoamYade.C:36-43, 77-111, 114-155, 239-243, 504-507, 537-549).  One instance per solver rank; every instance holds the same
deterministic workers (worker w owns records[lo_w:hi_w]) and answers its own rank's calls the way the Yade ranks would:

  * a worker sends a solver rank the particles that lie inside the bounding box THAT RANK sent at start-up (closed box, like a
    bounding-box intersection test), so a particle on a slab interface goes to both neighbours and one inside no box goes nowhere;
  * counts first (an int per solver rank, tag 1003), then the records of the intersecting workers (tag 1002);
  * search results (tag 1004) and forces (tag 1005) come back per (worker, rank) and are recorded;
  * the dt handshake goes through solver rank 0; `bcast_local` is supplied by the caller (a thread barrier for in-process virtual slabs,
    torch.distributed for one process per slab).
"""
import ctypes as C

import numpy as np

TAG_SZ, TAG_BBOX, TAG_DATA, TAG_FORCE, TAG_RES, TAG_FDT, TAG_YDT = 1003, 1001, 1002, 1005, 1004, 1050, 1060


class FakeParallelYade:
    def __init__(self, prod, workers, rank, n_solver_ranks, bcast_local=None, rec_len=10):
        self.W, self.rank, self.S, self.L = int(workers), int(rank), int(n_solver_ranks), rec_len
        self.records = None
        self.bbox = None
        self.sel = {}                          # worker -> indices (into the worker's slice) sent to this rank in the current step
        self.found, self.force = {}, {}        # worker -> what came back
        self.fluid_dt = []
        self._bcast_local = bcast_local
        T = prod.Transport()
        T.world_size = self.W + 1 + self.S
        T.world_rank = self.W + 1 + self.rank
        T.local_rank, T.local_size = self.rank, self.S
        self._cb = [prod._SEND(self.send), prod._RECV(self.recv), prod._BCAST(self.bcast_world), prod._BCAST(self.bcast_local), prod._ALLRED(self.allreduce)]
        T.send, T.recv, T.bcast_world, T.bcast_local, T.allreduce_world = self._cb
        self.T = T

    @staticmethod
    def _view(ptr, count, dtype):
        ct = C.c_int32 if dtype == 0 else C.c_double
        return np.ctypeslib.as_array((ct * count).from_address(ptr))

    def worker_range(self, w):                 # w = 1..W
        n = self.records.shape[0]
        return ((w - 1) * n) // self.W, (w * n) // self.W

    def set_records(self, rec):
        self.records = np.ascontiguousarray(rec, dtype=np.float64).reshape(-1, self.L)
        self.sel.clear(); self.found.clear(); self.force.clear()

    def _selection(self, w):
        if w not in self.sel:
            lo, hi = self.worker_range(w)
            p = self.records[lo:hi, 0:3]
            b = self.bbox
            inside = np.all((p >= b[0:3]) & (p <= b[3:6]), axis=1)
            self.sel[w] = np.nonzero(inside)[0]
        return self.sel[w]

    # ---- the transport
    def send(self, user, buf, count, dtype, dest, tag):
        v = self._view(buf, count, dtype).copy() if count else np.zeros(0)
        if tag == TAG_BBOX:
            self.bbox = v                       # (the same six doubles go to the master and to every worker)
        elif tag == TAG_RES:
            self.found[dest] = v
        elif tag == TAG_FORCE:
            self.force[dest] = v.reshape(-1, 6)
        elif tag == TAG_FDT:
            self.fluid_dt.append(float(v[0]))
        else:
            return 1
        return 0

    def recv(self, user, buf, count, dtype, src, tag):
        out = self._view(buf, count, dtype)
        if tag == TAG_SZ:
            out[:] = 0
            out[self.rank] = len(self._selection(src))       # (a real worker fills the other ranks' counts too; a rank reads only its own)
        elif tag == TAG_DATA:
            lo, _ = self.worker_range(src)
            out[:] = self.records[lo + self._selection(src)].ravel()
        elif tag == TAG_YDT:
            out[:] = 1.25e-5
        else:
            return 1
        return 0

    def bcast_world(self, user, buf, count, dtype, root):
        return 1                                # serial-Yade only

    def bcast_local(self, user, buf, count, dtype, root):
        if self._bcast_local is None:
            return 0 if self.S == 1 else 1
        self._bcast_local(self.rank, self._view(buf, count, dtype), root)
        return 0

    def allreduce(self, user, inp, out, count, dtype, op):
        return 1                                # serial-Yade only

    # ---- what the workers hold after a step: per particle of the full record set, found flag and force as this rank reported them
    def gathered(self):
        n = self.records.shape[0]
        found = np.zeros(n, dtype=np.int64)      # number of "found" answers (1) from this rank
        answers = np.zeros(n, dtype=np.int64)    # number of answers of any kind
        force = np.zeros((n, 6))
        for w, idx in self.sel.items():
            if len(idx) == 0:
                assert w not in self.found and w not in self.force      # nothing is sent to a worker that sent nothing (FoamYade.C:239-243)
                continue
            lo, _ = self.worker_range(w)
            f = self.found[w]
            assert f.shape[0] == len(idx) and self.force[w].shape[0] == len(idx)
            found[lo + idx] += (f == 1)
            answers[lo + idx] += 1
            force[lo + idx] += self.force[w]
        return found, answers, force
