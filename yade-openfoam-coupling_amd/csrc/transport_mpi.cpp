// fy_transport over real MPI (companion library, built only where mpi.h / libmpi exist).
#include <mpi.h>

#include <new>

#include "../../include/foamyade_mpi.h"

namespace {
struct MpiState { MPI_Comm foam; };
MPI_Datatype dt(int t) { return t == FY_T_INT ? MPI_INT : MPI_DOUBLE; }
int do_send(void*, const void* buf, int count, int dtype, int dest, int tag) {
    return MPI_Send(const_cast<void*>(buf), count, dt(dtype), dest, tag, MPI_COMM_WORLD) == MPI_SUCCESS ? 0 : 1;
}
int do_recv(void*, void* buf, int count, int dtype, int src, int tag) {
    MPI_Status st;
    return MPI_Recv(buf, count, dt(dtype), src, tag, MPI_COMM_WORLD, &st) == MPI_SUCCESS ? 0 : 1;
}
int do_bcast_world(void*, void* buf, int count, int dtype, int root) { return MPI_Bcast(buf, count, dt(dtype), root, MPI_COMM_WORLD) == MPI_SUCCESS ? 0 : 1; }
int do_bcast_local(void* u, void* buf, int count, int dtype, int root) {
    return MPI_Bcast(buf, count, dt(dtype), root, static_cast<MpiState*>(u)->foam) == MPI_SUCCESS ? 0 : 1;
}
int do_allreduce(void*, const void* in, void* out, int count, int dtype, int op) {
    return MPI_Allreduce(const_cast<void*>(in), out, count, dt(dtype), op == FY_OP_MAX ? MPI_MAX : MPI_SUM, MPI_COMM_WORLD) == MPI_SUCCESS ? 0 : 1;
}
}  // namespace

extern "C" {

int fy_mpi_transport_create(int n_yade_ranks, fy_transport* out) {
    if (!out || n_yade_ranks < 1) return FY_ERR_INVALID;
    int inited = 0;
    MPI_Initialized(&inited);
    if (!inited) return FY_ERR_TRANSPORT;
    MpiState* st = new (std::nothrow) MpiState();
    if (!st) return FY_ERR_INVALID;
    int wr = 0, ws = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &wr);
    MPI_Comm_size(MPI_COMM_WORLD, &ws);
    // colour split that leaves the Yade ranks first in WORLD (what the reference's patched Pstream provides)
    if (MPI_Comm_split(MPI_COMM_WORLD, wr < n_yade_ranks ? 2 : 1, wr, &st->foam) != MPI_SUCCESS) { delete st; return FY_ERR_TRANSPORT; }
    int lr = 0, ls = 0;
    MPI_Comm_rank(st->foam, &lr);
    MPI_Comm_size(st->foam, &ls);
    out->user = st;
    out->world_rank = wr; out->world_size = ws; out->local_rank = lr; out->local_size = ls;
    out->send = do_send; out->recv = do_recv; out->bcast_world = do_bcast_world; out->bcast_local = do_bcast_local;
    out->allreduce_world = do_allreduce;
    return FY_OK;
}

int fy_mpi_local_comm(const fy_transport* t, void* mpi_comm_out) {
    if (!t || !t->user || !mpi_comm_out) return FY_ERR_INVALID;
    *static_cast<MPI_Comm*>(mpi_comm_out) = static_cast<MpiState*>(t->user)->foam;
    return FY_OK;
}

namespace {
// fy_comm_callbacks over an MPI communicator of the solver ranks: the planes have been staged to host memory by the library
struct CommState { MPI_Comm c; int rank, size; };
int cb_sendrecv(void* u, const double* su, size_t nu, double* rd, size_t md, const double* sd, size_t nd, double* ru, size_t mu) {
    CommState* st = static_cast<CommState*>(u);
    MPI_Request rq[4];
    int n = 0;
    const int up = st->rank + 1, down = st->rank - 1;
    if (md) MPI_Irecv(rd, (int)md, MPI_DOUBLE, down, 71, st->c, &rq[n++]);
    if (mu) MPI_Irecv(ru, (int)mu, MPI_DOUBLE, up, 72, st->c, &rq[n++]);
    if (nu) MPI_Isend(const_cast<double*>(su), (int)nu, MPI_DOUBLE, up, 71, st->c, &rq[n++]);
    if (nd) MPI_Isend(const_cast<double*>(sd), (int)nd, MPI_DOUBLE, down, 72, st->c, &rq[n++]);
    return MPI_Waitall(n, rq, MPI_STATUSES_IGNORE) == MPI_SUCCESS ? 0 : 1;
}
int cb_allreduce(void* u, double* buf, int n, int is_max) {
    return MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, is_max ? MPI_MAX : MPI_SUM, static_cast<CommState*>(u)->c) == MPI_SUCCESS ? 0 : 1;
}
int cb_allgather(void* u, const double* send, double* recv, size_t cnt) {
    return MPI_Allgather(const_cast<double*>(send), (int)cnt, MPI_DOUBLE, recv, (int)cnt, MPI_DOUBLE, static_cast<CommState*>(u)->c) == MPI_SUCCESS ? 0 : 1;
}
}  // namespace

int fy_mpi_comm_create(const void* mpi_comm, int use_rccl, int device_ordinal, fy_comm** out) {
    if (!mpi_comm || !out) return FY_ERR_INVALID;
    const MPI_Comm c = *static_cast<const MPI_Comm*>(mpi_comm);
    int rank = 0, size = 1;
    MPI_Comm_rank(c, &rank);
    MPI_Comm_size(c, &size);
    if (use_rccl) {
        // one GPU per rank: the planes go over RCCL / xGMI; MPI only carries the communicator's 128-byte id
        unsigned char id[128] = {0};
        int rc = rank == 0 ? fy_rccl_unique_id(id) : FY_OK;
        int ok = rc == FY_OK ? 1 : 0;
        MPI_Bcast(&ok, 1, MPI_INT, 0, c);
        if (!ok) return rc != FY_OK ? rc : FY_ERR_TRANSPORT;
        MPI_Bcast(id, 128, MPI_BYTE, 0, c);
        return fy_comm_create_rccl(rank, size, id, device_ordinal, out);
    }
    CommState* st = new (std::nothrow) CommState{c, rank, size};
    if (!st) return FY_ERR_INVALID;
    fy_comm_callbacks cb{};
    cb.user = st; cb.sendrecv = cb_sendrecv; cb.allreduce = cb_allreduce; cb.allgather = cb_allgather;
    return fy_comm_create_host(rank, size, &cb, out);      // (st lives as long as the process: the communicator keeps the pointer)
}

int fy_mpi_transport_destroy(fy_transport* t) {
    if (!t || !t->user) return FY_OK;
    MpiState* st = static_cast<MpiState*>(t->user);
    MPI_Comm_free(&st->foam);
    delete st;
    t->user = nullptr;
    return FY_OK;
}

}  // extern "C"
