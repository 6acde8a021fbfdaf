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

int fy_mpi_transport_destroy(fy_transport* t) {
    if (!t || !t->user) return FY_OK;
    MpiState* st = static_cast<MpiState*>(t->user);
    MPI_Comm_free(&st->foam);
    delete st;
    t->user = nullptr;
    return FY_OK;
}

}  // extern "C"
