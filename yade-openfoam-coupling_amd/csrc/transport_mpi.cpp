// fy_transport over real MPI (companion library, built only where mpi.h / libmpi exist).
#include <mpi.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

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

// ================================================================================================ wire helpers (foamyade_mpi.h)
namespace {
enum { W_TAG_SZ_BUFF = 1003, W_TAG_GRID_BBOX = 1001, W_TAG_YADE_DATA = 1002, W_TAG_FORCE = 1005, W_TAG_SEARCH_RES = 1004,
       W_TAG_RESULT = 2100, W_TAG_PIECE = 2200 /* + worker */ };

struct Wire {
    MPI_Comm foam = MPI_COMM_NULL;
    int K = 1, me = 0, n_yade = 1, W = 0;            // solver-side ranks, my rank among them, Yade ranks (master + W workers)
    double origin[3] = {0, 0, 0}, dx = 0, bbox[6] = {0, 0, 0, 0, 0, 0};
    int n[3] = {0, 0, 0};
    int axis = 0;                                     // the helpers' slabs cut the block across this axis
    bool have_block = false, boxes_sent = false, stopped = false;
    std::vector<int> k0, k1;                          // [K] cell layers [k0, k1) across `axis` of helper h (h = 0, the computing rank: none)
    std::vector<int> counts;                          // [W][K] this step's counts, as every solver-side rank receives them
    std::vector<long long> base;                      // [W] first record of worker w's region of the arena; [W][K] offsets of the pieces in it
    std::vector<long long> off;
    // the arena: [records cap x 10 doubles | forces cap x 6 doubles | found flags cap ints]
    char name[64] = {0};
    int fd = -1;
    char* mem = nullptr;
    size_t bytes = 0;
    long long cap = 0;
    unsigned long long generation = 0;
    // mappings the arena has outgrown: the library may still hold them page-locked (hipHostRegister) until it has seen the new generation through
    // view_region -- unmapping registered memory is undefined for the HIP runtime -- so they are unmapped one view_region call later
    struct Retired { char* mem; size_t bytes; unsigned long long gen; };
    std::vector<Retired> retired;
    unsigned long long reported_generation = 0;
    double* rec() const { return reinterpret_cast<double*>(mem); }
    double* force() const { return reinterpret_cast<double*>(mem) + 10 * cap; }
    int* found() const { return reinterpret_cast<int*>(reinterpret_cast<double*>(mem) + 16 * cap); }
    int world_of(int h) const { return n_yade + h; }
};

int wire_map(Wire& w, long long cap, bool create) {
    if (w.mem) {
        if (create) w.retired.push_back({w.mem, w.bytes, w.generation + 1});      // (the computing rank: see Wire::retired; helpers never register theirs)
        else munmap(w.mem, w.bytes);
        w.mem = nullptr;
    }
    const size_t bytes = (size_t)cap * (16 * sizeof(double) + sizeof(int)) + 4096;
    if (create) {
        if (w.fd < 0) {
            std::snprintf(w.name, sizeof(w.name), "/foamyade_wire_%ld", (long)getpid());
            shm_unlink(w.name);
            w.fd = shm_open(w.name, O_CREAT | O_RDWR, 0600);
            if (w.fd < 0) return 1;
        }
        if (ftruncate(w.fd, (off_t)bytes) != 0) return 1;
    } else {
        if (w.fd >= 0) close(w.fd);
        w.fd = shm_open(w.name, O_RDWR, 0600);
        if (w.fd < 0) return 1;
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, w.fd, 0);
    if (p == MAP_FAILED) return 1;
    w.mem = static_cast<char*>(p); w.bytes = bytes; w.cap = cap; ++w.generation;
    return 0;
}

// every solver-side rank holds the same counts, so every one reaches the same verdict: the arena grows together (collective over foam)
int wire_layout(Wire& w) {
    long long need = 0;
    w.base.assign((size_t)w.W, 0); w.off.assign((size_t)w.W * w.K, 0);
    for (int q = 0; q < w.W; ++q) {
        w.base[(size_t)q] = need;
        long long o = 0;
        for (int h = 0; h < w.K; ++h) { w.off[(size_t)q * w.K + h] = o; o += w.counts[(size_t)q * w.K + h]; }
        need += o;
    }
    if (need > w.cap) {
        const long long cap = need + need / 4 + 1024;
        int bad = 0;
        if (w.me == 0) bad = wire_map(w, cap, true);
        MPI_Bcast(w.name, (int)sizeof(w.name), MPI_CHAR, 0, w.foam);
        MPI_Barrier(w.foam);
        if (w.me != 0) bad = wire_map(w, cap, false);
        int any = 0;
        MPI_Allreduce(&bad, &any, 1, MPI_INT, MPI_MAX, w.foam);
        if (any) return 1;
    }
    return 0;
}

// cut the block's planes among the helpers and tell Yade: helpers announce their slab's box, the computing rank a box nothing can touch
int wire_announce(Wire& w) {
    double d[8] = {w.origin[0], w.origin[1], w.origin[2], w.dx, (double)w.n[0], (double)w.n[1], (double)w.n[2], w.have_block ? 1.0 : 0.0};
    MPI_Bcast(d, 8, MPI_DOUBLE, 0, w.foam);
    if (d[7] == 0.0) return 1;
    for (int a = 0; a < 3; ++a) { w.origin[a] = d[a]; w.n[a] = (int)d[4 + a]; }
    w.dx = d[3]; w.have_block = true;
    // Which way to cut: a settled or fluidised bed is stratified along gravity and even across it, and a helper's share of the wire is its
    // slab's share of the particles -- so across the longer HORIZONTAL axis, taking z for the vertical (FOAMYADE_WIRE_CUT_AXIS = 0 | 1 | 2 overrides;
    // measured on the C3 cloud, lower 60 % of the box filled: cut across z, two of four helpers carry 83 % of the records)
    w.axis = w.n[1] > w.n[0] ? 1 : 0;
    if (const char* e = getenv("FOAMYADE_WIRE_CUT_AXIS")) { const int a = atoi(e); if (a >= 0 && a <= 2) w.axis = a; }
    const int H = w.K - 1, na = w.n[w.axis];
    if (na < H) return 1;
    w.k0.assign((size_t)w.K, 0); w.k1.assign((size_t)w.K, 0);
    for (int h = 1; h < w.K; ++h) { w.k0[(size_t)h] = (int)((long long)na * (h - 1) / H); w.k1[(size_t)h] = (int)((long long)na * h / H); }
    double box[6];
    if (w.me == 0) {
        for (int a = 0; a < 6; ++a) box[a] = 1e30;                // FoamYade.C:81-95 sends min / max of mesh.points(): a point far from every particle
    } else {
        for (int a = 0; a < 3; ++a) { box[a] = w.origin[a]; box[3 + a] = w.origin[a] + w.n[a] * w.dx; }
        box[w.axis] = w.origin[w.axis] + w.k0[(size_t)w.me] * w.dx;
        box[3 + w.axis] = w.origin[w.axis] + w.k1[(size_t)w.me] * w.dx;
    }
    for (int r = 0; r < w.n_yade; ++r)
        if (MPI_Send(box, 6, MPI_DOUBLE, r, W_TAG_GRID_BBOX, MPI_COMM_WORLD) != MPI_SUCCESS) return 1;      // FoamYade.C:99-108
    w.boxes_sent = true;
    return 0;
}

int wire_recv_counts(Wire& w, int worker /* 1 .. W */) {
    if ((int)w.counts.size() != w.W * w.K) w.counts.assign((size_t)w.W * w.K, 0);
    MPI_Status st;
    return MPI_Recv(&w.counts[(size_t)(worker - 1) * w.K], w.K, MPI_INT, worker, W_TAG_SZ_BUFF, MPI_COMM_WORLD, &st) == MPI_SUCCESS ? 0 : 1;   // FoamYade.C:122-125
}

// ---- the computing rank's callbacks
int h_send(void* u, const void* buf, int count, int dtype, int dest, int tag) {
    Wire& w = *static_cast<Wire*>(u);
    if (tag == W_TAG_GRID_BBOX) {
        if (!w.boxes_sent && wire_announce(w) != 0) return 1;     // (sends this rank's box to every Yade rank; the library's own loop over them is then a no-op)
        return 0;
    }
    return MPI_Send(const_cast<void*>(buf), count, dt(dtype), dest, tag, MPI_COMM_WORLD) == MPI_SUCCESS ? 0 : 1;
}
int h_recv(void* u, void* buf, int count, int dtype, int src, int tag) {
    Wire& w = *static_cast<Wire*>(u);
    if (tag == W_TAG_SZ_BUFF) {
        if (dtype != FY_T_INT || count != 1 || src < 1 || src > w.W) return 1;
        if (src == 1) { int go = 1; MPI_Bcast(&go, 1, MPI_INT, 0, w.foam); }        // a step begins: the helpers start receiving too
        if (wire_recv_counts(w, src) != 0) return 1;
        int tot = 0;
        for (int h = 0; h < w.K; ++h) tot += w.counts[(size_t)(src - 1) * w.K + h];
        *static_cast<int*>(buf) = tot;
        if (src == w.W && wire_layout(w) != 0) return 1;
        return 0;
    }
    if (tag == W_TAG_YADE_DATA) return 1;                       // (records come as views)
    MPI_Status st;
    return MPI_Recv(buf, count, dt(dtype), src, tag, MPI_COMM_WORLD, &st) == MPI_SUCCESS ? 0 : 1;
}
int h_bcast_world(void*, void*, int, int, int) { return 1; }                         // serial-Yade calls: not with helpers
int h_allreduce(void*, const void*, void*, int, int, int) { return 1; }
// FoamYade.C:547 broadcasts yadeDT over the solver ranks: the computing rank, the only one that uses it, is the root and already holds it -- and a
// collective here would make it wait until the last helper has delivered the last answer, while it could be solving the fluid step
int h_bcast_local(void*, void*, int, int, int) { return 0; }
int h_describe_block(void* u, const double origin[3], double dx, const int32_t n[3]) {
    Wire& w = *static_cast<Wire*>(u);
    for (int a = 0; a < 3; ++a) { w.origin[a] = origin[a]; w.n[a] = n[a]; }
    w.dx = dx; w.have_block = dx > 0 && n[0] >= w.K - 1 && n[1] >= w.K - 1 && n[2] >= w.K - 1;
    return w.have_block ? 0 : 1;
}
// where worker src's records lie in the arena and how the helpers cut them (known as soon as the counts are); wait = also until every piece has landed
int view_of(Wire& w, const void** buf, int count, int dtype, int src, int tag, fy_wire_pieces* pieces, bool wait) {
    if (tag != W_TAG_YADE_DATA || dtype != FY_T_DOUBLE || src < 1 || src > w.W) return 1;
    const size_t row = (size_t)(src - 1) * w.K;
    long long tot = 0;
    int np = 0;
    for (int h = 1; h < w.K; ++h) {
        const int c = w.counts[row + h];
        if (c <= 0) continue;
        if (wait) {
            int dummy = 0;
            MPI_Status st;
            if (MPI_Recv(&dummy, 1, MPI_INT, h, W_TAG_PIECE + src, w.foam, &st) != MPI_SUCCESS || dummy != src) return 1;     // helper h has this worker's piece
        }
        if (pieces && np < FY_WIRE_MAX_PIECES) { pieces->start[np] = (int32_t)w.off[row + h]; pieces->k0[np] = w.k0[(size_t)h]; pieces->k1[np] = w.k1[(size_t)h]; }
        ++np; tot += c;
    }
    if (np > FY_WIRE_MAX_PIECES || (long long)count != 10 * tot) return 1;
    if (pieces) { pieces->n = np; pieces->axis = w.axis; }
    *buf = w.rec() + 10 * w.base[(size_t)(src - 1)];
    return 0;
}
int h_recv_view(void* u, const void** buf, int count, int dtype, int src, int tag, fy_wire_pieces* pieces) {
    return view_of(*static_cast<Wire*>(u), buf, count, dtype, src, tag, pieces, true);
}
int h_recv_view_layout(void* u, const void** buf, int count, int dtype, int src, int tag, fy_wire_pieces* pieces) {
    return view_of(*static_cast<Wire*>(u), buf, count, dtype, src, tag, pieces, false);
}
// the next piece any helper reports, whichever worker it belongs to (the helpers' reports are the only point-to-point traffic towards this rank)
int h_recv_view_next(void* u, int* src, int* piece) {
    Wire& w = *static_cast<Wire*>(u);
    int q = 0;
    MPI_Status st;
    if (MPI_Recv(&q, 1, MPI_INT, MPI_ANY_SOURCE, MPI_ANY_TAG, w.foam, &st) != MPI_SUCCESS) return 1;
    const int h = st.MPI_SOURCE;
    if (st.MPI_TAG != W_TAG_PIECE + q || q < 1 || q > w.W || h < 1 || h >= w.K) return 1;
    int idx = 0;
    for (int g = 1; g < h; ++g) idx += w.counts[(size_t)(q - 1) * w.K + g] > 0 ? 1 : 0;
    *src = q; *piece = idx;
    return 0;
}
int h_send_reserve(void* u, void** buf, int count, int dtype, int dest, int tag) {
    Wire& w = *static_cast<Wire*>(u);
    if (dest < 1 || dest > w.W) return 1;
    const long long b = w.base[(size_t)(dest - 1)];
    if (tag == W_TAG_SEARCH_RES && dtype == FY_T_INT) *buf = w.found() + b;
    else if (tag == W_TAG_FORCE && dtype == FY_T_DOUBLE) *buf = w.force() + 6 * b;
    else return 1;
    (void)count;
    return 0;
}
int h_send_commit(void* u, const void*, int, int, int dest, int tag) {
    Wire& w = *static_cast<Wire*>(u);
    if (tag != W_TAG_FORCE) return 0;                            // flags and forces are committed back to back: the helpers are told once, after the second
    for (int h = 1; h < w.K; ++h)
        if (w.counts[(size_t)(dest - 1) * w.K + h] > 0 && MPI_Send(&dest, 1, MPI_INT, h, W_TAG_RESULT, w.foam) != MPI_SUCCESS) return 1;
    return 0;
}
int h_view_region(void* u, void** base, size_t* bytes, uint64_t* generation) {
    Wire& w = *static_cast<Wire*>(u);
    // what was retired before the generation the library saw LAST time has been unregistered by it since (Coupling::lock_view_region)
    for (size_t q = 0; q < w.retired.size();) {
        if (w.retired[q].gen <= w.reported_generation) { munmap(w.retired[q].mem, w.retired[q].bytes); w.retired.erase(w.retired.begin() + (long)q); }
        else ++q;
    }
    *base = w.mem; *bytes = w.mem ? w.bytes : 0; *generation = w.generation;
    w.reported_generation = w.generation;
    return 0;
}
}  // namespace

extern "C" {

int fy_mpi_transport_create_wire_helpers(int n_yade_ranks, fy_transport* out, int* is_helper) {
    if (!out || !is_helper || n_yade_ranks == 0 || n_yade_ranks == 1) return FY_ERR_INVALID;     // (a master and at least one worker: the parallel-Yade protocol; < 0: derive)
    int inited = 0;
    MPI_Initialized(&inited);
    if (!inited) return FY_ERR_TRANSPORT;
    Wire* w = new (std::nothrow) Wire();
    if (!w) return FY_ERR_INVALID;
    int wr = 0, ws = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &wr);
    MPI_Comm_size(MPI_COMM_WORLD, &ws);
    if (MPI_Comm_split(MPI_COMM_WORLD, (n_yade_ranks > 0 && wr < n_yade_ranks) ? 2 : 1, wr, &w->foam) != MPI_SUCCESS) { delete w; return FY_ERR_TRANSPORT; }
    MPI_Comm_rank(w->foam, &w->me);
    MPI_Comm_size(w->foam, &w->K);
    if (n_yade_ranks < 0) n_yade_ranks = ws - w->K;
    if (n_yade_ranks < 2) { MPI_Comm_free(&w->foam); delete w; return FY_ERR_INVALID; }
    w->n_yade = n_yade_ranks; w->W = n_yade_ranks - 1;
    if (w->K < 2 || w->K - 1 > FY_WIRE_MAX_PIECES) { MPI_Comm_free(&w->foam); delete w; return FY_ERR_INVALID; }
    std::memset(out, 0, sizeof(*out));
    out->user = w;
    // what the library sees: ONE solver rank behind the Yade ranks
    out->world_rank = n_yade_ranks; out->world_size = n_yade_ranks + 1; out->local_rank = 0; out->local_size = 1;
    out->send = h_send; out->recv = h_recv; out->bcast_world = h_bcast_world; out->bcast_local = h_bcast_local; out->allreduce_world = h_allreduce;
    out->describe_block = h_describe_block; out->recv_view = h_recv_view; out->send_reserve = h_send_reserve; out->send_commit = h_send_commit;
    out->view_region = h_view_region; out->recv_view_layout = h_recv_view_layout; out->recv_view_next = h_recv_view_next;
    *is_helper = w->me != 0;
    return FY_OK;
}

// a helper's life: announce my slab's box, then per step receive the counts, my pieces of every worker's records (straight into the arena),
// tell the computing rank, and send each worker's flags and forces back out of the arena when the computing rank says they are there
int fy_mpi_wire_helper_serve(fy_transport* t) {
    if (!t || !t->user) return FY_ERR_INVALID;
    Wire& w = *static_cast<Wire*>(t->user);
    if (w.me == 0) return FY_ERR_INVALID;
    if (wire_announce(w) != 0) return FY_ERR_TRANSPORT;
    MPI_Status st;
    for (;;) {
        int go = 0;
        MPI_Bcast(&go, 1, MPI_INT, 0, w.foam);
        if (!go) break;
        for (int q = 1; q <= w.W; ++q) if (wire_recv_counts(w, q) != 0) return FY_ERR_TRANSPORT;
        if (wire_layout(w) != 0) return FY_ERR_TRANSPORT;
        // FoamYade.C:127-153 receives worker after worker.  Here every worker's piece is posted at once: K - 1 helpers that all wait for the
        // same worker form a convoy behind that worker's one sending core (measured at C3, 4 workers x 4 helpers: 47 ms in worker order for
        // what the senders can deliver in 22), so each piece is taken as it comes and reported to the computing rank under its worker's tag
        {
            std::vector<MPI_Request> rq;
            std::vector<int> who;
            for (int q = 1; q <= w.W; ++q) {
                const int c = w.counts[(size_t)(q - 1) * w.K + w.me];
                if (c <= 0) continue;
                double* dst = w.rec() + 10 * (w.base[(size_t)(q - 1)] + w.off[(size_t)(q - 1) * w.K + w.me]);
                rq.emplace_back();
                if (MPI_Irecv(dst, 10 * c, MPI_DOUBLE, q, W_TAG_YADE_DATA, MPI_COMM_WORLD, &rq.back()) != MPI_SUCCESS) return FY_ERR_TRANSPORT;
                who.push_back(q);
            }
            for (size_t left = rq.size(); left > 0; --left) {
                int idx = MPI_UNDEFINED;
                if (MPI_Waitany((int)rq.size(), rq.data(), &idx, &st) != MPI_SUCCESS || idx == MPI_UNDEFINED) return FY_ERR_TRANSPORT;
                if (MPI_Send(&who[(size_t)idx], 1, MPI_INT, 0, W_TAG_PIECE + who[(size_t)idx], w.foam) != MPI_SUCCESS) return FY_ERR_TRANSPORT;
            }
        }
        // FoamYade.C:239-243, 504-507: flags and forces back, per worker, as the computing rank reports them there.  Non-blocking: a worker takes
        // its solver ranks' answers one after the other, and a helper that blocked on a worker still busy with another helper's answer would hold
        // back the answers it has for the other workers
        {
            std::vector<MPI_Request> rq;
            std::vector<char> served((size_t)w.W, 0);
            for (int q = 1; q <= w.W; ++q) {
                const int c = w.counts[(size_t)(q - 1) * w.K + w.me];
                if (c <= 0) continue;
                // the computing rank reports a worker's answers when their copy has landed -- in ascending worker order as it is written, but nothing
                // here depends on that: whichever worker is named is served (once)
                int which = 0;
                if (MPI_Recv(&which, 1, MPI_INT, 0, W_TAG_RESULT, w.foam, &st) != MPI_SUCCESS || which < 1 || which > w.W) return FY_ERR_TRANSPORT;
                const int cw = w.counts[(size_t)(which - 1) * w.K + w.me];
                if (cw <= 0 || served[(size_t)(which - 1)]) return FY_ERR_TRANSPORT;
                served[(size_t)(which - 1)] = 1;
                const long long at = w.base[(size_t)(which - 1)] + w.off[(size_t)(which - 1) * w.K + w.me];
                rq.emplace_back();
                if (MPI_Isend(w.found() + at, cw, MPI_INT, which, W_TAG_SEARCH_RES, MPI_COMM_WORLD, &rq.back()) != MPI_SUCCESS) return FY_ERR_TRANSPORT;
                rq.emplace_back();
                if (MPI_Isend(w.force() + 6 * at, 6 * cw, MPI_DOUBLE, which, W_TAG_FORCE, MPI_COMM_WORLD, &rq.back()) != MPI_SUCCESS) return FY_ERR_TRANSPORT;
            }
            if (!rq.empty() && MPI_Waitall((int)rq.size(), rq.data(), MPI_STATUSES_IGNORE) != MPI_SUCCESS) return FY_ERR_TRANSPORT;
        }
    }
    if (w.mem) munmap(w.mem, w.bytes);
    if (w.fd >= 0) close(w.fd);
    MPI_Comm_free(&w.foam);
    delete &w;
    t->user = nullptr;
    return FY_OK;
}

}  // extern "C"

extern "C" {

int fy_mpi_transport_create(int n_yade_ranks, fy_transport* out) {
    if (!out || n_yade_ranks == 0) return FY_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));          // the optional callbacks (views, pieces, describe_block) are tested for null by the library
    int inited = 0;
    MPI_Initialized(&inited);
    if (!inited) return FY_ERR_TRANSPORT;
    MpiState* st = new (std::nothrow) MpiState();
    if (!st) return FY_ERR_INVALID;
    int wr = 0, ws = 0;
    MPI_Comm_rank(MPI_COMM_WORLD, &wr);
    MPI_Comm_size(MPI_COMM_WORLD, &ws);
    // colour split that leaves the Yade ranks first in WORLD (what the reference's patched Pstream provides).  n_yade_ranks < 0: the caller IS a
    // solver rank and the number of Yade ranks is what the reference derives, world size - solver communicator size (FoamYade.C:28)
    if (MPI_Comm_split(MPI_COMM_WORLD, (n_yade_ranks > 0 && wr < n_yade_ranks) ? 2 : 1, wr, &st->foam) != MPI_SUCCESS) { delete st; return FY_ERR_TRANSPORT; }
    int lr = 0, ls = 0;
    MPI_Comm_rank(st->foam, &lr);
    MPI_Comm_size(st->foam, &ls);
    const bool derived = n_yade_ranks < 0;
    if (derived) n_yade_ranks = ws - ls;
    if (derived && n_yade_ranks == 0) {
        // every rank of the world is a solver rank (a fluid-only -parallel run under a launcher that sets no MPI_APPNUM): nobody to couple with.
        // *out keeps null callbacks -- the caller passes a NULL transport to the library -- and fy_mpi_local_comm still hands out the communicator
        out->user = st;
        out->world_rank = wr; out->world_size = ws; out->local_rank = lr; out->local_size = ls;
        return FY_OK;
    }
    if (n_yade_ranks < 1 || wr < n_yade_ranks) { MPI_Comm_free(&st->foam); delete st; return FY_ERR_INVALID; }
    out->user = st;
    out->world_rank = wr; out->world_size = ws; out->local_rank = lr; out->local_size = ls;
    out->send = do_send; out->recv = do_recv; out->bcast_world = do_bcast_world; out->bcast_local = do_bcast_local;
    out->allreduce_world = do_allreduce;
    return FY_OK;
}

int fy_mpi_local_comm(const fy_transport* t, void* mpi_comm_out) {
    if (!t || !t->user || !mpi_comm_out) return FY_ERR_INVALID;
    if (t->recv_view == h_recv_view) { *static_cast<MPI_Comm*>(mpi_comm_out) = static_cast<Wire*>(t->user)->foam; return FY_OK; }
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
    if (use_rccl == 1) {
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
    // use_rccl == 2: the peer-store communicator (fy_comm_create_ipc) -- the ranks of ONE node map each other's device windows, the library's kernels store
    // planes and all-reduce operands into them; MPI carries the bootstrap only (the window handles, the closing barrier)
    if (use_rccl == 2) return fy_comm_create_ipc(rank, size, &cb, device_ordinal, out);
    return fy_comm_create_host(rank, size, &cb, out);      // (st lives as long as the process: the communicator keeps the pointer)
}

int fy_mpi_transport_destroy(fy_transport* t) {
    if (!t || !t->user) return FY_OK;
    if (t->recv_view == h_recv_view) {                  // the computing rank of a wire-helper group: release the helpers, drop the arena
        Wire* w = static_cast<Wire*>(t->user);
        int go = 0;
        MPI_Bcast(&go, 1, MPI_INT, 0, w->foam);
        if (w->mem) munmap(w->mem, w->bytes);
        for (auto& r : w->retired) munmap(r.mem, r.bytes);
        if (w->fd >= 0) { close(w->fd); shm_unlink(w->name); }
        MPI_Comm_free(&w->foam);
        delete w;
        t->user = nullptr;
        return FY_OK;
    }
    MpiState* st = static_cast<MpiState*>(t->user);
    MPI_Comm_free(&st->foam);
    delete st;
    t->user = nullptr;
    return FY_OK;
}

}  // extern "C"
