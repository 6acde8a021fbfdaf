#include "comm.hpp"

#include <algorithm>
#include <array>

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace fy {

namespace {
__global__ void k_fold_gathered(const double* __restrict__ g, int size, int n, int stride, unsigned max_mask, double* __restrict__ out) {
    const int q = threadIdx.x;
    if (q >= n) return;
    const bool mx = (max_mask >> q) & 1u;
    double x = g[q];
    for (int r = 1; r < size; ++r) { const double y = g[(size_t)r * stride + q]; x = mx ? (x > y ? x : y) : x + y; }
    out[q] = x;
}
}  // namespace

int Comm::allreduce_ops(hipStream_t s, double* dev, int n, unsigned max_mask) {
    if (size == 1 || n <= 0) return FY_OK;
    if (n > 32) return fail(FY_ERR_INVALID, "allreduce_ops: at most 32 slots");
    const unsigned all = n == 32 ? 0xffffffffu : ((1u << n) - 1u);
    if ((max_mask & all) == 0) return allreduce(s, dev, n, false);
    if ((max_mask & all) == all) return allreduce(s, dev, n, true);
    const size_t need = (size_t)size * (size_t)n;
    if (ops_scratch_n < need) {
        if (ops_scratch) (void)hipFree(ops_scratch);
        ops_scratch = nullptr; ops_scratch_n = 0;
        FY_HIP(hipMalloc((void**)&ops_scratch, std::max(need, (size_t)size * 32) * sizeof(double)));
        ops_scratch_n = std::max(need, (size_t)size * 32);
    }
    const uint64_t g0 = n_allgather;
    auto it = by_tag.find(tag);
    const uint64_t t0 = it == by_tag.end() ? 0 : it->second[2];
    FY_TRY(allgather(s, dev, ops_scratch, (size_t)n));
    // (counted as what it stands for: one all-reduce)
    n_allreduce += n_allgather - g0; n_allgather = g0;
    auto& bt = by_tag[tag];
    bt[1] += bt[2] - t0; bt[2] = t0;
    hipLaunchKernelGGL(k_fold_gathered, dim3(1), dim3(32), 0, s, ops_scratch, size, n, n, max_mask, dev);
    if (hipGetLastError() != hipSuccess) return fail(FY_ERR_HIP, "allreduce_ops: fold launch failed");
    return FY_OK;
}

int SelfComm::allgather(hipStream_t s, const double* send, double* recv, size_t n) {
    if (send != recv) FY_HIP(hipMemcpyAsync(recv, send, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    return FY_OK;
}

// ================================================================================================ LocalComm
namespace {

struct Barrier {
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0;
    unsigned long gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct LocalShared {
    int n;
    Barrier bar;
    std::vector<const Comm::Xchg*> lists;
    std::vector<const double*> gather_src;
    std::vector<std::vector<double> > red;     // per-rank host staging for all-reduce (host-synchronous mode)
    // stream-ordered mode (the ranks are NOT serialised by the host, as under RCCL: a dependency the solver's schedule forgets shows as
    // different bits): every collective is a pair of events per rank -- `ready` (my send buffers are final on my stream) and
    // `done` (my reads of the others' buffers have been enqueued up to here) -- taken from a ring that is indexed by the rank's own
    // collective sequence number (all ranks issue the same collectives in the same order).  The host barriers only order the
    // hipEventRecord / hipStreamWaitEvent CALLS; nothing waits for the GPU.
    static constexpr int RING = 16;
    std::vector<std::array<hipEvent_t, RING> > ready, done;
    std::vector<int> device;
    double* red_all = nullptr;                 // [n x 32] device slots of the all-reduce, folded in rank order by every rank
    bool stream_ordered = false;               // FOAMYADE_LOCALCOMM_STREAM=1 (measured: no faster at 2 slabs, slower at 8 -- DESIGN.md 8)
    // FOAMYADE_LOCALCOMM_TURNS=1 (profiling aid): between two collectives only ONE rank at a time enqueues and runs its work -- a rank takes the turn when it leaves
    // a collective and gives it up, its device work drained, when it enters the next.  The slabs then do not share the GPU kernel by kernel, so a kernel trace shows
    // every kernel of a slab at the duration it has with the GPU to itself (tools/r05/slab_kernels.sh); the wall time means nothing in this mode.
    bool turns = false;
    std::timed_mutex turn_m;
    std::mutex init_m;
    explicit LocalShared(int n_) : n(n_), bar(n_), lists(n_), gather_src(n_), red(n_), ready(n_), done(n_), device(n_, -1) {
        for (auto& r : ready) r.fill(nullptr);
        for (auto& d : done) d.fill(nullptr);
        const char* e = std::getenv("FOAMYADE_LOCALCOMM_STREAM");
        if (e && e[0] == '1') stream_ordered = true;
        const char* t = std::getenv("FOAMYADE_LOCALCOMM_TURNS");
        if (t && t[0] == '1') { turns = true; stream_ordered = false; }
    }
    ~LocalShared() {
        for (auto& r : ready) for (hipEvent_t e : r) if (e) (void)hipEventDestroy(e);
        for (auto& d : done) for (hipEvent_t e : d) if (e) (void)hipEventDestroy(e);
        if (red_all) (void)hipFree(red_all);
    }
};

struct LocalComm : Comm {
    std::shared_ptr<LocalShared> sh;
    uint64_t seq = 0;                                          // collectives issued by this rank so far
    bool inited = false, my_turn = false;
    struct Turn {                                              // scoped: give the turn up on entry (device drained), take it again on exit
        LocalComm* c;
        explicit Turn(LocalComm* c_) : c(c_) {
            if (!c->sh->turns) return;
            (void)hipDeviceSynchronize();
            if (c->my_turn) { c->my_turn = false; c->sh->turn_m.unlock(); }
        }
        ~Turn() {
            if (!c->sh->turns) return;
            (void)hipDeviceSynchronize();                      // (the collective's own copies)
            // (bounded wait: the rank that holds the turn when its step ends only gives it up in the next step's first collective)
            c->my_turn = c->sh->turn_m.try_lock_for(std::chrono::milliseconds(100));
        }
    };
    ~LocalComm() override { if (my_turn) { my_turn = false; sh->turn_m.unlock(); } }
    int init_rank() {
        if (inited) return FY_OK;
        int dev = 0;
        FY_HIP(hipGetDevice(&dev));
        for (int q = 0; q < LocalShared::RING; ++q) {
            FY_HIP(hipEventCreateWithFlags(&sh->ready[rank][q], hipEventDisableTiming));
            FY_HIP(hipEventCreateWithFlags(&sh->done[rank][q], hipEventDisableTiming));
        }
        {
            std::lock_guard<std::mutex> lk(sh->init_m);
            sh->device[rank] = dev;
            if (!sh->red_all) FY_HIP(hipMalloc((void**)&sh->red_all, (size_t)size * 32 * sizeof(double)));
        }
        inited = true;
        sh->bar.wait();                                        // every rank's events exist before anyone waits on one
        for (int r = 0; r < size; ++r)
            if (sh->device[r] != dev) sh->stream_ordered = false;      // slabs on different devices: the host-synchronous path
        sh->bar.wait();
        return FY_OK;
    }
    // open a collective: my buffers are final at this point of my stream
    int open(hipStream_t s, int& slot) {
        FY_TRY(init_rank());
        slot = (int)(seq++ % LocalShared::RING);
        FY_HIP(hipEventRecord(sh->ready[rank][slot], s));
        return FY_OK;
    }
    // close it: nobody's later work may overwrite a buffer that `lo..hi` are still reading
    int close(hipStream_t s, int slot, int lo, int hi) {
        FY_HIP(hipEventRecord(sh->done[rank][slot], s));
        sh->bar.wait();                                        // every rank has recorded `done`; the posted lists may go
        for (int r = lo; r <= hi; ++r)
            if (r != rank && r >= 0 && r < size) FY_HIP(hipStreamWaitEvent(s, sh->done[r][slot], 0));
        return FY_OK;
    }
    int exchange_many(hipStream_t s, const Xchg* x, size_t n) override {
        count(0);
        for (size_t q = 0; q < n; ++q) exchange_bytes += sizeof(double) * ((has_up() ? x[q].su() : 0) + (has_down() ? x[q].sd() : 0));
        FY_TRY(init_rank());
        if (!sh->stream_ordered) return exchange_many_sync(s, x, n);
        int slot = 0;
        FY_TRY(open(s, slot));
        sh->lists[rank] = x;                                   // every rank posts the same number of items in the same order
        sh->bar.wait();
        if (has_down()) FY_HIP(hipStreamWaitEvent(s, sh->ready[rank - 1][slot], 0));
        if (has_up()) FY_HIP(hipStreamWaitEvent(s, sh->ready[rank + 1][slot], 0));
        int rc = FY_OK;
        for (size_t q = 0; q < n && rc == FY_OK; ++q) {
            if (has_down() && x[q].recv_from_down && x[q].rd()) {
                if (sh->lists[rank - 1][q].su() != x[q].rd()) rc = FY_ERR_TRANSPORT;
                else if (hipMemcpyAsync(x[q].recv_from_down, sh->lists[rank - 1][q].send_up, x[q].rd() * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = FY_ERR_HIP;
            }
            if (rc == FY_OK && has_up() && x[q].recv_from_up && x[q].ru()) {
                if (sh->lists[rank + 1][q].sd() != x[q].ru()) rc = FY_ERR_TRANSPORT;
                else if (hipMemcpyAsync(x[q].recv_from_up, sh->lists[rank + 1][q].send_down, x[q].ru() * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) rc = FY_ERR_HIP;
            }
        }
        // (a failing rank still takes part in the closing barrier: the others must not hang on it)
        const int rc2 = close(s, slot, rank - 1, rank + 1);
        if (rc == FY_ERR_TRANSPORT) return fail(rc, "neighbour exchange: send/receive sizes differ");
        if (rc != FY_OK) return fail(rc, "neighbour exchange: device copy failed");
        return rc2;
    }
    int exchange_many_sync(hipStream_t s, const Xchg* x, size_t n) {
        Turn turn(this);
        FY_HIP(hipStreamSynchronize(s));                       // my planes are final
        sh->lists[rank] = x;
        sh->bar.wait();
        for (size_t q = 0; q < n; ++q) {
            if (has_down() && x[q].recv_from_down && x[q].rd()) {
                if (sh->lists[rank - 1][q].su() != x[q].rd()) return fail(FY_ERR_TRANSPORT, "neighbour exchange: send/receive sizes differ");
                FY_HIP(hipMemcpyAsync(x[q].recv_from_down, sh->lists[rank - 1][q].send_up, x[q].rd() * sizeof(double), hipMemcpyDeviceToDevice, s));
            }
            if (has_up() && x[q].recv_from_up && x[q].ru()) {
                if (sh->lists[rank + 1][q].sd() != x[q].ru()) return fail(FY_ERR_TRANSPORT, "neighbour exchange: send/receive sizes differ");
                FY_HIP(hipMemcpyAsync(x[q].recv_from_up, sh->lists[rank + 1][q].send_down, x[q].ru() * sizeof(double), hipMemcpyDeviceToDevice, s));
            }
        }
        FY_HIP(hipStreamSynchronize(s));
        sh->bar.wait();                                        // nobody overwrites a send buffer that is still being read
        return FY_OK;
    }
    int allreduce(hipStream_t s, double* dev, int n, bool is_max) override {
        count(1);
        FY_TRY(init_rank());
        if (sh->stream_ordered && n <= 32) {
            // every rank parks its values in its slot of one device array and folds all slots itself, in rank order => identical
            // bits on every rank (and the bits of the host fold below)
            FY_HIP(hipMemcpyAsync(sh->red_all + (size_t)rank * 32, dev, n * sizeof(double), hipMemcpyDeviceToDevice, s));
            int slot = 0;
            FY_TRY(open(s, slot));
            sh->bar.wait();
            for (int r = 0; r < size; ++r)
                if (r != rank) FY_HIP(hipStreamWaitEvent(s, sh->ready[r][slot], 0));
            hipLaunchKernelGGL(k_fold_gathered, dim3(1), dim3(32), 0, s, sh->red_all, size, n, 32, is_max ? 0xffffffffu : 0u, dev);
            const bool bad = hipGetLastError() != hipSuccess;
            FY_TRY(close(s, slot, 0, size - 1));
            return bad ? fail(FY_ERR_HIP, "all-reduce: fold launch failed") : FY_OK;
        }
        Turn turn(this);
        std::vector<double>& mine = sh->red[rank];
        mine.resize((size_t)n);
        FY_HIP(hipMemcpyAsync(mine.data(), dev, n * sizeof(double), hipMemcpyDeviceToHost, s));
        FY_HIP(hipStreamSynchronize(s));
        sh->bar.wait();
        std::vector<double> acc(sh->red[0].begin(), sh->red[0].begin() + n);     // fixed rank order => identical on every rank
        for (int r = 1; r < size; ++r)
            for (int q = 0; q < n; ++q) acc[q] = is_max ? (acc[q] > sh->red[r][q] ? acc[q] : sh->red[r][q]) : acc[q] + sh->red[r][q];
        sh->bar.wait();                                        // all ranks have read every staging vector
        FY_HIP(hipMemcpyAsync(dev, acc.data(), n * sizeof(double), hipMemcpyHostToDevice, s));
        FY_HIP(hipStreamSynchronize(s));
        return FY_OK;
    }
    int allgather(hipStream_t s, const double* send, double* recv, size_t cnt) override {
        count(2);
        FY_TRY(init_rank());
        if (sh->stream_ordered) {
            int slot = 0;
            FY_TRY(open(s, slot));
            sh->gather_src[rank] = send;
            sh->bar.wait();
            bool bad = false;
            for (int r = 0; r < size; ++r) {
                if (r != rank) FY_HIP(hipStreamWaitEvent(s, sh->ready[r][slot], 0));
                if (recv + (size_t)r * cnt == sh->gather_src[r]) continue;        // gathered in place
                bad = bad || hipMemcpyAsync(recv + (size_t)r * cnt, sh->gather_src[r], cnt * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess;
            }
            FY_TRY(close(s, slot, 0, size - 1));
            return bad ? fail(FY_ERR_HIP, "all-gather: device copy failed") : FY_OK;
        }
        Turn turn(this);
        FY_HIP(hipStreamSynchronize(s));
        sh->gather_src[rank] = send;
        sh->bar.wait();
        for (int r = 0; r < size; ++r)
            FY_HIP(hipMemcpyAsync(recv + (size_t)r * cnt, sh->gather_src[r], cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
        FY_HIP(hipStreamSynchronize(s));
        sh->bar.wait();
        return FY_OK;
    }
    int barrier(hipStream_t s) override {
        Turn turn(this);
        FY_HIP(hipStreamSynchronize(s));
        sh->bar.wait();
        return FY_OK;
    }
};

}  // namespace

int local_comm_group_create(int n, Comm** out) {
    if (n < 1 || !out) return fail(FY_ERR_INVALID, "bad local comm group size");
    auto sh = std::make_shared<LocalShared>(n);
    for (int r = 0; r < n; ++r) {
        LocalComm* c = new LocalComm();
        c->rank = r; c->size = n; c->sh = sh;
        out[r] = c;
    }
    return FY_OK;
}

// ================================================================================================ HostComm
namespace {

struct HostComm : Comm {
    fy_comm_callbacks cb{};
    HostBuf<double> h_su, h_sd, h_rd, h_ru, h_small, h_gather;
    int exchange_many(hipStream_t s, const Xchg* x, size_t n) override {
        count(0);
        // every item of a group in the order it was posted: all ranks post the same items in the same order
        for (size_t q = 0; q < n; ++q) {
            const size_t su = has_up() && x[q].send_up ? x[q].su() : 0, sd = has_down() && x[q].send_down ? x[q].sd() : 0;
            const size_t rd = has_down() && x[q].recv_from_down ? x[q].rd() : 0, ru = has_up() && x[q].recv_from_up ? x[q].ru() : 0;
            exchange_bytes += sizeof(double) * (su + sd);
            FY_TRY(h_su.reserve(su + 1)); FY_TRY(h_sd.reserve(sd + 1)); FY_TRY(h_rd.reserve(rd + 1)); FY_TRY(h_ru.reserve(ru + 1));
            if (su) FY_HIP(hipMemcpyAsync(h_su.p, x[q].send_up, su * sizeof(double), hipMemcpyDeviceToHost, s));
            if (sd) FY_HIP(hipMemcpyAsync(h_sd.p, x[q].send_down, sd * sizeof(double), hipMemcpyDeviceToHost, s));
            FY_HIP(hipStreamSynchronize(s));
            if (cb.sendrecv(cb.user, su ? h_su.p : nullptr, su, rd ? h_rd.p : nullptr, rd, sd ? h_sd.p : nullptr, sd, ru ? h_ru.p : nullptr, ru) != 0)
                return fail(FY_ERR_TRANSPORT, "host communicator: sendrecv callback failed");
            if (rd) FY_HIP(hipMemcpyAsync(x[q].recv_from_down, h_rd.p, rd * sizeof(double), hipMemcpyHostToDevice, s));
            if (ru) FY_HIP(hipMemcpyAsync(x[q].recv_from_up, h_ru.p, ru * sizeof(double), hipMemcpyHostToDevice, s));
            FY_HIP(hipStreamSynchronize(s));             // the staging buffers are reused by the next item
        }
        return FY_OK;
    }
    int allreduce(hipStream_t s, double* dev, int n, bool is_max) override {
        count(1);
        FY_TRY(h_small.reserve((size_t)n + 1));
        FY_HIP(hipMemcpyAsync(h_small.p, dev, n * sizeof(double), hipMemcpyDeviceToHost, s));
        FY_HIP(hipStreamSynchronize(s));
        if (cb.allreduce(cb.user, h_small.p, n, is_max ? 1 : 0) != 0) return fail(FY_ERR_TRANSPORT, "host communicator: allreduce callback failed");
        FY_HIP(hipMemcpyAsync(dev, h_small.p, n * sizeof(double), hipMemcpyHostToDevice, s));
        FY_HIP(hipStreamSynchronize(s));
        return FY_OK;
    }
    int allgather(hipStream_t s, const double* send, double* recv, size_t cnt) override {
        count(2);
        FY_TRY(h_su.reserve(cnt + 1)); FY_TRY(h_gather.reserve(cnt * (size_t)size + 1));
        FY_HIP(hipMemcpyAsync(h_su.p, send, cnt * sizeof(double), hipMemcpyDeviceToHost, s));
        FY_HIP(hipStreamSynchronize(s));
        if (cb.allgather(cb.user, h_su.p, h_gather.p, cnt) != 0) return fail(FY_ERR_TRANSPORT, "host communicator: allgather callback failed");
        FY_HIP(hipMemcpyAsync(recv, h_gather.p, cnt * (size_t)size * sizeof(double), hipMemcpyHostToDevice, s));
        FY_HIP(hipStreamSynchronize(s));
        return FY_OK;
    }
    int barrier(hipStream_t s) override {
        double z = 0.0;
        FY_HIP(hipStreamSynchronize(s));
        return cb.allreduce(cb.user, &z, 1, 0) == 0 ? FY_OK : fail(FY_ERR_TRANSPORT, "host communicator: barrier failed");
    }
};

}  // namespace

int host_comm_create(int rank, int size, const fy_comm_callbacks* cb, Comm** out) {
    if (!out || !cb || !cb->sendrecv || !cb->allreduce || !cb->allgather || size < 1 || rank < 0 || rank >= size) return fail(FY_ERR_INVALID, "bad host communicator arguments");
    HostComm* c = new HostComm();
    c->rank = rank; c->size = size; c->cb = *cb;
    *out = c;
    return FY_OK;
}

// ================================================================================================ RcclComm
namespace {

// the handful of RCCL entry points we use, resolved with dlopen so that libfoamyade_hip has no link-time dependency on librccl
typedef struct { char internal[128]; } UniqueId;
typedef void* NcclComm;
struct RcclApi {
    void* h = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*CommSplit)(NcclComm, int, int, NcclComm*, void*) = nullptr;     // optional (RCCL >= 2.18)
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
const int kNcclDouble = 8;      // ncclFloat64 (rccl.h ncclDataType_t)
const int kNcclSum = 0, kNcclMax = 2;

RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.h ? &api : nullptr;
    tried = true;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return nullptr;
    api.GetUniqueId = (int (*)(UniqueId*))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(NcclComm*, int, UniqueId, int))dlsym(h, "ncclCommInitRank");
    api.CommDestroy = (int (*)(NcclComm))dlsym(h, "ncclCommDestroy");
    api.CommSplit = (int (*)(NcclComm, int, int, NcclComm*, void*))dlsym(h, "ncclCommSplit");
    api.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    api.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    api.Send = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))dlsym(h, "ncclSend");
    api.Recv = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))dlsym(h, "ncclRecv");
    api.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))dlsym(h, "ncclAllReduce");
    api.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))dlsym(h, "ncclAllGather");
    api.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.AllReduce || !api.AllGather)
        return nullptr;
    api.h = h;
    return &api;
}

#define FY_NCCL(expr)                                                                                          \
    do {                                                                                                       \
        int _r = (expr);                                                                                       \
        if (_r != 0) return fail(FY_ERR_TRANSPORT, "%s failed: %s", #expr, A->GetErrorString ? A->GetErrorString(_r) : "rccl error"); \
    } while (0)

struct RcclComm : Comm {
    RcclApi* A = nullptr;
    NcclComm comm = nullptr;
    // a second communicator of the same ranks (ncclCommSplit, colour 0) for the exchanges that overlap interior work on another stream:
    // operations on ONE ncclComm are meant to be issued from one stream at a time, two communicators may progress independently
    NcclComm comm_aux = nullptr;
    hipStream_t aux_stream = nullptr;
    ~RcclComm() override {
        if (A && comm_aux && A->CommDestroy) A->CommDestroy(comm_aux);
        if (A && comm && A->CommDestroy) A->CommDestroy(comm);
    }
    void set_aux_stream(hipStream_t s) override { aux_stream = s; }
    hipStream_t get_aux_stream() const override { return aux_stream; }
    NcclComm comm_for(hipStream_t s) const { return (comm_aux && aux_stream && s == aux_stream) ? comm_aux : comm; }
    // Both directions in ONE group: every rank posts all its sends and receives before any of them has to complete, so the
    // pairing cannot deadlock whatever order the ranks reach this call in.
    int exchange_many(hipStream_t s, const Xchg* x, size_t n) override {
        count(0);
        const NcclComm cm = comm_for(s);
        FY_NCCL(A->GroupStart());
        for (size_t q = 0; q < n; ++q) {
            exchange_bytes += sizeof(double) * ((has_up() ? x[q].su() : 0) + (has_down() ? x[q].sd() : 0));
            if (has_up() && x[q].send_up && x[q].su()) FY_NCCL(A->Send(x[q].send_up, x[q].su(), kNcclDouble, rank + 1, cm, s));
            if (has_down() && x[q].recv_from_down && x[q].rd()) FY_NCCL(A->Recv(x[q].recv_from_down, x[q].rd(), kNcclDouble, rank - 1, cm, s));
            if (has_down() && x[q].send_down && x[q].sd()) FY_NCCL(A->Send(x[q].send_down, x[q].sd(), kNcclDouble, rank - 1, cm, s));
            if (has_up() && x[q].recv_from_up && x[q].ru()) FY_NCCL(A->Recv(x[q].recv_from_up, x[q].ru(), kNcclDouble, rank + 1, cm, s));
        }
        FY_NCCL(A->GroupEnd());
        return FY_OK;
    }
    int allreduce(hipStream_t s, double* dev, int n, bool is_max) override {
        count(1);
        FY_NCCL(A->AllReduce(dev, dev, (size_t)n, kNcclDouble, is_max ? kNcclMax : kNcclSum, comm, s));
        return FY_OK;
    }
    int allgather(hipStream_t s, const double* send, double* recv, size_t cnt) override {
        count(2);
        FY_NCCL(A->AllGather(send, recv, cnt, kNcclDouble, comm, s));
        return FY_OK;
    }
    int barrier(hipStream_t s) override {
        FY_HIP(hipStreamSynchronize(s));
        return FY_OK;
    }
};

}  // namespace

int rccl_unique_id(void* out128) {
    RcclApi* A = rccl_api();
    if (!A) return fail(FY_ERR_UNSUPPORTED, "librccl could not be loaded");
    UniqueId id;
    FY_NCCL(A->GetUniqueId(&id));
    std::memcpy(out128, id.internal, 128);
    return FY_OK;
}

int rccl_comm_create(int rank, int size, const void* id128, int device, Comm** out) {
    RcclApi* A = rccl_api();
    if (!A) return fail(FY_ERR_UNSUPPORTED, "librccl could not be loaded");
    if (!out || !id128 || rank < 0 || rank >= size) return fail(FY_ERR_INVALID, "bad rccl comm arguments");
    FY_HIP(hipSetDevice(device));
    UniqueId id;
    std::memcpy(id.internal, id128, 128);
    RcclComm* c = new RcclComm();
    c->A = A; c->rank = rank; c->size = size;
    int r = A->CommInitRank(&c->comm, size, id, rank);
    if (r != 0) { delete c; return fail(FY_ERR_TRANSPORT, "ncclCommInitRank failed: %s", A->GetErrorString ? A->GetErrorString(r) : "?"); }
    // the second communicator (collective over the same ranks); without ncclCommSplit, or if it fails, the overlapped exchanges share
    // the first one as in round 1
    if (A->CommSplit && !options().no_aux_comm) {
        if (A->CommSplit(c->comm, 0, rank, &c->comm_aux, nullptr) != 0) c->comm_aux = nullptr;
    }
    *out = c;
    return FY_OK;
}


// Exercise every operation the slab solver uses on this communicator with known values: two fields in ONE grouped neighbour exchange
// (both directions), a sum and a max all-reduce, an all-gather.  Collective: every rank of the communicator calls it.  Meant to run in a
// throw-away process before the real run commits to the communicator (bench.py), so that a fabric that cannot carry the pattern shows
// up as an error or a time-out there instead of as a hung benchmark.
int comm_selftest(Comm* c, int device) {
    if (!c) return fail(FY_ERR_INVALID, "null communicator");
    FY_HIP(hipSetDevice(device));
    hipStream_t s = nullptr;
    FY_HIP(hipStreamCreate(&s));
    const size_t n = 25600;                       // one 160 x 160 plane
    const int R = c->rank, S = c->size;
    std::vector<double> h(8 * n), out(8 * n + 64 + 8 * (size_t)S);
    for (int f = 0; f < 2; ++f) for (size_t q = 0; q < n; ++q) { h[(2 * f) * n + q] = 1000.0 * R + 10.0 * f + 1.0 + 1e-3 * (double)q; h[(2 * f + 1) * n + q] = 1000.0 * R + 10.0 * f + 2.0 + 1e-3 * (double)q; }
    DevBuf<double> d, red, gat;
    FY_TRY(d.alloc_exact(8 * n)); FY_TRY(red.alloc_exact(8)); FY_TRY(gat.alloc_exact(8 * (size_t)S + 8));
    FY_HIP(hipMemcpyAsync(d.p, h.data(), 4 * n * sizeof(double), hipMemcpyHostToDevice, s));          // [f][up|down] send planes
    FY_HIP(hipMemsetAsync(d.p + 4 * n, 0, 4 * n * sizeof(double), s));                                // [f][from_down|from_up] receive planes
    c->group_begin();
    for (int f = 0; f < 2; ++f)
        FY_TRY(c->neighbour_exchange(s, d.p + (2 * f) * n, d.p + (4 + 2 * f) * n, d.p + (2 * f + 1) * n, d.p + (4 + 2 * f + 1) * n, n));
    FY_TRY(c->group_end(s));
    const double r0[4] = {(double)(R + 1), 0.5 * (double)R, -3.0, (double)((R * 7) % S)};
    FY_HIP(hipMemcpyAsync(red.p, r0, sizeof(r0), hipMemcpyHostToDevice, s));
    FY_TRY(c->allreduce(s, red.p, 3, false));
    FY_TRY(c->allreduce(s, red.p + 3, 1, true));
    {   // a diagnostics group in one collective: {sum, sum, max, sum}
        const double m0[4] = {(double)(R + 1), 0.25 * (double)R, (double)((R * 5) % S), -1.0};
        FY_HIP(hipMemcpyAsync(red.p + 4, m0, sizeof(m0), hipMemcpyHostToDevice, s));
        FY_TRY(c->allreduce_ops(s, red.p + 4, 4, 4u));
        double mo[4];
        FY_HIP(hipMemcpyAsync(mo, red.p + 4, sizeof(mo), hipMemcpyDeviceToHost, s));
        FY_HIP(hipStreamSynchronize(s));
        double a0 = 0, a1 = 0, mxv = -1;
        for (int r = 0; r < S; ++r) { a0 += r + 1; a1 += 0.25 * r; mxv = std::max(mxv, (double)((r * 5) % S)); }
        if (mo[0] != a0 || mo[1] != a1 || mo[2] != mxv || mo[3] != -1.0 * S) return fail(FY_ERR_TRANSPORT, "comm self-test: mixed sum / max all-reduce wrong on rank %d", R);
    }
    const double g0[8] = {1.0 * R, 2.0 * R, 3.0 * R, 4.0 * R, 5.0 * R, 6.0 * R, 7.0 * R, 8.0 * R};
    FY_HIP(hipMemcpyAsync(gat.p + 8 * (size_t)S, g0, sizeof(g0), hipMemcpyHostToDevice, s));
    FY_TRY(c->allgather(s, gat.p + 8 * (size_t)S, gat.p, 8));
    FY_HIP(hipMemcpyAsync(out.data(), d.p, 8 * n * sizeof(double), hipMemcpyDeviceToHost, s));
    FY_HIP(hipMemcpyAsync(out.data() + 8 * n, red.p, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    FY_HIP(hipMemcpyAsync(out.data() + 8 * n + 64, gat.p, 8 * (size_t)S * sizeof(double), hipMemcpyDeviceToHost, s));
    FY_HIP(hipStreamSynchronize(s));
    (void)hipStreamDestroy(s);
    for (int f = 0; f < 2; ++f) for (size_t q = 0; q < n; q += 997) {
        // what my lower neighbour sent UP lands in from_down, what my upper neighbour sent DOWN lands in from_up
        const double want_dn = c->has_down() ? 1000.0 * (R - 1) + 10.0 * f + 1.0 + 1e-3 * (double)q : 0.0;
        const double want_up = c->has_up() ? 1000.0 * (R + 1) + 10.0 * f + 2.0 + 1e-3 * (double)q : 0.0;
        if (out[(4 + 2 * f) * n + q] != want_dn || out[(4 + 2 * f + 1) * n + q] != want_up)
            return fail(FY_ERR_TRANSPORT, "comm self-test: neighbour exchange delivered wrong data on rank %d (field %d, element %zu)", R, f, q);
    }
    double s0 = 0, s1 = 0, mx = -1;
    for (int r = 0; r < S; ++r) { s0 += r + 1; s1 += 0.5 * r; mx = std::max(mx, (double)((r * 7) % S)); }
    const double* ro = out.data() + 8 * n;
    if (ro[0] != s0 || ro[1] != s1 || ro[2] != -3.0 * S || ro[3] != mx) return fail(FY_ERR_TRANSPORT, "comm self-test: all-reduce wrong on rank %d", R);
    for (int r = 0; r < S; ++r) for (int q = 0; q < 8; ++q)
        if (out[8 * n + 64 + 8 * (size_t)r + q] != (double)(q + 1) * r) return fail(FY_ERR_TRANSPORT, "comm self-test: all-gather wrong on rank %d", R);
    // the overlapped halo of the smoother goes over the communicator's second channel (RcclComm: the ncclCommSplit communicator) on
    // another stream while the first stream keeps working: one plane each way on an auxiliary stream, an all-reduce on a main stream
    {
        // the two streams die on every exit path, and a solver's own auxiliary stream (registered in its constructor) is put back afterwards
        struct Streams {
            Comm* c; hipStream_t s1 = nullptr, s2 = nullptr, prev = nullptr;
            ~Streams() { c->set_aux_stream(prev); if (s1) (void)hipStreamDestroy(s1); if (s2) (void)hipStreamDestroy(s2); }
        } st{c};
        st.prev = c->get_aux_stream();
        FY_HIP(hipStreamCreate(&st.s1));
        FY_HIP(hipStreamCreateWithFlags(&st.s2, hipStreamNonBlocking));
        hipStream_t s1 = st.s1, s2 = st.s2;
        c->set_aux_stream(s2);
        FY_HIP(hipMemsetAsync(d.p + 4 * n, 0, 2 * n * sizeof(double), s2));
        FY_TRY(c->neighbour_exchange(s2, d.p, d.p + 4 * n, d.p + n, d.p + 5 * n, n));
        const double one = 1.0;
        FY_HIP(hipMemcpyAsync(red.p + 4, &one, sizeof(one), hipMemcpyHostToDevice, s1));
        FY_TRY(c->allreduce(s1, red.p + 4, 1, false));
        double cnt = 0.0;
        FY_HIP(hipMemcpyAsync(&cnt, red.p + 4, sizeof(cnt), hipMemcpyDeviceToHost, s1));
        FY_HIP(hipMemcpyAsync(out.data(), d.p + 4 * n, 2 * n * sizeof(double), hipMemcpyDeviceToHost, s2));
        FY_HIP(hipStreamSynchronize(s2));
        FY_HIP(hipStreamSynchronize(s1));
        if (cnt != (double)S) return fail(FY_ERR_TRANSPORT, "comm self-test: all-reduce beside the auxiliary exchange wrong on rank %d", R);
        for (size_t q = 0; q < n; q += 997) {
            const double want_dn = c->has_down() ? 1000.0 * (R - 1) + 1.0 + 1e-3 * (double)q : 0.0;
            const double want_up = c->has_up() ? 1000.0 * (R + 1) + 2.0 + 1e-3 * (double)q : 0.0;
            if (out[q] != want_dn || out[n + q] != want_up)
                return fail(FY_ERR_TRANSPORT, "comm self-test: exchange on the auxiliary stream delivered wrong data on rank %d (element %zu)", R, q);
        }
    }
    return FY_OK;
}

}  // namespace fy
