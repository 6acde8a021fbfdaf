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
    bool ordered_decided = false;              // stream_ordered's final value is set (LocalComm::init_rank, once, under init_m)
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
        // (a rank whose HIP call fails still takes part in both barriers -- the others must not hang on it -- and reports afterwards)
        int dev = 0;
        hipError_t err = hipGetDevice(&dev);
        for (int q = 0; q < LocalShared::RING && err == hipSuccess; ++q) {
            err = hipEventCreateWithFlags(&sh->ready[rank][q], hipEventDisableTiming);
            if (err == hipSuccess) err = hipEventCreateWithFlags(&sh->done[rank][q], hipEventDisableTiming);
        }
        {
            std::lock_guard<std::mutex> lk(sh->init_m);
            sh->device[rank] = err == hipSuccess ? dev : -2;
            if (!sh->red_all && err == hipSuccess) err = hipMalloc((void**)&sh->red_all, (size_t)size * 32 * sizeof(double));
        }
        inited = true;
        sh->bar.wait();                                        // every rank's events exist before anyone waits on one
        {
            // slabs on different devices (or a rank that failed above): the host-synchronous path -- decided once, by whoever gets here first
            std::lock_guard<std::mutex> lk(sh->init_m);
            if (!sh->ordered_decided) {
                for (int r = 0; r < size; ++r)
                    if (sh->device[r] != sh->device[0] || sh->device[r] < 0) sh->stream_ordered = false;
                sh->ordered_decided = true;
            }
        }
        sh->bar.wait();
        if (err != hipSuccess) return fail(FY_ERR_HIP, "local communicator: set-up failed on rank %d: %s", rank, hipGetErrorString(err));
        return FY_OK;
    }
    // HIP calls between two barriers of a collective: the error is kept, the barriers are still taken, the failure is returned at the end
    hipError_t pend = hipSuccess;
    void note(hipError_t e) { if (e != hipSuccess && pend == hipSuccess) pend = e; }
    int settle(const char* what) {
        if (pend == hipSuccess) return FY_OK;
        const hipError_t e = pend; pend = hipSuccess;
        return fail(FY_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    }
    // open a collective: my buffers are final at this point of my stream
    int open(hipStream_t s, int& slot) {
        FY_TRY(init_rank());
        slot = (int)(seq++ % LocalShared::RING);
        note(hipEventRecord(sh->ready[rank][slot], s));
        return FY_OK;
    }
    // close it: nobody's later work may overwrite a buffer that `lo..hi` are still reading
    int close(hipStream_t s, int slot, int lo, int hi) {
        note(hipEventRecord(sh->done[rank][slot], s));
        sh->bar.wait();                                        // every rank has recorded `done`; the posted lists may go
        for (int r = lo; r <= hi; ++r)
            if (r != rank && r >= 0 && r < size) note(hipStreamWaitEvent(s, sh->done[r][slot], 0));
        return settle("local communicator: event call failed");
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
        if (has_down()) note(hipStreamWaitEvent(s, sh->ready[rank - 1][slot], 0));
        if (has_up()) note(hipStreamWaitEvent(s, sh->ready[rank + 1][slot], 0));
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
            note(hipMemcpyAsync(sh->red_all + (size_t)rank * 32, dev, n * sizeof(double), hipMemcpyDeviceToDevice, s));
            int slot = 0;
            FY_TRY(open(s, slot));
            sh->bar.wait();
            for (int r = 0; r < size; ++r)
                if (r != rank) note(hipStreamWaitEvent(s, sh->ready[r][slot], 0));
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
                if (r != rank) note(hipStreamWaitEvent(s, sh->ready[r][slot], 0));
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

// ================================================================================================ IpcComm
// One process per slab, the planes and the scalars written STRAIGHT INTO THE PEER'S MEMORY by this rank's kernels ("prefer direct peer stores
// over the fully connected xGMI mesh"): every rank exports one device window (hipIpcGetMemHandle), opens the others' (hipIpcOpenMemHandle) and from then on
// no library stands between two GPUs -- a transfer is a copy kernel whose stores land in the neighbour's window, a flag behind them, and a copy kernel on the
// neighbour's stream that waits for the flag.  The same code runs whether the peers are the eight GPUs of an xGMI node or N processes sharing one GPU (which
// RCCL refuses: "duplicate GPU"), so this is the transport the one-GPU test box can execute with N > 1.
//   channel   one direction between two ranks: two slots in the receiver's window (message m uses slot m & 1), a flag word per slot in the receiver's window
//             (= m + 1 once message m has landed) and an acknowledgement word in the SENDER's window (= messages consumed so far).
//   push      [wait: ack >= m - 1, i.e. the slot is free] -> copy segments into the peer's slot -> the last block stores the flag (system-scope release)
//   pull      [wait: flag == m + 1] -> copy the slot out to its destinations -> the last block stores the acknowledgement into the sender's window
//   exchange  per neighbour a channel each way on each of two lanes (main stream / the overlapped-halo stream); a group larger than a slot goes in chunks,
//             push and pull interleaved chunk by chunk (a push of chunk c + 2 waits for the peer's pull of chunk c, which lies BEFORE the peer's push of
//             chunk c + 1 in its stream: every wait points at an earlier position of the other stream, so no cycle)
//   <= 32 doubles to everybody (all-reduce, the diagnostics group, small all-gathers): ONE kernel -- lane group p stores this rank's values into rank p's
//             window, flags, waits for p's, reads them, acknowledges; the fold then runs in rank order on every rank (identical bits everywhere)
// Every wait is bounded (FOAMYADE_IPC_TIMEOUT_MS, default 20 s): a peer that never arrives makes the kernel give up and set an error word the host sees at
// its next call, instead of a GPU that spins for ever.
namespace {

typedef unsigned long long u64;
constexpr int kIpcMaxSeg = 12;
struct IpcSeg { const double* src; double* dst; size_t n; };              // n doubles
struct IpcXfer {
    IpcSeg seg[kIpcMaxSeg];
    int nseg;
    const u64* wait; u64 wait_min;             // a word in MY window the peer stores to (nullptr: nothing to wait for)
    u64* signal; u64 signal_val;               // a word in the PEER's window
    unsigned int* counter;                     // blocks done (local; the last block puts it back to zero)
    unsigned int* err; long long timeout_ticks;
};

__device__ __forceinline__ bool ipc_wait(const u64* w, u64 need, long long timeout_ticks) {
    // relaxed polls (a load that bypasses the caches; an ACQUIRE per poll would invalidate them every time round), one acquire when the word has arrived
    const long long t0 = wall_clock64();
    bool ok = true;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout_ticks) { ok = false; break; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return ok;
}

__global__ __launch_bounds__(256) void k_ipc_xfer(IpcXfer d) {
    if (threadIdx.x == 0 && d.wait && !ipc_wait(d.wait, d.wait_min, d.timeout_ticks)) *d.err = 1u;
    __syncthreads();
    for (int q = 0; q < d.nseg; ++q) {
        const double* __restrict__ src = d.seg[q].src;
        double* __restrict__ dst = d.seg[q].dst;
        const size_t n = d.seg[q].n;
        if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
            const size_t n2 = n >> 1;
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256)
                reinterpret_cast<double2*>(dst)[i] = reinterpret_cast<const double2*>(src)[i];
            if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
        } else {
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();                                        // this block's stores are out before it is counted
        const unsigned int prev = atomicAdd(d.counter, 1u);
        if (prev == gridDim.x - 1) {
            *d.counter = 0u;
            __threadfence_system();
            if (d.signal) __hip_atomic_store(d.signal, d.signal_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// <= 32 doubles from every rank to every rank in one launch: thread (p, j) = p * 32 + j serves peer p, value j.  out: [size x n] in rank order
struct IpcSmall {
    int rank, size, n;
    const double* mine; double* out;
    const u64* ack_in; u64 ack_min;            // [size] in my window: what each peer has consumed of my messages
    double* peer_slot[8]; u64* peer_flag[8];   // where MY values and flag go in peer p's window (slot of this message)
    const double* my_slot[8]; const u64* my_flag[8];      // where peer p's values arrive in my window
    u64* peer_ack[8];                          // my acknowledgement word in peer p's window
    u64 msg;                                   // flag value = msg + 1, acknowledgement value = msg + 1
    unsigned int* err; long long timeout_ticks;
    // the fold (all-reduce): slots in max_mask are maxima, the others sums; fold_to == nullptr: plain all-gather
    double* fold_to; unsigned max_mask;
};
__global__ __launch_bounds__(256) void k_ipc_small(IpcSmall d) {
    const int p = threadIdx.x >> 5, j = threadIdx.x & 31;
    __shared__ double got[8 * 32];
    const bool peer = p < d.size && p != d.rank;
    double v = 0.0;
    if (p == d.rank && j < d.n) v = d.mine[j];
    if (peer) {
        if (j == 0 && !ipc_wait(d.ack_in + p, d.ack_min, d.timeout_ticks)) *d.err = 2u;       // the slot in p's window is free again
    }
    __syncthreads();
    if (peer && j < d.n) d.peer_slot[p][j] = d.mine[j];
    __syncthreads();
    if (peer && j == 0) {
        __threadfence_system();
        __hip_atomic_store(d.peer_flag[p], d.msg + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!ipc_wait(d.my_flag[p], d.msg + 1, d.timeout_ticks)) *d.err = 3u;
    }
    __syncthreads();
    if (peer && j < d.n) v = __hip_atomic_load(d.my_slot[p] + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (p < d.size && j < d.n) { got[p * 32 + j] = v; if (!d.fold_to) d.out[(size_t)p * d.n + j] = v; }
    __syncthreads();
    if (peer && j == 0) __hip_atomic_store(d.peer_ack[p], d.msg + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (d.fold_to && threadIdx.x < d.n) {
        const bool mx = (d.max_mask >> threadIdx.x) & 1u;
        double x = got[threadIdx.x];
        for (int r = 1; r < d.size; ++r) { const double y = got[r * 32 + threadIdx.x]; x = mx ? (x > y ? x : y) : x + y; }
        d.fold_to[threadIdx.x] = x;
    }
}

struct IpcComm : Comm {
    fy_comm_callbacks cb{};
    int device = 0;
    // window layout (identical on every rank): header of 64-bit words, then the slots
    static constexpr int kLanes = 2, kDirs = 2, kSlots = 2;
    size_t slot_nb = 0, slot_coll = 0;          // doubles per neighbour slot / per collective slot
    size_t win_bytes = 0;
    char* win = nullptr;                       // my window
    std::vector<char*> peer_win;               // [size] the others' windows as mapped here (nullptr: me / not a peer I talk to)
    bool ext_alloc = false;
    unsigned int* counters = nullptr;          // ring of block counters
    size_t counter_next = 0;
    static constexpr size_t kCounters = 256;
    unsigned int* err_host = nullptr; unsigned int* err_dev = nullptr;
    long long timeout_ticks = 0;
    hipStream_t aux_stream = nullptr;
    // message counts per channel
    u64 nb_tx[kLanes][kDirs] = {}, nb_rx[kLanes][kDirs] = {};      // dir 0: the neighbour below, 1: above
    std::vector<u64> coll_tx, coll_rx, small_msg;                    // bulk collective channels per peer; the small kernel's message count (one for all)
    double* small_scratch = nullptr;           // [size x 32]

    // ---- offsets inside a window
    //   words: nb_flag[lane][dir][slot] (the message that arrived FROM direction dir), nb_ack[lane][dir] (what the neighbour in direction dir has consumed
    //   of MY messages), coll_flag[src][slot], coll_ack[dst], small_flag[src][slot], small_ack[dst]
    size_t w_nb_flag(int lane, int dir, int slot) const { return ((size_t)lane * kDirs + dir) * kSlots + slot; }
    size_t w_nb_ack(int lane, int dir) const { return (size_t)kLanes * kDirs * kSlots + (size_t)lane * kDirs + dir; }
    size_t w_base2() const { return (size_t)kLanes * kDirs * kSlots + (size_t)kLanes * kDirs; }
    size_t w_coll_flag(int src, int slot) const { return w_base2() + (size_t)src * kSlots + slot; }
    size_t w_coll_ack(int dst) const { return w_base2() + (size_t)size * kSlots + dst; }
    size_t w_small_flag(int src, int slot) const { return w_base2() + (size_t)size * (kSlots + 1) + (size_t)src * kSlots + slot; }
    size_t w_small_ack(int dst) const { return w_base2() + (size_t)size * (2 * kSlots + 1) + dst; }
    size_t n_words() const { return w_base2() + (size_t)size * (2 * kSlots + 2); }
    size_t header_bytes() const { return (n_words() * sizeof(u64) + 4095) & ~(size_t)4095; }
    size_t d_nb_slot(int lane, int dir, int slot) const { return (((size_t)lane * kDirs + dir) * kSlots + slot) * slot_nb; }            // in doubles, behind the header
    size_t d_coll_slot(int src, int slot) const { return (size_t)kLanes * kDirs * kSlots * slot_nb + ((size_t)src * kSlots + slot) * slot_coll; }
    size_t d_small_slot(int src, int slot) const { return (size_t)kLanes * kDirs * kSlots * slot_nb + (size_t)size * kSlots * slot_coll + ((size_t)src * kSlots + slot) * 32; }
    size_t data_doubles() const { return (size_t)kLanes * kDirs * kSlots * slot_nb + (size_t)size * kSlots * (slot_coll + 32); }
    u64* words(char* w) const { return reinterpret_cast<u64*>(w); }
    double* data(char* w) const { return reinterpret_cast<double*>(w + header_bytes()); }

    ~IpcComm() override {
        (void)hipDeviceSynchronize();
        // nobody may unmap a window a peer's kernel could still be storing to: one last host round through the bootstrap callbacks
        if (cb.allreduce) { double z = 0.0; (void)cb.allreduce(cb.user, &z, 1, 0); }
        for (char* w : peer_win) if (w) (void)hipIpcCloseMemHandle(w);
        if (win) (void)hipFree(win);
        if (counters) (void)hipFree(counters);
        if (small_scratch) (void)hipFree(small_scratch);
        if (err_host) (void)hipHostFree(err_host);
    }
    void set_aux_stream(hipStream_t s) override { aux_stream = s; }
    hipStream_t get_aux_stream() const override { return aux_stream; }

    int check() const {
        const unsigned int e = err_host ? __atomic_load_n(err_host, __ATOMIC_RELAXED) : 0u;
        if (e) return fail(FY_ERR_TRANSPORT, "ipc communicator: a wait for a peer timed out on rank %d (code %u): the ranks did not issue the same collectives, or a peer died", rank, e);
        return FY_OK;
    }
    // (the kernels of one stream run one after the other: a small ring per lane is never shared by two kernels in flight)
    unsigned int* next_counter(hipStream_t s) {
        const size_t lane = (aux_stream && s == aux_stream) ? 1 : 0;
        return counters + lane * (kCounters / 2) + (counter_next++ % (kCounters / 2));
    }
    int launch(hipStream_t s, IpcXfer& d, size_t doubles) {
        d.counter = next_counter(s); d.err = err_dev; d.timeout_ticks = timeout_ticks;
        const unsigned blocks = (unsigned)std::min<size_t>(std::max<size_t>((doubles / 2 + 255) / 256, 1), 512);
        hipLaunchKernelGGL(k_ipc_xfer, dim3(blocks), dim3(256), 0, s, d);
        if (hipGetLastError() != hipSuccess) return fail(FY_ERR_HIP, "ipc communicator: launch failed");
        return FY_OK;
    }

    // one direction of a (grouped) transfer as a list of pieces; chunk c of it = the pieces of the byte range [c * slot, (c + 1) * slot), cut where more than
    // kIpcMaxSeg pieces would be needed (the receiver walks the same sizes, so it cuts at the same places)
    struct Piece { const double* src; double* dst; size_t n; };
    static size_t n_chunks(const std::vector<Piece>& v, size_t slot) {
        size_t chunks = 0, fill = 0; int segs = 0;
        for (const Piece& pc : v) {
            size_t left = pc.n;
            while (left) {
                if (fill == slot || segs == kIpcMaxSeg) { ++chunks; fill = 0; segs = 0; }
                const size_t take = std::min(left, slot - fill);
                fill += take; left -= take; ++segs;
            }
        }
        return chunks + (fill ? 1 : 0);
    }
    // fills d.seg / d.nseg with chunk `c`: push (to_slot = the peer's slot, pieces' src) or pull (from_slot = my slot, pieces' dst); returns the doubles in the chunk
    static size_t chunk_segs(const std::vector<Piece>& v, size_t slot, size_t c, double* slot_ptr, bool push, IpcXfer& d) {
        size_t chunk = 0, fill = 0, total = 0; int segs = 0;
        d.nseg = 0;
        for (const Piece& pc : v) {
            size_t left = pc.n, done = 0;
            while (left) {
                if (fill == slot || segs == kIpcMaxSeg) { ++chunk; fill = 0; segs = 0; }
                const size_t take = std::min(left, slot - fill);
                if (chunk == c) {
                    IpcSeg& sg = d.seg[d.nseg++];
                    if (push) { sg.src = pc.src + done; sg.dst = slot_ptr + fill; }
                    else { sg.src = slot_ptr + fill; sg.dst = pc.dst + done; }
                    sg.n = take; total += take;
                }
                fill += take; left -= take; done += take; ++segs;
            }
            if (chunk > c) break;
        }
        return total;
    }

    int push_nb(hipStream_t s, int lane, int dir, const std::vector<Piece>& v, size_t c) {
        const int peer = dir ? rank + 1 : rank - 1;
        const int their_dir = dir ? 0 : 1;                      // I am the neighbour BELOW the rank above me
        const u64 m = nb_tx[lane][dir]++;
        IpcXfer d{};
        const size_t n = chunk_segs(v, slot_nb, c, data(peer_win[peer]) + d_nb_slot(lane, their_dir, (int)(m & 1)), true, d);
        d.wait = words(win) + w_nb_ack(lane, dir); d.wait_min = m >= 1 ? m - 1 : 0;
        if (m < 2) d.wait = nullptr;
        d.signal = words(peer_win[peer]) + w_nb_flag(lane, their_dir, (int)(m & 1)); d.signal_val = m + 1;
        return launch(s, d, n);
    }
    int pull_nb(hipStream_t s, int lane, int dir, const std::vector<Piece>& v, size_t c) {
        const int peer = dir ? rank + 1 : rank - 1;
        const int their_dir = dir ? 0 : 1;
        const u64 m = nb_rx[lane][dir]++;
        IpcXfer d{};
        const size_t n = chunk_segs(v, slot_nb, c, data(win) + d_nb_slot(lane, dir, (int)(m & 1)), false, d);
        d.wait = words(win) + w_nb_flag(lane, dir, (int)(m & 1)); d.wait_min = m + 1;
        d.signal = words(peer_win[peer]) + w_nb_ack(lane, their_dir); d.signal_val = m + 1;
        return launch(s, d, n);
    }

    int exchange_many(hipStream_t s, const Xchg* x, size_t n) override {
        count(0);
        FY_TRY(check());
        const int lane = (aux_stream && s == aux_stream) ? 1 : 0;
        std::vector<Piece> tx[2], rx[2];                       // [dir]
        for (size_t q = 0; q < n; ++q) {
            const size_t su = has_up() ? x[q].su() : 0, sd = has_down() ? x[q].sd() : 0, rd = has_down() ? x[q].rd() : 0, ru = has_up() ? x[q].ru() : 0;
            exchange_bytes += sizeof(double) * (su + sd);
            if ((su && !x[q].send_up) || (sd && !x[q].send_down) || (rd && !x[q].recv_from_down) || (ru && !x[q].recv_from_up))
                return fail(FY_ERR_INVALID, "ipc communicator: a neighbour exchange with a count but no buffer");
            if (su) tx[1].push_back(Piece{x[q].send_up, nullptr, su});
            if (sd) tx[0].push_back(Piece{x[q].send_down, nullptr, sd});
            if (rd) rx[0].push_back(Piece{nullptr, x[q].recv_from_down, rd});
            if (ru) rx[1].push_back(Piece{nullptr, x[q].recv_from_up, ru});
        }
        size_t ct[2] = {n_chunks(tx[0], slot_nb), n_chunks(tx[1], slot_nb)}, cr[2] = {n_chunks(rx[0], slot_nb), n_chunks(rx[1], slot_nb)};
        const size_t rounds = std::max(std::max(ct[0], ct[1]), std::max(cr[0], cr[1]));
        for (size_t c = 0; c < rounds; ++c) {
            for (int dir = 1; dir >= 0; --dir) if (c < ct[dir]) FY_TRY(push_nb(s, lane, dir, tx[dir], c));
            for (int dir = 0; dir < 2; ++dir) if (c < cr[dir]) FY_TRY(pull_nb(s, lane, dir, rx[dir], c));
        }
        return FY_OK;
    }

    // <= 32 doubles per rank
    int small(hipStream_t s, const double* mine, int n, double* out, double* fold_to, unsigned max_mask) {
        IpcSmall d{};
        d.rank = rank; d.size = size; d.n = n; d.mine = mine; d.out = out; d.fold_to = fold_to; d.max_mask = max_mask;
        const u64 m = small_msg[0]++;
        d.msg = m;
        d.ack_in = words(win) + w_small_ack(0); d.ack_min = m >= 1 ? m - 1 : 0;
        for (int p = 0; p < size; ++p) {
            if (p == rank) continue;
            d.peer_slot[p] = data(peer_win[p]) + d_small_slot(rank, (int)(m & 1)); d.peer_flag[p] = words(peer_win[p]) + w_small_flag(rank, (int)(m & 1));
            d.my_slot[p] = data(win) + d_small_slot(p, (int)(m & 1)); d.my_flag[p] = words(win) + w_small_flag(p, (int)(m & 1));
            d.peer_ack[p] = words(peer_win[p]) + w_small_ack(rank);
        }
        d.err = err_dev; d.timeout_ticks = timeout_ticks;
        hipLaunchKernelGGL(k_ipc_small, dim3(1), dim3(256), 0, s, d);
        if (hipGetLastError() != hipSuccess) return fail(FY_ERR_HIP, "ipc communicator: launch failed");
        return FY_OK;
    }
    int allreduce(hipStream_t s, double* dev, int n, bool is_max) override {
        count(1);
        FY_TRY(check());
        if (n > 32) return fail(FY_ERR_INVALID, "ipc communicator: all-reduce of more than 32 doubles");
        // (the kernel reads `mine` = dev before any thread stores the fold to dev: the stores sit behind three block barriers)
        return small(s, dev, n, nullptr, dev, is_max ? 0xffffffffu : 0u);
    }
    int allgather(hipStream_t s, const double* send, double* recv, size_t cnt) override {
        count(2);
        FY_TRY(check());
        if (cnt <= 32 && send != recv + (size_t)rank * cnt) return small(s, send, (int)cnt, recv, nullptr, 0u);
        if (cnt <= 32) {                                         // gathered in place: through the scratch
            FY_HIP(hipMemcpyAsync(small_scratch, send, cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
            return small(s, small_scratch, (int)cnt, recv, nullptr, 0u);
        }
        std::vector<Piece> tx{Piece{send, nullptr, cnt}};
        const size_t chunks = n_chunks(tx, slot_coll);
        for (size_t c = 0; c < chunks; ++c) {
            for (int p = 0; p < size; ++p) {
                if (p == rank) continue;
                const u64 m = coll_tx[(size_t)p]++;
                IpcXfer d{};
                const size_t nn = chunk_segs(tx, slot_coll, c, data(peer_win[(size_t)p]) + d_coll_slot(rank, (int)(m & 1)), true, d);
                d.wait = m >= 2 ? words(win) + w_coll_ack(p) : nullptr; d.wait_min = m >= 1 ? m - 1 : 0;
                d.signal = words(peer_win[(size_t)p]) + w_coll_flag(rank, (int)(m & 1)); d.signal_val = m + 1;
                FY_TRY(launch(s, d, nn));
            }
            for (int p = 0; p < size; ++p) {
                if (p == rank) continue;
                const u64 m = coll_rx[(size_t)p]++;
                std::vector<Piece> rx{Piece{nullptr, recv + (size_t)p * cnt, cnt}};
                IpcXfer d{};
                const size_t nn = chunk_segs(rx, slot_coll, c, data(win) + d_coll_slot(p, (int)(m & 1)), false, d);
                d.wait = words(win) + w_coll_flag(p, (int)(m & 1)); d.wait_min = m + 1;
                d.signal = words(peer_win[(size_t)p]) + w_coll_ack(rank); d.signal_val = m + 1;
                FY_TRY(launch(s, d, nn));
            }
        }
        if (recv + (size_t)rank * cnt != send) FY_HIP(hipMemcpyAsync(recv + (size_t)rank * cnt, send, cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
        return FY_OK;
    }
    int barrier(hipStream_t s) override {
        FY_HIP(hipStreamSynchronize(s));
        double z = 0.0;
        if (cb.allreduce(cb.user, &z, 1, 0) != 0) return fail(FY_ERR_TRANSPORT, "ipc communicator: barrier failed");
        return check();
    }
};

}  // namespace

int ipc_comm_create(int rank, int size, const fy_comm_callbacks* cb, int device, Comm** out) {
    if (!out || !cb || !cb->allreduce || !cb->allgather || size < 1 || size > 8 || rank < 0 || rank >= size)
        return fail(FY_ERR_INVALID, "bad ipc communicator arguments (1 .. 8 ranks; the callbacks carry the bootstrap: allgather + allreduce)");
    FY_HIP(hipSetDevice(device));
    std::unique_ptr<IpcComm> c(new IpcComm());
    c->rank = rank; c->size = size; c->cb = *cb; c->device = device;
    const char* e = std::getenv("FOAMYADE_IPC_SLOT_MB");
    const size_t mb = e && std::atoi(e) > 0 ? (size_t)std::atoi(e) : 8;
    c->slot_nb = mb * (1u << 20) / sizeof(double);
    const char* ek = std::getenv("FOAMYADE_IPC_SLOT_KB");      // (tests: slots small enough that the test meshes' groups travel in chunks)
    if (ek && std::atoi(ek) > 0) c->slot_nb = (size_t)std::atoi(ek) * 1024 / sizeof(double);
    c->slot_coll = std::min<size_t>((256u << 10) / sizeof(double), c->slot_nb);
    const char* t = std::getenv("FOAMYADE_IPC_TIMEOUT_MS");
    const long long ms = t && std::atoll(t) > 0 ? std::atoll(t) : 20000;
    c->timeout_ticks = ms * 100000;                               // wall_clock64: 100 MHz
    c->win_bytes = c->header_bytes() + c->data_doubles() * sizeof(double);
    // uncached device memory where the runtime offers it (a peer GPU's stores must not meet stale lines of this GPU's L2), plain device memory otherwise
    hipIpcMemHandle_t mine;
    void* w = nullptr;
    if (hipExtMallocWithFlags(&w, c->win_bytes, hipDeviceMallocUncached) == hipSuccess && hipIpcGetMemHandle(&mine, w) == hipSuccess) c->ext_alloc = true;
    else {
        (void)hipGetLastError();
        if (w) (void)hipFree(w);
        w = nullptr;
        FY_HIP(hipMalloc(&w, c->win_bytes));
        FY_HIP(hipIpcGetMemHandle(&mine, w));
    }
    c->win = static_cast<char*>(w);
    FY_HIP(hipMemset(c->win, 0, c->header_bytes()));
    FY_HIP(hipMalloc((void**)&c->counters, IpcComm::kCounters * sizeof(unsigned int)));
    FY_HIP(hipMemset(c->counters, 0, IpcComm::kCounters * sizeof(unsigned int)));
    FY_HIP(hipMalloc((void**)&c->small_scratch, 32 * sizeof(double)));
    FY_HIP(hipHostMalloc((void**)&c->err_host, sizeof(unsigned int), hipHostMallocMapped));
    *c->err_host = 0u;
    FY_HIP(hipHostGetDevicePointer((void**)&c->err_dev, c->err_host, 0));
    FY_HIP(hipDeviceSynchronize());
    // the handles travel as doubles through the caller's all-gather (64 bytes = 8 doubles; the bit patterns are only copied)
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    std::vector<double> send(8), all(8 * (size_t)size);
    std::memcpy(send.data(), &mine, 64);
    if (cb->allgather(cb->user, send.data(), all.data(), 8) != 0) return fail(FY_ERR_TRANSPORT, "ipc communicator: the bootstrap all-gather failed");
    c->peer_win.assign((size_t)size, nullptr);
    c->coll_tx.assign((size_t)size, 0); c->coll_rx.assign((size_t)size, 0); c->small_msg.assign(1, 0);
    int bad = 0;
    for (int p = 0; p < size; ++p) {
        if (p == rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, all.data() + 8 * (size_t)p, 64);
        void* pw = nullptr;
        if (hipIpcOpenMemHandle(&pw, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); bad = 1; break; }
        c->peer_win[(size_t)p] = static_cast<char*>(pw);
    }
    // every rank learns whether every rank could map its peers (and nobody starts storing into a window that is not zeroed yet)
    double ok = bad ? 0.0 : 1.0, worst = ok;
    {
        std::vector<double> flags((size_t)size);
        if (cb->allgather(cb->user, &ok, flags.data(), 1) != 0) return fail(FY_ERR_TRANSPORT, "ipc communicator: the bootstrap all-gather failed");
        for (double f : flags) worst = std::min(worst, f);
    }
    if (worst < 1.0) return fail(FY_ERR_TRANSPORT, "ipc communicator: hipIpcOpenMemHandle failed on rank %s (peers in ONE process cannot map each other: use the local group there)", bad ? "this" : "another");
    *out = c.release();
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
