#include "comm.hpp"

#include <dlfcn.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace fy {

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
    std::vector<std::vector<double> > red;     // per-rank host staging for all-reduce
    explicit LocalShared(int n_) : n(n_), bar(n_), lists(n_), gather_src(n_), red(n_) {}
};

struct LocalComm : Comm {
    std::shared_ptr<LocalShared> sh;
    int exchange_many(hipStream_t s, const Xchg* x, size_t n) override {
        ++n_exchange;
        for (size_t q = 0; q < n; ++q) exchange_bytes += sizeof(double) * ((has_up() ? x[q].su() : 0) + (has_down() ? x[q].sd() : 0));
        FY_HIP(hipStreamSynchronize(s));                       // my planes are final
        sh->lists[rank] = x;                                   // every rank posts the same number of items in the same order
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
        ++n_allreduce;
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
        ++n_allgather;
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
    ~RcclComm() override { if (A && comm && A->CommDestroy) A->CommDestroy(comm); }
    // Both directions in ONE group: every rank posts all its sends and receives before any of them has to complete, so the
    // pairing cannot deadlock whatever order the ranks reach this call in.
    int exchange_many(hipStream_t s, const Xchg* x, size_t n) override {
        ++n_exchange;
        FY_NCCL(A->GroupStart());
        for (size_t q = 0; q < n; ++q) {
            exchange_bytes += sizeof(double) * ((has_up() ? x[q].su() : 0) + (has_down() ? x[q].sd() : 0));
            if (has_up() && x[q].send_up && x[q].su()) FY_NCCL(A->Send(x[q].send_up, x[q].su(), kNcclDouble, rank + 1, comm, s));
            if (has_down() && x[q].recv_from_down && x[q].rd()) FY_NCCL(A->Recv(x[q].recv_from_down, x[q].rd(), kNcclDouble, rank - 1, comm, s));
            if (has_down() && x[q].send_down && x[q].sd()) FY_NCCL(A->Send(x[q].send_down, x[q].sd(), kNcclDouble, rank - 1, comm, s));
            if (has_up() && x[q].recv_from_up && x[q].ru()) FY_NCCL(A->Recv(x[q].recv_from_up, x[q].ru(), kNcclDouble, rank + 1, comm, s));
        }
        FY_NCCL(A->GroupEnd());
        return FY_OK;
    }
    int allreduce(hipStream_t s, double* dev, int n, bool is_max) override {
        ++n_allreduce;
        FY_NCCL(A->AllReduce(dev, dev, (size_t)n, kNcclDouble, is_max ? kNcclMax : kNcclSum, comm, s));
        return FY_OK;
    }
    int allgather(hipStream_t s, const double* send, double* recv, size_t cnt) override {
        ++n_allgather;
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
    *out = c;
    return FY_OK;
}

}  // namespace fy
