// Slab communicator: what the z-slab decomposition needs from "the other GPUs" (SURVEY.md 8e):
//   * neighbour exchange of contiguous z-plane blocks (FV halos 1 plane, particle halos 5 planes, reverse sums),
//   * tiny all-reduces (Krylov scalars, Courant number, continuity errors, residual norms),
//   * an all-gather (coarse multigrid right-hand sides).
// Back-ends behind one interface:
//   RcclComm   one process per GPU, RCCL (ncclSend/ncclRecv/ncclAllReduce/ncclAllGather) on the solver's stream over xGMI;
//              librccl is dlopen()ed on first use so the library still loads on hosts without it;
//   IpcComm    one process per GPU -- or N processes on ONE GPU -- with direct peer stores into hipIpc-mapped windows (no library between the GPUs);
//   LocalComm  N "virtual slabs" inside one process (one host thread per slab, device-to-device copies, host barriers): the
//              same solver code runs unmodified, which is how the decomposition is tested on a single-GPU box.
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <cstddef>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/foamyade_hip.h"

namespace fy {

struct Comm {
    int rank = 0, size = 1;
    uint64_t n_exchange = 0, n_allreduce = 0, n_allgather = 0, exchange_bytes = 0;   // call counters (fy_comm_stats)
    // the same counters by the phase that issued the call (the solver names the phase it is in: fy_comm_stats_by_tag, tools/comm_count.py)
    const char* tag = "";
    std::map<std::string, std::array<uint64_t, 3> > by_tag;
    void count(int kind) { ++(kind == 0 ? n_exchange : kind == 1 ? n_allreduce : n_allgather); ++by_tag[tag][(size_t)kind]; }
    struct Tag {                                         // scoped phase name
        Comm* c; const char* prev;
        Tag(Comm* c_, const char* t) : c(c_), prev(c_->tag) { c->tag = t; }
        ~Tag() { c->tag = prev; }
    };
    double* ops_scratch = nullptr;                       // [size x n] landing zone of allreduce_ops
    size_t ops_scratch_n = 0;
    virtual ~Comm() { if (ops_scratch) (void)hipFree(ops_scratch); }
    bool has_down() const { return rank > 0; }           // neighbour owning the planes below mine
    bool has_up() const { return rank + 1 < size; }
    // send `count` doubles starting at send_up to the upper neighbour (it receives them at ITS recv_from_down), and so on.
    // Pointers for a missing neighbour are ignored.  Between group_begin() and group_end() the calls are only recorded and then
    // issued as ONE exchange (one RCCL group = one launch and one round of handshakes for fields that travel together); nothing may
    // depend on the received planes before group_end().
    // count: doubles per direction; n_* (all zero = use `count` for every direction) give each transfer its own size -- a rank's
    // recv_from_down size must equal its lower neighbour's send_up size (particle migration: the sizes were exchanged first)
    struct Xchg {
        const double* send_up; double* recv_from_down; const double* send_down; double* recv_from_up; size_t count;
        size_t n_send_up = 0, n_recv_down = 0, n_send_down = 0, n_recv_up = 0;
        bool sized() const { return n_send_up || n_recv_down || n_send_down || n_recv_up; }
        size_t su() const { return sized() ? n_send_up : count; }
        size_t rd() const { return sized() ? n_recv_down : count; }
        size_t sd() const { return sized() ? n_send_down : count; }
        size_t ru() const { return sized() ? n_recv_up : count; }
    };
    int neighbour_exchange(hipStream_t s, const double* send_up, double* recv_from_down, const double* send_down, double* recv_from_up,
                           size_t count) {
        Xchg x{send_up, recv_from_down, send_down, recv_from_up, count};
        if (grouping) { pending.push_back(x); return 0; }
        return exchange_many(s, &x, 1);
    }
    int neighbour_exchange_sized(hipStream_t s, const double* send_up, size_t n_send_up, double* recv_from_down, size_t n_recv_down,
                                 const double* send_down, size_t n_send_down, double* recv_from_up, size_t n_recv_up) {
        Xchg x{send_up, recv_from_down, send_down, recv_from_up, 0};
        x.n_send_up = n_send_up; x.n_recv_down = n_recv_down; x.n_send_down = n_send_down; x.n_recv_up = n_recv_up;
        // (no early return when all four sizes are zero: the exchange is collective -- LocalComm pairs host barriers -- and an unsized
        //  entry already reads as four zero counts)
        if (grouping) { pending.push_back(x); return 0; }
        return exchange_many(s, &x, 1);
    }
    void group_begin() { grouping = true; }
    int group_end(hipStream_t s) {
        grouping = false;
        const int rc = pending.empty() ? 0 : exchange_many(s, pending.data(), pending.size());
        pending.clear();
        return rc;
    }
    // exchanges enqueued on `aux` run beside the ones on the main stream (the smoother's overlapped halo): a back-end whose handle may
    // not be driven from two streams at once (RCCL: one ncclComm per stream in flight) routes them over a second communicator
    virtual void set_aux_stream(hipStream_t) {}
    virtual hipStream_t get_aux_stream() const { return nullptr; }
    virtual int exchange_many(hipStream_t s, const Xchg* x, size_t n) = 0;
    std::vector<Xchg> pending;
    bool grouping = false;
    virtual int allreduce(hipStream_t s, double* dev, int n, bool is_max) = 0;   // in place; identical result on every rank
    // n <= 32 scalars of which the slots in `max_mask` are maxima and the others sums, in ONE collective: the ranks' values are all-gathered and
    // folded locally in rank order (a diagnostics group such as {sum, sum, max, sum} cost three latency-bound all-reduces before)
    int allreduce_ops(hipStream_t s, double* dev, int n, unsigned max_mask);
    virtual int allgather(hipStream_t s, const double* send, double* recv, size_t count_per_rank) = 0;
    virtual int barrier(hipStream_t s) = 0;
};

struct SelfComm : Comm {                                   // size 1: every call is a no-op
    int exchange_many(hipStream_t, const Xchg*, size_t) override { return 0; }
    int allreduce(hipStream_t, double*, int, bool) override { return 0; }
    int allgather(hipStream_t s, const double* send, double* recv, size_t n) override;
    int barrier(hipStream_t) override { return 0; }
};

// HostComm: one process per slab, planes staged through pinned host memory, the inter-process transport is the caller's (callbacks)
int host_comm_create(int rank, int size, const fy_comm_callbacks* cb, Comm** out);
// IpcComm: one process per slab, peer stores into each other's device windows (hipIpc); the callbacks carry the bootstrap only (all-gather of the handles)
int ipc_comm_create(int rank, int size, const fy_comm_callbacks* cb, int device, Comm** out);
// LocalComm group: create `n` communicators that talk to each other inside this process
int local_comm_group_create(int n, Comm** out /* [n] */);
// RCCL: unique id = 128 opaque bytes produced on rank 0 (fy_rccl_unique_id) and broadcast by the launcher
int rccl_unique_id(void* out128);
int rccl_comm_create(int rank, int size, const void* id128, int device, Comm** out);
// known-answer run of every operation the slab solver uses (grouped two-field neighbour exchange, sum / max all-reduce, all-gather); collective
int comm_selftest(Comm* c, int device);

}  // namespace fy

struct fy_comm { fy::Comm* c; };
