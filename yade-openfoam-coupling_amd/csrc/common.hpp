// Shared host-side helpers for libfoamyade_hip.so (product code; never includes anything from oracle/).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/foamyade_hip.h"

namespace fy {

// Run-time switches: environment variables, read here and nowhere else, when an object is created.  Deployment and
// diagnosis knobs only -- timing experiments are build variants (tools/build_variant.sh), not switches in the shipped library.
struct Options {
    bool explicit_tree;          // FOAMYADE_EXPLICIT_TREE=1     32-byte explicit tree nodes even on a lattice block (what a general mesh gets)
    bool no_locate_lists;        // FOAMYADE_NO_LOCATE_LISTS=1   the plain tree walk places every particle (what a general mesh gets); tests/test_locate_paths.py
    std::string tree_cache_dir;  // FOAMYADE_TREE_CACHE_DIR      ranks of one node share the k-d build through this directory ("" = off)
    int rebin_interval;          // FOAMYADE_REBIN_INTERVAL      counting sort of the particles every this many steps (default 32; locality only)
    bool no_halo_overlap;        // FOAMYADE_NO_HALO_OVERLAP=1   slab smoother: exchange, then sweep (serial schedule; identical results)
    bool halo_overlap;           // FOAMYADE_HALO_OVERLAP=0       every slab exchange is followed by its consumer (serial schedule; identical results); default 1
    bool no_aux_comm;            // FOAMYADE_NO_AUX_COMM=1       slab mode: no second RCCL communicator for the overlapped halo
    bool no_deep_vcycle;         // FOAMYADE_NO_DEEP_VCYCLE=1    slab multigrid: one exchange per sweep (round 3's schedule) instead of one per level and cycle
    bool no_fused_corrector;     // FOAMYADE_NO_FUSED_CORRECTOR=1  the corrector as five sweeps (rounds 1 - 4) instead of the two fused ones (A/B switch, identical results)
    int strip_blocks;            // FOAMYADE_STRIP_BLOCKS=n       blocks per strip of the FV cell sweeps (0: off; unset: ~8 rows of cells)
    bool faces_from_arrays;      // FOAMYADE_FACES_FROM_ARRAYS=1   the fused sweeps stream rAUf / alphacf from their face arrays instead of re-forming them from rAU / alpha
    bool no_pairs;               // FOAMYADE_NO_PAIRS=1          pressure solver: one cell per thread in every sweep (fv_kernels.hip "two cells per thread"; identical results)
};
Options options();

// thread-local last-error text behind fy_last_error()
std::string& last_error();
int fail(int code, const char* fmt, ...);
// FOAMYADE_NO_PAIRS=1: the pressure solver's one-cell-per-thread kernels everywhere (A/B switch of the two-cell kernels, same bits); read when options() is
bool pairs_disabled();

#define FY_HIP(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return ::fy::fail(FY_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define FY_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != FY_OK) return _rc; \
    } while (0)

// Owning device allocation.  All product state lives in HBM; there is no host fallback.
template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    // grow-only (keeps the allocation across steps; sized for the largest batch seen)
    int reserve(size_t count) {
        if (count <= n) return FY_OK;
        release();
        size_t cap = count + count / 8 + 64;
        hipError_t e = hipMalloc((void**)&p, cap * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(FY_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", cap * sizeof(T), hipGetErrorString(e));
        }
        n = cap;
        return FY_OK;
    }
    int alloc_exact(size_t count) {
        release();
        if (count == 0) return FY_OK;
        hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            return fail(FY_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return FY_OK;
    }
};

// Owning PINNED host allocation (hipHostMalloc): the staging side of every host<->device copy on the drop-in path.  Copies from or to
// pageable memory are staged by the runtime through its own small pinned buffers and block the caller; from pinned memory they run
// at PCIe rate, asynchronously, on whatever stream they are put on.
template <class T>
struct HostBuf {
    T* p = nullptr;
    size_t n = 0;
    HostBuf() = default;
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    ~HostBuf() { release(); }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        n = 0;
    }
    T* data() { return p; }
    T& operator[](size_t i) { return p[i]; }
    int reserve(size_t count) {          // grow-only, contents not kept
        if (count <= n) return FY_OK;
        release();
        const size_t cap = count + count / 8 + 64;
        hipError_t e = hipHostMalloc((void**)&p, cap * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(FY_ERR_HIP, "hipHostMalloc(%zu bytes) failed: %s", cap * sizeof(T), hipGetErrorString(e));
        }
        n = cap;
        return FY_OK;
    }
};

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    bool armed = false;
    int init() {
        FY_HIP(hipEventCreate(&a));
        FY_HIP(hipEventCreate(&b));
        return FY_OK;
    }
    void destroy() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        a = b = nullptr;
    }
    void start(hipStream_t s) { (void)hipEventRecord(a, s); armed = true; }
    void stop(hipStream_t s) { (void)hipEventRecord(b, s); }
    double ms() {
        if (!armed) return 0.0;
        float t = 0.f;
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&t, a, b);
        armed = false;
        return (double)t;
    }
};

// consecutive phases on one stream share their boundary events: N + 1 records for N phases (an event record between two kernels idles
// the stream for several microseconds, so every pair that can be saved is)
struct PhaseMarks {
    static constexpr int kMax = 8;
    hipEvent_t ev[kMax] = {};
    bool set_[kMax] = {};
    int init() { for (auto& e : ev) FY_HIP(hipEventCreate(&e)); return FY_OK; }
    void destroy() { for (auto& e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; } }
    void clear() { for (auto& b : set_) b = false; }
    void mark(int i, hipStream_t s) { (void)hipEventRecord(ev[i], s); set_[i] = true; }
    double ms(int i, int j) {
        if (!set_[i] || !set_[j]) return 0.0;
        float t = 0.f;
        (void)hipEventSynchronize(ev[j]);
        (void)hipEventElapsedTime(&t, ev[i], ev[j]);
        return (double)t;
    }
};

// accumulating per-kernel clock: one HIP event pair per launch, recorded on the launch stream, harvested once per step
struct KernelClock {
    std::vector<hipEvent_t> a, b;
    size_t used = 0;
    double total_ms = 0.0;
    int64_t launches = 0;
    bool on = false;
    // An event record between two kernels costs the stream 5 - 10 us of idle GPU (kernel trace: 10.9 us between two level-0 sweeps with
    // the clock on, 0 without).  The clock therefore SAMPLES: the first `per_collect` launches after each collect() are timed -- every
    // launch of a category does the same work -- so what it measures does not slow down what it measures by more than that.
    size_t per_collect = 1;
    bool open_ = false;
    void begin(hipStream_t s) {
        open_ = false;
        if (!on || used >= per_collect) return;
        if (used == a.size()) { hipEvent_t x, y; (void)hipEventCreate(&x); (void)hipEventCreate(&y); a.push_back(x); b.push_back(y); }
        (void)hipEventRecord(a[used], s);
        open_ = true;
    }
    void end(hipStream_t s) {
        if (!open_) return;
        (void)hipEventRecord(b[used], s);
        ++used;
        open_ = false;
    }
    void collect() {       // call after the stream has been synchronised
        for (size_t i = 0; i < used; ++i) { float t = 0.f; (void)hipEventElapsedTime(&t, a[i], b[i]); total_ms += t; }
        launches += (int64_t)used;
        used = 0;
    }
    void reset() { used = 0; total_ms = 0.0; launches = 0; }
    void destroy() { for (auto e : a) (void)hipEventDestroy(e); for (auto e : b) (void)hipEventDestroy(e); a.clear(); b.clear(); }
};

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace fy
