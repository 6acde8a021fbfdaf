// Micro-benchmark (round 5): what a reducing sweep pays for folding its own block partials in the block that finishes last, instead of
// leaving them to a second kernel (k_reduce_finalize: 8 us x 20 calls per C3 step).  The sweep: a 7-point Laplacian of a 160^3 field
// (the PCG's A p with p . A p, k_p_apply_dot's shape), one partial per 256-cell block.
//   0  sweep + separate fold kernel (the shipped scheme)
//   1  sweep; every block: agent-scope RELEASE fence + relaxed counter increment; the last block: ACQUIRE fence, fold
//   2  sweep; partials written with agent-scope atomic stores (write-through), s_waitcnt + barrier, relaxed counter increment;
//      the last block reads them with agent-scope atomic loads, fold          (no cache-wide write-back per block)
//   3  as 1 with a full __threadfence() in every block (the textbook form)
//   4  as 0 with the fold kernel on 256 threads (the 1024-thread kernel's order kept)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/reduce_tail.hip -o tools/micro/reduce_tail ; run: ./reduce_tail [n=160] [reps=200]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int swz_block(int bid, int nblk) { return (nblk % 8) ? bid : (bid % 8) * (nblk / 8) + bid / 8; }

__device__ __forceinline__ double block_sum(double x) {
    __shared__ double sh[4];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// the fold of k_reduce_finalize by 256 threads, in the 1024-thread kernel's order: virtual thread v = 64 w + lane sums partials v, v + 1024, ...;
// shuffle tree per virtual wave; the 16 wave sums in order
template <bool ATOMIC_LD>
__device__ __forceinline__ double fold_as_1024(const double* partials, int nblocks) {
    __shared__ double wsum[16];
    const int lane = threadIdx.x & 63, rw = threadIdx.x >> 6;
    for (int vw = rw; vw < 16; vw += 4) {
        double x = 0.0;
        for (int b = vw * 64 + lane; b < nblocks; b += 1024)
            x += ATOMIC_LD ? __hip_atomic_load(&partials[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partials[b];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if (lane == 0) wsum[vw] = x;
    }
    __syncthreads();
    double r = wsum[0];
    for (int w = 1; w < 16; ++w) r += wsum[w];
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_sweep(int nx, int ny, int nz, const double* __restrict__ p, double* __restrict__ Ap, double* partials,
                                               unsigned int* counter, double* out) {
    const int lb = swz_block(blockIdx.x, gridDim.x);
    const int c = lb * 256 + threadIdx.x, N = nx * ny * nz;
    double v = 0.0;
    if (c < N) {
        const int i = c % nx, j = (c / nx) % ny, k = c / (nx * ny);
        const double pc = p[c];
        double s = 0.0;
        if (i > 0) s += p[c - 1] - pc;
        if (i < nx - 1) s += p[c + 1] - pc;
        if (j > 0) s += p[c - nx] - pc;
        if (j < ny - 1) s += p[c + nx] - pc;
        if (k > 0) s += p[c - nx * ny] - pc;
        if (k < nz - 1) s += p[c + nx * ny] - pc;
        Ap[c] = s;
        v = pc * s;
    }
    const double bs = block_sum(v);
    if (MODE == 0 || MODE == 4) { if (threadIdx.x == 0) partials[lb] = bs; return; }
    __shared__ int last;
    if (MODE == 2) {
        if (threadIdx.x == 0) {
            __hip_atomic_store(&partials[lb], bs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_s_waitcnt(0);                       // (vmcnt(0): the write-through store has been acknowledged)
            const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t == gridDim.x - 1);
        }
    } else {
        if (threadIdx.x == 0) {
            partials[lb] = bs;
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __threadfence();
            const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t == gridDim.x - 1);
        }
    }
    __syncthreads();
    if (!last) return;
    if (MODE != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const double r = (MODE == 2) ? fold_as_1024<true>(partials, (int)gridDim.x) : fold_as_1024<false>(partials, (int)gridDim.x);
    if (threadIdx.x == 0) { out[0] = r; *counter = 0u; }
}

__global__ __launch_bounds__(1024) void k_fold(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
    __shared__ double sh[16];
    double x = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 1024) x += partials[b];
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) { double r = sh[0]; for (int w = 1; w < 16; ++w) r += sh[w]; out[0] = r; }
}

// the separate fold on 256 threads, in the 1024-thread kernel's order (same bits): fewer waves to start and a 4-wave barrier
__global__ __launch_bounds__(256) void k_fold256(const double* __restrict__ partials, int nblocks, double* __restrict__ out) {
    const double r = fold_as_1024<false>(partials, nblocks);
    if (threadIdx.x == 0) out[0] = r;
}

// a dependent follow-up (what the PCG does next with the scalar): y += out * x
__global__ __launch_bounds__(256) void k_axpy(int N, const double* __restrict__ out, const double* __restrict__ x, double* __restrict__ y) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) y[c] += out[0] * 1e-9 * x[c];
}

template <int MODE>
static void run(int n, int reps, const double* p, double* Ap, double* y, double* partials, unsigned int* counter, double* out) {
    const int N = n * n * n, nb = (N + 255) / 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double res[2] = {0, 0};
    float best = 1e30f;
    for (int trial = 0; trial < 3; ++trial) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(k_sweep<MODE>, dim3(nb), dim3(256), 0, 0, n, n, n, p, Ap, partials, counter, out);
            if (MODE == 0) hipLaunchKernelGGL(k_fold, dim3(1), dim3(1024), 0, 0, partials, nb, out);
            if (MODE == 4) hipLaunchKernelGGL(k_fold256, dim3(1), dim3(256), 0, 0, partials, nb, out);
            hipLaunchKernelGGL(k_axpy, dim3(nb), dim3(256), 0, 0, N, out, p, y);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipMemcpy(res, out, sizeof(double), hipMemcpyDeviceToHost));
    printf("mode %d: %.2f us per (sweep + fold + dependent sweep); p.Ap = %.17g\n", MODE, best * 1e3 / reps, res[0]);
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 160, reps = argc > 2 ? atoi(argv[2]) : 200;
    const int N = n * n * n, nb = (N + 255) / 256;
    std::vector<double> h(N);
    unsigned s = 12345u;
    for (int c = 0; c < N; ++c) { s = s * 1664525u + 1013904223u; h[c] = (double)(s >> 8) / (1 << 24); }
    double *p, *Ap, *y, *partials, *out; unsigned int* counter;
    CK(hipMalloc(&p, N * sizeof(double))); CK(hipMalloc(&Ap, N * sizeof(double))); CK(hipMalloc(&y, N * sizeof(double)));
    CK(hipMalloc(&partials, nb * sizeof(double))); CK(hipMalloc(&out, 64)); CK(hipMalloc(&counter, 64));
    CK(hipMemcpy(p, h.data(), N * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMemset(y, 0, N * sizeof(double))); CK(hipMemset(counter, 0, 64));
    printf("n = %d (%d blocks)\n", n, nb);
    run<0>(n, reps, p, Ap, y, partials, counter, out);
    run<1>(n, reps, p, Ap, y, partials, counter, out);
    run<2>(n, reps, p, Ap, y, partials, counter, out);
    run<3>(n, reps, p, Ap, y, partials, counter, out);
    run<0>(n, reps, p, Ap, y, partials, counter, out);
    run<4>(n, reps, p, Ap, y, partials, counter, out);
    run<0>(n, reps, p, Ap, y, partials, counter, out);
    run<4>(n, reps, p, Ap, y, partials, counter, out);
    return 0;
}
