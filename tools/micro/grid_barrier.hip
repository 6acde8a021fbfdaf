// micro-benchmark: cost of a device-wide barrier on MI355X (8 XCDs, per-XCD L2): atomic counter + agent-scope fences, co-resident blocks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_barrier(unsigned int* counter, unsigned int nblocks, unsigned int& phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
        phase += nblocks;
        __threadfence();                                              // release: this block's writes are visible device-wide
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) { __builtin_amdgcn_s_sleep(1); }
        __threadfence();                                              // acquire
    }
    __syncthreads();
}

__global__ void k_bar(unsigned int* counter, int iters, double* data, int n) {
    unsigned int phase = 0;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (t < n) data[t] = data[(t + 977) % n] * 0.5 + 1.0;         // a little work that crosses blocks
        grid_barrier(counter, gridDim.x, phase);
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = 256, iters = 200, n = blocks * threads;
    unsigned int* c; double* d;
    hipMalloc(&c, 4); hipMalloc(&d, n * 8); hipMemset(d, 0, n * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(c, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_bar, dim3(blocks), dim3(threads), 0, 0, c, iters, d, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("blocks %d: %.2f us per barrier (+ trivial work)\n", blocks, 1e3 * ms / iters);
    }
    // for comparison: the same work as separate launches
    hipEventRecord(a);
    for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(k_bar, dim3(blocks), dim3(threads), 0, 0, c, 0, d, n);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("empty kernel launches back to back: %.2f us each\n", 1e3 * ms / iters);
    return 0;
}
