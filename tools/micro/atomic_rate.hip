// micro-benchmark (development tool): throughput of no-return global_atomic_add_f64 on MI355X
//   random cells of an N-cell array (32-byte records, 4 atomics per record or 1), from every CU
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/micro/atomic_rate.hip -o tools/micro/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k_atomics(double* acc, unsigned n_cells, int per_lane, int comps, unsigned seed, int local) {
    unsigned x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    for (int it = 0; it < per_lane; ++it) {
        x = x * 1664525u + 1013904223u;
        unsigned c = (x >> 4) % n_cells;
        if (local) c = (blockIdx.x * 997u + ((x >> 4) % 4096u)) % n_cells;      // each block hits a 4096-cell window
        for (int q = 0; q < comps; ++q) unsafeAtomicAdd(&acc[4 * (size_t)c + q], 1.0);
    }
}
__global__ __launch_bounds__(256) void k_stores(double* acc, unsigned n_cells, int per_lane, int comps, unsigned seed) {
    unsigned x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + seed;
    for (int it = 0; it < per_lane; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned c = (x >> 4) % n_cells;
        for (int q = 0; q < comps; ++q) acc[4 * (size_t)c + q] = 1.0;
    }
}
int main() {
    const unsigned n_cells = 4096000;
    double* acc; hipMalloc(&acc, (size_t)n_cells * 32); hipMemset(acc, 0, (size_t)n_cells * 32);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {2048, 8192, 32768})
        for (int comps : {1, 4})
            for (int local = 0; local < 2; ++local) {
                const int per_lane = 8;
                k_atomics<<<blocks, 256>>>(acc, n_cells, per_lane, comps, 1u, local);
                hipEventRecord(a);
                k_atomics<<<blocks, 256>>>(acc, n_cells, per_lane, comps, 7u, local);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                const double n = (double)blocks * 256 * per_lane * comps;
                printf("atomics blocks %6d comps %d local %d: %.3f ms, %.1f G atomics/s\n", blocks, comps, local, ms, n / ms * 1e-6);
            }
    for (int comps : {1, 4}) {
        hipEventRecord(a);
        k_stores<<<8192, 256>>>(acc, n_cells, 8, comps, 3u);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("plain stores comps %d: %.3f ms, %.1f G stores/s\n", comps, ms, 8192.0 * 256 * 8 * comps / ms * 1e-6);
    }
    return 0;
}
