// micro-benchmark (development tool): ds_add_f64 throughput per CU, random slots of a 2048-double LDS table vs few hot slots
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k_lds(double* out, int iters, int mode, int width, int soa) {
    __shared__ double t[2048];
    for (int q = threadIdx.x; q < 2048; q += 512) t[q] = 0.0;
    __syncthreads();
    unsigned x = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        unsigned s = mode == 0 ? (x >> 8) % 512u : (mode == 1 ? ((x >> 8) % 16u) : (threadIdx.x % 512u));   // random / 16 hot cells / conflict-free
        for (int q = 0; q < width; ++q) unsafeAtomicAdd(&t[soa ? 512 * q + s : 4 * s + q], 1.0);   // slot-major (32-byte stride: 4 of 16 bank pairs per component) / component-major
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = t[0] + t[5];
}
int main() {
    double* out; hipMalloc(&out, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int soa = 0; soa < 2; ++soa)
    for (int mode = 0; mode < 3; ++mode)
        for (int width : {1, 4}) {
            const int blocks = 256 * 4, iters = 200;
            k_lds<<<blocks, 512>>>(out, iters, mode, width, soa);
            hipEventRecord(a);
            k_lds<<<blocks, 512>>>(out, iters, mode, width, soa);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double n = (double)blocks * 512 * iters * width;
            printf("%s mode %d (%s) width %d: %.3f ms, %.1f G lds-atomics/s = %.2f per clk per CU\n", soa ? "component-major" : "slot-major", mode, mode == 0 ? "random 512 cells" : mode == 1 ? "16 hot cells" : "conflict-free",
                   width, ms, n / ms * 1e-6, n / ms * 1e-6 / 256 / 2.4);
        }
    return 0;
}
