// Micro-benchmark (round 5): what a dependent FP64 operation costs a LONE wave (the coarsest multigrid level's band solve is one wave walking 250 columns of
// lane read -> multiply -> multiply -> subtract): chains of dependent v_fma_f64, with and without a v_readlane in the chain, one 64-thread block on an idle GPU.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/lone_wave.hip -o tools/micro/lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_chain(double* out, int n, double a, double b) {
    double v = (double)threadIdx.x;
    for (int i = 0; i < n; ++i) v = v * a + b;                      // (mul + add: two dependent operations, no contraction)
    out[threadIdx.x] = v;
}
__global__ void k_chain_lane(double* out, int n, double a) {
    double v = (double)threadIdx.x + 1.0;
    for (int i = 0; i < n; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), i & 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), i & 63);
        const double y = __hiloint2double(hi, lo) * a;               // lane read -> multiply
        v = v - 1e-9 * y;                                            // -> multiply -> subtract
    }
    out[threadIdx.x] = v;
}
int main() {
    double* out; CK(hipMalloc(&out, 1024 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 100000;
    for (int threads : {64, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            float ms;
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, out, n, 1.0000001, 1e-9); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%4d threads: mul+add chain          %.2f ns per dependent operation\n", threads, ms * 1e6 / (2.0 * n));
            CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chain_lane, dim3(1), dim3(threads), 0, 0, out, n, 1.0000001); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%4d threads: readlane+mul+mul+sub   %.2f ns per column (4 dependent operations)\n", threads, ms * 1e6 / n);
        }
    }
    return 0;
}
