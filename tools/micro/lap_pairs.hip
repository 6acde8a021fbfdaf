// Micro-benchmark (round 5): the pEqn Laplacian apply y = A x (k_p_apply: diag + three upper-face coefficient arrays, 48 B/cell) with ONE cell per
// thread and 8-byte loads (the shipped form) against TWO / FOUR consecutive cells per thread and 16-byte loads -- six 8-byte streams reach 4.2 TB/s past the
// Infinity Cache where a float4 copy reaches 6.3.  Same row arithmetic in the same order (p_row): the outputs must be the same bits.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/lap_pairs.hip -o tools/micro/lap_pairs ; run: ./lap_pairs [n=320] [reps=40]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct PMat { const double *diag, *ux, *uy, *uz; int nx, ny, nz, N, ntot; };

__device__ __forceinline__ int swz_block(int bid, int nblk) { return (nblk % 8) ? bid : (bid % 8) * (nblk / 8) + bid / 8; }

__device__ __forceinline__ double p_row(const PMat& A, const double* __restrict__ x, int c) {
    const int sy = A.nx, sz = A.nx * A.ny, last = A.ntot - 1;
    const int xm = max(c - 1, 0), xp = min(c + 1, last), ym = max(c - sy, 0), yp = min(c + sy, last), zm = max(c - sz, 0), zp = min(c + sz, last);
    const double uxc = A.ux[c], uyc = A.uy[c], uzc = A.uz[c];
    const double t0 = A.ux[xm] * x[xm], t1 = uxc * x[xp], t2 = A.uy[ym] * x[ym], t3 = uyc * x[yp], t4 = A.uz[zm] * x[zm], t5 = uzc * x[zp];
    double a = A.diag[c] * x[c];
    a = (c >= 1) ? a - t0 : a;
    a = (c + 1 < A.ntot) ? a - t1 : a;
    a = (c >= sy) ? a - t2 : a;
    a = (c + sy < A.ntot) ? a - t3 : a;
    a = (c >= sz) ? a - t4 : a;
    a = (c + sz < A.ntot) ? a - t5 : a;
    return a;
}

__global__ __launch_bounds__(256) void k_apply1(PMat A, const double* __restrict__ x, double* __restrict__ y) {
    const int c = swz_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (c < A.N) y[c] = p_row(A, x, c);
}

// W consecutive cells per thread (W = 2 or 4; nx % W == 0, so they share a row and every W-wide load is aligned): the coefficient and x values of the
// cells and of their y / z neighbours come as 16-byte loads, the x-neighbours of the group's ends as two 8-byte loads
template <int W>
__global__ __launch_bounds__(256) void k_applyW(PMat A, const double* __restrict__ x, double* __restrict__ y) {
    const int g = swz_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int c = g * W;
    if (c >= A.N) return;
    const int sy = A.nx, sz = A.nx * A.ny, last = A.ntot - 1;
    double dg[W], ux[W], uy[W], uz[W], xc[W], uym[W], uzm[W], xym[W], xyp[W], xzm[W], xzp[W];
    const int ym = max(c - sy, 0), yp = min(c + sy, last - (W - 1)), zm = max(c - sz, 0), zp = min(c + sz, last - (W - 1));
#pragma unroll
    for (int h = 0; h < W; h += 2) {
        const double2 a = *reinterpret_cast<const double2*>(A.diag + c + h); dg[h] = a.x; dg[h + 1] = a.y;
        const double2 b = *reinterpret_cast<const double2*>(A.ux + c + h); ux[h] = b.x; ux[h + 1] = b.y;
        const double2 d = *reinterpret_cast<const double2*>(A.uy + c + h); uy[h] = d.x; uy[h + 1] = d.y;
        const double2 e = *reinterpret_cast<const double2*>(A.uz + c + h); uz[h] = e.x; uz[h + 1] = e.y;
        const double2 f = *reinterpret_cast<const double2*>(x + c + h); xc[h] = f.x; xc[h + 1] = f.y;
        const double2 p = *reinterpret_cast<const double2*>(A.uy + ym + h); uym[h] = p.x; uym[h + 1] = p.y;
        const double2 q = *reinterpret_cast<const double2*>(A.uz + zm + h); uzm[h] = q.x; uzm[h + 1] = q.y;
        const double2 r = *reinterpret_cast<const double2*>(x + ym + h); xym[h] = r.x; xym[h + 1] = r.y;
        const double2 s = *reinterpret_cast<const double2*>(x + yp + h); xyp[h] = s.x; xyp[h + 1] = s.y;
        const double2 t = *reinterpret_cast<const double2*>(x + zm + h); xzm[h] = t.x; xzm[h + 1] = t.y;
        const double2 u = *reinterpret_cast<const double2*>(x + zp + h); xzp[h] = u.x; xzp[h + 1] = u.y;
    }
    const int xmi = max(c - 1, 0), xpi = min(c + W, last);
    const double uxm = A.ux[xmi], xm = x[xmi], xp = x[xpi];
    double out[W];
#pragma unroll
    for (int h = 0; h < W; ++h) {
        const int cc = c + h;
        const double t0 = (h == 0 ? uxm : ux[h - 1]) * (h == 0 ? xm : xc[h - 1]);
        const double t1 = ux[h] * (h == W - 1 ? xp : xc[h + 1]);
        const double t2 = uym[h] * xym[h], t3 = uy[h] * xyp[h], t4 = uzm[h] * xzm[h], t5 = uz[h] * xzp[h];
        double a = dg[h] * xc[h];
        a = (cc >= 1) ? a - t0 : a;
        a = (cc + 1 < A.ntot) ? a - t1 : a;
        a = (cc >= sy) ? a - t2 : a;
        a = (cc + sy < A.ntot) ? a - t3 : a;
        a = (cc >= sz) ? a - t4 : a;
        a = (cc + sz < A.ntot) ? a - t5 : a;
        out[h] = a;
    }
#pragma unroll
    for (int h = 0; h < W; h += 2) *reinterpret_cast<double2*>(y + c + h) = make_double2(out[h], out[h + 1]);
}

// two PAIRS per thread, 512 cells apart within a 1024-cell block: every load instruction is still 64 lanes x 16 consecutive bytes, twice the loads in flight per thread
__device__ __forceinline__ void pair_row(const PMat& A, const double* __restrict__ x, double* __restrict__ y, int c) {
    const int sy = A.nx, sz = A.nx * A.ny, last = A.ntot - 1;
    const int ym = max(c - sy, 0), yp = min(c + sy, last - 1), zm = max(c - sz, 0), zp = min(c + sz, last - 1);
    const double2 dg = *reinterpret_cast<const double2*>(A.diag + c), ux = *reinterpret_cast<const double2*>(A.ux + c), uy = *reinterpret_cast<const double2*>(A.uy + c),
                  uz = *reinterpret_cast<const double2*>(A.uz + c), xc = *reinterpret_cast<const double2*>(x + c), uym = *reinterpret_cast<const double2*>(A.uy + ym),
                  uzm = *reinterpret_cast<const double2*>(A.uz + zm), xym = *reinterpret_cast<const double2*>(x + ym), xyp = *reinterpret_cast<const double2*>(x + yp),
                  xzm = *reinterpret_cast<const double2*>(x + zm), xzp = *reinterpret_cast<const double2*>(x + zp);
    const int xmi = max(c - 1, 0), xpi = min(c + 2, last);
    const double uxm = A.ux[xmi], xm = x[xmi], xp = x[xpi];
    double a = dg.x * xc.x;
    a = (c >= 1) ? a - uxm * xm : a;
    a = (c + 1 < A.ntot) ? a - ux.x * xc.y : a;
    a = (c >= sy) ? a - uym.x * xym.x : a;
    a = (c + sy < A.ntot) ? a - uy.x * xyp.x : a;
    a = (c >= sz) ? a - uzm.x * xzm.x : a;
    a = (c + sz < A.ntot) ? a - uz.x * xzp.x : a;
    double b = dg.y * xc.y;
    b = b - ux.x * xc.x;                                        // (c + 1 >= 1 always)
    b = (c + 2 < A.ntot) ? b - ux.y * xp : b;
    b = (c + 1 >= sy) ? b - uym.y * xym.y : b;
    b = (c + 1 + sy < A.ntot) ? b - uy.y * xyp.y : b;
    b = (c + 1 >= sz) ? b - uzm.y * xzm.y : b;
    b = (c + 1 + sz < A.ntot) ? b - uz.y * xzp.y : b;
    *reinterpret_cast<double2*>(y + c) = make_double2(a, b);
}
__global__ __launch_bounds__(256) void k_apply2x2(PMat A, const double* __restrict__ x, double* __restrict__ y) {
    const int base = swz_block(blockIdx.x, gridDim.x) * 1024 + 2 * (int)threadIdx.x;
    if (base < A.N) pair_row(A, x, y, base);
    if (base + 512 < A.N) pair_row(A, x, y, base + 512);
}

__global__ void k_copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 320, reps = argc > 2 ? atoi(argv[2]) : 40;
    const int N = n * n * n;
    std::vector<double> h(N);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1 << 24); };
    double *d[5], *y1, *y2, *y4;
    for (int a = 0; a < 5; ++a) {
        for (int c = 0; c < N; ++c) {
            const int i = c % n, j = (c / n) % n, k = c / (n * n);
            double v = 0.5 + rnd();
            if (a == 0) v += 6.0;
            if ((a == 1 && i == n - 1) || (a == 2 && j == n - 1) || (a == 3 && k == n - 1)) v = 0.0;      // high-side boundary faces store 0
            h[c] = v;
        }
        CK(hipMalloc(&d[a], (size_t)N * sizeof(double)));
        CK(hipMemcpy(d[a], h.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&y1, (size_t)N * 8)); CK(hipMalloc(&y2, (size_t)N * 8)); CK(hipMalloc(&y4, (size_t)N * 8));
    PMat A{d[0], d[1], d[2], d[3], n, n, n, N, N};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* nm, auto launch, double bytes) {
        float best = 1e30f;
        for (int t = 0; t < 3; ++t) {
            launch();
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-28s %8.1f us  %7.1f GB/s\n", nm, best * 1e3 / reps, bytes / (best * 1e-3 / reps) / 1e9);
    };
    const double bytes = 48.0 * N;
    printf("n = %d: %d cells, %.2f GB per apply\n", n, N, bytes / 1e9);
    timeit("1 cell / thread (shipped)", [&] { hipLaunchKernelGGL(k_apply1, dim3((N + 255) / 256), dim3(256), 0, 0, A, d[4], y1); }, bytes);
    timeit("2 cells / thread", [&] { hipLaunchKernelGGL(k_applyW<2>, dim3((N / 2 + 255) / 256), dim3(256), 0, 0, A, d[4], y2); }, bytes);
    timeit("4 cells / thread", [&] { hipLaunchKernelGGL(k_applyW<4>, dim3((N / 4 + 255) / 256), dim3(256), 0, 0, A, d[4], y4); }, bytes);
    double* y5; CK(hipMalloc(&y5, (size_t)N * 8));
    timeit("2 x 2 cells / thread", [&] { hipLaunchKernelGGL(k_apply2x2, dim3((N + 1023) / 1024), dim3(256), 0, 0, A, d[4], y5); }, bytes);
    { std::vector<double> r1(N), r5(N); hipLaunchKernelGGL(k_apply1, dim3((N + 255) / 256), dim3(256), 0, 0, A, d[4], y1);
      CK(hipMemcpy(r1.data(), y1, (size_t)N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r5.data(), y5, (size_t)N * 8, hipMemcpyDeviceToHost));
      printf("bits: 2 x 2 cells %s\n", memcmp(r1.data(), r5.data(), (size_t)N * 8) ? "DIFFER" : "equal"); }
    float4 *ca, *cb;
    const size_t n4 = (size_t)N * 24 / 16;                 // 24 B/cell read + 24 B/cell written = the apply's 48
    CK(hipMalloc(&ca, n4 * 16)); CK(hipMalloc(&cb, n4 * 16)); CK(hipMemset(ca, 0, n4 * 16));
    timeit("float4 copy, 24 + 24 B/cell", [&] { hipLaunchKernelGGL(k_copy4, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4*)ca, cb, n4); }, bytes);
    hipLaunchKernelGGL(k_apply1, dim3((N + 255) / 256), dim3(256), 0, 0, A, d[4], y1);
    std::vector<double> r1(N), r2(N), r4(N);
    CK(hipMemcpy(r1.data(), y1, (size_t)N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), y2, (size_t)N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(r4.data(), y4, (size_t)N * 8, hipMemcpyDeviceToHost));
    printf("bits: 2 cells %s, 4 cells %s\n", memcmp(r1.data(), r2.data(), (size_t)N * 8) ? "DIFFER" : "equal", memcmp(r1.data(), r4.data(), (size_t)N * 8) ? "DIFFER" : "equal");
    return 0;
}
