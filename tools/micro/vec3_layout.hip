// Micro-benchmark (round 5): the H-operator sweep of the corrector (k_HbyA: out = rAU (src - sum_nb a_nb U_nb) / V, 128 B/cell) with its
// three vec3 fields stored (a) AoS [Nc][3] as the solver has them, (b) as component planes [3][Nc], (c) AoS staged through LDS with coalesced loads.
// Question: are the AoS sweeps bound by the L1's delivery of 24-byte records (a wave's component load touches 12 lines of 128 B for 512 useful bytes)?
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/micro/vec3_layout.hip -o tools/micro/vec3_layout ; run: ./vec3_layout [n=160]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int swz_block(int bid, int nblk) { return (nblk % 8) ? bid : (bid % 8) * (nblk / 8) + bid / 8; }

struct Geo { int nx, ny, nz, N; };

template <int LAYOUT>   // 0 AoS, 1 planes
__device__ __forceinline__ double ld(const double* __restrict__ F, size_t n, int c, int q) { return LAYOUT == 0 ? F[3 * (size_t)c + q] : F[(size_t)q * n + c]; }
template <int LAYOUT>
__device__ __forceinline__ void st(double* __restrict__ F, size_t n, int c, int q, double v) { if (LAYOUT == 0) F[3 * (size_t)c + q] = v; else F[(size_t)q * n + c] = v; }

template <int LAYOUT>
__global__ __launch_bounds__(256) void k_H(Geo g, const double* __restrict__ a0, const double* __restrict__ a1, const double* __restrict__ a2, const double* __restrict__ a3,
                                           const double* __restrict__ a4, const double* __restrict__ a5, const double* __restrict__ src, const double* __restrict__ U,
                                           const double* __restrict__ rAU, double* __restrict__ out) {
    const int t = swz_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= g.N) return;
    const int i = t % g.nx, q_ = t / g.nx, j = q_ % g.ny, k = q_ / g.ny;
    const int c = t;
    const size_t n = g.N;
    double acc[3] = {ld<LAYOUT>(src, n, c, 0), ld<LAYOUT>(src, n, c, 1), ld<LAYOUT>(src, n, c, 2)};
    const double* an[6] = {a0, a1, a2, a3, a4, a5};
    const int st_[3] = {1, g.nx, g.nx * g.ny};
    const int qd[3] = {i, j, k}, nd[3] = {g.nx, g.ny, g.nz};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (s ? qd[d] < nd[d] - 1 : qd[d] > 0) {
                const int nb = c + (s ? st_[d] : -st_[d]);
                const double a = an[2 * d + s][c];
                for (int q = 0; q < 3; ++q) acc[q] -= a * ld<LAYOUT>(U, n, nb, q);
            }
    const double r = rAU[c];
    for (int q = 0; q < 3; ++q) st<LAYOUT>(out, n, c, q, r * (acc[q] * 0.5));
}

// AoS in memory, but a wave fetches its 64 records as three fully coalesced 512-byte rows and transposes them through LDS
__device__ __forceinline__ void ld3_coalesced(const double* __restrict__ F, int cbase /* first cell of the wave */, int lane, int nvalid, double* lds /* 192 doubles of this wave */, double (&o)[3]) {
    const double* p = F + 3 * (size_t)cbase;
    const int tot = 3 * nvalid;
#pragma unroll
    for (int r = 0; r < 3; ++r) { const int e = r * 64 + lane; if (e < tot) lds[e] = p[e]; }
    __builtin_amdgcn_wave_barrier();
    o[0] = lds[3 * lane]; o[1] = lds[3 * lane + 1]; o[2] = lds[3 * lane + 2];
    __builtin_amdgcn_wave_barrier();
}
__global__ __launch_bounds__(256) void k_H_lds(Geo g, const double* __restrict__ a0, const double* __restrict__ a1, const double* __restrict__ a2, const double* __restrict__ a3,
                                               const double* __restrict__ a4, const double* __restrict__ a5, const double* __restrict__ src, const double* __restrict__ U,
                                               const double* __restrict__ rAU, double* __restrict__ out) {
    __shared__ double sh[4][192];
    const int t = swz_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int cbase = t - lane;
    if (cbase >= g.N) return;
    const int nvalid = min(64, g.N - cbase);
    const bool live = t < g.N;
    const int tt = live ? t : g.N - 1;
    const int i = tt % g.nx, q_ = tt / g.nx, j = q_ % g.ny, k = q_ / g.ny;
    const int c = tt;
    double acc[3];
    ld3_coalesced(src, cbase, lane, nvalid, sh[wv], acc);
    const double* an[6] = {a0, a1, a2, a3, a4, a5};
    const int st_[3] = {1, g.nx, g.nx * g.ny};
    const int qd[3] = {i, j, k}, nd[3] = {g.nx, g.ny, g.nz};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            // the neighbours of the wave's 64 consecutive cells are 64 consecutive cells (shifted by the stride): one coalesced fetch as well
            const int off = s ? st_[d] : -st_[d];
            const int nb0 = cbase + off;
            double u[3] = {0, 0, 0};
            if (nb0 >= 0 && nb0 + nvalid <= g.N) ld3_coalesced(U, nb0, lane, nvalid, sh[wv], u);
            else if (live && c + off >= 0 && c + off < g.N) { u[0] = U[3 * (size_t)(c + off)]; u[1] = U[3 * (size_t)(c + off) + 1]; u[2] = U[3 * (size_t)(c + off) + 2]; }
            if (live && (s ? qd[d] < nd[d] - 1 : qd[d] > 0)) {
                const double a = an[2 * d + s][c];
                for (int q = 0; q < 3; ++q) acc[q] -= a * u[q];
            }
        }
    const double r = live ? rAU[c] : 0.0;
    double o[3];
    for (int q = 0; q < 3; ++q) o[q] = r * (acc[q] * 0.5);
    // coalesced store through LDS
    double* l = sh[wv];
    l[3 * lane] = o[0]; l[3 * lane + 1] = o[1]; l[3 * lane + 2] = o[2];
    __builtin_amdgcn_wave_barrier();
    double* p = out + 3 * (size_t)cbase;
    const int tot = 3 * nvalid;
#pragma unroll
    for (int r2 = 0; r2 < 3; ++r2) { const int e = r2 * 64 + lane; if (e < tot) p[e] = l[e]; }
}

// (d) AoS, TWO consecutive cells per thread: a pair of vec3 records is 48 contiguous, 16-byte aligned bytes = three 16-byte loads that a wave issues fully coalesced;
// the scalar coefficient arrays come as one 16-byte load per pair
struct V2 { double a[3], b[3]; };
__device__ __forceinline__ V2 ldv2(const double* __restrict__ F, int c) {
    const double2* p = reinterpret_cast<const double2*>(F + 3 * (size_t)c);
    const double2 x = p[0], y = p[1], z = p[2];
    return V2{{x.x, x.y, y.x}, {y.y, z.x, z.y}};
}
__global__ __launch_bounds__(256) void k_H_pairs(Geo g, const double* __restrict__ a0, const double* __restrict__ a1, const double* __restrict__ a2, const double* __restrict__ a3,
                                                 const double* __restrict__ a4, const double* __restrict__ a5, const double* __restrict__ src, const double* __restrict__ U,
                                                 const double* __restrict__ rAU, double* __restrict__ out) {
    const int c = swz_block(blockIdx.x, gridDim.x) * 512 + 2 * (int)threadIdx.x;
    if (c >= g.N) return;
    const int i = c % g.nx, q_ = c / g.nx, j = q_ % g.ny, k = q_ / g.ny;
    const int sy = g.nx, sz = g.nx * g.ny, last = g.N - 2;
    // gather first: every pair from an always-valid, aligned address
    const V2 S = ldv2(src, c), Uc = ldv2(U, c);
    const V2 Uym = ldv2(U, max(c - sy, 0)), Uyp = ldv2(U, min(c + sy, last)), Uzm = ldv2(U, max(c - sz, 0)), Uzp = ldv2(U, min(c + sz, last));
    const int xm = max(c - 1, 0), xp = min(c + 2, g.N - 1);
    const double uxm[3] = {U[3 * (size_t)xm], U[3 * (size_t)xm + 1], U[3 * (size_t)xm + 2]}, uxp[3] = {U[3 * (size_t)xp], U[3 * (size_t)xp + 1], U[3 * (size_t)xp + 2]};
    const double2 A0 = *reinterpret_cast<const double2*>(a0 + c), A1 = *reinterpret_cast<const double2*>(a1 + c), A2 = *reinterpret_cast<const double2*>(a2 + c),
                  A3 = *reinterpret_cast<const double2*>(a3 + c), A4 = *reinterpret_cast<const double2*>(a4 + c), A5 = *reinterpret_cast<const double2*>(a5 + c),
                  R = *reinterpret_cast<const double2*>(rAU + c);
    double acc0[3] = {S.a[0], S.a[1], S.a[2]}, acc1[3] = {S.b[0], S.b[1], S.b[2]};
    // same order of the six faces per cell as k_H: x-, x+, y-, y+, z-, z+
    if (i > 0) for (int q = 0; q < 3; ++q) acc0[q] -= A0.x * uxm[q];
    for (int q = 0; q < 3; ++q) acc0[q] -= A1.x * Uc.b[q];                       // (i + 1 <= nx - 1: the pair shares a row)
    if (j > 0) for (int q = 0; q < 3; ++q) acc0[q] -= A2.x * Uym.a[q];
    if (j < g.ny - 1) for (int q = 0; q < 3; ++q) acc0[q] -= A3.x * Uyp.a[q];
    if (k > 0) for (int q = 0; q < 3; ++q) acc0[q] -= A4.x * Uzm.a[q];
    if (k < g.nz - 1) for (int q = 0; q < 3; ++q) acc0[q] -= A5.x * Uzp.a[q];
    for (int q = 0; q < 3; ++q) acc1[q] -= A0.y * Uc.a[q];
    if (i + 1 < g.nx - 1) for (int q = 0; q < 3; ++q) acc1[q] -= A1.y * uxp[q];
    if (j > 0) for (int q = 0; q < 3; ++q) acc1[q] -= A2.y * Uym.b[q];
    if (j < g.ny - 1) for (int q = 0; q < 3; ++q) acc1[q] -= A3.y * Uyp.b[q];
    if (k > 0) for (int q = 0; q < 3; ++q) acc1[q] -= A4.y * Uzm.b[q];
    if (k < g.nz - 1) for (int q = 0; q < 3; ++q) acc1[q] -= A5.y * Uzp.b[q];
    double2* o = reinterpret_cast<double2*>(out + 3 * (size_t)c);
    o[0] = make_double2(R.x * (acc0[0] * 0.5), R.x * (acc0[1] * 0.5));
    o[1] = make_double2(R.x * (acc0[2] * 0.5), R.y * (acc1[0] * 0.5));
    o[2] = make_double2(R.y * (acc1[1] * 0.5), R.y * (acc1[2] * 0.5));
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 160;
    Geo g{n, n, n, n * n * n};
    const size_t N = g.N;
    double *a[6], *src, *U, *rAU, *out;
    std::vector<double> h(3 * N);
    for (size_t q = 0; q < 3 * N; ++q) h[q] = 1.0 + 1e-3 * (double)(q % 977);
    for (auto& p : a) { CK(hipMalloc(&p, N * 8)); CK(hipMemcpy(p, h.data(), N * 8, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&src, 3 * N * 8)); CK(hipMalloc(&U, 3 * N * 8)); CK(hipMalloc(&out, 3 * N * 8)); CK(hipMalloc(&rAU, N * 8));
    CK(hipMemcpy(src, h.data(), 3 * N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(U, h.data(), 3 * N * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rAU, h.data(), N * 8, hipMemcpyHostToDevice));
    const int nb = ((int)((N + 255) / 256) + 7) & ~7;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 4; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it) {
                if (variant == 0) hipLaunchKernelGGL(k_H<0>, dim3(nb), dim3(256), 0, 0, g, a[0], a[1], a[2], a[3], a[4], a[5], src, U, rAU, out);
                else if (variant == 1) hipLaunchKernelGGL(k_H<1>, dim3(nb), dim3(256), 0, 0, g, a[0], a[1], a[2], a[3], a[4], a[5], src, U, rAU, out);
                else if (variant == 2) hipLaunchKernelGGL(k_H_lds, dim3(nb), dim3(256), 0, 0, g, a[0], a[1], a[2], a[3], a[4], a[5], src, U, rAU, out);
                else hipLaunchKernelGGL(k_H_pairs, dim3((((int)((N / 2 + 255) / 256)) + 7) & ~7), dim3(256), 0, 0, g, a[0], a[1], a[2], a[3], a[4], a[5], src, U, rAU, out);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / 5 < best) best = ms / 5;
        }
        double sum = 0;
        CK(hipMemcpy(h.data(), out, 3 * N * 8, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < 3 * N; ++q) sum += h[q];
        printf("%-28s %8.1f us   %6.0f GB/s of 128 B/cell   checksum %.10e\n", variant == 0 ? "AoS (component loads)" : variant == 1 ? "component planes" : variant == 2 ? "AoS, LDS-staged coalesced" : "AoS, two cells per thread",
               best * 1e3, 128.0 * N / (best * 1e-3) / 1e9, sum);
    }
    return 0;
}
