// FV half of the hot path as hand-written HIP for gfx950: the finite-volume operators behind icoFoamYade.C:65-149 and
// pimpleFoamYade.C:60-114 / UcEqn.H / pEqn.H on a uniform hex block.  One lane per cell (or per face), consecutive lanes on
// consecutive x-cells => every array access is a coalesced stream; y/z neighbours are re-reads served by L2.  FP64, no MFMA.
// Operator semantics follow OpenFOAM-6 (see DESIGN.md "FV discretisation"); parity for this half is UNPINNED (no OpenFOAM here).
// This source is compiled TWICE: as it stands (FY_FVK_GRADED 0: the uniform block of cubes, every kernel works with the constants dx, Af, V --
// namespace fy) and through fv_kernels_graded.hip (FY_FVK_GRADED 1: a graded, rectilinear block with per-axis cell sizes -- namespace fy::gr).
// Every geometric quantity goes through the geo_* functions below; in the uniform build they ARE the constants, in the expressions the
// kernels were written and measured with.
#include "fv_kernels.hpp"

#include "common.hpp"

#ifndef FY_FVK_GRADED
#define FY_FVK_GRADED 0
#endif

// register caps of the gather-first sweeps (waves per SIMD the compiler must leave room for; 0 = its own choice): build-time constants, measured per kernel
#ifndef FY_WPE_FRONT
#define FY_WPE_FRONT 4
#endif
#ifndef FY_WPE_BACK
#define FY_WPE_BACK 0
#endif
#ifndef FY_WPE_PRE
#define FY_WPE_PRE 0
#endif
#ifndef FY_WPE_BMOM
#define FY_WPE_BMOM 0
#endif
#define FY_WPE_ATTR_(n) __attribute__((amdgpu_waves_per_eu(n)))
#define FY_WPE_ATTR(n) FY_WPE_ATTR_IF(n)
#define FY_WPE_ATTR_IF(n) FY_WPE_PICK_##n
#define FY_WPE_PICK_0
#define FY_WPE_PICK_2 FY_WPE_ATTR_(2)
#define FY_WPE_PICK_3 FY_WPE_ATTR_(3)
#define FY_WPE_PICK_4 FY_WPE_ATTR_(4)
#define FY_WPE_PICK_5 FY_WPE_ATTR_(5)
#define FY_WPE_PICK_6 FY_WPE_ATTR_(6)
#define FY_WPE_PICK_7 FY_WPE_ATTR_(7)
#define FY_WPE_PICK_8 FY_WPE_ATTR_(8)
namespace fy {
#if FY_FVK_GRADED
namespace gr {
#endif
namespace {

constexpr double kSmall = 1e-15;     // OpenFOAM `small`

// ------------------------------------------------------------------------------------------------ index helpers
// t = owned-cell number in [0, Nc): lattice coordinates (k local) and storage index c = t + c0
__device__ __forceinline__ void ijk_of(const FvGeo& g, int t, int& i, int& j, int& k) {
    i = t % g.nx;
    const int q = t / g.nx;
    j = q % g.ny;
    k = q / g.ny;
}
__device__ __forceinline__ int cidx(const FvGeo& g, int i, int j, int k) { return i + g.nx * (j + g.ny * (k + g.gz)); }
__device__ __forceinline__ int stride_of(const FvGeo& g, int d) { return d == 0 ? 1 : d == 1 ? g.nx : g.nx * g.ny; }
// doubles between the rows of the stress tensor field G (stored as three vec3 fields over the whole storage, ghost planes included)
__device__ __forceinline__ size_t g_row_stride(const FvGeo& g) { return 3 * (size_t)g.nx * g.ny * (size_t)(g.nz + 2 * g.gz); }
__device__ __forceinline__ int ndim(const FvGeo& g, int d) { return d == 0 ? g.nx : d == 1 ? g.ny : g.nz; }
__device__ __forceinline__ int fid(const FvGeo& g, int d, int i, int j, int k) {
    return d == 0 ? i + (g.nx + 1) * (j + g.ny * k) : d == 1 ? i + g.nx * (j + (g.ny + 1) * k) : i + g.nx * (j + g.ny * k);
}
__device__ __forceinline__ int cface(const FvGeo& g, int d, int s, int i, int j, int k) {
    return fid(g, d, i + (d == 0 ? s : 0), j + (d == 1 ? s : 0), k + (d == 2 ? s : 0));
}
// ------------------------------------------------------------------------------------------------ geometry
// q = index along axis d: of a cell (geo_h, geo_rh, geo_rhalf, geo_lerp_side's qc) or of a face (geo_lerp, geo_rdelta: face q lies between
// cells q - 1 and q).  Linear-interpolation weight of the low-side cell at an internal face = distance from the face to the high-side
// centre over the centre distance [OF-6 surfaceInterpolation::weights]; |Sf| / |d| from the face area and the centre distance, centre to
// face at a boundary [OF-6 deltaCoeffs / fvPatch::deltaCoeffs].  kBfac: the uniform code writes a boundary coefficient as 2 x (gamma dx).
#if FY_FVK_GRADED
[[maybe_unused]] __device__ __forceinline__ double geo_h(const FvGeo& g, int d, int q) { return g.h[d][q]; }
__device__ __forceinline__ double geo_rh(const FvGeo& g, int d, int q) { return 1.0 / g.h[d][q]; }
__device__ __forceinline__ double geo_rhalf(const FvGeo& g, int d, int q) { return 2.0 / g.h[d][q]; }
__device__ __forceinline__ double geo_rdelta(const FvGeo& g, int d, int q) { return 2.0 / (g.h[d][q - 1] + g.h[d][q]); }
__device__ __forceinline__ double geo_V(const FvGeo& g, int i, int j, int k) { return (g.h[0][i] * g.h[1][j]) * g.h[2][k]; }
__device__ __forceinline__ double geo_rV(const FvGeo& g, int i, int j, int k) { return 1.0 / geo_V(g, i, j, k); }
// area of a face normal to d whose transverse indices are those of (i, j, k) (cell or face indices: the normal one is not used)
__device__ __forceinline__ double geo_Af(const FvGeo& g, int d, int i, int j, int k) { return d == 0 ? g.h[1][j] * g.h[2][k] : d == 1 ? g.h[0][i] * g.h[2][k] : g.h[0][i] * g.h[1][j]; }
__device__ __forceinline__ double geo_wlow(const FvGeo& g, int d, int q) { return g.h[d][q] / (g.h[d][q - 1] + g.h[d][q]); }
__device__ __forceinline__ double geo_lerp(const FvGeo& g, int d, int q, double lo, double hi) { const double w = geo_wlow(g, d, q); return w * lo + (1.0 - w) * hi; }
constexpr double kBfac = 1.0;
#else
[[maybe_unused]] __device__ __forceinline__ double geo_h(const FvGeo& g, int, int) { return g.dx; }
__device__ __forceinline__ double geo_rh(const FvGeo& g, int, int) { return g.rdx; }
__device__ __forceinline__ double geo_rhalf(const FvGeo& g, int, int) { return g.rhdx; }
__device__ __forceinline__ double geo_rdelta(const FvGeo& g, int, int) { return g.rdx; }
__device__ __forceinline__ double geo_V(const FvGeo& g, int, int, int) { return g.V; }
__device__ __forceinline__ double geo_rV(const FvGeo& g, int, int, int) { return g.rV; }
__device__ __forceinline__ double geo_Af(const FvGeo& g, int, int, int, int) { return g.Af; }
[[maybe_unused]] __device__ __forceinline__ double geo_wlow(const FvGeo&, int, int) { return 0.5; }
__device__ __forceinline__ double geo_lerp(const FvGeo&, int, int, double lo, double hi) { return 0.5 * (lo + hi); }
constexpr double kBfac = 2.0;
#endif
// value at face (d, s) of the cell with index qc along d, from the cell's own value and its neighbour's across that face
__device__ __forceinline__ double geo_lerp_side(const FvGeo& g, int d, int s, int qc, double own, double nb) {
#if FY_FVK_GRADED
    return s ? geo_lerp(g, d, qc + 1, own, nb) : geo_lerp(g, d, qc, nb, own);
#else
    return 0.5 * (own + nb);
#endif
}
// linear interpolate at face (d, s) of a cell from its own value and the value across the face (q = the FACE's index along d)
__device__ __forceinline__ double lerp_face(const FvGeo& g, int d, int s, int q, double own, double nbv) {
    return s ? geo_lerp(g, d, q, own, nbv) : geo_lerp(g, d, q, nbv, own);
}
// weight of the cell's own value at its face (d, s)
__device__ __forceinline__ double geo_wown(const FvGeo& g, int d, int s, int qc) {
#if FY_FVK_GRADED
    return s ? geo_wlow(g, d, qc + 1) : 1.0 - geo_wlow(g, d, qc);
#else
    return 0.5;
#endif
}
// 1 / (centre distance) across face (d, s) of the cell (i, j, k): centre to face on a physical boundary
__device__ __forceinline__ bool onb(const FvGeo& g, int d, int s, int i, int j, int k);
__device__ __forceinline__ double geo_rdist(const FvGeo& g, int d, int s, int i, int j, int k) {
#if FY_FVK_GRADED
    const int qc = d == 0 ? i : d == 1 ? j : k;
    return onb(g, d, s, i, j, k) ? geo_rhalf(g, d, qc) : geo_rdelta(g, d, qc + s);
#else
    return onb(g, d, s, i, j, k) ? g.rhdx : g.rdx;
#endif
}
// |Sf| / |d| of face (d, s) of cell (i, j, k).  Uniform block: dx for EVERY face -- the boundary's factor 2 is kBfac, where the uniform code has it
__device__ __forceinline__ double geo_sfd(const FvGeo& g, int d, int s, int i, int j, int k) {
#if FY_FVK_GRADED
    return geo_Af(g, d, i, j, k) * geo_rdist(g, d, s, i, j, k);
#else
    return g.dx;
#endif
}
// the same for a FACE given by its own indices (fi, fj, fk) (q = its index along d): boundary faces have q = 0 or q = n
__device__ __forceinline__ double geo_sfd_face(const FvGeo& g, int d, int q, int nq, int fi, int fj, int fk) {
#if FY_FVK_GRADED
    const double rd = q == 0 ? geo_rhalf(g, d, 0) : (q == nq ? geo_rhalf(g, d, nq - 1) : geo_rdelta(g, d, q));
    return geo_Af(g, d, fi, fj, fk) * rd;
#else
    return g.dx;
#endif
}
// cell size along d of the cell with STORAGE index c
__device__ __forceinline__ double geo_hc(const FvGeo& g, int d, int c) {
#if FY_FVK_GRADED
    const int t = c - g.c0;
    return g.h[d][d == 0 ? t % g.nx : d == 1 ? (t / g.nx) % g.ny : t / (g.nx * g.ny)];
#else
    return g.dx;
#endif
}
// LESdelta cubeRootVol [OF-6 cubeRootVolDelta.C]: deltaCoeff * cbrt(V) of the cell (storage index c); the uniform block's is one number
__device__ __forceinline__ double geo_delta(const FvGeo& g, double uniform_delta, int c) {
#if FY_FVK_GRADED
    return g.turb_dcoeff * cbrt((geo_hc(g, 0, c) * geo_hc(g, 1, c)) * geo_hc(g, 2, c));
#else
    return uniform_delta;
#endif
}
// nearWallDist: distance of the cell centre from its face on side d (half the cell's extent along d)
__device__ __forceinline__ double geo_ywall(const FvGeo& g, int d, int c) { return 0.5 * geo_hc(g, d, c); }

// is face (d, s) of owned cell (i,j,k) on a PHYSICAL boundary?  (slab interfaces in z are interior faces)
__device__ __forceinline__ bool onb(const FvGeo& g, int d, int s, int i, int j, int k) {
    if (d == 2) { const int kg = k + g.kglob0; return s ? kg == g.nzglob - 1 : kg == 0; }
    const int q = d == 0 ? i : j;
    return s ? q == ndim(g, d) - 1 : q == 0;
}
// face coordinate q along d (0..ndim): physical low / high boundary?
__device__ __forceinline__ bool face_low_b(const FvGeo& g, int d, int q) { return d == 2 ? (q + g.kglob0 == 0) : (q == 0); }
__device__ __forceinline__ bool face_high_b(const FvGeo& g, int d, int q) { return d == 2 ? (q + g.kglob0 == g.nzglob) : (q == ndim(g, d)); }
__device__ __forceinline__ void Ub(const FvGeo& g, const double* F, int c, int patch, double* out) {
    if (g.u_bc[patch] == 0) { out[0] = g.u_val[patch][0]; out[1] = g.u_val[patch][1]; out[2] = g.u_val[patch][2]; }
    else { out[0] = F[3 * (size_t)c]; out[1] = F[3 * (size_t)c + 1]; out[2] = F[3 * (size_t)c + 2]; }
    // symmetryPlane / slip on a planar, axis-aligned patch [OF-6 basicSymmetryFvPatchField::evaluate]: the cell value without its normal component
    if (g.u_bc[patch] == 2) out[patch >> 1] = 0.0;
}
__device__ __forceinline__ double pbv(const FvGeo& g, const double* p, const CFace3& psn, int c, int d, int s, int face) {
    const int patch = 2 * d + s;
    if (g.p_bc[patch] == 1) return g.p_val[patch];
    if (g.p_bc[patch] == 2) return p[c] + (s ? 0.5 : -0.5) * geo_hc(g, d, c) * psn.a[d][face];
    return p[c];
}
// boundary value of nut on patch `patch` next to cell c (FvGeo: nut_bc 0 zeroGradient, 1 fixedValue, 2 nutkWallFunction)
__device__ __forceinline__ double nut_boundary(const FvGeo& g, int patch, int c) {
    const int t = g.nut_bc[patch];
    if (t == 1 || ((t == 2 || t == 3) && !g.nut_wall_live)) return g.nut_val[patch];
    if (t == 3) {                                       // calculated: the model's expression on the boundary values
        const double kb = g.k_bc[patch] == 1 ? g.k_val[patch] : g.kturb[c];
        if (g.turb_model == 2) return g.turb_ck * sqrt(kb) * geo_delta(g, g.turb_delta, c);
        const double eb = g.eps_bc[patch] == 1 ? g.eps_val[patch] : g.epsturb[c];
        return g.turb_cmu * (kb * kb) / eb;
    }
    if (t == 2) {
        const double y = geo_ywall(g, patch >> 1, c);
        const double yPlus = g.wf_cmu25 * y * sqrt(g.kturb[c]) / g.nu;
        return yPlus > g.wf_yPlusLam ? g.nu * (yPlus * g.wf_kappa / log(g.wf_E * yPlus) - 1.0) : 0.0;
    }
    return g.nut[c];
}
// Value held by the previous / next lane of the wave (undefined in lane 0 / 63): one DPP move per dword.  A wave's lanes are consecutive
// x-cells, so the x-neighbours of a stencil are already in registers next door -- two more full-wave loads of an AoS vector field
// (24 cache lines each way for the address unit) become two single-lane loads at the wave's ends.
__device__ __forceinline__ double wave_prev(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);      // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);      // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// x-neighbours of an AoS vector field from the neighbouring lanes; call in wave-uniform control flow (every lane that holds a cell)
__device__ __forceinline__ void x_neighbours3(const double* __restrict__ F, int c, int i, int nx, const double (&own)[3], double (&L)[3], double (&R)[3]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 3; ++q) { L[q] = wave_prev(own[q]); R[q] = wave_next(own[q]); }
    if (lane == 0 && i > 0) for (int q = 0; q < 3; ++q) L[q] = F[3 * (size_t)(c - 1) + q];
    if (lane == 63 && i + 1 < nx) for (int q = 0; q < 3; ++q) R[q] = F[3 * (size_t)(c + 1) + q];
}
// XCD-aware block order (guide T1): block b runs on XCD b % 8; give every XCD one contiguous z-slab of the grid so that the
// y/z-neighbour re-reads of a stencil hit that XCD's own L2.  Pure speed: any mapping is correct.
__device__ __forceinline__ int swz_block(int bid, int nblk) {
    if (nblk % 8) return bid;
    return (bid % 8) * (nblk / 8) + bid / 8;
}
// Strip order (round 5): on top of the XCD's contiguous z-slab, its blocks run strip by strip -- a strip = strip_B consecutive 256-cell blocks of a
// plane (a few rows), followed by the same strip of the next plane, and so on up the slab, before the next strip starts.  The cell above /
// below the one a block is working on was then touched strip_B blocks ago instead of a whole plane of blocks ago: with a dozen arrays in a sweep a
// plane of them (5 - 6 MB at 160^2) does not survive in the XCD's 4 MB L2 until the next plane comes by, a strip of them (300 KB) does.
// Host side (Solver::create): strips only when planes are whole numbers of blocks and every XCD gets whole planes; FOAMYADE_STRIP_BLOCKS.
// Returns the LOGICAL block (cells [256 b, 256 b + 256)); partial sums are stored under it, so folds do not depend on the order.
__device__ __forceinline__ int fv_block(const FvGeo& g, int bid, int nblk) {
    if (g.win_nblk > 0) return g.win_blk0 + bid;            // a window of whole z-planes (its blocks in storage order)
    if (g.strip_B <= 0) return swz_block(bid, nblk);
    const int x = bid & 7, l = bid >> 3;
    const int per = g.strip_nzx * g.strip_B;
    const int s = l / per, rem = l - s * per;
    const int kk = rem / g.strip_B, b = rem - kk * g.strip_B;
    return (x * g.strip_nzx + kk) * g.strip_bp + s * g.strip_B + b;
}

// ------------------------------------------------------------------------------------------------ block reductions (256 threads)
template <int N>
__device__ __forceinline__ void block_reduce_store(double (&v)[N], const int (&is_max)[N], double* partials, int lb = -1, int stride = 0) {
    if (lb < 0) lb = (int)blockIdx.x;
    if (stride <= 0) stride = (int)gridDim.x;              // partials of slot q start at q * (blocks of the WHOLE sweep)
    __shared__ double sh[4][N];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double x = v[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double y = __shfl_down(x, o, 64);
            x = is_max[q] ? fmax(x, y) : x + y;
        }
        if (lane == 0) sh[wv][q] = x;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int q = threadIdx.x;
        double x = sh[0][q];
        for (int w = 1; w < 4; ++w) x = is_max[q] ? fmax(x, sh[w][q]) : x + sh[w][q];
        partials[(size_t)q * stride + lb] = x;
    }
}

// fold the per-block partials of one slot per workgroup, in a fixed order: 1024 threads stride over the partials (16 000 of them at
// 160^3), then a shuffle + LDS tree.  (256 threads took 17 us per call x 23 calls per step.)
// flag != nullptr (out and flag in mapped host memory): the result is followed by a system-scope fence and flag[slot] = seq, which is
// what the host spins on instead of waiting for the stream to drain (a stream synchronisation costs ~15 us of idle GPU per read-back).
__global__ __launch_bounds__(1024) void k_reduce_finalize(const double* __restrict__ partials, int nblocks, const int* __restrict__ ops,
                                                          double* __restrict__ out, unsigned long long* flag, unsigned long long seq) {
    __shared__ double sh[16];
    const int slot = blockIdx.x;
    const int mx = ops ? ops[slot] : 0;
    double x = mx ? -1e300 : 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 1024) {
        const double y = partials[(size_t)slot * nblocks + b];
        x = mx ? fmax(x, y) : x + y;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double y = __shfl_down(x, o, 64);
        x = mx ? fmax(x, y) : x + y;
    }
    if (lane == 0) sh[wv] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = sh[0];
        for (int w = 1; w < 16; ++w) r = mx ? fmax(r, sh[w]) : r + sh[w];
        out[slot] = r;
        if (flag) {
            __threadfence_system();
            __hip_atomic_store(&flag[slot], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Reducing kernels: one thread per cell (like the plain stencil kernels -- a 1024-block grid-stride loop reached only ~3 TB/s where
// the one-thread-per-cell smoother reaches 5.4), XCD-aware block order, one partial per block; k_reduce_finalize folds the
// red_blocks(n) partials of a slot in a fixed order, so results are reproducible from run to run.
#define FY_RED_LOOP(t, n) const int t = swz_block(blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x; if (t < (n))
// the same over the owned cells of g, blocks in strip order; fy_lb = the logical block (pass it to block_reduce_store)
#define FY_RED_LOOP_G(g, t) const int fy_lb = fv_block(g, blockIdx.x, gridDim.x); const int fy_stride = (g).win_nblk > 0 ? (g).red_stride : (int)gridDim.x; \
    const int t = fy_lb * 256 + (int)threadIdx.x; if (t < (g).Nc)

// ------------------------------------------------------------------------------------------------ face kernels
// generic face iteration: thread -> (d fixed per launch, face index f) -> local (i,j,k) of the face
__device__ __forceinline__ bool face_ijk(const FvGeo& g, int d, size_t f, int& i, int& j, int& k) {
    const int ex = g.nx + (d == 0), ey = g.ny + (d == 1), ez = g.nz + (d == 2);
    if (f >= (size_t)ex * ey * ez) return false;
    i = (int)(f % ex);
    const size_t t = f / ex;
    j = (int)(t % ey);
    k = (int)(t / ey);
    return true;
}

// fvc::flux(F) = linearInterpolate(F) & Sf with U's boundary conditions (createPhi; flux(HbyA) with constrainHbyA)
__device__ __forceinline__ double face_flux_vec(const FvGeo& g, const double* F, int d, int i, int j, int k) {
    const int q = d == 0 ? i : d == 1 ? j : k;
    double v;
    if (face_low_b(g, d, q)) { double b[3]; Ub(g, F, cidx(g, i, j, k), 2 * d, b); v = b[d]; }
    else if (face_high_b(g, d, q)) { double b[3]; Ub(g, F, cidx(g, i - (d == 0), j - (d == 1), k - (d == 2)), 2 * d + 1, b); v = b[d]; }
    else { const int c = cidx(g, i, j, k); v = geo_lerp(g, d, q, F[3 * (size_t)(c - stride_of(g, d)) + d], F[3 * (size_t)c + d]); }
    return v * geo_Af(g, d, i, j, k);
}

template <int D>
__global__ __launch_bounds__(256) void k_flux_of(FvGeo g, const double* __restrict__ F, double* __restrict__ out) {
    const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
    int i, j, k;
    if (!face_ijk(g, D, f, i, j, k)) return;
    out[f] = face_flux_vec(g, F, D, i, j, k);
}

template <int D>
__device__ __forceinline__ void interp_alpha_face(const FvGeo& g, size_t f, int i, int j, int k, const double* __restrict__ alpha, double* __restrict__ af) {
    const int q = D == 0 ? i : D == 1 ? j : k;
    if (face_low_b(g, D, q) || face_high_b(g, D, q)) af[f] = 1.0;     // calculated patch, value 1 (`alpha = 1.0`, FoamYade.C:68)
    else { const int c = cidx(g, i, j, k); af[f] = geo_lerp(g, D, q, alpha[c - stride_of(g, D)], alpha[c]); }
}

template <int D>
__global__ __launch_bounds__(256) void k_interp_rAU(FvGeo g, const double* __restrict__ rAU, double* __restrict__ rf) {
    const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
    int i, j, k;
    if (!face_ijk(g, D, f, i, j, k)) return;
    const int q = D == 0 ? i : D == 1 ? j : k;
    if (face_low_b(g, D, q)) rf[f] = rAU[cidx(g, i, j, k)];
    else if (face_high_b(g, D, q)) rf[f] = rAU[cidx(g, i - (D == 0), j - (D == 1), k - (D == 2))];
    else { const int c = cidx(g, i, j, k); rf[f] = geo_lerp(g, D, q, rAU[c - stride_of(g, D)], rAU[c]); }
}

// Every face of the block exactly once from a cell-centred sweep: an owned cell does its three low faces, and the high face where it
// is the last cell of its row / column / slab.  One pass over the cell fields instead of one per direction (the three per-direction
// launches each pulled the whole AoS vector fields through for one component: 264 -> 168 B/cell for phiHbyA); same arithmetic per face.
#define FY_CELL_FACES(g, i, j, k, CALL)                     \
    do {                                                    \
        CALL(0, i, j, k); if (i == g.nx - 1) CALL(0, i + 1, j, k); \
        CALL(1, i, j, k); if (j == g.ny - 1) CALL(1, i, j + 1, k); \
        CALL(2, i, j, k); if (k == g.nz - 1) CALL(2, i, j, k + 1); \
    } while (0)

// does cell (i, j, k) store face (D, S)?  its low faces always, a high face where no cell lies beyond it in this domain
__device__ __forceinline__ bool owns_face(const FvGeo& g, int d, int s, int i, int j, int k) {
    return s == 0 || (d == 0 ? i == g.nx - 1 : d == 1 ? j == g.ny - 1 : k == g.nz - 1);
}

// rAUcf = fvc::interpolate(rAUc) and phicForces = fvc::flux(rAUc*uSource) + rAUcf*(g & Sf) in one cell-centred sweep (UcEqn.H:15-20;
// uSource's calculated boundary value is 0)
template <int D>
__device__ __forceinline__ void rAUf_phi_forces_face(const FvGeo& g, size_t f, int i, int j, int k, const double* __restrict__ rAU,
                                                     const double* __restrict__ uSource, double* __restrict__ rf, double* __restrict__ out) {
    const int q = D == 0 ? i : D == 1 ? j : k;
    double r, fl = 0.0;
    if (face_low_b(g, D, q)) r = rAU[cidx(g, i, j, k)];
    else if (face_high_b(g, D, q)) r = rAU[cidx(g, i - (D == 0), j - (D == 1), k - (D == 2))];
    else {
        const int c = cidx(g, i, j, k), cm = c - stride_of(g, D);
        const double rm = rAU[cm], rc = rAU[c];
        r = geo_lerp(g, D, q, rm, rc);
        fl = geo_lerp(g, D, q, rm * uSource[3 * (size_t)cm + D], rc * uSource[3 * (size_t)c + D]) * geo_Af(g, D, i, j, k);
    }
    rf[f] = r;
    out[f] = fl + r * (g.g[D] * geo_Af(g, D, i, j, k));
}
__global__ __launch_bounds__(256) void k_rAUf_phi_forces_cells(FvGeo g, const double* __restrict__ rAU, const double* __restrict__ uSource, Face3 rf, Face3 out) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
#define FY_CALL(D, fi, fj, fk) { const size_t f = (size_t)fid(g, D, fi, fj, fk); rAUf_phi_forces_face<D>(g, f, fi, fj, fk, rAU, uSource, rf.a[D], out.a[D]); }
    FY_CELL_FACES(g, i, j, k, FY_CALL);
#undef FY_CALL
}

__global__ __launch_bounds__(256) void k_interp_alpha_cells(FvGeo g, const double* __restrict__ alpha, Face3 af) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
#define FY_CALL(D, fi, fj, fk) { const size_t f = (size_t)fid(g, D, fi, fj, fk); interp_alpha_face<D>(g, f, fi, fj, fk, alpha, af.a[D]); }
    FY_CELL_FACES(g, i, j, k, FY_CALL);
#undef FY_CALL
}

// phiHbyA = fvc::flux(HbyA) + [alphacf*]rAUf*fvc::ddtCorr(U, phi) [+ phicForces]; constrainPressure on fixedFluxPressure patches
// (icoFoamYade.C:101-111, pEqn.H:4-21).  ddtCorr = EulerDdtScheme::fvcDdtPhiCorr with fvcDdtPhiCoeff.
// KEEP: the ddtCorr term [alphacf] rAUf ddtCorr(U, phi) is made of old-time fields and of rAUf / alphacf, none of which changes between the correctors of one
// momentum assembly: the first corrector (KEEP 1) stores it per face, the later ones (KEEP 2) read it back instead of U.oldTime, phi.oldTime, rAUf and alphacf
// (96 B per cell less; the same value added in the same place, so the same bits).  KEEP 0: neither (no scratch field given)
template <int D, int KEEP>
__device__ __forceinline__ void phiHbyA_face(const FvGeo& g, size_t f, int i, int j, int k, const double* __restrict__ HbyA, const double* __restrict__ U,
                                             const double* __restrict__ Uold, const double* __restrict__ phiOld,
                                             const double* __restrict__ rAUf, const double* __restrict__ alphaf,
                                             const double* __restrict__ phiForces, double* __restrict__ out, double* __restrict__ psn, double* __restrict__ ddtc) {
    const int q = D == 0 ? i : D == 1 ? j : k;
    double v = face_flux_vec(g, HbyA, D, i, j, k);
    int bpatch = -1, bc = -1;
    if (face_low_b(g, D, q)) { bpatch = 2 * D; bc = cidx(g, i, j, k); }
    else if (face_high_b(g, D, q)) { bpatch = 2 * D + 1; bc = cidx(g, i - (D == 0), j - (D == 1), k - (D == 2)); }
    const double Afc = geo_Af(g, D, i, j, k);
    double add;
    if (KEEP == 2) {
        add = ddtc[f];
    } else {
        double uf;
        bool fixes = false;
        if (bpatch >= 0) { fixes = g.u_bc[bpatch] == 0; double b[3]; Ub(g, Uold, bc, bpatch, b); uf = b[D] * Afc; }
        else { const int c = cidx(g, i, j, k); uf = geo_lerp(g, D, q, Uold[3 * (size_t)(c - stride_of(g, D)) + D], Uold[3 * (size_t)c + D]) * Afc; }
        const double po = phiOld[f];
        const double phiCorr = po - uf;
        const double coef = fixes ? 0.0 : 1.0 - fmin(fabs(phiCorr) / (fabs(po) + kSmall), 1.0);
        add = rAUf[f] * (coef * (1.0 / g.dt) * phiCorr);
        if (g.pimple) add *= alphaf[f];
        if (KEEP == 1) ddtc[f] = add;
    }
    v += add;
    if (g.pimple) v += phiForces[f];
    out[f] = v;
    if (bpatch >= 0 && g.p_bc[bpatch] == 2) {
        double ub[3]; Ub(g, U, bc, bpatch, ub);
        psn[f] = (v - ub[D] * Afc) / (rAUf[f] * Afc);
    }
}

template <int KEEP>
__global__ __launch_bounds__(256) void k_phiHbyA_cells(FvGeo g, const double* __restrict__ HbyA, const double* __restrict__ U,
                                                       const double* __restrict__ Uold, CFace3 phiOld, CFace3 rAUf, CFace3 alphaf,
                                                       CFace3 phiForces, Face3 out, Face3 psn, Face3 ddtc) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
#define FY_CALL(D, fi, fj, fk) { const size_t f = (size_t)fid(g, D, fi, fj, fk); \
        phiHbyA_face<D, KEEP>(g, f, fi, fj, fk, HbyA, U, Uold, phiOld.a[D], rAUf.a[D], alphaf.a[D], phiForces.a[D], out.a[D], psn.a[D], ddtc.a[D]); }
    FY_CELL_FACES(g, i, j, k, FY_CALL);
#undef FY_CALL
}

// ---- adjustPhi(phiHbyA, U, p) (icoFoamYade.C:108, pEqn.H:13-16) [OF-6 cfdTools/general/adjustPhi/adjustPhi.C].  Acts when no patch
// fixes the pressure: the outflow through the patches that do not fix U is scaled by massCorr = (massIn - fixedMassOut) / adjustableMassOut
// so that the pressure equation is solvable.  In pimpleFoamYade it is applied BEFORE phicForces are added (pEqn.H:13-18); phiHbyA here
// already carries them, so the flux it works on is phiHbyA - phicForces.  slots: 0 massIn, 1 fixedMassOut, 2 adjustableMassOut,
// 3 sum |flux| over the internal faces (the normalisation of the two tests)
template <int D>
__device__ __forceinline__ void adjust_phi_face(const FvGeo& g, size_t f, int i, int j, int k, const double* __restrict__ phiHbyA,
                                                const double* __restrict__ phiForces, double* v) {
    const int q = D == 0 ? i : D == 1 ? j : k;
    const double fl = phiHbyA[f] - (g.pimple ? phiForces[f] : 0.0);
    const bool lo = face_low_b(g, D, q), hi = face_high_b(g, D, q);
    if (lo || hi) {
        const double outw = lo ? -fl : fl;
        const bool fixes = g.u_bc[2 * D + (hi ? 1 : 0)] == 0;
        if (outw < 0.0) v[0] -= outw;
        else if (fixes) v[1] += outw;
        else v[2] += outw;
    } else if (!(D == 2 && k == g.nz)) {                    // a slab's top interface face belongs to the slab above (counted there)
        v[3] += fabs(fl);
    }
}
__global__ __launch_bounds__(256) void k_adjust_phi_sums(FvGeo g, CFace3 phiHbyA, CFace3 phiForces, double* __restrict__ partials) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    FY_RED_LOOP_G(g, t) {
        int i, j, k; ijk_of(g, t, i, j, k);
#define FY_CALL(D, fi, fj, fk) { const size_t f = (size_t)fid(g, D, fi, fj, fk); adjust_phi_face<D>(g, f, fi, fj, fk, phiHbyA.a[D], phiForces.a[D], v); }
        FY_CELL_FACES(g, i, j, k, FY_CALL);
#undef FY_CALL
    }
    const int mx[4] = {0, 0, 0, 0};
    block_reduce_store<4>(v, mx, partials, fy_lb, fy_stride);
}
// one thread per boundary face of the block (both sides of the three directions); err: set when OpenFOAM would stop with
// "Continuity error cannot be removed by adjusting the outflow"
__global__ __launch_bounds__(256) void k_adjust_phi_apply(FvGeo g, const double* __restrict__ sums, Face3 phiHbyA, CFace3 phiForces, CFace3 rAUf,
                                                          const double* __restrict__ U, Face3 psn, int* __restrict__ err) {
    const double massIn = sums[0], fixedOut = sums[1], adjOut = sums[2], total = 1e-300 + sums[3];
    double massCorr = 1.0;
    if (fabs(adjOut) > 1e-300 && fabs(adjOut) / total > kSmall) massCorr = (massIn - fixedOut) / adjOut;
    else if (fabs(fixedOut - massIn) / total > 1e-8) { if (blockIdx.x == 0 && threadIdx.x == 0) *err = 1; return; }
    if (massCorr == 1.0) return;
    const int nb[3] = {g.ny * g.nz, g.nx * g.nz, g.nx * g.ny};
    int t = blockIdx.x * 256 + threadIdx.x;
    int d = 0, s = 0;
    for (; d < 3; ++d) { if (t < 2 * nb[d]) { s = t >= nb[d]; t -= s * nb[d]; break; } t -= 2 * nb[d]; }
    if (d == 3) return;
    int i, j, k;
    if (d == 0) { j = t % g.ny; k = t / g.ny; i = s ? g.nx : 0; }
    else if (d == 1) { i = t % g.nx; k = t / g.nx; j = s ? g.ny : 0; }
    else { i = t % g.nx; j = t / g.nx; k = s ? g.nz : 0; }
    const int q = d == 0 ? i : d == 1 ? j : k;
    if (!(s ? face_high_b(g, d, q) : face_low_b(g, d, q))) return;        // (a slab's interface plane is not a patch)
    const int patch = 2 * d + s;
    if (g.u_bc[patch] == 0) return;                                        // fixes the value: not adjustable
    const size_t f = (size_t)fid(g, d, i, j, k);
    const double pf = g.pimple ? phiForces.a[d][f] : 0.0;
    const double fl = phiHbyA.a[d][f] - pf;
    const double outw = s ? fl : -fl;
    if (!(outw > 0.0)) return;
    const double v = fl * massCorr + pf;
    phiHbyA.a[d][f] = v;
    if (g.p_bc[patch] == 2) {                                              // constrainPressure sees the adjusted flux (pEqn.H:21)
        const int c = cidx(g, i - (d == 0 && s), j - (d == 1 && s), k - (d == 2 && s));
        double ub[3]; Ub(g, U, c, patch, ub);
        const double Afc = geo_Af(g, d, i, j, k);
        psn.a[d][f] = (v - ub[d] * Afc) / (rAUf.a[d][f] * Afc);
    }
}

// pEqn.flux() and phi = phiHbyA - pEqn.flux()[/alphacf]   (icoFoamYade.C:129, pEqn.H:39), all three directions in one cell-centred sweep
// (three per-direction launches: 3 x 36 us at 160^3; the pressure field went through three times).
// pimple: what is kept per face is not pEqn.flux() itself but the face term of the velocity reconstruction that follows (pEqn.H:43-45),
// (phicForces - pEqn.flux() / alphacf) / rAUcf -- its only reader, k_U_correct, then streams ONE face field instead of four (the same
// operations on the same operands in the same order, so the same bits; 48 B per cell and corrector less traffic)
template <int D>
__device__ __forceinline__ void flux_correct_face(const FvGeo& g, size_t f, int i, int j, int k, const double* __restrict__ p, const double* __restrict__ phiHbyA,
                                                  const double* __restrict__ rAUf, const double* __restrict__ alphaf, const double* __restrict__ psn,
                                                  const double* __restrict__ phiForces, double* __restrict__ pflux, double* __restrict__ phi) {
    const int q = D == 0 ? i : D == 1 ? j : k;
    const double af = g.pimple ? alphaf[f] : 1.0;
    double fl = 0.0;
    const bool lo = face_low_b(g, D, q), hi = face_high_b(g, D, q);
    if (lo || hi) {
        const int s = lo ? 0 : 1, patch = 2 * D + s;
        const int c = cidx(g, i - (D == 0 && s), j - (D == 1 && s), k - (D == 2 && s));
        if (g.p_bc[patch] == 1) { const double gb = kBfac * af * rAUf[f] * geo_sfd_face(g, D, q, ndim(g, D), i, j, k); fl = s ? gb * (g.p_val[patch] - p[c]) : gb * (p[c] - g.p_val[patch]); }
        else if (g.p_bc[patch] == 2) fl = af * rAUf[f] * geo_Af(g, D, i, j, k) * psn[f];
    } else {
        const int c = cidx(g, i, j, k);
        fl = af * rAUf[f] * geo_sfd_face(g, D, q, ndim(g, D), i, j, k) * (p[c] - p[c - stride_of(g, D)]);
    }
    const double fa = fl / af;
    pflux[f] = g.pimple ? (phiForces[f] - fa) / rAUf[f] : fl;
    phi[f] = phiHbyA[f] - fa;
}
__global__ __launch_bounds__(256) void k_flux_correct_cells(FvGeo g, const double* __restrict__ p, CFace3 phiHbyA, CFace3 rAUf, CFace3 alphaf, CFace3 psn,
                                                            CFace3 phiForces, Face3 pflux, Face3 phi) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
#define FY_CALL(D, fi, fj, fk) { const size_t f = (size_t)fid(g, D, fi, fj, fk); \
        flux_correct_face<D>(g, f, fi, fj, fk, p, phiHbyA.a[D], rAUf.a[D], alphaf.a[D], psn.a[D], phiForces.a[D], pflux.a[D], phi.a[D]); }
    FY_CELL_FACES(g, i, j, k, FY_CALL);
#undef FY_CALL
}

// ------------------------------------------------------------------------------------------------ cell kernels
// CourantNo.H:32-49: sumPhi = fvc::surfaceSum(mag(phi)); slots: 0 = max(sumPhi/V), 1 = sum(sumPhi)
__global__ __launch_bounds__(256) void k_courant(FvGeo g, CFace3 phi, double* __restrict__ partials) {
    double v[2] = {0.0, 0.0};
    FY_RED_LOOP_G(g, t) {
        int i, j, k; ijk_of(g, t, i, j, k);
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) s += fabs(phi.a[d][cface(g, d, sd, i, j, k)]);
        v[0] = fmax(v[0], s * geo_rV(g, i, j, k));
        v[1] += s;
    }
    const int mx[2] = {1, 0};
    block_reduce_store<2>(v, mx, partials, fy_lb, fy_stride);
}

// vGrad = fvc::grad(U) (icoFoamYade.C:71, pimpleFoamYade.C:76); pimple also gradP = fvc::grad(p) (:74) and
// divT = 2 nu fvc::laplacian(alphac, Uc) (:75)
// In pimple mode the only consumer of grad(U) is the explicit stress term of divDevRhoReff (the Gaussian torque that would read vGrad
// is disabled in the reference, FoamYade.C:618), so the kernel can emit G = alpha nu dev2(T(grad U)) directly (Gout != nullptr) and
// skip the 72 B/cell vGrad store (write_vgrad = 0): one stencil pass instead of two plus a tensor round trip through HBM.
__global__ __launch_bounds__(256) FY_WPE_ATTR(FY_WPE_PRE) void k_pre_coupling(FvGeo g, const double* __restrict__ U, const double* __restrict__ p,
                                                      const double* __restrict__ alpha, CFace3 psn, double* __restrict__ vGrad,
                                                      double* __restrict__ gradP, double* __restrict__ divT, double* __restrict__ Gout,
                                                      int write_vgrad, int write_pfields, CFace3 phi, double* __restrict__ ddtU, double* __restrict__ Uold_out,
                                                      double* __restrict__ cellrec, double rec_two_nu, double rec_rhoF, Face3 dcorr) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    const bool pf = g.pimple && write_pfields;          // gradP and divT wanted
    // ---- gather first (round 5): every neighbour value is loaded up front from an address that is always valid (across a boundary face the
    // "neighbour" is the cell itself), the faces are evaluated from registers afterwards.  With the loads inside the per-face branches a wave made
    // one dependent memory round trip per face: 240 us for this sweep's 200 B/cell
    const int ijk[3] = {i, j, k};
    bool bnd[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) bnd[d][s] = onb(g, d, s, i, j, k);
    const double uc[3] = {U[3 * (size_t)c], U[3 * (size_t)c + 1], U[3 * (size_t)c + 2]};
    const double pc = pf ? p[c] : 0.0, ac = pf ? alpha[c] : 0.0;
    double un6[3][2][3], pn6[3][2], an6[3][2];
#pragma unroll
    for (int d = 1; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int nb = bnd[d][s] ? c : c + (s ? stride_of(g, d) : -stride_of(g, d));
            un6[d][s][0] = U[3 * (size_t)nb]; un6[d][s][1] = U[3 * (size_t)nb + 1]; un6[d][s][2] = U[3 * (size_t)nb + 2];
            pn6[d][s] = pf ? p[nb] : 0.0; an6[d][s] = pf ? alpha[nb] : 0.0;
        }
    const bool want_phi = ddtU != nullptr || dcorr.a[0] != nullptr;
    double phf[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) phf[d][s] = want_phi ? phi.a[d][cface(g, d, s, i, j, k)] : 0.0;
    // x-neighbours from the neighbouring lanes (wave_prev / wave_next); the wave's end lanes fetch theirs
    const int lane = threadIdx.x & 63;
    {
        const int xm = (lane == 0 && i > 0) ? c - 1 : c, xp = (lane == 63 && i + 1 < g.nx) ? c + 1 : c;      // (only the end lanes use what they load)
        double em[3] = {0, 0, 0}, ep[3] = {0, 0, 0}, epm = 0, epp = 0, eam = 0, eap = 0;
        if (lane == 0 || lane == 63) {
            for (int q = 0; q < 3; ++q) { em[q] = U[3 * (size_t)xm + q]; ep[q] = U[3 * (size_t)xp + q]; }
            if (pf) { epm = p[xm]; epp = p[xp]; eam = alpha[xm]; eap = alpha[xp]; }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { un6[0][0][q] = wave_prev(uc[q]); un6[0][1][q] = wave_next(uc[q]); }
        pn6[0][0] = wave_prev(pc); pn6[0][1] = wave_next(pc); an6[0][0] = wave_prev(ac); an6[0][1] = wave_next(ac);
        if (lane == 0 && i > 0) { for (int q = 0; q < 3; ++q) un6[0][0][q] = em[q]; pn6[0][0] = epm; an6[0][0] = eam; }
        if (lane == 63 && i + 1 < g.nx) { for (int q = 0; q < 3; ++q) un6[0][1][q] = ep[q]; pn6[0][1] = epp; an6[0][1] = eap; }
    }
    if (Uold_out) { Uold_out[3 * (size_t)c] = uc[0]; Uold_out[3 * (size_t)c + 1] = uc[1]; Uold_out[3 * (size_t)c + 2] = uc[2]; }    // runTime++: U.oldTime() (single domain: no ghost planes to copy)
    double lap[3] = {0, 0, 0}, T[9], conv[3] = {0, 0, 0}, gp3[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double fv[2][3], fp[2] = {0, 0};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (bnd[d][s]) {
                const int patch = 2 * d + s;                       // Ub() on the cell's own value
                if (g.u_bc[patch] == 0) { fv[s][0] = g.u_val[patch][0]; fv[s][1] = g.u_val[patch][1]; fv[s][2] = g.u_val[patch][2]; }
                else { fv[s][0] = uc[0]; fv[s][1] = uc[1]; fv[s][2] = uc[2]; }
                if (g.u_bc[patch] == 2) fv[s][d] = 0.0;
                if (pf) {
                    fp[s] = pbv(g, p, psn, c, d, s, cface(g, d, s, i, j, k));
                    for (int q = 0; q < 3; ++q) lap[q] += 1.0 * geo_Af(g, d, i, j, k) * (fv[s][q] - uc[q]) * geo_rhalf(g, d, ijk[d]);     // alphaf = 1 on the boundary
                }
            } else {
                const double* un = un6[d][s];
                const int qc = ijk[d];
                for (int q = 0; q < 3; ++q) fv[s][q] = geo_lerp_side(g, d, s, qc, uc[q], un[q]);
                if (pf) {
                    fp[s] = geo_lerp_side(g, d, s, qc, pc, pn6[d][s]);
                    const double af = geo_lerp_side(g, d, s, qc, ac, an6[d][s]);
                    for (int q = 0; q < 3; ++q) lap[q] += af * geo_Af(g, d, i, j, k) * (un[q] - uc[q]) * geo_rdelta(g, d, qc + s);
                }
            }
        }
        if (dcorr.a[0]) {
            // the old-time part of ddtCorr(U, phi) (EulerDdtScheme::fvcDdtPhiCorr with fvcDdtPhiCoeff; phiHbyA_face): coef / deltaT (phi.old - flux(U.old)) per
            // face, by the face's owner.  This sweep runs at the start of the step, where U is U.oldTime() and `phi` is phi.oldTime(), and already holds
            // U's face values; the correctors then need neither U.oldTime() nor phi.oldTime() (fused sweep k_corr_front)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (owns_face(g, d, s, i, j, k)) {
                    const int fi = i + (d == 0 ? s : 0), fj = j + (d == 1 ? s : 0), fk = k + (d == 2 ? s : 0);
                    const int f = cface(g, d, s, i, j, k);
                    const double uf = fv[s][d] * geo_Af(g, d, fi, fj, fk);
                    const double po = phf[d][s];
                    const double phiCorr = po - uf;
                    const bool fixes = bnd[d][s] && g.u_bc[2 * d + s] == 0;
                    const double coef = fixes ? 0.0 : 1.0 - fmin(fabs(phiCorr) / (fabs(po) + kSmall), 1.0);
                    dcorr.a[d][f] = coef * (1.0 / g.dt) * phiCorr;
                }
        }
        const double rhd = geo_rh(g, d, ijk[d]);
        for (int q = 0; q < 3; ++q) T[3 * d + q] = (fv[1][q] - fv[0][q]) * rhd;
        if (pf) gp3[d] = (fp[1] - fp[0]) * rhd;
        if (ddtU) {     // fvc::div(phic, Uc), Gauss linear: the face values are the ones the gradient just used
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const double flux = (s ? 1.0 : -1.0) * phf[d][s];
                for (int q = 0; q < 3; ++q) conv[q] += flux * fv[s][q];
            }
        }
    }
    // pimpleFoamYade.C:73 ddtU_f = fvc::ddt(Uc) + fvc::div(phic, Uc); the ddt term is identically zero there (Uc.oldTime() is
    // stored on that access, before Uc is written in the new step), only consumer: addedMassForce (fy_set_force_models)
    if (ddtU)
        for (int q = 0; q < 3; ++q) ddtU[3 * (size_t)c + q] = conv[q] * geo_rV(g, i, j, k);
    if (write_vgrad)
        for (int q = 0; q < 9; ++q) vGrad[9 * (size_t)c + q] = T[q];
    if (pf)
        for (int q = 0; q < 3; ++q) gradP[3 * (size_t)c + q] = gp3[q];
    if (pf) {
        double dT[3];
        for (int q = 0; q < 3; ++q) { dT[q] = 2 * g.nu * (lap[q] * geo_rV(g, i, j, k)); divT[3 * (size_t)c + q] = dT[q]; }
        if (cellrec) {
            // the 64-byte record the force pass gathers per stencil cell, {U, alpha, 2 nu rho_f divT - gradP, V}, written from here instead of by
            // a pass of its own over U / gradP / divT (k_pack_cells: same operands, same operations, same bits; single domain only)
            double2* r = reinterpret_cast<double2*>(cellrec + 8 * (size_t)c);
            r[0] = make_double2(uc[0], uc[1]);
            r[1] = make_double2(uc[2], ac);
            r[2] = make_double2(((rec_two_nu * dT[0]) * rec_rhoF) - gp3[0], ((rec_two_nu * dT[1]) * rec_rhoF) - gp3[1]);
            r[3] = make_double2(((rec_two_nu * dT[2]) * rec_rhoF) - gp3[2], geo_V(g, i, j, k));
        }
    }
    if (Gout) {
        const double tr = T[0] + T[4] + T[8];
        const double an = g.nut ? alpha[c] * (g.nu + g.nut[c]) : alpha[c] * g.nu;      // alpha nuEff (nuEff = nut + nu [OF-6 eddyViscosity/linearViscousStress])
        // stored BY ROWS (three vec3 arrays): k_div_G takes row d from the +-d neighbours only, so it streams 24-byte records
        // instead of picking 24 bytes out of 72-byte ones (fewer lines per load instruction, and the z-planes it re-reads fit L2)
        const size_t rs = g_row_stride(g);
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) Gout[(size_t)a * rs + 3 * (size_t)c + b] = an * (T[3 * b + a] - (a == b ? (2.0 / 3.0) * tr : 0.0));
    }
}

// divergence of the explicit part of divDevRhoReff (laminar Stokes), G = alpha nu dev2(T(grad U)) formed by k_pre_coupling, stored by rows
__global__ __launch_bounds__(256) void k_div_G(FvGeo g, const double* __restrict__ G, double* __restrict__ divG) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    const int ijk[3] = {i, j, k};
    // gather first (round 5): row d of the tensor at the cell and at its +-d neighbours (across a boundary face the cell itself), then the arithmetic
    const double g0[3] = {G[3 * (size_t)c], G[3 * (size_t)c + 1], G[3 * (size_t)c + 2]};      // row x at this cell; its x-neighbours come from the lanes next door
    double gx[2][3];
    x_neighbours3(G, c, i, g.nx, g0, gx[0], gx[1]);
    bool bnd[3][2];
    double own[3][3], nbv[3][2][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { own[0][q] = g0[q]; nbv[0][0][q] = gx[0][q]; nbv[0][1][q] = gx[1][q]; }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) bnd[d][s] = onb(g, d, s, i, j, k);
#pragma unroll
    for (int d = 1; d < 3; ++d) {
        const double* Gd = G + (size_t)d * g_row_stride(g);           // row d of the tensor field
#pragma unroll
        for (int q = 0; q < 3; ++q) own[d][q] = Gd[3 * (size_t)c + q];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int nb = bnd[d][s] ? c : c + (s ? stride_of(g, d) : -stride_of(g, d));
#pragma unroll
            for (int q = 0; q < 3; ++q) nbv[d][s][q] = Gd[3 * (size_t)nb + q];
        }
    }
    double acc[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double fv[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            for (int q = 0; q < 3; ++q) fv[s][q] = bnd[d][s] ? own[d][q] : geo_lerp_side(g, d, s, ijk[d], own[d][q], nbv[d][s][q]);
        for (int q = 0; q < 3; ++q) acc[q] += (fv[1][q] - fv[0][q]) * geo_rh(g, d, ijk[d]);
    }
    for (int q = 0; q < 3; ++q) divG[3 * (size_t)c + q] = acc[q];
}

// continuousPhaseTurbulence->correct() (pimpleFoamYade.C:101-104) for LESModel Smagorinsky (DPMTurbulenceModels.C:73-74) [OF-6
// Smagorinsky.C: k(gradU), correctNut()]: D = symm(grad U); a = Ce/delta; b = (2/3) tr(D); c = 2 Ck delta (dev(D) && D);
// k = sqr((-b + sqrt(sqr(b) + 4 a c)) / (2 a)); nut = Ck delta sqrt(k).  grad U is the Gauss-linear gradient k_pre_coupling just wrote.
__global__ __launch_bounds__(256) void k_smagorinsky_nut(FvGeo g, const double* __restrict__ vGrad, double ck, double ce, double delta, double* __restrict__ nut) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Nc) return;
    const int c = t + g.c0;
    delta = geo_delta(g, delta, c);
    double T[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) T[q] = vGrad[9 * (size_t)c + q];
    const double Dxx = T[0], Dyy = T[4], Dzz = T[8];
    const double Dxy = 0.5 * (T[1] + T[3]), Dxz = 0.5 * (T[2] + T[6]), Dyz = 0.5 * (T[5] + T[7]);
    const double trD = Dxx + Dyy + Dzz;
    const double a = ce / delta;
    const double b = (2.0 / 3.0) * trD;
    const double third = (1.0 / 3.0) * trD;
    // dev(D) && D over the nine components of the symmetric tensors
    const double dd = (Dxx - third) * Dxx + (Dyy - third) * Dyy + (Dzz - third) * Dzz + 2.0 * (Dxy * Dxy) + 2.0 * (Dxz * Dxz) + 2.0 * (Dyz * Dyz);
    const double cc = 2.0 * ck * delta * dd;
    const double r = (-b + sqrt(b * b + 4.0 * a * cc)) / (2.0 * a);
    const double kk = r * r;
    nut[c] = ck * delta * sqrt(kk);
}

// epsilonWallFunction [OF-6 epsilonWallFunctionFvPatchScalarField::calculate]: the value imposed on a cell with wall faces is the average over
// them of Cmu^3/4 k^3/2 / (kappa y_w) (cornerWeights = 1 / number of wall faces); on the uniform block every y_w is dx / 2
__device__ __forceinline__ double wall_epsilon(const FvGeo& g, const TurbEqn& e, double kc, int c, int i, int j, int k) {
#if FY_FVK_GRADED
    double sum = 0.0;
    int W = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (e.wall[2 * d + s] && onb(g, d, s, i, j, k)) { sum += e.cmu75 * pow(kc, 1.5) / (e.kappa * geo_ywall(g, d, c)); ++W; }
    return sum / (double)W;
#else
    return e.cmu75 * pow(kc, 1.5) / (e.kappa * geo_ywall(g, 0, c));
#endif
}

// Transport equations of LESModel kEqn (DPMTurbulenceModels.C:76-77) [OF-6 LES/kEqn/kEqn.C correct()] and RASModel kEpsilon
// (DPMTurbulenceModels.C:70-71) [OF-6 RAS/kEpsilon/kEpsilon.C correct()], see TurbEqn in fv_kernels.hpp:
//   fvm::ddt(alpha, rho, X) + fvm::div(alphaRhoPhi, X) - fvm::laplacian(alpha rho DXEff, X) == Su - fvm::SuSp(c1, X) - fvm::Sp(c2, X)
// then relax(), solve, bound(X, XMin), correctNut().  alpha.oldTime() == alpha (quirk F-Q1, as in UcEqn).  Face diffusivity: linear
// interpolate of the cell field alpha (nu + nut / sigma), boundary alpha_b (nu + nut_b / sigma), as in the momentum equation.
// [OF-6 fvm::SuSp: diag += V max(susp, 0), source -= V min(susp, 0) psi; fvm::Sp: diag += V sp; fvMatrix == volField: source += V field]
__global__ __launch_bounds__(256) void k_assemble_turb(FvGeo g, TurbEqn e, const double* __restrict__ kf, const double* __restrict__ ef,
                                                       const double* __restrict__ alpha, CFace3 alphaf, CFace3 phi, const double* __restrict__ vGrad,
                                                       const double* __restrict__ U, Mom7 M, double* __restrict__ b3, double* __restrict__ x3) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    const double nu = g.nu, dt = g.dt, V = geo_V(g, i, j, k);
    const double aP = alpha[c], nutc = g.nut[c];
    const double* X = e.mode == 1 ? ef : kf;
    const double xc = X[c];
    double dg = aP * V / dt;                                    // fvm::ddt(alpha, X)
    double src = aP * V * xc / dt;
    double sumPhi = 0.0, an[6];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int f = cface(g, d, s, i, j, k);
            const double af = alphaf.a[d][f];
            const double pv = (s ? 1.0 : -1.0) * phi.a[d][f];
            const double phio = af * pv;                          // alphaRhoPhi, outward
            sumPhi += pv;
            if (onb(g, d, s, i, j, k)) {
                an[2 * d + s] = 0.0;
                const int patch = 2 * d + s;
                const double nb = nut_boundary(g, patch, c);
                const double gam = (af * (nu + nb / e.sigma)) * geo_sfd(g, d, s, i, j, k);
                if (e.bc[patch] == 1) {
                    const double gb = kBfac * gam;
                    dg += gb;
                    src += (-phio + gb) * e.val[patch];
                } else {
                    dg += phio;
                }
            } else {
                const int nbc = c + (s ? stride_of(g, d) : -stride_of(g, d));
                const int qc = d == 0 ? i : d == 1 ? j : k;
                const double gam = geo_lerp_side(g, d, s, qc, aP * (nu + nutc / e.sigma), alpha[nbc] * (nu + g.nut[nbc] / e.sigma)) * geo_sfd(g, d, s, i, j, k);
                const double wP = geo_wown(g, d, s, qc);
#if FY_FVK_GRADED
                const double cP = e.upwind ? fmax(phio, 0.0) : wP * phio, cN = e.upwind ? fmin(phio, 0.0) : (1.0 - wP) * phio;
#else
                const double cP = e.upwind ? fmax(phio, 0.0) : wP * phio, cN = e.upwind ? fmin(phio, 0.0) : wP * phio;
#endif
                dg += cP + gam;
                an[2 * d + s] = cN - gam;
            }
        }
    // G = nut (gradU && dev(twoSymm(gradU)))
    double T[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) T[q] = vGrad[9 * (size_t)c + q];
    const double tr2 = 2.0 * (T[0] + T[4] + T[8]);              // tr(twoSymm(T))
    double GG = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) GG += T[3 * a + b] * ((T[3 * a + b] + T[3 * b + a]) - (a == b ? (1.0 / 3.0) * tr2 : 0.0));
    double G = nutc * GG;
    // epsilonWallFunction: wall value of G, and (in the epsilon equation) the imposed cell value
    int Wc = 0;
    double Gw = 0.0;
    if (e.mode != 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (e.wall[2 * d + s] && onb(g, d, s, i, j, k)) {
                    const int patch = 2 * d + s;
                    double ub[3];
                    Ub(g, U, c, patch, ub);
                    const double ywall = geo_ywall(g, d, c);
                    const double d0 = (ub[0] - U[3 * (size_t)c]) / ywall, d1 = (ub[1] - U[3 * (size_t)c + 1]) / ywall, d2 = (ub[2] - U[3 * (size_t)c + 2]) / ywall;
                    const double magGradUw = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
                    Gw += (nut_boundary(g, patch, c) + nu) * magGradUw * e.cmu25 * sqrt(kf[c]) / (e.kappa * ywall);
                    ++Wc;
                }
        if (Wc) G = Gw / (double)Wc;
    }
    const double divU = sumPhi * geo_rV(g, i, j, k);
    double Su, c1, c2;
    if (e.mode == 0) { Su = aP * G; c1 = (2.0 / 3.0) * aP * divU; c2 = e.ce * aP * sqrt(xc) / geo_delta(g, e.delta, c); }
    else if (e.mode == 1) { const double kc = kf[c]; Su = e.c1 * aP * G * xc / kc; c1 = ((2.0 / 3.0) * e.c1 - e.c3) * aP * divU; c2 = e.c2 * aP * xc / kc; }
    else { Su = aP * G; c1 = (2.0 / 3.0) * aP * divU; c2 = aP * ef[c] / xc; }
    dg += V * (fmax(c1, 0.0) + c2);                             // fvm::SuSp (implicit where it stabilises) + fvm::Sp
    src += V * Su - V * fmin(c1, 0.0) * xc;
    if (e.relax > 0) {                                          // fvMatrix::relax as in UcEqn
        double so = 0.0;
        for (int q = 0; q < 6; ++q) so += fabs(an[q]);
        const double dn = fmax(fabs(dg), so) / e.relax;
        src += (dn - dg) * xc;
        dg = dn;
    }
    double x0 = xc;
    if (e.mode == 1) {
        // epsEqn.boundaryManipulate -> fvMatrix::setValues(faceCells, value) [OF-6 fvMatrix.C setValuesFromList]
        if (Wc) {
            const double v = wall_epsilon(g, e, kf[c], c, i, j, k);
            for (int q = 0; q < 6; ++q) an[q] = 0.0;
            src = dg * v;
            x0 = v;
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (!onb(g, d, s, i, j, k)) {
                        const int ni = i + (d == 0 ? (s ? 1 : -1) : 0), nj = j + (d == 1 ? (s ? 1 : -1) : 0), nk = k + (d == 2 ? (s ? 1 : -1) : 0);
                        bool nwall = false;
#pragma unroll
                        for (int d2 = 0; d2 < 3; ++d2)
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) nwall = nwall || (e.wall[2 * d2 + s2] && onb(g, d2, s2, ni, nj, nk));
                        if (nwall) {
                            const int nbc = c + (s ? stride_of(g, d) : -stride_of(g, d));
                            src -= an[2 * d + s] * wall_epsilon(g, e, kf[nbc], nbc, ni, nj, nk);
                            an[2 * d + s] = 0.0;
                        }
                    }
        }
    }
    M.diag[c] = dg;
    for (int q = 0; q < 6; ++q) M.an[q][c] = an[q];
    b3[3 * (size_t)c] = src; b3[3 * (size_t)c + 1] = 0.0; b3[3 * (size_t)c + 2] = 0.0;
    x3[3 * (size_t)c] = x0; x3[3 * (size_t)c + 1] = 0.0; x3[3 * (size_t)c + 2] = 0.0;
}

// bound(X, XMin) [OF-6 finiteVolume/cfdTools/general/bound/bound.C]: X = max(max(X, fvc::average(max(X, XMin)) pos0(-X)), XMin) -- the
// identity where X >= XMin, so it is applied without the global min test OpenFOAM guards it with -- then correctNut (see launch_turb_finish)
__global__ __launch_bounds__(256) void k_turb_finish(FvGeo g, TurbEqn e, const double* __restrict__ x3, double* __restrict__ Xf, int nut_mode, double cmu,
                                                     const double* __restrict__ ef, double* __restrict__ nut) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    const double xc = x3[3 * (size_t)c];
    double xb = xc;
    if (!(xc > 0.0)) {                                          // pos0(-X) = 1: the face-area average of the bounded neighbourhood
        const double mP = fmax(xc, e.xmin);
        double av = 0.0;
#if FY_FVK_GRADED
        double asum = 0.0;                                      // fvc::average: sum |Sf| x_f / sum |Sf|
#endif
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                double xf;
                if (onb(g, d, s, i, j, k)) {
                    const int patch = 2 * d + s;
                    xf = fmax(e.bc[patch] == 1 ? e.val[patch] : xc, e.xmin);
                } else {
                    const int nbc = c + (s ? stride_of(g, d) : -stride_of(g, d));
                    xf = geo_lerp_side(g, d, s, d == 0 ? i : d == 1 ? j : k, mP, fmax(x3[3 * (size_t)nbc], e.xmin));
                }
#if FY_FVK_GRADED
                const double Afc = geo_Af(g, d, i, j, k);
                av += Afc * xf; asum += Afc;
#else
                av += xf;
#endif
            }
#if FY_FVK_GRADED
        xb = fmax(xc, av / asum);
#else
        xb = fmax(xc, av / 6.0);
#endif
    }
    xb = fmax(xb, e.xmin);
    Xf[c] = xb;
    if (nut_mode == 1) nut[c] = e.ck * sqrt(xb) * geo_delta(g, e.delta, c);
    else if (nut_mode == 2) nut[c] = cmu * (xb * xb) / ef[c];
}

// UEqn (icoFoamYade.C:79-85) / UcEqn + relax (UcEqn.H:3-12): diag, 6 neighbour coefficients, source (no pressure term), rAU = 1/A
// ---- NVD / TVD limited convection schemes for div(phi,U) [OF-6 LimitedScheme<vector, Limiter<NVDTVD>, limitFuncs::magSqr>, NVDTVD.H]: one
// limiter per face from the scalar lPhi = magSqr(U): r = 2 (d . grad(lPhi)_C) / (lPhi_N - lPhi_P) - 1, C the upwind cell of the face flux, grad
// the Gauss-linear gradient (boundary value magSqr(U_b)); the weight of the OWNER's value is limiter * w_linear + (1 - limiter) * pos0(faceFlux)
// [limitedSurfaceInterpolationScheme::weights], used implicitly in the matrix.
__device__ __forceinline__ double magsqr3(const double* __restrict__ F, int c) {
    const double a = F[3 * (size_t)c], b = F[3 * (size_t)c + 1], e = F[3 * (size_t)c + 2];
    return (a * a + b * b) + e * e;
}
__global__ __launch_bounds__(256) void k_grad_magsqr(FvGeo g, const double* __restrict__ U, double* __restrict__ gradL) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    const double lc = magsqr3(U, c);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double fv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (onb(g, d, s, i, j, k)) { double b[3]; Ub(g, U, c, 2 * d + s, b); fv[s] = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]; }
            else fv[s] = geo_lerp_side(g, d, s, d == 0 ? i : d == 1 ? j : k, lc, magsqr3(U, c + (s ? stride_of(g, d) : -stride_of(g, d))));
        }
        gradL[3 * (size_t)c + d] = (fv[1] - fv[0]) * geo_rh(g, d, d == 0 ? i : d == 1 ? j : k);
    }
}
__device__ __forceinline__ double limiter_fn(int scheme, double twoByk, double r) {
    switch (scheme) {
        case 3: return fmax(fmin(twoByk * r, 1.0), 0.0);                                   // limitedLinear k: twoByk = 2 / max(k, small)
        case 4: return (r + fabs(r)) / (1.0 + fabs(r));                                    // vanLeer
        case 5: return fmax(fmin(fmin(2.0 * r, 0.5 * r + 0.5), 2.0), 0.0);                 // MUSCL
        case 6: return fmax(fmin(fmin(r, 1.0), 2.0), 0.0);                                 // Minmod
        case 7: return fmax(fmax(fmin(2.0 * r, 1.0), fmin(r, 2.0)), 0.0);                  // SuperBee
        // QUICK (8) [OF-6 QUICK.H]: QLimiter = (phif - phiU) / (phiCD - phiU) with phif = (phiCD + phiU + (1 - w) d.grad(phi)_U) / 2, which is
        // (3 + r) / 4 whatever the linear weight w; limited "between upwind and downwind" only -- max(min(., 2), 0) -- so NOT a TVD limiter: it stays
        // positive down to r = -3 and has no 2 r bound (rounds 2 - 3 had min(2 r, .) here: another scheme under QUICK's name)
        default: return fmax(fmin((3.0 + r) / 4.0, 2.0), 0.0);
    }
}
// weight of the owner's (low cell's) value on the interior face between own and nei (axis d; qn = the neighbour's index along d); flux: owner -> neighbour
__device__ __forceinline__ double limited_weight(const FvGeo& g, const double* __restrict__ U, const double* __restrict__ gradL, int d, int qn, int own,
                                                 int nei, double flux) {
    const double gradf = magsqr3(U, nei) - magsqr3(U, own);
    const double gradcf = (1.0 / geo_rdelta(g, d, qn)) * gradL[3 * (size_t)(flux > 0 ? own : nei) + d];
    double r;
    if (fabs(gradcf) >= 1000.0 * fabs(gradf)) r = 2.0 * 1000.0 * (gradcf >= 0 ? 1.0 : -1.0) * (gradf >= 0 ? 1.0 : -1.0) - 1.0;
    else r = 2.0 * (gradcf / gradf) - 1.0;
    const double lim = limiter_fn(g.upwind, g.lim_twoByk, r);
#if FY_FVK_GRADED
    const double wl = geo_wlow(g, d, qn);
#else
    const double wl = 0.5;
#endif
    return lim * wl + (1.0 - lim) * (flux >= 0 ? 1.0 : 0.0);
}

template <bool LIMITED>
__global__ __launch_bounds__(256) void k_assemble_momentum(FvGeo g, const double* __restrict__ U, const double* __restrict__ Uold,
                                                           const double* __restrict__ alpha, const double* __restrict__ alphaOld, CFace3 alphaf,
                                                           CFace3 phi, const double* __restrict__ uSource, const double* __restrict__ uSourceDrag,
                                                           const double* __restrict__ divG, const double* __restrict__ vGrad, Mom7 M,
                                                           double* __restrict__ src, double* __restrict__ rAU) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    const double nu = g.nu, dt = g.dt, V = geo_V(g, i, j, k);
    const bool pim = g.pimple != 0;
    const double aP = pim ? alpha[c] : 1.0, aP0 = pim ? alphaOld[c] : 1.0;
    double dg = aP * V / dt;
    double s3[3];
    for (int q = 0; q < 3; ++q) s3[q] = aP0 * V * Uold[3 * (size_t)c + q] / dt;
    double divAPhi = 0.0, an[6];
    double bd[3] = {0.0, 0.0, 0.0};               // per-component boundary diagonal (slip patches; M.bd)
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int f = cface(g, d, s, i, j, k);
            // alphacf streamed from its face array, or (no array kept: alphaf.a[0] == nullptr) re-formed from alpha as interp_alpha_face forms it
            double af = 1.0;
            if (pim) {
                if (alphaf.a[0]) af = alphaf.a[d][f];
                else if (!onb(g, d, s, i, j, k)) af = lerp_face(g, d, s, (d == 0 ? i : d == 1 ? j : k) + s, aP, alpha[c + (s ? stride_of(g, d) : -stride_of(g, d))]);
            }
            const double phio = (s ? 1.0 : -1.0) * af * phi.a[d][f];
            divAPhi += phio;
            // fvm::laplacian(alpha nuEff, U): the face diffusivity is the linear interpolate of the cell field alpha (nu + nut) [OF-6
            // linearViscousStress::divDevRhoReff, gaussLaplacianScheme]; its boundary value is alpha_b (nu + nut_b).  Laminar: nu alphaf
            const double geo = geo_sfd(g, d, s, i, j, k);          // |Sf| / |d| (uniform block: dx; a boundary's factor 2 is kBfac)
            double gam = nu * af * geo;
            if (g.nut) {
                if (onb(g, d, s, i, j, k)) {
                    const double nb = nut_boundary(g, 2 * d + s, c);
                    gam = (af * (nu + nb)) * geo;
                } else {
                    const int nbc = c + (s ? stride_of(g, d) : -stride_of(g, d));
                    gam = geo_lerp_side(g, d, s, d == 0 ? i : d == 1 ? j : k, aP * (nu + g.nut[c]), alpha[nbc] * (nu + g.nut[nbc])) * geo;
                }
            }
            if (onb(g, d, s, i, j, k)) {
                an[2 * d + s] = 0.0;
                const int patch = 2 * d + s;
                if (g.u_bc[patch] == 0) {
                    const double gb = kBfac * gam;
                    dg += gb;
                    for (int q = 0; q < 3; ++q) s3[q] += (-phio + gb) * g.u_val[patch][q];
                } else {
                    // symmetryPlane / slip [OF-6 transformFvPatchField::gradientInternalCoeffs with basicSymmetry's snGradTransformDiag]: the
                    // NORMAL component sees a fixed value 0 (implicit coefficient only), the tangential ones a zero gradient
                    if (g.u_bc[patch] == 2) bd[d] += kBfac * gam;
                    dg += phio;
                }
            } else {
                // fvm::div(phi, U): Gauss linear puts half the face flux on either side; Gauss upwind takes the whole of an
                // outgoing flux on the diagonal and the whole of an incoming one on the neighbour [OF-6 gaussConvectionScheme]
                const double wP = geo_wown(g, d, s, d == 0 ? i : d == 1 ? j : k);      // Gauss linear: w_P U_P + (1 - w_P) U_N at the face
#if FY_FVK_GRADED
                double cP = g.upwind ? fmax(phio, 0.0) : wP * phio, cN = g.upwind ? fmin(phio, 0.0) : (1.0 - wP) * phio;
#else
                double cP = g.upwind ? fmax(phio, 0.0) : wP * phio, cN = g.upwind ? fmin(phio, 0.0) : wP * phio;
#endif
                if (LIMITED) {                                 // (vGrad carries grad(magSqr(U)), three per cell)
                    const int nb = c + (s ? stride_of(g, d) : -stride_of(g, d));
                    const int qc = d == 0 ? i : d == 1 ? j : k;
                    const double w = limited_weight(g, U, vGrad, d, qc + s, s ? c : nb, s ? nb : c, af * phi.a[d][f]);
                    cP = s ? phio * w : phio * (1.0 - w);
                    cN = s ? phio * (1.0 - w) : phio * w;
                }
                dg += cP + gam;
                an[2 * d + s] = cN - gam;
                if (g.upwind == 2) {
                    // linearUpwind: implicit upwind + explicit (C_f - C_upwind) . grad(U)_upwind with the current Gauss-linear gradient
                    const int nb = c + (s ? stride_of(g, d) : -stride_of(g, d));
                    const int uw = phio > 0.0 ? c : nb;
                    // (C_f - C_upwind) along d: half the UPWIND cell's extent, towards the face
                    const double half = (phio > 0.0 ? (s ? 0.5 : -0.5) : (s ? -0.5 : 0.5)) * geo_hc(g, d, uw);
                    for (int q = 0; q < 3; ++q) s3[q] -= phio * (half * vGrad[9 * (size_t)uw + 3 * d + q]);
                }
            }
        }
    if (pim) {
        const double S = (alpha[c] - alphaOld[c]) / dt + divAPhi / V;
        dg -= V * S;
        dg -= V * uSourceDrag[c];
        for (int q = 0; q < 3; ++q) s3[q] += V * divG[3 * (size_t)c + q];
        double so = 0.0;
        for (int q = 0; q < 6; ++q) so += fabs(an[q]);
        if (g.u_relax > 0) {
            // fvMatrix::relax(alpha) [OF-6 fvMatrix.C]: D = max(|D|, sum|offdiag|) / alpha (the boundary coefficients, part of dg here all
            // along, take part in the dominance test), source += (D_new - D_old) psi; no factor for the equation: relax() does nothing
            // (a slip patch's coefficient differs by component: relax() adds cmptMax(cmptMag(internalCoeffs)) before the test and takes
            // cmptMin (= 0) off afterwards -- all of it joins the test, the scalar diagonal keeps what the test gave)
            const double dn = fmax(fabs(dg + ((bd[0] + bd[1]) + bd[2])), so) / g.u_relax;
            for (int q = 0; q < 3; ++q) s3[q] += (dn - dg) * U[3 * (size_t)c + q];
            dg = dn;
        }
    } else {
        for (int q = 0; q < 3; ++q) s3[q] += V * uSource[3 * (size_t)c + q];
    }
    M.diag[c] = dg;
    for (int q = 0; q < 6; ++q) M.an[q][c] = an[q];
    for (int q = 0; q < 3; ++q) src[3 * (size_t)c + q] = s3[q];
    // 1 / UEqn.A(): fvMatrix::A() adds the COMPONENT AVERAGE of the boundary diagonal
    double bav = 0.0;
    if (M.bd) { for (int q = 0; q < 3; ++q) M.bd[3 * (size_t)c + q] = bd[q]; bav = ((bd[0] + bd[1]) + bd[2]) / 3.0; }
    rAU[c] = 1.0 / ((dg + bav) / V);
}

// RHS of the momentum predictor: ico  src - V grad(p)            (icoFoamYade.C:91-94)
//                                pimple src + V reconstruct(phicForces/rAUcf - snGrad(p) magSf)   (UcEqn.H:22-33)
__global__ __launch_bounds__(256) void k_bmom(FvGeo g, const double* __restrict__ src, const double* __restrict__ p, CFace3 psn,
                                              CFace3 phiForces, CFace3 rAUf, double* __restrict__ bmom) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    double out[3];      // stored together at the end: component stores issued far apart reach HBM as three partial-line writes
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (!g.pimple) {
            double fv[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (onb(g, d, s, i, j, k)) fv[s] = pbv(g, p, psn, c, d, s, cface(g, d, s, i, j, k));
                else fv[s] = geo_lerp_side(g, d, s, d == 0 ? i : d == 1 ? j : k, p[c], p[c + (s ? stride_of(g, d) : -stride_of(g, d))]);
            }
            out[d] = src[3 * (size_t)c + d] - geo_V(g, i, j, k) * ((fv[1] - fv[0]) * geo_rh(g, d, d == 0 ? i : d == 1 ? j : k));
        } else {
            double sm = 0;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int f = cface(g, d, s, i, j, k);
                double sng;
                const double rds = geo_rdist(g, d, s, i, j, k);
                if (onb(g, d, s, i, j, k)) { const double pbd = pbv(g, p, psn, c, d, s, f); sng = s ? (pbd - p[c]) * rds : (p[c] - pbd) * rds; }
                else sng = s ? (p[c + stride_of(g, d)] - p[c]) * rds : (p[c] - p[c - stride_of(g, d)]) * rds;
                sm += phiForces.a[d][f] / rAUf.a[d][f] - sng * geo_Af(g, d, i, j, k);
            }
            out[d] = src[3 * (size_t)c + d] + geo_V(g, i, j, k) * (sm / (2.0 * geo_Af(g, d, i, j, k)));
        }
    }
    for (int d = 0; d < 3; ++d) bmom[3 * (size_t)c + d] = out[d];
}

// pimple: rAUcf / phicForces (UcEqn.H:15-20) and the momentum predictor's right-hand side (UcEqn.H:22-33) in one gather-first sweep:
// k_rAUf_phi_forces_cells + k_bmom without the two face fields being read back; a cell forms its six faces' rAUcf and phicForces itself (same
// expressions as rAUf_phi_forces_face), the face's owner stores phicForces (and rAUcf where somebody still streams it: rf_out.a[0] != nullptr)
__global__ __launch_bounds__(256) FY_WPE_ATTR(FY_WPE_BMOM) void k_bmom_faces(FvGeo g, const double* __restrict__ rAU, const double* __restrict__ uSource, const double* __restrict__ src,
                                                    const double* __restrict__ p, CFace3 psn, Face3 rf_out, Face3 pf_out, double* __restrict__ bmom) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    const int ijk[3] = {i, j, k};
    bool bnd[3][2]; int nb[3][2], fx[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bnd[d][s] = onb(g, d, s, i, j, k);
            nb[d][s] = bnd[d][s] ? c : c + (s ? stride_of(g, d) : -stride_of(g, d));
            fx[d][s] = cface(g, d, s, i, j, k);
        }
    const double rc = rAU[c], pc = p[c];
    const double us[3] = {uSource[3 * (size_t)c], uSource[3 * (size_t)c + 1], uSource[3 * (size_t)c + 2]};
    const double sr[3] = {src[3 * (size_t)c], src[3 * (size_t)c + 1], src[3 * (size_t)c + 2]};
    double rn[3][2], un[3][2], pn[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) { rn[d][s] = rAU[nb[d][s]]; un[d][s] = uSource[3 * (size_t)nb[d][s] + d]; pn[d][s] = p[nb[d][s]]; }
    double out[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double sm = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int q = ijk[d] + s;
            const int fi = i + (d == 0 ? s : 0), fj = j + (d == 1 ? s : 0), fk = k + (d == 2 ? s : 0);
            const bool b = bnd[d][s];
            const double Afc = geo_Af(g, d, fi, fj, fk);
            const double r = b ? rc : lerp_face(g, d, s, q, rc, rn[d][s]);
            const double fl = b ? 0.0 : lerp_face(g, d, s, q, rc * us[d], rn[d][s] * un[d][s]) * Afc;
            const double pfv = fl + r * (g.g[d] * Afc);
            if (owns_face(g, d, s, i, j, k)) { pf_out.a[d][fx[d][s]] = pfv; if (rf_out.a[0]) rf_out.a[d][fx[d][s]] = r; }
            double sng;
            const double rds = geo_rdist(g, d, s, i, j, k);
            if (b) { const double pbd = pbv(g, p, psn, c, d, s, fx[d][s]); sng = s ? (pbd - pc) * rds : (pc - pbd) * rds; }
            else sng = s ? (pn[d][s] - pc) * rds : (pc - pn[d][s]) * rds;
            sm += pfv / r - sng * geo_Af(g, d, i, j, k);
        }
        out[d] = sr[d] + geo_V(g, i, j, k) * (sm / (2.0 * geo_Af(g, d, i, j, k)));
    }
    for (int d = 0; d < 3; ++d) bmom[3 * (size_t)c + d] = out[d];
}

// One fused Jacobi pass over the 3 velocity components: with x the current iterate, accumulate the L1 residual |b - A x|
// (slots 0..2), the lduMatrix normalisation sum |A x - A xbar| + |b - A xbar| (slots 3..5) and write the next iterate xn.
// WITH_H: the pass also leaves HbyA = rAU H(x) / V of ITS iterate x (k_HbyA's expression in k_HbyA's order: the same bits) -- the pass that finds x
// converged has then done the first corrector's H-operator sweep on the side, for 56 B/cell instead of a sweep of 128
template <bool WITH_H>
__global__ __launch_bounds__(256) void k_mom_pass(FvGeo g, Mom7 M, const double* __restrict__ b, const double* __restrict__ x,
                                                  double* __restrict__ xn, const double* __restrict__ xsum, double n_glob, double* __restrict__ partials,
                                                  const double* __restrict__ hsrc, const double* __restrict__ rAU, double* __restrict__ HbyA) {
    double v[6] = {0, 0, 0, 0, 0, 0};
    // xbar = average(x) (lduMatrix::solver::normFactor): the component sums stay on the device (k_sum3 + fold [+ all-reduce]); dividing
    // them here saves the host round trip the average used to make
    const double xb[3] = {xsum[0] / n_glob, xsum[1] / n_glob, xsum[2] / n_glob};
    FY_RED_LOOP_G(g, t) {
        int i, j, k; ijk_of(g, t, i, j, k);
        const int c = t + g.c0;
        const double dg = M.diag[c];
        double dq[3] = {dg, dg, dg};                      // (fvMatrix::solveSegregated: addBoundaryDiag per component)
        if (M.bd) for (int q = 0; q < 3; ++q) dq[q] += M.bd[3 * (size_t)c + q];
        double off[3] = {0, 0, 0}, rowsum = dg;
        const double xc[3] = {x[3 * (size_t)c], x[3 * (size_t)c + 1], x[3 * (size_t)c + 2]};
        double hacc[3] = {0, 0, 0};
        if (WITH_H) { hacc[0] = hsrc[3 * (size_t)c]; hacc[1] = hsrc[3 * (size_t)c + 1]; hacc[2] = hsrc[3 * (size_t)c + 2]; }
        double xx[2][3];
        x_neighbours3(x, c, i, g.nx, xc, xx[0], xx[1]);
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (!onb(g, d, s, i, j, k)) {
                    const int nb = c + (s ? stride_of(g, d) : -stride_of(g, d));
                    const double a = M.an[2 * d + s][c];
                    rowsum += a;
                    double xv[3];
                    if (d == 0) for (int q = 0; q < 3; ++q) xv[q] = xx[s][q];
                    else for (int q = 0; q < 3; ++q) xv[q] = x[3 * (size_t)nb + q];
                    for (int q = 0; q < 3; ++q) off[q] += a * xv[q];
                    if (WITH_H) for (int q = 0; q < 3; ++q) hacc[q] -= a * xv[q];
                }
        if (WITH_H) {
            if (M.bd) {
                const double b0 = M.bd[3 * (size_t)c], b1 = M.bd[3 * (size_t)c + 1], b2 = M.bd[3 * (size_t)c + 2];
                const double bav = ((b0 + b1) + b2) / 3.0;
                hacc[0] += (bav - b0) * xc[0]; hacc[1] += (bav - b1) * xc[1]; hacc[2] += (bav - b2) * xc[2];
            }
            const double r = rAU[c];
            for (int q = 0; q < 3; ++q) HbyA[3 * (size_t)c + q] = r * (hacc[q] * geo_rV(g, i, j, k));
        }
        for (int q = 0; q < 3; ++q) {
            const double bq = b[3 * (size_t)c + q];
            const double Ax = dq[q] * xc[q] + off[q];
            const double Aref = (rowsum + (dq[q] - dg)) * xb[q];
            v[q] += fabs(bq - Ax);
            v[3 + q] += fabs(Ax - Aref) + fabs(bq - Aref);
            xn[3 * (size_t)c + q] = (bq - off[q]) / dq[q];
        }
    }
    const int mx[6] = {0, 0, 0, 0, 0, 0};
    block_reduce_store<6>(v, mx, partials, fy_lb, fy_stride);
}

// component sums over a contiguous range of n vectors starting at x
__global__ __launch_bounds__(256) void k_sum3(const double* __restrict__ x, int n, double* __restrict__ partials) {
    double v[3] = {0, 0, 0};
    FY_RED_LOOP(c, n)
        for (int q = 0; q < 3; ++q) v[q] += x[3 * (size_t)c + q];
    const int mx[3] = {0, 0, 0};
    block_reduce_store<3>(v, mx, partials);
}

// HbyA = rAU * UEqn.H() (icoFoamYade.C:100, pEqn.H:2)
__global__ __launch_bounds__(256) void k_HbyA(FvGeo g, Mom7 M, const double* __restrict__ src, const double* __restrict__ U,
                                              const double* __restrict__ rAU, double* __restrict__ HbyA) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    double acc[3] = {src[3 * (size_t)c], src[3 * (size_t)c + 1], src[3 * (size_t)c + 2]};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            if (!onb(g, d, s, i, j, k)) {
                const int nb = c + (s ? stride_of(g, d) : -stride_of(g, d));
                const double a = M.an[2 * d + s][c];
                for (int q = 0; q < 3; ++q) acc[q] -= a * U[3 * (size_t)nb + q];
            }
    if (M.bd) {
        // fvMatrix::H(): per component (component-averaged boundary diagonal - that component's) * psi
        const double b0 = M.bd[3 * (size_t)c], b1 = M.bd[3 * (size_t)c + 1], b2 = M.bd[3 * (size_t)c + 2];
        const double bav = ((b0 + b1) + b2) / 3.0;
        acc[0] += (bav - b0) * U[3 * (size_t)c]; acc[1] += (bav - b1) * U[3 * (size_t)c + 1]; acc[2] += (bav - b2) * U[3 * (size_t)c + 2];
    }
    const double r = rAU[c];
    for (int q = 0; q < 3; ++q) HbyA[3 * (size_t)c + q] = r * (acc[q] * geo_rV(g, i, j, k));
}

// pEqn in SPD form: sum_f g_f (p_P - p_nb) [+ g_b (p_P - p_b)] = -(ddt(alpha) V + sum_out alphaf phiHbyA)   (icoFoamYade.C:118-123, pEqn.H:26-33)
// MATRIX = false: the right-hand side alone -- the second PISO corrector (and the non-orthogonal correctors) solve with the SAME matrix
// (rAU, alphacf have not changed), so the four coefficient arrays are neither recomputed nor rewritten and rAUf is read only where the
// right-hand side itself needs it (fixedFluxPressure / fixedValue boundary faces, the reference cell).  Same expressions, same bits.
template <bool MATRIX>
__global__ __launch_bounds__(256) void k_assemble_pressure(FvGeo g, CFace3 phiHbyA, CFace3 rAUf, CFace3 alphaf, CFace3 psn,
                                                           const double* __restrict__ alpha, const double* __restrict__ alphaOld, PMat A,
                                                           double* __restrict__ rhs) {
    const int t = fv_block(g, blockIdx.x, gridDim.x) * 256 + (int)threadIdx.x;
    if (t >= g.Nc) return;
    int i, j, k; ijk_of(g, t, i, j, k);
    const int c = t + g.c0;
    // fvMatrix::setReference: p_ref_cell is a GLOBAL cell number (i + nx*(j + ny*kglob))
    const bool refc = g.need_ref && (i + g.nx * (j + g.ny * (k + g.kglob0))) == g.p_ref_cell;
    double dg = 0.0, r = 0.0, up[3] = {0, 0, 0};
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int f = cface(g, d, s, i, j, k);
            const double af = g.pimple ? alphaf.a[d][f] : 1.0;
            double ph = (s ? 1.0 : -1.0) * af * phiHbyA.a[d][f];
            const bool b = onb(g, d, s, i, j, k);
            if (b && g.p_bc[2 * d + s] == 2) ph = (s ? 1.0 : -1.0) * af * (phiHbyA.a[d][f] - rAUf.a[d][f] * geo_Af(g, d, i, j, k) * psn.a[d][f]);
            r -= ph;
            if (b) {
                const int patch = 2 * d + s;
                if (g.p_bc[patch] == 1) { const double gb = kBfac * af * rAUf.a[d][f] * geo_sfd(g, d, s, i, j, k); dg += gb; r += gb * g.p_val[patch]; }
            } else if (MATRIX || refc) {
                const double gg = af * rAUf.a[d][f] * geo_sfd(g, d, s, i, j, k);
                dg += gg;
                if (s) up[d] = gg;
            }
        }
    if (g.pimple) r -= geo_V(g, i, j, k) * (alpha[c] - alphaOld[c]) / g.dt;
    if (refc) { r += dg * g.p_ref_value; dg += dg; }
    if (MATRIX) { A.diag[c] = dg; A.ux[c] = up[0]; A.uy[c] = up[1]; A.uz[c] = up[2]; }
    rhs[c] = r;
}

// slab interface below the first owned plane: its z-face coefficient goes to the ghost cell under it (read by p_row as uz[c - sz])
__global__ __launch_bounds__(256) void k_p_ghost_uz(FvGeo g, CFace3 rAUf, CFace3 alphaf, PMat A) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.nx * g.ny) return;
    const double af = g.pimple ? alphaf.a[2][t] : 1.0;           // z-face plane kf = 0
    A.uz[g.c0 - g.nx * g.ny + t] = af * rAUf.a[2][t] * g.dx;
}

// continuityErrs.H:32-46: contErr = [ddt(alpha) +] div([alphaf] phi); slots 0 = sum |contErr| V, 1 = sum contErr V
__global__ __launch_bounds__(256) void k_cont_err(FvGeo g, CFace3 phi, CFace3 alphaf, const double* __restrict__ alpha,
                                                  const double* __restrict__ alphaOld, double* __restrict__ partials) {
    double v[2] = {0, 0};
    FY_RED_LOOP_G(g, t) {
        int i, j, k; ijk_of(g, t, i, j, k);
        const int c = t + g.c0;
        double dv = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) { const int f = cface(g, d, s, i, j, k); dv += (s ? 1.0 : -1.0) * (g.pimple ? alphaf.a[d][f] : 1.0) * phi.a[d][f]; }
        double ce = dv * geo_rV(g, i, j, k);
        if (g.pimple) ce += (alpha[c] - alphaOld[c]) / g.dt;
        v[0] += fabs(ce) * geo_V(g, i, j, k);
        v[1] += ce * geo_V(g, i, j, k);
    }
    const int mx[2] = {0, 0};
    block_reduce_store<2>(v, mx, partials, fy_lb, fy_stride);
}

// ico:    U = HbyA - rAU*fvc::grad(p)                                                         icoFoamYade.C:136
// pimple: Uc = HbyA + rAUc*fvc::reconstruct((phicForces - pEqn.flux()/alphacf)/rAUcf)         pEqn.H:43-45
// DIAG: the same pass also forms the two per-cell diagnostics of the corrected flux that used to be sweeps of their own -- the continuity
// errors (continuityErrs.H:32-46, k_cont_err; slots 0, 1) and the Courant sums of the NEXT pass of the time loop (CourantNo.H:32-49,
// k_courant; slots 2 = max, 3 = sum: phi does not change between the last corrector and the next runTime++).  Same per-cell expressions
// and the same block partition as the stand-alone kernels, so the folded values are the same bits.
template <bool DIAG>
__global__ __launch_bounds__(256) void k_U_correct(FvGeo g, const double* __restrict__ HbyA, const double* __restrict__ rAU,
                                                   const double* __restrict__ p, CFace3 psn, CFace3 phiForces, CFace3 pflux, CFace3 alphaf,
                                                   CFace3 rAUf, double* __restrict__ U, CFace3 phi, const double* __restrict__ alpha,
                                                   const double* __restrict__ alphaOld, double* __restrict__ partials) {
    double v[4] = {0, 0, 0, 0};
    FY_RED_LOOP_G(g, t) {
        int i, j, k; ijk_of(g, t, i, j, k);
        const int c = t + g.c0;
        const double r = rAU[c];
        double out[3];      // see k_bmom
        double dv = 0.0, sp = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (!g.pimple) {
                double fv[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (onb(g, d, s, i, j, k)) fv[s] = pbv(g, p, psn, c, d, s, cface(g, d, s, i, j, k));
                    else fv[s] = geo_lerp_side(g, d, s, d == 0 ? i : d == 1 ? j : k, p[c], p[c + (s ? stride_of(g, d) : -stride_of(g, d))]);
                }
                out[d] = HbyA[3 * (size_t)c + d] - r * ((fv[1] - fv[0]) * geo_rh(g, d, d == 0 ? i : d == 1 ? j : k));
            } else {
                double sm = 0;
#pragma unroll
                for (int s = 0; s < 2; ++s) sm += pflux.a[d][cface(g, d, s, i, j, k)];          // (phicForces - pEqn.flux() / alphacf) / rAUcf, formed by flux_correct_face
                out[d] = HbyA[3 * (size_t)c + d] + r * (sm / (2.0 * geo_Af(g, d, i, j, k)));
            }
            if (DIAG) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int f = cface(g, d, s, i, j, k);
                    const double ph = phi.a[d][f];
                    dv += (s ? 1.0 : -1.0) * (g.pimple ? alphaf.a[d][f] : 1.0) * ph;
                    sp += fabs(ph);
                }
            }
        }
        for (int d = 0; d < 3; ++d) U[3 * (size_t)c + d] = out[d];
        if (DIAG) {
            double ce = dv * geo_rV(g, i, j, k);
            if (g.pimple) ce += (alpha[c] - alphaOld[c]) / g.dt;
            v[0] += fabs(ce) * geo_V(g, i, j, k);
            v[1] += ce * geo_V(g, i, j, k);
            v[2] = fmax(v[2], sp * geo_rV(g, i, j, k));
            v[3] += sp;
        }
    }
    if (DIAG) {
        const int mx[4] = {0, 0, 1, 0};
        block_reduce_store<4>(v, mx, partials, fy_lb, fy_stride);
    }
}

// ------------------------------------------------------------------------------------------------ fused corrector sweeps (round 5)
// The corrector used to be five cell-centred sweeps -- phiHbyA, pressure assembly, PCG's r0 = b - A p, flux correction, velocity correction --
// each of which handed its face fields to the next through HBM (phiHbyA 24, psn, the flux term 24, rAUf / alphaf streamed by every one of them,
// the matrix read back by k_p_init: SURVEY.md 8(d) prices a corrector at 504 B/cell, the five sweeps moved 730 - 800).  Here a cell forms the
// values of ALL SIX of its faces itself -- same expressions, same operands, same order as the face functions above, so the same bits on both
// sides of a face -- and uses them at once; only what a later pass needs is stored, by the face's owner (the cell on its high side; the low-side
// cell for the last face of a row).  The neighbour values a cell needs twice as often come out of L1 / L2, not HBM.
// FaceSrc: a linear face interpolate (alphacf, rAUf) either streamed from its face array or re-formed from the cell field (8 B/cell instead
// of 24; the expression of the kernel that fills the array)
struct FaceSrc { CFace3 arr; const double* cell; };
template <int D>
__device__ __forceinline__ double alphaf_at(const FvGeo& g, const FaceSrc& a, size_t f, int fi, int fj, int fk) {
    if (!a.cell) return a.arr.a[D][f];
    const int q = D == 0 ? fi : D == 1 ? fj : fk;
    if (face_low_b(g, D, q) || face_high_b(g, D, q)) return 1.0;      // interp_alpha_face
    const int c = cidx(g, fi, fj, fk);
    return geo_lerp(g, D, q, a.cell[c - stride_of(g, D)], a.cell[c]);
}
template <int D>
__device__ __forceinline__ double rAUf_at(const FvGeo& g, const FaceSrc& a, size_t f, int fi, int fj, int fk) {
    if (!a.cell) return a.arr.a[D][f];
    const int q = D == 0 ? fi : D == 1 ? fj : fk;
    if (face_low_b(g, D, q)) return a.cell[cidx(g, fi, fj, fk)];     // k_interp_rAU / rAUf_phi_forces_face
    if (face_high_b(g, D, q)) return a.cell[cidx(g, fi - (D == 0), fj - (D == 1), fk - (D == 2))];
    const int c = cidx(g, fi, fj, fk);
    return geo_lerp(g, D, q, a.cell[c - stride_of(g, D)], a.cell[c]);
}
// CALL(D, S, fi, fj, fk): the six faces of cell (i, j, k) in the order every per-cell sum over faces uses (x-, x+, y-, y+, z-, z+)
#define FY_ALL_FACES(i, j, k, CALL) \
    do { CALL(0, 0, i, j, k); CALL(0, 1, i + 1, j, k); CALL(1, 0, i, j, k); CALL(1, 1, i, j + 1, k); CALL(2, 0, i, j, k); CALL(2, 1, i, j, k + 1); } while (0)

// Both sweeps are written GATHER FIRST: every operand of the six faces is loaded up front from an address that is always valid (across a
// boundary face the "neighbour" is the cell itself), then the faces are evaluated from registers.  Written face by face -- a branch per face
// with its loads inside -- a wave makes one dependent memory round trip per face and more (the first version of these kernels: 385 us for what
// the three sweeps it replaced did in 396; 825 VALU instructions and 70 loads per wave, 3 % of the cycles waiting on anything but memory).
// Only what boundary faces alone need (psn, U at a fixedFluxPressure patch) stays behind a branch: interior waves skip it.

// value of a vector field's component d at a boundary face, from the field's own-cell value (Ub, component d: patch >> 1 == d here)
__device__ __forceinline__ double ub_normal(const FvGeo& g, int patch, int d, double own_d) {
    return g.u_bc[patch] == 0 ? g.u_val[patch][d] : (g.u_bc[patch] == 2 ? 0.0 : own_d);
}
// flux correction + velocity correction [+ continuity errors + the next pass's Courant sums] in one sweep (icoFoamYade.C:127-137, pEqn.H:39-45,
// continuityErrs.H, CourantNo.H): k_flux_correct_cells and k_U_correct<DIAG> without the face field between them
template <bool DIAG, bool FFC>
__global__ __launch_bounds__(256) FY_WPE_ATTR(FY_WPE_BACK) void k_corr_back(FvGeo g, const double* __restrict__ p, CFace3 phiHbyA, FaceSrc rAUf, FaceSrc alphaf, CFace3 psn,
                                                   CFace3 phiForces, Face3 phi, const double* __restrict__ HbyA, const double* __restrict__ rAU,
                                                   double* __restrict__ U, const double* __restrict__ alpha, const double* __restrict__ alphaOld,
                                                   double* __restrict__ partials) {
    double v[4] = {0, 0, 0, 0};
    FY_RED_LOOP_G(g, t) {
        int i, j, k; ijk_of(g, t, i, j, k);
        const int c = t + g.c0;
        const int ijk[3] = {i, j, k};
        const bool pim = g.pimple != 0;
        // ---- gather
        bool bnd[3][2]; int nb[3][2], fx[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bnd[d][s] = onb(g, d, s, i, j, k);
                nb[d][s] = bnd[d][s] ? c : c + (s ? stride_of(g, d) : -stride_of(g, d));
                fx[d][s] = cface(g, d, s, i, j, k);
            }
        const double pc = p[c], rc = rAU[c], ac = pim ? alpha[c] : 1.0;
        const double hb[3] = {HbyA[3 * (size_t)c], HbyA[3 * (size_t)c + 1], HbyA[3 * (size_t)c + 2]};
        double pn[3][2], rfv[3][2], afv[3][2], phh[3][2], pfo[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                pn[d][s] = p[nb[d][s]];
                phh[d][s] = phiHbyA.a[d][fx[d][s]];
                pfo[d][s] = pim ? phiForces.a[d][fx[d][s]] : 0.0;
                if (FFC) { rfv[d][s] = rAU[nb[d][s]]; afv[d][s] = pim ? alpha[nb[d][s]] : 1.0; }
                else { rfv[d][s] = rAUf.arr.a[d][fx[d][s]]; afv[d][s] = pim ? alphaf.arr.a[d][fx[d][s]] : 1.0; }
            }
        // ---- the six faces (flux_correct_face's expressions)
        double pfl[3][2], ph[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int q = ijk[d] + s;                       // the face's index along d
                const int fi = i + (d == 0 ? s : 0), fj = j + (d == 1 ? s : 0), fk = k + (d == 2 ? s : 0);
                const bool b = bnd[d][s];
                if (FFC) {
                    afv[d][s] = pim ? (b ? 1.0 : lerp_face(g, d, s, q, ac, afv[d][s])) : 1.0;          // interp_alpha_face
                    rfv[d][s] = b ? rc : lerp_face(g, d, s, q, rc, rfv[d][s]);                           // k_interp_rAU / rAUf_phi_forces_face
                }
                const double af = afv[d][s], rf = rfv[d][s];
                double fl = 0.0;
                if (b) {
                    const int patch = 2 * d + s;
                    if (g.p_bc[patch] == 1) { const double gb = kBfac * af * rf * geo_sfd_face(g, d, q, ndim(g, d), fi, fj, fk); fl = s ? gb * (g.p_val[patch] - pc) : gb * (pc - g.p_val[patch]); }
                    else if (g.p_bc[patch] == 2) fl = af * rf * geo_Af(g, d, fi, fj, fk) * psn.a[d][fx[d][s]];
                } else {
                    fl = af * rf * geo_sfd_face(g, d, q, ndim(g, d), fi, fj, fk) * (s ? pn[d][s] - pc : pc - pn[d][s]);
                }
                const double fa = fl / af;
                pfl[d][s] = pim ? (pfo[d][s] - fa) / rf : fl;
                ph[d][s] = phh[d][s] - fa;
                if (owns_face(g, d, s, i, j, k)) phi.a[d][fx[d][s]] = ph[d][s];
            }
        // ---- the velocity correction and the diagnostics (k_U_correct's expressions)
        double out[3];
        double dv = 0.0, sp = 0.0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (!pim) {
                double fv[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (bnd[d][s]) fv[s] = pbv(g, p, psn, c, d, s, fx[d][s]);
                    else fv[s] = geo_lerp_side(g, d, s, ijk[d], pc, pn[d][s]);
                }
                out[d] = hb[d] - rc * ((fv[1] - fv[0]) * geo_rh(g, d, ijk[d]));
            } else {
                double sm = 0;
#pragma unroll
                for (int s = 0; s < 2; ++s) sm += pfl[d][s];
                out[d] = hb[d] + rc * (sm / (2.0 * geo_Af(g, d, i, j, k)));
            }
            if (DIAG) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    dv += (s ? 1.0 : -1.0) * (pim ? afv[d][s] : 1.0) * ph[d][s];
                    sp += fabs(ph[d][s]);
                }
            }
        }
        for (int d = 0; d < 3; ++d) U[3 * (size_t)c + d] = out[d];
        if (DIAG) {
            double ce = dv * geo_rV(g, i, j, k);
            if (pim) ce += (ac - alphaOld[c]) / g.dt;
            v[0] += fabs(ce) * geo_V(g, i, j, k);
            v[1] += ce * geo_V(g, i, j, k);
            v[2] = fmax(v[2], sp * geo_rV(g, i, j, k));
            v[3] += sp;
        }
    }
    if (DIAG) {
        const int mx[4] = {0, 0, 1, 0};
        block_reduce_store<4>(v, mx, partials, fy_lb, fy_stride);
    }
}

// phiHbyA + constrainPressure, the pressure equation's assembly and PCG's first residual r0 = b - A p with its norm factor in one sweep
// (icoFoamYade.C:101-123, pEqn.H:4-33, PCG.C [OF-6]): k_phiHbyA_cells<KEEP>, k_assemble_pressure and k_p_init without phiHbyA, psn and the matrix
// being read back in between.  The residual's row uses the face coefficients in p_row's order (x-, x+, y-, y+, z-, z+; a boundary face's stored
// coefficient is 0 there, skipped here) and the block partition of k_p_init, so r0 and the two sums are k_p_init's bits.
// dcorr: the old-time part of the ddtCorr term per face, left by k_pre_coupling at the start of the step (coef / deltaT (phi.old - flux(U.old)));
// the term itself is rAUf dcorr [alphacf], the same product in every corrector.  STORE_A = false: a later corrector of the same momentum
// assembly -- the matrix in A stands (same rAU, same alphacf)
template <bool STORE_A, bool FFC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FY_WPE_FRONT))) void k_corr_front(FvGeo g, const double* __restrict__ HbyA, const double* __restrict__ U, CFace3 dcorr,
                                                    FaceSrc rAUf, FaceSrc alphaf, CFace3 phiForces, Face3 phiHbyA, Face3 psn,
                                                    const double* __restrict__ rAU, const double* __restrict__ alpha, const double* __restrict__ alphaOld, PMat A,
                                                    double* __restrict__ rhs, const double* __restrict__ x, const double* __restrict__ xbar_dev, double xsum_val,
                                                    double inv_n, double* __restrict__ res, double* __restrict__ partials) {
    double v[2] = {0, 0};
    const double xbar = (xbar_dev ? xbar_dev[0] : xsum_val) * inv_n;
    FY_RED_LOOP_G(g, t) {
        int i, j, k; ijk_of(g, t, i, j, k);
        const int c = t + g.c0;
        const int ijk[3] = {i, j, k};
        const bool pim = g.pimple != 0;
        const bool refc = g.need_ref && (i + g.nx * (j + g.ny * (k + g.kglob0))) == g.p_ref_cell;
        // ---- gather
        bool bnd[3][2]; int nb[3][2], fx[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bnd[d][s] = onb(g, d, s, i, j, k);
                nb[d][s] = bnd[d][s] ? c : c + (s ? stride_of(g, d) : -stride_of(g, d));
                fx[d][s] = cface(g, d, s, i, j, k);
            }
        const double xc = x[c], ac = pim ? alpha[c] : 1.0;
        const double rc = FFC ? rAU[c] : 0.0;
        const double hb[3] = {HbyA[3 * (size_t)c], HbyA[3 * (size_t)c + 1], HbyA[3 * (size_t)c + 2]};
        double hn[3][2], dc[3][2], pfo[3][2], xn[3][2], rfv[3][2], afv[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                hn[d][s] = HbyA[3 * (size_t)nb[d][s] + d];
                dc[d][s] = dcorr.a[d][fx[d][s]];
                pfo[d][s] = pim ? phiForces.a[d][fx[d][s]] : 0.0;
                xn[d][s] = x[nb[d][s]];
                if (FFC) { rfv[d][s] = rAU[nb[d][s]]; afv[d][s] = pim ? alpha[nb[d][s]] : 1.0; }
                else { rfv[d][s] = rAUf.arr.a[d][fx[d][s]]; afv[d][s] = pim ? alphaf.arr.a[d][fx[d][s]] : 1.0; }
            }
        // ---- the six faces: phiHbyA_face, then k_assemble_pressure's terms
        double dg = 0.0, r = 0.0, up[3] = {0, 0, 0};
        double gg6[3][2];                          // coefficient towards the neighbour across face (d, s); 0 on a physical boundary
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int q = ijk[d] + s;
                const int fi = i + (d == 0 ? s : 0), fj = j + (d == 1 ? s : 0), fk = k + (d == 2 ? s : 0);
                const bool b = bnd[d][s];
                const int patch = 2 * d + s;
                if (FFC) {
                    afv[d][s] = pim ? (b ? 1.0 : lerp_face(g, d, s, q, ac, afv[d][s])) : 1.0;
                    rfv[d][s] = b ? rc : lerp_face(g, d, s, q, rc, rfv[d][s]);
                }
                const double af = afv[d][s], rf = rfv[d][s];
                const double Afc = geo_Af(g, d, fi, fj, fk);
                double pv = (b ? ub_normal(g, patch, d, hb[d]) : lerp_face(g, d, s, q, hb[d], hn[d][s])) * Afc;      // face_flux_vec(HbyA)
                double add = rf * dc[d][s];
                if (pim) add *= af;
                pv += add;
                if (pim) pv += pfo[d][s];
                const bool own = owns_face(g, d, s, i, j, k);
                if (own) phiHbyA.a[d][fx[d][s]] = pv;
                double ph = (s ? 1.0 : -1.0) * af * pv;
                gg6[d][s] = 0.0;
                if (b) {
                    if (g.p_bc[patch] == 2) {               // constrainPressure (fixedFluxPressure): snGrad(p) from the flux that must pass
                        const double ubd = ub_normal(g, patch, d, g.u_bc[patch] == 1 ? U[3 * (size_t)c + d] : 0.0);
                        const double psn_v = (pv - ubd * Afc) / (rf * Afc);
                        if (own) psn.a[d][fx[d][s]] = psn_v;
                        ph = (s ? 1.0 : -1.0) * af * (pv - rf * geo_Af(g, d, i, j, k) * psn_v);
                    }
                    r -= ph;
                    if (g.p_bc[patch] == 1) { const double gb = kBfac * af * rf * geo_sfd(g, d, s, i, j, k); dg += gb; r += gb * g.p_val[patch]; }
                } else {
                    r -= ph;
                    const double gg = af * rf * geo_sfd(g, d, s, i, j, k);
                    dg += gg; gg6[d][s] = gg;
                    if (s) up[d] = gg;
                }
            }
        if (pim) r -= geo_V(g, i, j, k) * (ac - alphaOld[c]) / g.dt;
        if (refc) { r += dg * g.p_ref_value; dg += dg; }
        if (STORE_A) { A.diag[c] = dg; A.ux[c] = up[0]; A.uy[c] = up[1]; A.uz[c] = up[2]; }
        rhs[c] = r;
        // ---- r0 = b - A x, sum |r0|, the norm factor's sum (k_p_init)
        double Ax = dg * xc, rs = dg;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                if (!bnd[d][s]) { Ax -= gg6[d][s] * xn[d][s]; rs -= gg6[d][s]; }
        const double Aref = rs * xbar;
        const double rr = r - Ax;
        res[c] = rr;
        v[0] += fabs(rr);
        v[1] += fabs(Ax - Aref) + fabs(r - Aref);
    }
    const int mx[2] = {0, 0};
    block_reduce_store<2>(v, mx, partials, fy_lb, fy_stride);
}

// ------------------------------------------------------------------------------------------------ pressure solver
// y = A x.  The pEqn Laplacian apply: 48 algorithmic bytes per cell (diag 8 + ux,uy,uz 24 + x 8 + y 8); the low-side
// coefficients ux[c-1], uy[c-nx], uz[c-nx*ny] and the six neighbour x values are re-reads served by L1/L2.
// High-side boundary faces store 0, so the wrapped low-side reads (e.g. ux[c-1] at i = 0) multiply by 0 and only the
// array ends need an index guard.  c is a STORAGE index; with ghost planes the z-neighbours always exist (their coefficient
// is 0 on a physical boundary, the interface coefficient otherwise) and the ghost x values come from the halo exchange.
__device__ __forceinline__ double p_row(const PMat& A, const double* __restrict__ x, int c) {
    // Guarded terms are loaded from clamped (always valid) addresses and SELECTED: written as `if (c >= 1) a -= ...` every guard
    // becomes a divergent branch with its loads inside, i.e. seven dependent memory round trips per row instead of one.
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
__device__ __forceinline__ double p_rowsum(const PMat& A, int c) {
    const int sy = A.nx, sz = A.nx * A.ny;
    double rs = A.diag[c];
    if (c >= 1) rs -= A.ux[c - 1];
    if (c + 1 < A.ntot) rs -= A.ux[c];
    if (c >= sy) rs -= A.uy[c - sy];
    if (c + sy < A.ntot) rs -= A.uy[c];
    if (c >= sz) rs -= A.uz[c - sz];
    if (c + sz < A.ntot) rs -= A.uz[c];
    return rs;
}

__global__ __launch_bounds__(256) void k_p_apply(PMat A, const double* __restrict__ x, double* __restrict__ y) {
    const int t = swz_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= A.N) return;
    const int c = t + A.c0;
    y[c] = p_row(A, x, c);
}

// w = A u inside the single-reduction PCG (fv_pressure.cpp): slot 1 = u.w (delta); WITH_R: also slot 0 = u.r (gamma) -- on a single domain
// the V-cycle's last sweep has left gamma's partials in slot 0 already (k_mg_smooth_dot: same blocks, same order) and only delta is formed here
template <bool WITH_R>
__global__ __launch_bounds__(256) void k_p_apply_dot(PMat A, const double* __restrict__ x, const double* __restrict__ r, double* __restrict__ y, double* __restrict__ partials) {
    double v[2] = {0, 0};
    FY_RED_LOOP(t, A.N) {
        const int c = t + A.c0;
        const double a = p_row(A, x, c);
        y[c] = a;
        const double xc = x[c];
        if (WITH_R) v[0] += xc * r[c];
        v[1] += a * xc;
    }
    if (WITH_R) {
        const int mx[2] = {0, 0};
        block_reduce_store<2>(v, mx, partials);
    } else {
        double v1[1] = {v[1]};
        const int mx[1] = {0};
        block_reduce_store<1>(v1, mx, partials + gridDim.x);
    }
}

// r = b - A x ; slot 0 = sum|r| ; slot 1 = sum(|A x - A xbar| + |b - A xbar|)   (lduMatrix::solver::normFactor)
__global__ __launch_bounds__(256) void k_p_init(PMat A, const double* __restrict__ b, const double* __restrict__ x, const double* __restrict__ xbar_dev,
                                                double xsum_val, double inv_n, double* __restrict__ r, double* __restrict__ partials) {
    double v[2] = {0, 0};
    const double xbar = (xbar_dev ? xbar_dev[0] : xsum_val) * inv_n;      // sum(x) from the device, or the value the last update of x left with the host
    FY_RED_LOOP(t, A.N) {
        const int c = t + A.c0;
        const double Ax = p_row(A, x, c);
        const double Aref = p_rowsum(A, c) * xbar;
        const double rr = b[c] - Ax;
        r[c] = rr;
        v[0] += fabs(rr);
        v[1] += fabs(Ax - Aref) + fabs(b[c] - Aref);
    }
    const int mx[2] = {0, 0};
    block_reduce_store<2>(v, mx, partials);
}

// dot product / plain sum over the owned range [c0, c0 + n) of arrays given by their storage base
__global__ __launch_bounds__(256) void k_dot(int n, int c0, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ partials) {
    double v[1] = {0};
    FY_RED_LOOP(t, n) v[0] += a[t + c0] * (b ? b[t + c0] : 1.0);
    const int mx[1] = {0};
    block_reduce_store<1>(v, mx, partials);
}

// The vector update of the single-reduction (Chronopoulos-Gear) form of PCG.C's loop [OF-6]: with u = M^-1 r, w = A u and the ONE reduction
// gamma = u.r, delta = u.w per iteration,
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old)      (beta = 0, alpha = gamma / delta in the first iteration)
//     p = u + beta p,  s = w + beta s  (= A p),  x += alpha p,  r -= alpha s
// -- the iterates of the textbook loop in exact arithmetic, with the search direction's image s carried by recurrence instead of a second dot
// product + all-reduce after the matrix-vector product.  sc[0] = gamma, sc[1] = delta (this iteration's fold); {gamma_old, alpha_old} live in
// sc[2 + 2 q], sc[3 + 2 q] with q = it & 1: an iteration reads set q and leaves set 1 - q, so no block reads what another one has rewritten.
// slot 0 = sum|r|, slot 1 = sum(x): the next solve's xbar (normFactor) -- k_dot's partition and order, so k_dot's bits, without k_dot's pass.
// FIRST: p = u and s = w need no pass of their own -- the host lets the two pairs of buffers trade places afterwards (fv_pressure.cpp)
template <bool FIRST>
__global__ __launch_bounds__(256) void k_pcg_cg_update(int n, int c0, const double* __restrict__ u, const double* __restrict__ w, double* __restrict__ p,
                                                       double* __restrict__ sv, double* __restrict__ x, double* __restrict__ r, double* __restrict__ sc, int it,
                                                       double* __restrict__ partials) {
    double v[2] = {0, 0};
    const double gamma = sc[0], delta = sc[1];
    double beta = 0.0, al = gamma / delta;
    if (!FIRST) {
        const double gold = sc[2 + 2 * (it & 1)], aold = sc[3 + 2 * (it & 1)];
        beta = gamma / gold;
        al = gamma / (delta - beta * gamma / aold);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[2 + 2 * ((it + 1) & 1)] = gamma; sc[3 + 2 * ((it + 1) & 1)] = al; }
    FY_RED_LOOP(t, n) {
        const int c = t + c0;
        double pn = u[c], sn = w[c];
        if (!FIRST) {
            pn = pn + beta * p[c]; sn = sn + beta * sv[c];
            p[c] = pn; sv[c] = sn;
        }
        const double xn = x[c] + al * pn;
        x[c] = xn;
        const double rr = r[c] - al * sn;
        r[c] = rr;
        v[0] += fabs(rr);
        v[1] += xn;
    }
    const int mx[2] = {0, 0};
    block_reduce_store<2>(v, mx, partials);
}

__global__ __launch_bounds__(256) void k_jacobi_precond(PMat A, const double* __restrict__ r, double* __restrict__ z) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < A.N) { const int c = t + A.c0; z[c] = r[c] / A.diag[c]; }
}

// A_coarse = 1/2 P^T A P, piecewise-constant P over 2x2x2 aggregates (gather form: one thread per owned coarse cell).
// The z-face above the top owned fine plane is a slab interface (or a physical boundary with coefficient 0): always "crossing".
// ref_c / ref_term: fvMatrix::setReference adds a POINT term a_ref to the reference cell's diagonal (k_assemble_pressure); it is what
// makes the closed-box operator non-singular, i.e. 1^T A 1 = a_ref.  The Galerkin product with the factor 1/2 -- right for the Laplacian
// under piecewise-constant transfer -- would halve that term on every level, and an EXACT coarse solve would then over-correct the constant
// mode by 2^levels (measured: 2.0 -> 3.85 PCG iterations per step at C3 when the 120 Jacobi sweeps, which never touched that mode, became
// the direct solve).  So the aggregate that holds the reference cell gets the missing half back: every level carries a_ref unscaled.
__global__ __launch_bounds__(256) void k_mg_coarsen(PMat F, PMat C, int ref_c, const double* __restrict__ ref_term) {
    const int tc = blockIdx.x * 256 + threadIdx.x;
    if (tc >= C.N) return;
    const int I = tc % C.nx, q = tc / C.nx, J = q % C.ny, K = q / C.ny;
    double dg = 0, ux = 0, uy = 0, uz = 0;
    for (int dk = 0; dk < 2; ++dk) {
        const int k = 2 * K + dk; if (k >= F.nz) break;
        for (int dj = 0; dj < 2; ++dj) {
            const int j = 2 * J + dj; if (j >= F.ny) break;
            for (int di = 0; di < 2; ++di) {
                const int i = 2 * I + di; if (i >= F.nx) break;
                const int c = F.c0 + i + F.nx * (j + F.ny * k);
                dg += 0.5 * F.diag[c];
                if (di == 0 && i + 1 < F.nx) dg -= F.ux[c]; else ux += 0.5 * F.ux[c];
                if (dj == 0 && j + 1 < F.ny) dg -= F.uy[c]; else uy += 0.5 * F.uy[c];
                if (dk == 0 && k + 1 < F.nz) dg -= F.uz[c]; else uz += 0.5 * F.uz[c];
            }
        }
    }
    if (tc == ref_c) dg += 0.5 * ref_term[0];
    const int cc = tc + C.c0;
    C.diag[cc] = dg; C.ux[cc] = ux; C.uy[cc] = uy; C.uz[cc] = uz;
}

// a_ref of the comment above: half of the (doubled) diagonal of the reference cell at level 0; 0 on a rank that does not own the cell
__global__ void k_mg_ref_term(PMat A0, int ref_local, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = ref_local >= 0 ? 0.5 * A0.diag[A0.c0 + ref_local] : 0.0;
}

// coarse ghost plane under the first owned coarse plane: uz = 1/2 sum of the fine ghost-plane uz of its 2x2 footprint
__global__ __launch_bounds__(256) void k_mg_coarsen_ghost(PMat F, PMat C) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= C.nx * C.ny) return;
    const int I = t % C.nx, J = t / C.nx;
    double uz = 0;
    for (int dj = 0; dj < 2; ++dj) {
        const int j = 2 * J + dj; if (j >= F.ny) break;
        for (int di = 0; di < 2; ++di) {
            const int i = 2 * I + di; if (i >= F.nx) break;
            uz += 0.5 * F.uz[F.c0 - F.nx * F.ny + i + F.nx * j];
        }
    }
    C.uz[C.c0 - C.nx * C.ny + t] = uz;
}

__global__ __launch_bounds__(256) void k_mg_smooth_first(PMat A, const double* __restrict__ b, double* __restrict__ x, double w) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < A.N) { const int c = t + A.c0; x[c] = w * b[c] / A.diag[c]; }
}

// smooth_first + one smooth in a single pass (non-distributed levels >= 1, which are launch-latency bound): the neighbours' first iterate
// x1 = w b / diag is recomputed inline instead of being stored and re-read -- the same operations on the same operands, so x2 is bit-identical
__global__ __launch_bounds__(256) void k_mg_smooth_two_from_zero(PMat A, const double* __restrict__ b, double* __restrict__ xn, double w, double w2) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= A.N) return;
    const int c = t + A.c0;
    const int sy = A.nx, sz = A.nx * A.ny, last = A.ntot - 1;
    const int xm = max(c - 1, 0), xp = min(c + 1, last), ym = max(c - sy, 0), yp = min(c + sy, last), zm = max(c - sz, 0), zp = min(c + sz, last);
    const double dc = A.diag[c], bc = b[c];
    const double x1c = w * bc / dc;
    const double t0 = A.ux[xm] * (w * b[xm] / A.diag[xm]), t1 = A.ux[c] * (w * b[xp] / A.diag[xp]);
    const double t2 = A.uy[ym] * (w * b[ym] / A.diag[ym]), t3 = A.uy[c] * (w * b[yp] / A.diag[yp]);
    const double t4 = A.uz[zm] * (w * b[zm] / A.diag[zm]), t5 = A.uz[c] * (w * b[zp] / A.diag[zp]);
    double a = dc * x1c;                                   // p_row(A, x1, c), same order
    a = (c >= 1) ? a - t0 : a;
    a = (c + 1 < A.ntot) ? a - t1 : a;
    a = (c >= sy) ? a - t2 : a;
    a = (c + sy < A.ntot) ? a - t3 : a;
    a = (c >= sz) ? a - t4 : a;
    a = (c + sz < A.ntot) ? a - t5 : a;
    xn[c] = x1c + w2 * (bc - a) / dc;
}

// the last level-0 sweep of a V-cycle used as PCG preconditioner: z = xn, and PCG wants z.r next -- r is this level's b, already in a
// register -- so the block partials of the dot product (k_dot's, same blocks, same order) come out of the same pass
__global__ __launch_bounds__(256) void k_mg_smooth_dot(PMat A, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ xn, double w,
                                                       double* __restrict__ partials) {
    double v[1] = {0};
    FY_RED_LOOP(t, A.N) {
        const int c = t + A.c0;
        const double bc = b[c];
        const double z = x[c] + w * (bc - p_row(A, x, c)) / A.diag[c];
        xn[c] = z;
        v[0] += z * bc;
    }
    const int mx[1] = {0};
    block_reduce_store<1>(v, mx, partials);
}
__global__ __launch_bounds__(256) void k_mg_smooth(PMat A, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ xn, double w) {
    const int t = swz_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (t >= A.N) return;
    const int c = t + A.c0;
    xn[c] = x[c] + w * (b[c] - p_row(A, x, c)) / A.diag[c];
}

// First post-smoothing sweep fused with the prolongation: the sweep reads x + P e -- its own cell's and its six neighbours' -- with e
// taken from the (8 x smaller, cache-resident) coarse solution, instead of a separate x += P e pass over the level (one launch and
// 16 B/cell fewer).  Same additions and the same row arithmetic as k_mg_prolong_add followed by k_mg_smooth (p_row's clamped-and-
// selected neighbour terms): bit-identical.  Levels without ghost planes only (c0 = 0).
__global__ __launch_bounds__(256) void k_mg_smooth_prolong(PMat A, const double* __restrict__ b, const double* __restrict__ x, PMat C,
                                                           const double* __restrict__ xc, double* __restrict__ xn, double w) {
    const int c = swz_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (c >= A.N) return;
    const int i = c % A.nx, q = c / A.nx, j = q % A.ny, k = q / A.ny;
    const int sy = A.nx, sz = A.nx * A.ny, last = A.ntot - 1;
    const int I = i >> 1, J = j >> 1, K = k >> 1;
    const int Im = max(i - 1, 0) >> 1, Ip = min(i + 1, A.nx - 1) >> 1, Jm = max(j - 1, 0) >> 1, Jp = min(j + 1, A.ny - 1) >> 1,
              Km = max(k - 1, 0) >> 1, Kp = min(k + 1, A.nz - 1) >> 1;
    const double* e = xc + C.c0;
    const int rowc = C.nx * (J + C.ny * K);
    const int xm = max(c - 1, 0), xp = min(c + 1, last), ym = max(c - sy, 0), yp = min(c + sy, last), zm = max(c - sz, 0), zp = min(c + sz, last);
    // (a clamped neighbour index pairs with a clamped parent: its coefficient is zero or the term is deselected below, as in p_row)
    const double vc = x[c] + e[I + rowc];
    const double vxm = x[xm] + e[Im + rowc], vxp = x[xp] + e[Ip + rowc];
    const double vym = x[ym] + e[I + C.nx * (Jm + C.ny * K)], vyp = x[yp] + e[I + C.nx * (Jp + C.ny * K)];
    const double vzm = x[zm] + e[I + C.nx * (J + C.ny * Km)], vzp = x[zp] + e[I + C.nx * (J + C.ny * Kp)];
    const double uxc = A.ux[c], uyc = A.uy[c], uzc = A.uz[c];
    const double t0 = A.ux[xm] * vxm, t1 = uxc * vxp, t2 = A.uy[ym] * vym, t3 = uyc * vyp, t4 = A.uz[zm] * vzm, t5 = uzc * vzp;
    const double dg = A.diag[c];
    double a = dg * vc;
    a = (c >= 1) ? a - t0 : a;
    a = (c + 1 < A.ntot) ? a - t1 : a;
    a = (c >= sy) ? a - t2 : a;
    a = (c + sy < A.ntot) ? a - t3 : a;
    a = (c >= sz) ? a - t4 : a;
    a = (c + sz < A.ntot) ? a - t5 : a;
    xn[c] = vc + w * (b[c] - a) / dg;
}

__global__ __launch_bounds__(256) void k_mg_residual_restrict(PMat A, const double* __restrict__ b, const double* __restrict__ x, PMat C,
                                                              double* __restrict__ bc) {
    const int tc = blockIdx.x * 256 + threadIdx.x;
    if (tc >= C.N) return;
    const int I = tc % C.nx, q = tc / C.nx, J = q % C.ny, K = q / C.ny;
    double acc = 0;
    for (int dk = 0; dk < 2; ++dk) {
        const int k = 2 * K + dk; if (k >= A.nz) break;
        for (int dj = 0; dj < 2; ++dj) {
            const int j = 2 * J + dj; if (j >= A.ny) break;
            for (int di = 0; di < 2; ++di) {
                const int i = 2 * I + di; if (i >= A.nx) break;
                const int c = A.c0 + i + A.nx * (j + A.ny * k);
                acc += b[c] - p_row(A, x, c);
            }
        }
    }
    bc[tc + C.c0] = acc;
}

// Coalesced restriction for the big levels: a 256-thread block covers a 64 x 2 x 2 tile of FINE cells (one lane per fine cell,
// consecutive lanes on consecutive x: every load is a coalesced row), residuals meet in LDS and 32 lanes fold the 8 children of
// each coarse cell.  The gather form above (one thread per coarse cell, 8 strided stencil evaluations each) measured 101 us at
// 160^3 against 41 us for a smoother sweep of the same level.
__global__ __launch_bounds__(256) void k_mg_residual_restrict_tiled(PMat A, const double* __restrict__ b, const double* __restrict__ x, PMat C,
                                                                    double* __restrict__ bc) {
    __shared__ double r[2][2][64];
    const int tx = threadIdx.x & 63, ty = (threadIdx.x >> 6) & 1, tz = threadIdx.x >> 7;
    const int i = blockIdx.x * 64 + tx, j = blockIdx.y * 2 + ty, k = blockIdx.z * 2 + tz;
    double v = 0.0;
    if (i < A.nx && j < A.ny && k < A.nz) {
        const int c = A.c0 + i + A.nx * (j + A.ny * k);
        v = b[c] - p_row(A, x, c);
    }
    r[tz][ty][tx] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int I = blockIdx.x * 32 + (int)threadIdx.x, J = blockIdx.y, K = blockIdx.z;
        if (I < C.nx) {
            const int q = 2 * (int)threadIdx.x;
            bc[C.c0 + I + C.nx * (J + C.ny * K)] = ((r[0][0][q] + r[0][0][q + 1]) + (r[0][1][q] + r[0][1][q + 1])) +
                                                   ((r[1][0][q] + r[1][0][q + 1]) + (r[1][1][q] + r[1][1][q + 1]));
        }
    }
}

// ------------------------------------------------------------------------------------------------ two cells per thread (round 5)
// The scalar-field sweeps of the pressure solver with TWO consecutive cells per thread: every coefficient and field value of the pair and of its
// y / z neighbours comes as one 16-byte load (the x-neighbours of the pair's ends as 8-byte ones), half the load instructions for the same bytes.
// tools/micro/lap_pairs.hip: the Laplacian apply 365 -> 323 us at 320^3 (4.3 -> 4.9 TB/s), 29.4 -> 27.0 us at 160^3, 3.6 -> 2.9 us at 64^3; four cells
// per thread lose (lanes 32 bytes apart).  Needs an even row length, even c0 / N / ntot and 16-byte aligned arrays (pairs_ok); the rows are p_row's
// operations in p_row's order and the block partials are folded in the one-cell kernels' order: the same bits, by construction and by test.
__device__ __forceinline__ double2 ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ void st2(double* p, double a, double b) { *reinterpret_cast<double2*>(p) = make_double2(a, b); }
struct PairIdx { int c, ym, yp, zm, zp, xm, xp; };
__device__ __forceinline__ PairIdx pair_idx(const PMat& A, int c) {
    const int sy = A.nx, sz = A.nx * A.ny, last = A.ntot - 1;
    // (a clamped index is that of a deselected term, or pairs with a zero coefficient -- as in p_row; the pair loads stay inside the array and aligned)
    return PairIdx{c, max(c - sy, 0), min(c + sy, last - 1), max(c - sz, 0), min(c + sz, last - 1), max(c - 1, 0), min(c + 2, last)};
}
struct P2 { double2 c, ym, yp, zm, zp; double xm, xp; };            // a field at the pair, at its y / z neighbour pairs and at the cells left and right of it
__device__ __forceinline__ P2 ld_p2(const double* __restrict__ f, const PairIdx& q) {
    return P2{ld2(f + q.c), ld2(f + q.ym), ld2(f + q.yp), ld2(f + q.zm), ld2(f + q.zp), f[q.xm], f[q.xp]};
}
struct C2 { double2 dg, ux, uy, uz, uym, uzm; double uxm; };
__device__ __forceinline__ C2 ld_c2(const PMat& A, const PairIdx& q) {
    return C2{ld2(A.diag + q.c), ld2(A.ux + q.c), ld2(A.uy + q.c), ld2(A.uz + q.c), ld2(A.uy + q.ym), ld2(A.uz + q.zm), A.ux[q.xm]};
}
// rows c and c + 1 of A applied to the field whose values are X
__device__ __forceinline__ double2 pair_rows(const PMat& A, const C2& K, const P2& X, int c) {
    const int sy = A.nx, sz = A.nx * A.ny, d = c + 1;
    double a = K.dg.x * X.c.x;
    a = (c >= 1) ? a - K.uxm * X.xm : a;
    a = (c + 1 < A.ntot) ? a - K.ux.x * X.c.y : a;
    a = (c >= sy) ? a - K.uym.x * X.ym.x : a;
    a = (c + sy < A.ntot) ? a - K.uy.x * X.yp.x : a;
    a = (c >= sz) ? a - K.uzm.x * X.zm.x : a;
    a = (c + sz < A.ntot) ? a - K.uz.x * X.zp.x : a;
    double b = K.dg.y * X.c.y;
    b = b - K.ux.x * X.c.x;                                           // (d >= 1 always)
    b = (d + 1 < A.ntot) ? b - K.ux.y * X.xp : b;
    b = (d >= sy) ? b - K.uym.y * X.ym.y : b;
    b = (d + sy < A.ntot) ? b - K.uy.y * X.yp.y : b;
    b = (d >= sz) ? b - K.uzm.y * X.zm.y : b;
    b = (d + sz < A.ntot) ? b - K.uz.y * X.zp.y : b;
    return make_double2(a, b);
}
// 128 threads x 2 cells = one 256-cell block of the one-cell kernels; the partial is folded in THEIR order: a wave of theirs is a half-wave here (lane l of it =
// lane l / 2, component l & 1), their shuffle offsets 32 .. 2 are lane offsets 16 .. 1 per component, their offset 1 is x + y in the half-wave's first lane
template <int N>
__device__ __forceinline__ void block_reduce_store_pairs(double2 (&v)[N], const int (&is_max)[N], double* partials, int lb = -1, int stride = 0) {
    if (lb < 0) lb = (int)blockIdx.x;
    if (stride <= 0) stride = (int)gridDim.x;
    __shared__ double sh[4][N];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        double x = v[q].x, y = v[q].y;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const double x2 = __shfl_down(x, o, 64), y2 = __shfl_down(y, o, 64);
            x = is_max[q] ? fmax(x, x2) : x + x2;
            y = is_max[q] ? fmax(y, y2) : y + y2;
        }
        if ((lane & 31) == 0) sh[2 * wv + (lane >> 5)][q] = is_max[q] ? fmax(x, y) : x + y;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        const int q = threadIdx.x;
        double x = sh[0][q];
        for (int w = 1; w < 4; ++w) x = is_max[q] ? fmax(x, sh[w][q]) : x + sh[w][q];
        partials[(size_t)q * stride + lb] = x;
    }
}
// reducing pair kernels: 128 threads, the logical 256-cell block of FY_RED_LOOP; plain ones: 256 threads, 512 cells per block
#define FY_RED_LOOP2(t, n) const int t = swz_block(blockIdx.x, gridDim.x) * 256 + 2 * (int)threadIdx.x; if (t < (n))
#define FY_PAIR_LOOP(t, n) const int t = swz_block(blockIdx.x, gridDim.x) * 512 + 2 * (int)threadIdx.x; if (t < (n))

__global__ __launch_bounds__(256) void k_p_apply2(PMat A, const double* __restrict__ x, double* __restrict__ y) {
    FY_PAIR_LOOP(t, A.N) {
        const int c = t + A.c0;
        const PairIdx q = pair_idx(A, c);
        const double2 a = pair_rows(A, ld_c2(A, q), ld_p2(x, q), c);
        st2(y + c, a.x, a.y);
    }
}
template <bool WITH_R>
__global__ __launch_bounds__(128) void k_p_apply_dot2(PMat A, const double* __restrict__ x, const double* __restrict__ r, double* __restrict__ y, double* __restrict__ partials) {
    double2 v[2] = {make_double2(0, 0), make_double2(0, 0)};
    FY_RED_LOOP2(t, A.N) {
        const int c = t + A.c0;
        const PairIdx q = pair_idx(A, c);
        const P2 X = ld_p2(x, q);
        const double2 a = pair_rows(A, ld_c2(A, q), X, c);
        st2(y + c, a.x, a.y);
        if (WITH_R) { const double2 rc = ld2(r + c); v[0] = make_double2(X.c.x * rc.x, X.c.y * rc.y); }
        v[1] = make_double2(a.x * X.c.x, a.y * X.c.y);
    }
    if (WITH_R) {
        const int mx[2] = {0, 0};
        block_reduce_store_pairs<2>(v, mx, partials);
    } else {
        double2 v1[1] = {v[1]};
        const int mx[1] = {0};
        block_reduce_store_pairs<1>(v1, mx, partials + gridDim.x);
    }
}
template <bool FIRST>
__global__ __launch_bounds__(128) void k_pcg_cg_update2(int n, int c0, const double* __restrict__ u, const double* __restrict__ w, double* __restrict__ p,
                                                        double* __restrict__ sv, double* __restrict__ x, double* __restrict__ r, double* __restrict__ sc, int it,
                                                        double* __restrict__ partials) {
    double2 v[2] = {make_double2(0, 0), make_double2(0, 0)};
    const double gamma = sc[0], delta = sc[1];
    double beta = 0.0, al = gamma / delta;
    if (!FIRST) {
        const double gold = sc[2 + 2 * (it & 1)], aold = sc[3 + 2 * (it & 1)];
        beta = gamma / gold;
        al = gamma / (delta - beta * gamma / aold);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc[2 + 2 * ((it + 1) & 1)] = gamma; sc[3 + 2 * ((it + 1) & 1)] = al; }
    FY_RED_LOOP2(t, n) {
        const int c = t + c0;
        double2 pn = ld2(u + c), sn = ld2(w + c);
        if (!FIRST) {
            const double2 po = ld2(p + c), so = ld2(sv + c);
            pn = make_double2(pn.x + beta * po.x, pn.y + beta * po.y); sn = make_double2(sn.x + beta * so.x, sn.y + beta * so.y);
            st2(p + c, pn.x, pn.y); st2(sv + c, sn.x, sn.y);
        }
        const double2 xo = ld2(x + c), ro = ld2(r + c);
        const double xa = xo.x + al * pn.x, xb = xo.y + al * pn.y;
        st2(x + c, xa, xb);
        const double ra = ro.x - al * sn.x, rb = ro.y - al * sn.y;
        st2(r + c, ra, rb);
        v[0] = make_double2(fabs(ra), fabs(rb));
        v[1] = make_double2(xa, xb);
    }
    const int mx[2] = {0, 0};
    block_reduce_store_pairs<2>(v, mx, partials);
}
__global__ __launch_bounds__(256) void k_mg_smooth_two_from_zero2(PMat A, const double* __restrict__ b, double* __restrict__ xn, double w, double w2) {
    const int t = (int)blockIdx.x * 512 + 2 * (int)threadIdx.x;
    if (t >= A.N) return;
    const int c = t + A.c0;
    const PairIdx q = pair_idx(A, c);
    const P2 B = ld_p2(b, q), D = ld_p2(A.diag, q);
    C2 K = ld_c2(A, q);
    K.dg = D.c;
    // the first iterate x1 = w b / diag at the pair and at every neighbour, formed inline as by the one-cell kernel
    P2 X1;
    X1.c = make_double2(w * B.c.x / D.c.x, w * B.c.y / D.c.y);
    X1.ym = make_double2(w * B.ym.x / D.ym.x, w * B.ym.y / D.ym.y); X1.yp = make_double2(w * B.yp.x / D.yp.x, w * B.yp.y / D.yp.y);
    X1.zm = make_double2(w * B.zm.x / D.zm.x, w * B.zm.y / D.zm.y); X1.zp = make_double2(w * B.zp.x / D.zp.x, w * B.zp.y / D.zp.y);
    X1.xm = w * B.xm / D.xm; X1.xp = w * B.xp / D.xp;
    const double2 a = pair_rows(A, K, X1, c);
    st2(xn + c, X1.c.x + w2 * (B.c.x - a.x) / D.c.x, X1.c.y + w2 * (B.c.y - a.y) / D.c.y);
}
__global__ __launch_bounds__(128) void k_mg_smooth_dot2(PMat A, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ xn, double w,
                                                        double* __restrict__ partials) {
    double2 v[1] = {make_double2(0, 0)};
    FY_RED_LOOP2(t, A.N) {
        const int c = t + A.c0;
        const PairIdx q = pair_idx(A, c);
        const C2 K = ld_c2(A, q);
        const P2 X = ld_p2(x, q);
        const double2 bc = ld2(b + c), a = pair_rows(A, K, X, c);
        const double za = X.c.x + w * (bc.x - a.x) / K.dg.x, zb = X.c.y + w * (bc.y - a.y) / K.dg.y;
        st2(xn + c, za, zb);
        v[0] = make_double2(za * bc.x, zb * bc.y);
    }
    const int mx[1] = {0};
    block_reduce_store_pairs<1>(v, mx, partials);
}
__global__ __launch_bounds__(256) void k_mg_smooth2(PMat A, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ xn, double w) {
    FY_PAIR_LOOP(t, A.N) {
        const int c = t + A.c0;
        const PairIdx q = pair_idx(A, c);
        const C2 K = ld_c2(A, q);
        const P2 X = ld_p2(x, q);
        const double2 bc = ld2(b + c), a = pair_rows(A, K, X, c);
        st2(xn + c, X.c.x + w * (bc.x - a.x) / K.dg.x, X.c.y + w * (bc.y - a.y) / K.dg.y);
    }
}
// k_mg_smooth_prolong for a pair: the two cells share their parent and the parents of their y / z neighbours
__global__ __launch_bounds__(256) void k_mg_smooth_prolong2(PMat A, const double* __restrict__ b, const double* __restrict__ x, PMat C,
                                                            const double* __restrict__ xc, double* __restrict__ xn, double w) {
    FY_PAIR_LOOP(c, A.N) {
        const int i = c % A.nx, qq = c / A.nx, j = qq % A.ny, k = qq / A.ny;
        const int I = i >> 1, J = j >> 1, Kk = k >> 1;
        const int Im = max(i - 1, 0) >> 1, Ip = min(i + 2, A.nx - 1) >> 1, Jm = max(j - 1, 0) >> 1, Jp = min(j + 1, A.ny - 1) >> 1,
                  Km = max(k - 1, 0) >> 1, Kp = min(k + 1, A.nz - 1) >> 1;
        const double* e = xc + C.c0;
        const int rowc = C.nx * (J + C.ny * Kk);
        const PairIdx q = pair_idx(A, c);
        const C2 K = ld_c2(A, q);
        P2 V = ld_p2(x, q);
        const double ec = e[I + rowc], eym = e[I + C.nx * (Jm + C.ny * Kk)], eyp = e[I + C.nx * (Jp + C.ny * Kk)],
                     ezm = e[I + C.nx * (J + C.ny * Km)], ezp = e[I + C.nx * (J + C.ny * Kp)];
        V.c.x += ec; V.c.y += ec; V.ym.x += eym; V.ym.y += eym; V.yp.x += eyp; V.yp.y += eyp; V.zm.x += ezm; V.zm.y += ezm; V.zp.x += ezp; V.zp.y += ezp;
        V.xm += e[Im + rowc]; V.xp += e[Ip + rowc];
        const double2 bc = ld2(b + c), a = pair_rows(A, K, V, c);
        st2(xn + c, V.c.x + w * (bc.x - a.x) / K.dg.x, V.c.y + w * (bc.y - a.y) / K.dg.y);
    }
}
// k_mg_residual_restrict_tiled with a pair per lane: a block covers (2 PX) x TY x TZ fine cells (PX TY TZ = 256), a lane's two residuals are the first sum of its
// coarse cell's fold, the four lanes of a coarse cell meet in LDS.  The x-extent of the tile is chosen per level so that the blocks of a row are full
// (160 cells = 5 tiles of 32; with a fixed 128-cell tile the second block of a 160-cell row was a quarter full: 42.5 -> 37.3 us at 160^3)
template <int PX, int TY, int TZ>
__global__ __launch_bounds__(256) void k_mg_residual_restrict_tiled2(PMat A, const double* __restrict__ b, const double* __restrict__ x, PMat C,
                                                                     double* __restrict__ bc) {
    static_assert(PX * TY * TZ == 256 && TY % 2 == 0 && TZ % 2 == 0, "tile shape");
    __shared__ double r[TZ][TY][PX];
    const int tx = threadIdx.x % PX, ty = (threadIdx.x / PX) % TY, tz = threadIdx.x / (PX * TY);
    const int i = blockIdx.x * (2 * PX) + 2 * tx, j = blockIdx.y * TY + ty, k = blockIdx.z * TZ + tz;
    double v = 0.0;
    if (i < A.nx && j < A.ny && k < A.nz) {
        const int c = A.c0 + i + A.nx * (j + A.ny * k);
        const PairIdx q = pair_idx(A, c);
        const double2 bb = ld2(b + c), a = pair_rows(A, ld_c2(A, q), ld_p2(x, q), c);
        v = (bb.x - a.x) + (bb.y - a.y);
    }
    r[tz][ty][tx] = v;
    __syncthreads();
    constexpr int NC = PX * (TY / 2) * (TZ / 2);          // coarse cells of the tile
    if ((int)threadIdx.x < NC) {
        const int cx = threadIdx.x % PX, cy = (threadIdx.x / PX) % (TY / 2), cz = threadIdx.x / (PX * (TY / 2));
        const int I = blockIdx.x * PX + cx, J = blockIdx.y * (TY / 2) + cy, K = blockIdx.z * (TZ / 2) + cz;
        if (I < C.nx && J < C.ny && K < C.nz)
            bc[C.c0 + I + C.nx * (J + C.ny * K)] = (r[2 * cz][2 * cy][cx] + r[2 * cz][2 * cy + 1][cx]) + (r[2 * cz + 1][2 * cy][cx] + r[2 * cz + 1][2 * cy + 1][cx]);
    }
}

// The tail of the V-cycle -- every level with <= kMgTailCells cells (20^3 and below at 160^3) -- inside ONE 1024-thread workgroup:
// pre-smoothing, restriction, coarsest solve, prolongation and post-smoothing of up to kMgTailMax levels separated by
// __syncthreads() instead of kernel boundaries.  Those levels are launch/latency bound (4-15 us per launch for microseconds of
// work, ~24 launches per V-cycle); arithmetic and operation order are exactly those of the per-level kernels.
constexpr int kMgCoarseMax = 256;     // = the hierarchy's kMgCoarsest: the coarsest level a single wave solves out of LDS
struct MgTail {
    int n;                       // levels in the tail (level 0 of the tail is the finest of them)
    PMat A[kMgTailMax];
    double* x0[kMgTailMax];
    double* x1[kMgTailMax];
    double* b[kMgTailMax];
    const double* fac;           // banded Cholesky factor of the coarsest operator (k_mg_coarse_factor); null: Jacobi sweeps
    int cache_n, cache_off;      // cache_n > 0: the tail's first level (that many cells, not the coarsest) lives in LDS for the whole kernel, cache_off doubles into the dynamic LDS
};

// The coarsest level's exact solve from its banded Cholesky factor (k_mg_coarse_factor): fac = {N, bw, ok} as three doubles, then the band rows
// [N][bw + 1]: entry d of row i = L(i, i - d), d >= 1, and 1 / L(i, i) at d = 0.  ONE wave does both substitutions with the solution vector in
// REGISTERS (lane l holds rows l and l + 64: N <= 128): per column the owner's value is broadcast with a lane read, every lane in the band
// updates its row with one multiply-subtract -- no LDS round trip in the dependent chain, ~40 cycles per column, a few microseconds per solve
// (the 120 Jacobi sweeps they replace took ~60).  `Ls` (LDS, N * (bw + 1) doubles) must hold the factor; called by the wave tid < 64 only.
__device__ __forceinline__ double read_lane_f64(double v, int src_lane) {      // src_lane is wave-uniform: two v_readlane, no LDS crossbar
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}
// one sweep of eight-column groups over columns [ja, jb) (FWD: ascending, L y = b; else descending, L^T x = y) whose owners' values sit in v0 (LO) or v1.
// Per column the ONLY dependent chain is  lane read of the owner -> times 1 / L(j, j) -> times the row's coefficient -> subtract:  the coefficients come up front
// from clamped LDS addresses, zeroed where a row is not in the column's band (subtracting 0 * y leaves a row as it is), and an owner is NOT overwritten with its
// result inside the loop -- once its column has passed nothing changes it any more, so all owners are scaled by their 1 / L(i, i) in one operation afterwards
// (the same product the loop formed for the updates).  Round 5, measured with timing-only variants of the tail at C3: empty kernel 4.8 us, way down (first level in
// LDS) + 3.3, factor into LDS + 2.4, THIS solve + 30, way up + 5.5.  Factor entries fetched up front instead of in the chain: 51.4 -> 46.2 us; selects and the
// owner's write-back out of the chain (8 instructions per column, 4 of them dependent): 43.1.  What is left is ~27 ns per dependent operation of a lone wave.
// (DO0 / DO1: whether rows < 64 / rows >= 64 can lie in the band of the columns of this range at all -- a range that cannot touch one half skips its work)
template <bool FWD, bool LO, bool DO0, bool DO1>
__device__ __forceinline__ void band_columns(const double* Ls, int N, int bw, int ja, int jb, int lane, double& v0, double& v1) {
    if (jb <= ja) return;
    constexpr int U = 8;
    const int Wd = bw + 1, r0 = min(lane, N - 1), r1 = min(lane + 64, N - 1);
    for (int g = 0; g < jb - ja; g += U) {
        double dj[U], a0[U], a1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jr = FWD ? ja + g + u : jb - 1 - g - u;                 // the column (may run past the range: then its coefficients are zero)
            const bool live = FWD ? jr < jb : jr >= ja;
            const int j = min(max(jr, 0), N - 1);
            dj[u] = live ? Ls[(size_t)j * Wd] : 0.0;
            const int d0 = FWD ? lane - jr : jr - lane, d1 = FWD ? lane + 64 - jr : jr - (lane + 64);      // distance of my rows from the diagonal, on the side the sweep updates
            a0[u] = 0.0; a1[u] = 0.0;
            if (DO0) {
                const double c0 = FWD ? Ls[(size_t)r0 * Wd + min(max(d0, 0), bw)] : Ls[(size_t)j * Wd + min(max(d0, 0), bw)];
                a0[u] = (live && d0 >= 1 && d0 <= bw && lane < N) ? c0 : 0.0;
            }
            if (DO1) {
                const double c1 = FWD ? Ls[(size_t)r1 * Wd + min(max(d1, 0), bw)] : Ls[(size_t)j * Wd + min(max(d1, 0), bw)];
                a1[u] = (live && d1 >= 1 && d1 <= bw && lane + 64 < N) ? c1 : 0.0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jr = FWD ? ja + g + u : jb - 1 - g - u;
            const int j = min(max(jr, 0), N - 1);
            const double yj = read_lane_f64(LO ? v0 : v1, j & 63) * dj[u];
            if (DO0) v0 -= a0[u] * yj;
            if (DO1) v1 -= a1[u] * yj;
        }
    }
}
__device__ __forceinline__ void coarse_band_solve(const double* Ls, int N, int bw, const double* __restrict__ b, double* __restrict__ x, int lane) {
    const int Wd = bw + 1;
    double v0 = lane < N ? b[lane] : 0.0, v1 = lane + 64 < N ? b[lane + 64] : 0.0;
    const double dg0 = Ls[(size_t)min(lane, N - 1) * Wd], dg1 = Ls[(size_t)min(lane + 64, N - 1) * Wd];      // 1 / L(i, i) of my rows
    const int n0 = min(N, 64);
    // forward: L y = b.  Column j reaches rows j + 1 .. j + bw: rows >= 64 only from column 64 - bw on, rows < 64 only from columns < 64
    const int js = min(max(64 - bw, 0), n0);
    band_columns<true, true, true, false>(Ls, N, bw, 0, js, lane, v0, v1);
    band_columns<true, true, true, true>(Ls, N, bw, js, n0, lane, v0, v1);
    band_columns<true, false, false, true>(Ls, N, bw, 64, N, lane, v0, v1);
    v0 *= dg0; v1 *= dg1;
    // backward: L^T x = y.  Column j reaches rows j - bw .. j - 1: rows < 64 from columns < 64 + bw, rows >= 64 only from columns > 64
    const int jt = min(64 + bw, N);
    band_columns<false, false, false, true>(Ls, N, bw, jt, N, lane, v0, v1);
    band_columns<false, false, true, true>(Ls, N, bw, 64, jt, lane, v0, v1);
    band_columns<false, true, true, false>(Ls, N, bw, 0, n0, lane, v0, v1);
    v0 *= dg0; v1 *= dg1;
    if (lane < N) x[lane] = v0;
    if (lane + 64 < N) x[lane + 64] = v1;
}

__global__ __launch_bounds__(1024) void k_mg_tail(MgTail T, double w, int coarse_sweeps, MgWeights W) {
    const int tid = threadIdx.x;
    extern __shared__ double tail_lds[];
    // The tail's first level (<= kMgTailCells cells: one cell per thread) is touched by ~7 sweeps, each one global round trip and a barrier long: its operator,
    // right-hand side and both iterates are held in LDS instead (round 5; the same p_row on the same values, through generic pointers)
    const bool cached = T.cache_n > 0;
    double* const lc = tail_lds + T.cache_off;
    PMat AL = T.A[0];
    if (cached) {
        const int N0 = T.cache_n;
        AL.diag = lc; AL.ux = lc + N0; AL.uy = lc + 2 * N0; AL.uz = lc + 3 * N0;
        for (int c = tid; c < N0; c += 1024) {
            AL.diag[c] = T.A[0].diag[c]; AL.ux[c] = T.A[0].ux[c]; AL.uy[c] = T.A[0].uy[c]; AL.uz[c] = T.A[0].uz[c];
            lc[4 * N0 + c] = T.b[0][c];
        }
        __syncthreads();
    }
    const double* const lb = lc + 4 * T.cache_n;
    double* const lx0 = lc + 5 * T.cache_n;
    double* const lx1 = lc + 6 * T.cache_n;
    // ---- down: smooth_first, smooth, residual -> restricted rhs of the next level
    for (int l = 0; l + 1 < T.n; ++l) {
        const bool in_lds = cached && l == 0;
        const PMat A = in_lds ? AL : T.A[l];
        const double* b = in_lds ? lb : T.b[l];
        double* xa = in_lds ? lx0 : T.x0[l];
        double* xb = in_lds ? lx1 : T.x1[l];
        for (int c = tid; c < A.N; c += 1024) xa[c] = W.w[0] * b[c] / A.diag[c];
        __syncthreads();
        for (int s = 1; s < W.n; ++s) {                       // the iterate alternates between x0 and x1; W.n is even, so it ends in x1
            for (int c = tid; c < A.N; c += 1024) xb[c] = xa[c] + W.w[s] * (b[c] - p_row(A, xa, c)) / A.diag[c];
            __syncthreads();
            double* t = xa; xa = xb; xb = t;
        }
        { double* t = xa; xa = xb; xb = t; }                  // xb = the level's iterate (x1), xa the scratch (x0)
        const PMat Cc = T.A[l + 1];
        double* bc = T.b[l + 1];
        for (int cc = tid; cc < Cc.N; cc += 1024) {
            const int I = cc % Cc.nx, q = cc / Cc.nx, J = q % Cc.ny, K = q / Cc.ny;
            double acc = 0;
            for (int dk = 0; dk < 2; ++dk) {
                const int k = 2 * K + dk; if (k >= A.nz) break;
                for (int dj = 0; dj < 2; ++dj) {
                    const int j = 2 * J + dj; if (j >= A.ny) break;
                    for (int di = 0; di < 2; ++di) {
                        const int i = 2 * I + di; if (i >= A.nx) break;
                        const int c = i + A.nx * (j + A.ny * k);
                        acc += b[c] - p_row(A, xb, c);
                    }
                }
            }
            bc[cc] = acc;
        }
        __syncthreads();
    }
    // ---- coarsest level (<= kMgCoarseMax cells): damped-Jacobi sweeps from a zero guess, result in x0.  ONE wave does all sweeps out of
    // LDS: 40 sweeps behind a 16-wave workgroup barrier each cost ~48 of this kernel's 55 us; inside a single wave the LDS operations
    // are ordered by the hardware and no barrier is needed at all.  Same arithmetic in the same order as the multi-wave loop.
    {
        const int l = T.n - 1;
        const PMat A = T.A[l];
        const double* b = T.b[l];
        __shared__ double c_dg[kMgCoarseMax], c_ux[kMgCoarseMax], c_uy[kMgCoarseMax], c_uz[kMgCoarseMax], c_b[kMgCoarseMax];
        __shared__ double c_x[2][kMgCoarseMax];
        if (T.fac && A.N <= kMgDirectMax && T.fac[2] == 1.0) {        // (uniform) the level's exact solve from its banded Cholesky factor
            double* const fac_lds = tail_lds;
            const int bw = (int)T.fac[1], cnt = A.N * (bw + 1);
            for (int q = tid; q < cnt; q += 1024) fac_lds[q] = T.fac[3 + q];
            __syncthreads();
            if (tid < 64) coarse_band_solve(fac_lds, A.N, bw, b, T.x0[l], tid);
            __syncthreads();
        } else if (A.N <= kMgCoarseMax) {
            if (tid < 64) {
                const int N = A.N, sy = A.nx, sz = A.nx * A.ny;
                for (int c = tid; c < N; c += 64) {
                    c_dg[c] = A.diag[c]; c_ux[c] = A.ux[c]; c_uy[c] = A.uy[c]; c_uz[c] = A.uz[c]; c_b[c] = b[c];
                    c_x[0][c] = w * b[c] / A.diag[c];
                }
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
                int cur = 0;
                constexpr int R = kMgCoarseMax / 64;                    // cells per lane, processed together: their dependent FP64 chains
                for (int s = 1; s < coarse_sweeps; ++s) {               // (7 multiply-subtracts + a division) overlap instead of queueing
                    const double* xc = c_x[cur];
                    double* xn = c_x[cur ^ 1];
                    double a[R], xo[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int c = tid + 64 * r;
                        a[r] = 0.0; xo[r] = 0.0;
                        if (c < N) {
                            // p_row in the same order; the guarded terms are loaded from clamped addresses and selected, so the seven
                            // LDS reads go out together instead of one per divergent branch (that serialisation was the sweep's cost)
                            const int xm = max(c - 1, 0), xp = min(c + 1, N - 1), ym = max(c - sy, 0), yp = min(c + sy, N - 1);
                            const int zm = max(c - sz, 0), zp = min(c + sz, N - 1);
                            xo[r] = xc[c];
                            const double t0 = c_ux[xm] * xc[xm], t1 = c_ux[c] * xc[xp], t2 = c_uy[ym] * xc[ym], t3 = c_uy[c] * xc[yp];
                            const double t4 = c_uz[zm] * xc[zm], t5 = c_uz[c] * xc[zp];
                            double v = c_dg[c] * xo[r];
                            v = (c >= 1) ? v - t0 : v;
                            v = (c + 1 < N) ? v - t1 : v;
                            v = (c >= sy) ? v - t2 : v;
                            v = (c + sy < N) ? v - t3 : v;
                            v = (c >= sz) ? v - t4 : v;
                            v = (c + sz < N) ? v - t5 : v;
                            a[r] = v;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int c = tid + 64 * r;
                        if (c < N) xn[c] = xo[r] + w * (c_b[c] - a[r]) / c_dg[c];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
                    cur ^= 1;
                }
                for (int c = tid; c < N; c += 64) T.x0[l][c] = c_x[cur][c];
            }
            __syncthreads();
        } else {
            double* cur = T.x0[l];
            double* nxt = T.x1[l];
            for (int c = tid; c < A.N; c += 1024) cur[c] = w * b[c] / A.diag[c];
            __syncthreads();
            for (int s = 1; s < coarse_sweeps; ++s) {
                for (int c = tid; c < A.N; c += 1024) nxt[c] = cur[c] + w * (b[c] - p_row(A, cur, c)) / A.diag[c];
                __syncthreads();
                double* t = cur; cur = nxt; nxt = t;
            }
            if (cur != T.x0[l]) { for (int c = tid; c < A.N; c += 1024) T.x0[l][c] = cur[c]; __syncthreads(); }
        }
    }
    // ---- up: prolongation + two post-smoothing sweeps; a level's result ends in x1 (the coarsest's in x0)
    for (int l = T.n - 2; l >= 0; --l) {
        const bool in_lds = cached && l == 0;
        const PMat A = in_lds ? AL : T.A[l];
        const PMat Cc = T.A[l + 1];
        const double* b = in_lds ? lb : T.b[l];
        const double* xc = (l + 1 == T.n - 1) ? T.x0[l + 1] : T.x1[l + 1];
        double* xb = in_lds ? lx1 : T.x1[l];
        double* xa = in_lds ? lx0 : T.x0[l];
        for (int c = tid; c < A.N; c += 1024) {
            const int i = c % A.nx, q = c / A.nx, j = q % A.ny, k = q / A.ny;
            xb[c] += xc[(i >> 1) + Cc.nx * ((j >> 1) + Cc.ny * (k >> 1))];
        }
        __syncthreads();
        for (int s = W.n - 1; s >= 0; --s) {                  // post-smoothing: the weights in reverse; W.n even: the result is back in x1
            for (int c = tid; c < A.N; c += 1024) xa[c] = xb[c] + W.w[s] * (b[c] - p_row(A, xb, c)) / A.diag[c];
            __syncthreads();
            double* t = xa; xa = xb; xb = t;
        }
    }
    if (cached) for (int c = tid; c < T.cache_n; c += 1024) T.x1[0][c] = lx1[c];      // the level above prolongs from x1
}

__global__ __launch_bounds__(256) void k_mg_prolong_add(PMat A, double* __restrict__ x, PMat C, const double* __restrict__ xc) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= A.N) return;
    const int i = t % A.nx, q = t / A.nx, j = q % A.ny, k = q / A.ny;
    x[t + A.c0] += xc[C.c0 + (i >> 1) + C.nx * ((j >> 1) + C.ny * (k >> 1))];
}

// the same over a range of z-planes that starts `kofs` planes away from the level's first owned plane (negative: ghost planes below it; the
// communication-avoiding V-cycle of a z-slab, fv_pressure.cpp): C.c0 is the coarse cell under the fine level's first OWNED cell
__global__ __launch_bounds__(256) void k_mg_prolong_add_planes(PMat A, int kofs, double* __restrict__ x, PMat C, const double* __restrict__ xc) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= A.N) return;
    const int i = t % A.nx, q = t / A.nx, j = q % A.ny, k = q / A.ny + kofs;
    const int K = k >= 0 ? (k >> 1) : -((1 - k) >> 1);                  // floor(k / 2)
    x[t + A.c0] += xc[C.c0 + (i >> 1) + C.nx * ((j >> 1) + C.ny * K)];
}

// coarsest level (N <= 1024, never distributed: c0 = 0): all sweeps inside one workgroup
// The coarsest operator (<= kMgDirectMax = 128 cells) only changes when the pressure matrix is assembled, and every V-cycle in between solves
// with it: so it is FACTORED once per assembly -- banded Cholesky A = L L^T in LDS (band width = the operator's z stride, 25 for the 5^3
// level of C3: N bw^2 = 78 k multiply-adds; a dense 125^3 inversion was built first and cost 0.5 ms, LDS-bandwidth bound), one workgroup, one barrier per column -- and a V-cycle's coarse solve is two banded substitutions in one wave (coarse_band_solve) instead of
// 120 Jacobi sweeps that left the level's smoothest modes partly in.  The operator is a symmetric positive definite M-matrix (the
// reference cell or a fixed-value patch makes it non-singular); fac[2] = 0 if a pivot is not positive and finite (the sweeps then stand in).
template <int Q>
__global__ __launch_bounds__(1024) void k_mg_coarse_factor(PMat A, int bw, double* __restrict__ fac) {
    extern __shared__ double B[];                  // [N][bw + 1]: B[i][d] = A(i, i - d), overwritten by L
    __shared__ int bad;
    const int N = A.N, tid = threadIdx.x, NT = (int)blockDim.x, Wd = bw + 1, sy = A.nx, sz = A.nx * A.ny;
    if (tid == 0) bad = 0;
    for (int e = tid; e < N * Wd; e += NT) B[e] = 0.0;
    __syncthreads();
    for (int c = tid; c < N; c += NT) {          // the lower half of row c of p_row (zero coefficients at the walls); += : strides coincide on flat grids
        double* row = B + (size_t)c * Wd;
        row[0] += A.diag[c];
        if (c >= 1) row[1] -= A.ux[c - 1];
        if (A.ny > 1 && c >= sy && sy <= bw) row[sy] -= A.uy[c - sy];
        if (A.nz > 1 && c >= sz && sz <= bw) row[sz] -= A.uz[c - sz];
    }
    __syncthreads();
    // Right-looking elimination with the column entries left UNSCALED (they hold L(i, j) L(j, j)): the trailing update of column j,
    // A(i, k) -= A(i, j) A(k, j) / A(j, j) for j < k <= i <= j + m, reads column j and writes columns > j only -- ONE barrier per column --
    // and the division by L(j, j) is applied to every entry at the end.  A thread owns fixed positions (a, b) of the bw x bw update window
    // (bw <= 64: at most four), so there is no index arithmetic in the loop.
    // (round 5: only the lower triangle b <= a of the window does anything -- its bw (bw + 1) / 2 positions are dealt out, Q per thread; see the launcher)
    int ua[Q], ub[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int e = tid + NT * q;
        int a = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while ((a + 1) * (a + 2) / 2 <= e) ++a;             // (guards the rounding of the square root)
        while (a * (a + 1) / 2 > e) --a;
        ua[q] = a; ub[q] = e - a * (a + 1) / 2;
    }
    for (int j = 0; j < N; ++j) {
        const double d = B[(size_t)j * Wd];
        if (!(d > 0.0) || !(d < 1e300)) { if (tid == 0) bad = 1; break; }      // (uniform: every thread reads the same value)
        const double id = 1.0 / d;
        const int m = min(bw, N - 1 - j);           // rows below the diagonal in this column
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int a = ua[q], b = ub[q];
            if (a < m) B[(size_t)(j + 1 + a) * Wd + (a - b)] -= (B[(size_t)(j + 1 + a) * Wd + 1 + a] * B[(size_t)(j + 1 + b) * Wd + 1 + b]) * id;
        }
        __syncthreads();
    }
    __syncthreads();
    if (!bad) {
        for (int i = tid; i < N; i += NT) fac[3 + (size_t)i * Wd] = 1.0 / sqrt(B[(size_t)i * Wd]);      // 1 / L(i, i): the substitutions multiply
        for (int e = tid; e < N * Wd; e += NT) {
            const int i = e / Wd, dd = e - i * Wd;
            if (dd >= 1) fac[3 + e] = dd <= i ? B[e] / sqrt(B[(size_t)(i - dd) * Wd]) : 0.0;                   // L(i, i - dd) = stored / L(i - dd, i - dd)
        }
    }
    if (tid == 0) { fac[0] = (double)N; fac[1] = (double)bw; fac[2] = bad ? 0.0 : 1.0; }
}

__global__ __launch_bounds__(1024) void k_mg_coarse_solve(PMat A, const double* __restrict__ b, double* __restrict__ x, double* __restrict__ tmp,
                                                          int sweeps, double w, const double* __restrict__ fac) {
    if (fac && A.N <= kMgDirectMax && fac[2] == 1.0) {        // (uniform)
        extern __shared__ double fac_lds[];
        const int bw = (int)fac[1], cnt = A.N * (bw + 1);
        for (int q = threadIdx.x; q < cnt; q += 1024) fac_lds[q] = fac[3 + q];
        __syncthreads();
        if (threadIdx.x < 64) coarse_band_solve(fac_lds, A.N, bw, b, x, threadIdx.x);
        return;
    }
    const int c = threadIdx.x;
    const bool act = c < A.N;
    double* cur = x;
    double* nxt = tmp;
    if (act) cur[c] = w * b[c] / A.diag[c];
    __syncthreads();
    for (int s = 1; s < sweeps; ++s) {
        if (act) nxt[c] = cur[c] + w * (b[c] - p_row(A, cur, c)) / A.diag[c];
        __syncthreads();
        double* t = cur; cur = nxt; nxt = t;
    }
    if (cur != x) { if (act) x[c] = cur[c]; }
}

// GeometricField::relax(alpha), pEqn.H:41: p = prevIter + alpha (p - prevIter)
__global__ __launch_bounds__(256) void k_relax_field(double* __restrict__ x, const double* __restrict__ prev, double alpha, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = prev[i] + alpha * (x[i] - prev[i]);
}

__global__ __launch_bounds__(256) void k_copy(double* __restrict__ dst, const double* __restrict__ src, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_add(double* __restrict__ y, const double* __restrict__ x, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] += x[i];
}

// launch grids of the cell sweeps: the whole owned range, or the block window the geometry names (FvGeo::win_nblk: a range of z-planes, for the
// sweeps that run beside a halo exchange)
inline int fv_grid(const FvGeo& g) { return g.win_nblk > 0 ? g.win_nblk : div_up(g.Nc, 256); }
inline int fv_red_grid(const FvGeo& g) { return g.win_nblk > 0 ? g.win_nblk : red_blocks(g.Nc); }

#define FY_LAUNCH_CHECK()                                                                                     \
    do {                                                                                                      \
        hipError_t _e = hipGetLastError();                                                                    \
        if (_e != hipSuccess) return fail(FY_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

}  // namespace

int launch_reduce_finalize(hipStream_t s, const double* partials, int n_cells, int nslots, const int* ops, double* out, unsigned long long* flag,
                           unsigned long long seq) {
    hipLaunchKernelGGL(k_reduce_finalize, dim3(nslots), dim3(1024), 0, s, partials, red_blocks(n_cells), ops, out, flag, seq);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_flux_of(hipStream_t s, FvGeo g, const double* F, Face3 out) {
    hipLaunchKernelGGL(k_flux_of<0>, dim3(div_up(fv_fsize(g, 0), 256)), dim3(256), 0, s, g, F, out.a[0]);
    hipLaunchKernelGGL(k_flux_of<1>, dim3(div_up(fv_fsize(g, 1), 256)), dim3(256), 0, s, g, F, out.a[1]);
    hipLaunchKernelGGL(k_flux_of<2>, dim3(div_up(fv_fsize(g, 2), 256)), dim3(256), 0, s, g, F, out.a[2]);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_courant(hipStream_t s, FvGeo g, CFace3 phi, double* partials) {
    hipLaunchKernelGGL(k_courant, dim3(fv_red_grid(g)), dim3(256), 0, s, g, phi, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_pre_coupling(hipStream_t s, FvGeo g, const double* U, const double* p, const double* alpha, CFace3 psn, double* vGrad,
                        double* gradP, double* divT, double* Gout, int write_vgrad, int write_pfields, CFace3 phi, double* ddtU, double* Uold_out,
                        double* cellrec, double rec_nu, double rec_rhoF, Face3 dcorr) {
    hipLaunchKernelGGL(k_pre_coupling, dim3(fv_grid(g)), dim3(256), 0, s, g, U, p, alpha, psn, vGrad, gradP, divT, Gout, write_vgrad, write_pfields,
                       phi, ddtU, Uold_out, cellrec, 2.0 * rec_nu, rec_rhoF, dcorr);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_interp_alpha(hipStream_t s, FvGeo g, const double* alpha, Face3 af) {
    hipLaunchKernelGGL(k_interp_alpha_cells, dim3(fv_grid(g)), dim3(256), 0, s, g, alpha, af);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_div_G(hipStream_t s, FvGeo g, const double* G, double* divG) {
    hipLaunchKernelGGL(k_div_G, dim3(fv_grid(g)), dim3(256), 0, s, g, G, divG);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_assemble_turb(hipStream_t s, FvGeo g, TurbEqn e, const double* k, const double* eps, const double* alpha, CFace3 alphaf, CFace3 phi,
                         const double* vGrad, const double* U, Mom7 M, double* b3, double* x3) {
    hipLaunchKernelGGL(k_assemble_turb, dim3(fv_grid(g)), dim3(256), 0, s, g, e, k, eps, alpha, alphaf, phi, vGrad, U, M, b3, x3);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_turb_finish(hipStream_t s, FvGeo g, TurbEqn e, const double* x3, double* X, int nut_mode, double cmu, const double* eps, double* nut) {
    hipLaunchKernelGGL(k_turb_finish, dim3(fv_grid(g)), dim3(256), 0, s, g, e, x3, X, nut_mode, cmu, eps, nut);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_smagorinsky_nut(hipStream_t s, FvGeo g, const double* vGrad, double ck, double ce, double delta, double* nut) {
    hipLaunchKernelGGL(k_smagorinsky_nut, dim3(fv_grid(g)), dim3(256), 0, s, g, vGrad, ck, ce, delta, nut);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_assemble_momentum(hipStream_t s, FvGeo g, const double* U, const double* Uold, const double* alpha, const double* alphaOld,
                             CFace3 alphaf, CFace3 phi, const double* uSource, const double* uSourceDrag, const double* divG, const double* vGrad,
                             Mom7 M, double* src, double* rAU) {
    if (g.upwind >= 3) hipLaunchKernelGGL(k_assemble_momentum<true>, dim3(fv_grid(g)), dim3(256), 0, s, g, U, Uold, alpha, alphaOld, alphaf, phi, uSource,
                                          uSourceDrag, divG, vGrad, M, src, rAU);
    else hipLaunchKernelGGL(k_assemble_momentum<false>, dim3(fv_grid(g)), dim3(256), 0, s, g, U, Uold, alpha, alphaOld, alphaf, phi, uSource,
                            uSourceDrag, divG, vGrad, M, src, rAU);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_grad_magsqr(hipStream_t s, FvGeo g, const double* U, double* gradL) {
    hipLaunchKernelGGL(k_grad_magsqr, dim3(fv_grid(g)), dim3(256), 0, s, g, U, gradL);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_interp_rAU(hipStream_t s, FvGeo g, const double* rAU, Face3 rf) {
    hipLaunchKernelGGL(k_interp_rAU<0>, dim3(div_up(fv_fsize(g, 0), 256)), dim3(256), 0, s, g, rAU, rf.a[0]);
    hipLaunchKernelGGL(k_interp_rAU<1>, dim3(div_up(fv_fsize(g, 1), 256)), dim3(256), 0, s, g, rAU, rf.a[1]);
    hipLaunchKernelGGL(k_interp_rAU<2>, dim3(div_up(fv_fsize(g, 2), 256)), dim3(256), 0, s, g, rAU, rf.a[2]);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_bmom(hipStream_t s, FvGeo g, const double* src, const double* p, CFace3 psn, CFace3 phiForces, CFace3 rAUf, double* bmom) {
    hipLaunchKernelGGL(k_bmom, dim3(fv_grid(g)), dim3(256), 0, s, g, src, p, psn, phiForces, rAUf, bmom);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_bmom_faces(hipStream_t s, FvGeo g, const double* rAU, const double* uSource, const double* src, const double* p, CFace3 psn, Face3 rAUf_out,
                      Face3 phiForces, double* bmom) {
    hipLaunchKernelGGL(k_bmom_faces, dim3(fv_grid(g)), dim3(256), 0, s, g, rAU, uSource, src, p, psn, rAUf_out, phiForces, bmom);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mom_pass(hipStream_t s, FvGeo g, Mom7 M, const double* b, const double* x, double* xn, const double* xsum, double n_glob, double* partials,
                    const double* hsrc, const double* rAU, double* HbyA) {
    if (HbyA) hipLaunchKernelGGL(k_mom_pass<true>, dim3(fv_red_grid(g)), dim3(256), 0, s, g, M, b, x, xn, xsum, n_glob, partials, hsrc, rAU, HbyA);
    else hipLaunchKernelGGL(k_mom_pass<false>, dim3(fv_red_grid(g)), dim3(256), 0, s, g, M, b, x, xn, xsum, n_glob, partials, hsrc, rAU, HbyA);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_sum3(hipStream_t s, const double* x, int n, double* partials) {
    hipLaunchKernelGGL(k_sum3, dim3(red_blocks(n)), dim3(256), 0, s, x, n, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_HbyA(hipStream_t s, FvGeo g, Mom7 M, const double* src, const double* U, const double* rAU, double* HbyA) {
    hipLaunchKernelGGL(k_HbyA, dim3(fv_grid(g)), dim3(256), 0, s, g, M, src, U, rAU, HbyA);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_phiHbyA(hipStream_t s, FvGeo g, const double* HbyA, const double* U, const double* Uold, CFace3 phiOld, CFace3 rAUf,
                   CFace3 alphaf, CFace3 phiForces, Face3 out, Face3 psn, Face3 ddtc, int keep) {
    if (keep == 1) hipLaunchKernelGGL(k_phiHbyA_cells<1>, dim3(fv_grid(g)), dim3(256), 0, s, g, HbyA, U, Uold, phiOld, rAUf, alphaf, phiForces, out, psn, ddtc);
    else if (keep == 2) hipLaunchKernelGGL(k_phiHbyA_cells<2>, dim3(fv_grid(g)), dim3(256), 0, s, g, HbyA, U, Uold, phiOld, rAUf, alphaf, phiForces, out, psn, ddtc);
    else hipLaunchKernelGGL(k_phiHbyA_cells<0>, dim3(fv_grid(g)), dim3(256), 0, s, g, HbyA, U, Uold, phiOld, rAUf, alphaf, phiForces, out, psn, ddtc);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_adjust_phi_sums(hipStream_t s, FvGeo g, CFace3 phiHbyA, CFace3 phiForces, double* partials) {
    hipLaunchKernelGGL(k_adjust_phi_sums, dim3(fv_red_grid(g)), dim3(256), 0, s, g, phiHbyA, phiForces, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_adjust_phi_apply(hipStream_t s, FvGeo g, const double* sums, Face3 phiHbyA, CFace3 phiForces, CFace3 rAUf, const double* U, Face3 psn, int* err) {
    const size_t nbf = 2 * ((size_t)g.ny * g.nz + (size_t)g.nx * g.nz + (size_t)g.nx * g.ny);
    hipLaunchKernelGGL(k_adjust_phi_apply, dim3(div_up(nbf, 256)), dim3(256), 0, s, g, sums, phiHbyA, phiForces, rAUf, U, psn, err);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_rAUf_phi_forces(hipStream_t s, FvGeo g, const double* rAU, const double* uSource, Face3 rf, Face3 out) {
    hipLaunchKernelGGL(k_rAUf_phi_forces_cells, dim3(fv_grid(g)), dim3(256), 0, s, g, rAU, uSource, rf, out);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_assemble_pressure(hipStream_t s, FvGeo g, CFace3 phiHbyA, CFace3 rAUf, CFace3 alphaf, CFace3 psn, const double* alpha,
                             const double* alphaOld, PMat A, double* rhs, bool matrix) {
    if (matrix) hipLaunchKernelGGL(k_assemble_pressure<true>, dim3(fv_grid(g)), dim3(256), 0, s, g, phiHbyA, rAUf, alphaf, psn, alpha, alphaOld, A, rhs);
    else hipLaunchKernelGGL(k_assemble_pressure<false>, dim3(fv_grid(g)), dim3(256), 0, s, g, phiHbyA, rAUf, alphaf, psn, alpha, alphaOld, A, rhs);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_flux_correct(hipStream_t s, FvGeo g, const double* p, CFace3 phiHbyA, CFace3 rAUf, CFace3 alphaf, CFace3 psn, CFace3 phiForces, Face3 pflux, Face3 phi) {
    hipLaunchKernelGGL(k_flux_correct_cells, dim3(fv_grid(g)), dim3(256), 0, s, g, p, phiHbyA, rAUf, alphaf, psn, phiForces, pflux, phi);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_cont_err(hipStream_t s, FvGeo g, CFace3 phi, CFace3 alphaf, const double* alpha, const double* alphaOld, double* partials) {
    hipLaunchKernelGGL(k_cont_err, dim3(fv_red_grid(g)), dim3(256), 0, s, g, phi, alphaf, alpha, alphaOld, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_U_correct(hipStream_t s, FvGeo g, const double* HbyA, const double* rAU, const double* p, CFace3 psn, CFace3 phiForces,
                     CFace3 pflux, CFace3 alphaf, CFace3 rAUf, double* U) {
    hipLaunchKernelGGL(k_U_correct<false>, dim3(fv_red_grid(g)), dim3(256), 0, s, g, HbyA, rAU, p, psn, phiForces, pflux, alphaf, rAUf, U, CFace3{}, nullptr, nullptr, nullptr);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_U_correct_diag(hipStream_t s, FvGeo g, const double* HbyA, const double* rAU, const double* p, CFace3 psn, CFace3 phiForces,
                          CFace3 pflux, CFace3 alphaf, CFace3 rAUf, double* U, CFace3 phi, const double* alpha, const double* alphaOld, double* partials) {
    hipLaunchKernelGGL(k_U_correct<true>, dim3(fv_red_grid(g)), dim3(256), 0, s, g, HbyA, rAU, p, psn, phiForces, pflux, alphaf, rAUf, U, phi, alpha, alphaOld, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_corr_back(hipStream_t s, FvGeo g, const double* p, CFace3 phiHbyA, CFace3 rAUf, CFace3 alphaf, CFace3 psn, CFace3 phiForces, Face3 phi,
                     const double* HbyA, const double* rAU, double* U, const double* alpha, const double* alphaOld, double* partials, bool faces_from_cells) {
    const FaceSrc rs{rAUf, faces_from_cells ? rAU : nullptr}, as{alphaf, faces_from_cells ? alpha : nullptr};
    const dim3 grid(fv_red_grid(g)), blk(256);
#define FY_BACK(DG, FC) hipLaunchKernelGGL((k_corr_back<DG, FC>), grid, blk, 0, s, g, p, phiHbyA, rs, as, psn, phiForces, phi, HbyA, rAU, U, alpha, alphaOld, partials)
    if (partials) { if (faces_from_cells) FY_BACK(true, true); else FY_BACK(true, false); }
    else { if (faces_from_cells) FY_BACK(false, true); else FY_BACK(false, false); }
#undef FY_BACK
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_corr_front(hipStream_t s, FvGeo g, const double* HbyA, const double* U, CFace3 dcorr, CFace3 rAUf, CFace3 alphaf, CFace3 phiForces, Face3 phiHbyA,
                      Face3 psn, const double* rAU, const double* alpha, const double* alphaOld, PMat A, double* rhs, bool store_A, const double* x,
                      const double* xsum_dev, double xsum_val, double inv_n, double* res, double* partials, bool faces_from_cells) {
    const FaceSrc rs{rAUf, faces_from_cells ? rAU : nullptr}, as{alphaf, faces_from_cells ? alpha : nullptr};
    const dim3 grid(fv_red_grid(g)), blk(256);
#define FY_FRONT(SA, FC) hipLaunchKernelGGL((k_corr_front<SA, FC>), grid, blk, 0, s, g, HbyA, U, dcorr, rs, as, phiForces, phiHbyA, psn, rAU, alpha, alphaOld, A, rhs, \
                                            x, xsum_dev, xsum_val, inv_n, res, partials)
    if (store_A) { if (faces_from_cells) FY_FRONT(true, true); else FY_FRONT(true, false); }
    else { if (faces_from_cells) FY_FRONT(false, true); else FY_FRONT(false, false); }
#undef FY_FRONT
    FY_LAUNCH_CHECK();
    return FY_OK;
}

// two cells per thread where the level allows it (see "two cells per thread" above); FOAMYADE_NO_PAIRS=1: the one-cell kernels everywhere (A/B switch, same bits)
static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static bool pairs_ok(const PMat& A) {
    return !pairs_disabled() && A.N >= 2 && A.nx % 2 == 0 && A.c0 % 2 == 0 && A.N % 2 == 0 && A.ntot % 2 == 0 && al16(A.diag) && al16(A.ux) && al16(A.uy) && al16(A.uz);
}
static bool pairs_ok(int n, int c0) {
    return !pairs_disabled() && n >= 2 && n % 2 == 0 && c0 % 2 == 0;
}

int launch_p_apply(hipStream_t s, PMat A, const double* x, double* y) {
    if (pairs_ok(A) && al16(x) && al16(y)) {
        hipLaunchKernelGGL(k_p_apply2, dim3(div_up(A.N, 512)), dim3(256), 0, s, A, x, y);
        FY_LAUNCH_CHECK();
        return FY_OK;
    }
    hipLaunchKernelGGL(k_p_apply, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, x, y);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_p_apply_dot(hipStream_t s, PMat A, const double* x, const double* r, double* y, double* partials) {
    if (pairs_ok(A) && al16(x) && al16(y) && al16(r)) {
        if (r) hipLaunchKernelGGL(k_p_apply_dot2<true>, dim3(red_blocks(A.N)), dim3(128), 0, s, A, x, r, y, partials);
        else hipLaunchKernelGGL(k_p_apply_dot2<false>, dim3(red_blocks(A.N)), dim3(128), 0, s, A, x, r, y, partials);
    }
    else if (r) hipLaunchKernelGGL(k_p_apply_dot<true>, dim3(red_blocks(A.N)), dim3(256), 0, s, A, x, r, y, partials);
    else hipLaunchKernelGGL(k_p_apply_dot<false>, dim3(red_blocks(A.N)), dim3(256), 0, s, A, x, r, y, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_p_init(hipStream_t s, PMat A, const double* b, const double* x, const double* xsum_dev, double xsum_val, double inv_n, double* r, double* partials) {
    hipLaunchKernelGGL(k_p_init, dim3(red_blocks(A.N)), dim3(256), 0, s, A, b, x, xsum_dev, xsum_val, inv_n, r, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_dot(hipStream_t s, int n, int c0, const double* a, const double* b, double* partials) {
    hipLaunchKernelGGL(k_dot, dim3(red_blocks(n)), dim3(256), 0, s, n, c0, a, b, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_pcg_cg_update(hipStream_t s, int n, int c0, const double* u, const double* w, double* p, double* sv, double* x, double* r, double* sc, int it, double* partials) {
    if (pairs_ok(n, c0) && al16(u) && al16(w) && al16(p) && al16(sv) && al16(x) && al16(r)) {
        if (it == 0) hipLaunchKernelGGL(k_pcg_cg_update2<true>, dim3(red_blocks(n)), dim3(128), 0, s, n, c0, u, w, p, sv, x, r, sc, it, partials);
        else hipLaunchKernelGGL(k_pcg_cg_update2<false>, dim3(red_blocks(n)), dim3(128), 0, s, n, c0, u, w, p, sv, x, r, sc, it, partials);
    }
    else if (it == 0) hipLaunchKernelGGL(k_pcg_cg_update<true>, dim3(red_blocks(n)), dim3(256), 0, s, n, c0, u, w, p, sv, x, r, sc, it, partials);
    else hipLaunchKernelGGL(k_pcg_cg_update<false>, dim3(red_blocks(n)), dim3(256), 0, s, n, c0, u, w, p, sv, x, r, sc, it, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_jacobi_precond(hipStream_t s, PMat A, const double* r, double* z) {
    hipLaunchKernelGGL(k_jacobi_precond, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, r, z);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_ref_term(hipStream_t s, PMat A0, int ref_local, double* out) {
    hipLaunchKernelGGL(k_mg_ref_term, dim3(1), dim3(64), 0, s, A0, ref_local, out);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_coarsen(hipStream_t s, PMat F, PMat C, int ref_c, const double* ref_term) {
    hipLaunchKernelGGL(k_mg_coarsen, dim3(div_up(C.N, 256)), dim3(256), 0, s, F, C, ref_term ? ref_c : -1, ref_term);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_smooth_first(hipStream_t s, PMat A, const double* b, double* x, double w) {
    hipLaunchKernelGGL(k_mg_smooth_first, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, b, x, w);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_smooth_two_from_zero(hipStream_t s, PMat A, const double* b, double* xn, double w, double w2) {
    if (pairs_ok(A) && al16(b) && al16(xn)) hipLaunchKernelGGL(k_mg_smooth_two_from_zero2, dim3(div_up(A.N, 512)), dim3(256), 0, s, A, b, xn, w, w2);
    else hipLaunchKernelGGL(k_mg_smooth_two_from_zero, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, b, xn, w, w2);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_smooth_dot(hipStream_t s, PMat A, const double* b, const double* x, double* xn, double w, double* partials) {
    if (pairs_ok(A) && al16(b) && al16(x) && al16(xn)) hipLaunchKernelGGL(k_mg_smooth_dot2, dim3(red_blocks(A.N)), dim3(128), 0, s, A, b, x, xn, w, partials);
    else hipLaunchKernelGGL(k_mg_smooth_dot, dim3(red_blocks(A.N)), dim3(256), 0, s, A, b, x, xn, w, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_smooth(hipStream_t s, PMat A, const double* b, const double* x, double* xn, double w) {
    if (pairs_ok(A) && al16(b) && al16(x) && al16(xn)) hipLaunchKernelGGL(k_mg_smooth2, dim3(div_up(A.N, 512)), dim3(256), 0, s, A, b, x, xn, w);
    else hipLaunchKernelGGL(k_mg_smooth, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, b, x, xn, w);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

// band width of a level's operator = its largest neighbour stride
static int band_width(const PMat& A) { return A.nz > 1 ? A.nx * A.ny : (A.ny > 1 ? A.nx : 1); }
static size_t fac_lds_bytes(int N, int bw) { return (size_t)N * (size_t)(bw + 1) * sizeof(double); }
static int allow_big_lds(const void* fn) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fac_lds_bytes(kMgDirectMax, kMgDirectBand)) == hipSuccess ? FY_OK
           : fail(FY_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
}

int mg_coarse_factor_doubles(PMat A) { return 3 + A.N * (band_width(A) + 1); }
bool mg_coarse_direct_ok(PMat A) { return A.c0 == 0 && A.N <= kMgDirectMax && band_width(A) <= kMgDirectBand; }

int launch_mg_coarse_factor(hipStream_t s, PMat A, double* fac) {
    if (!(mg_coarse_direct_ok)(A)) return fail(FY_ERR_INVALID, "direct coarse solve: level of %d cells, band %d (ghost offset %d)", A.N, band_width(A), A.c0);
    static bool attr_set = false;
    if (!attr_set) {
        FY_TRY(allow_big_lds(reinterpret_cast<const void*>(k_mg_coarse_factor<1>)));
        FY_TRY(allow_big_lds(reinterpret_cast<const void*>(k_mg_coarse_factor<2>)));
        FY_TRY(allow_big_lds(reinterpret_cast<const void*>(k_mg_coarse_factor<3>)));
        attr_set = true;
    }
    const int bw = band_width(A);
    // the update window's lower triangle dealt out Q positions per thread.  The kernel is one barrier and one dependent LDS round trip per column: fewer
    // positions per thread shorten the round trip, fewer waves the barrier (measured at the 5^3 level, band 25, 325 positions: 1024 threads x 4 positions of the
    // full window 72 us, 256 x 4 of the full window 51; of the triangle: 64 x 6 81, 128 x 3 54, 192 x 2 44, 384 x 1 39 us)
    const int tri = bw * (bw + 1) / 2;
    static const int force_q = [] { const char* e = getenv("FOAMYADE_FACTOR_Q"); return e ? atoi(e) : 0; }();      // (experiments)
    const int q = force_q > 0 ? force_q : (tri <= 1024 ? 1 : (tri <= 2048 ? 2 : 3));
    const int nt = std::min(1024, ((tri + q - 1) / q + 63) / 64 * 64);
    if (q == 1) hipLaunchKernelGGL(k_mg_coarse_factor<1>, dim3(1), dim3(nt), fac_lds_bytes(A.N, bw), s, A, bw, fac);
    else if (q == 2) hipLaunchKernelGGL(k_mg_coarse_factor<2>, dim3(1), dim3(nt), fac_lds_bytes(A.N, bw), s, A, bw, fac);
    else hipLaunchKernelGGL(k_mg_coarse_factor<3>, dim3(1), dim3(nt), fac_lds_bytes(A.N, bw), s, A, bw, fac);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_tail(hipStream_t s, const PMat* A, double* const* x0, double* const* x1, double* const* b, int n, double w, int coarse_sweeps, MgWeights W,
                   const double* fac) {
    if (n < 1 || n > kMgTailMax) return fail(FY_ERR_INVALID, "bad multigrid tail depth %d", n);
    MgTail T;
    T.n = n;
    T.fac = fac;
    size_t lds = 0;
    static bool attr_set = false;
    if (!attr_set) {      // the factor of the coarsest level + seven arrays of the tail's first level
        const int want = (int)(fac_lds_bytes(kMgDirectMax, kMgDirectBand) + 7 * (size_t)kMgTailCells * sizeof(double));
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_mg_tail), hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess)
            return fail(FY_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
        attr_set = true;
    }
    if (fac) {
        if (!(mg_coarse_direct_ok)(A[n - 1])) return fail(FY_ERR_INVALID, "multigrid tail: a factor was handed over for a level it cannot belong to");
        lds = fac_lds_bytes(A[n - 1].N, band_width(A[n - 1]));
    }
    T.cache_n = 0; T.cache_off = (int)(lds / sizeof(double));
    const char* tc = getenv("FOAMYADE_NO_TAIL_CACHE");                                 // (A/B switch, read per launch: tests flip it inside one process)
    const bool tail_cache = !(tc && *tc && strcmp(tc, "0") != 0);
    if (tail_cache && n >= 2 && A[0].N <= kMgTailCells) { T.cache_n = A[0].N; lds += 7 * (size_t)A[0].N * sizeof(double); }
    for (int l = 0; l < n; ++l) {
        if (A[l].c0 != 0) return fail(FY_ERR_INVALID, "multigrid tail levels must not carry ghost planes");
        T.A[l] = A[l]; T.x0[l] = x0[l]; T.x1[l] = x1[l]; T.b[l] = b[l];
    }
    if (W.n < 2 || (W.n & 1)) return fail(FY_ERR_INVALID, "the multigrid tail needs an even number of smoothing sweeps");
    hipLaunchKernelGGL(k_mg_tail, dim3(1), dim3(1024), lds, s, T, w, coarse_sweeps, W);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_residual_restrict(hipStream_t s, PMat A, const double* b, const double* x, PMat C, double* bc) {
    if (C.N > 8192 && C.ny * 2 >= A.ny && C.nz * 2 >= A.nz) {       // big level: coalesced tile kernel (grid.y/z = coarse rows/planes)
        if (pairs_ok(A) && al16(b) && al16(x) && A.ny % 2 == 0 && A.nz % 2 == 0) {
            // the tile's x-extent (8, 16, 32 or 64 pairs) that wastes the fewest lanes on this row length
            const int px_opts[4] = {64, 32, 16, 8};
            int best = 64; double best_fill = 0.0;
            for (int px : px_opts) { const double fill = (double)A.nx / (double)(div_up(A.nx, 2 * px) * 2 * px); if (fill > best_fill + 1e-9) { best_fill = fill; best = px; } }
            if (best == 64) hipLaunchKernelGGL((k_mg_residual_restrict_tiled2<64, 2, 2>), dim3(div_up(A.nx, 128), div_up(A.ny, 2), div_up(A.nz, 2)), dim3(256), 0, s, A, b, x, C, bc);
            else if (best == 32) hipLaunchKernelGGL((k_mg_residual_restrict_tiled2<32, 4, 2>), dim3(div_up(A.nx, 64), div_up(A.ny, 4), div_up(A.nz, 2)), dim3(256), 0, s, A, b, x, C, bc);
            else if (best == 16) hipLaunchKernelGGL((k_mg_residual_restrict_tiled2<16, 4, 4>), dim3(div_up(A.nx, 32), div_up(A.ny, 4), div_up(A.nz, 4)), dim3(256), 0, s, A, b, x, C, bc);
            else hipLaunchKernelGGL((k_mg_residual_restrict_tiled2<8, 8, 4>), dim3(div_up(A.nx, 16), div_up(A.ny, 8), div_up(A.nz, 4)), dim3(256), 0, s, A, b, x, C, bc);
        }
        else hipLaunchKernelGGL(k_mg_residual_restrict_tiled, dim3(div_up(A.nx, 64), C.ny, C.nz), dim3(256), 0, s, A, b, x, C, bc);
        FY_LAUNCH_CHECK();
        return FY_OK;
    }
    hipLaunchKernelGGL(k_mg_residual_restrict, dim3(div_up(C.N, 256)), dim3(256), 0, s, A, b, x, C, bc);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_smooth_prolong(hipStream_t s, PMat A, const double* b, const double* x, PMat C, const double* xc, double* xn, double w) {
    if (A.c0 != 0 || A.ntot != A.N) return fail(FY_ERR_INVALID, "k_mg_smooth_prolong works on levels without ghost planes");
    if (pairs_ok(A) && al16(b) && al16(x) && al16(xn)) hipLaunchKernelGGL(k_mg_smooth_prolong2, dim3(div_up(A.N, 512)), dim3(256), 0, s, A, b, x, C, xc, xn, w);
    else hipLaunchKernelGGL(k_mg_smooth_prolong, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, b, x, C, xc, xn, w);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_prolong_add(hipStream_t s, PMat A, double* x, PMat C, const double* xc) {
    hipLaunchKernelGGL(k_mg_prolong_add, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, x, C, xc);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_prolong_add_planes(hipStream_t s, PMat A, int kofs, double* x, PMat C, const double* xc) {
    hipLaunchKernelGGL(k_mg_prolong_add_planes, dim3(div_up(A.N, 256)), dim3(256), 0, s, A, kofs, x, C, xc);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_coarse_solve(hipStream_t s, PMat A, const double* b, double* x, double* tmp, int sweeps, double w, const double* fac) {
    if (A.c0 != 0) return fail(FY_ERR_INVALID, "the coarsest multigrid level must be replicated (no ghost planes)");
    if (A.N > 1024) return fail(FY_ERR_INVALID, "coarsest multigrid level too large (%d cells)", A.N);
    size_t lds = 0;
    if (fac) {
        if (!(mg_coarse_direct_ok)(A)) return fail(FY_ERR_INVALID, "coarse solve: a factor was handed over for a level it cannot belong to");
        static bool attr_set = false;
        if (!attr_set) { FY_TRY(allow_big_lds(reinterpret_cast<const void*>(k_mg_coarse_solve))); attr_set = true; }
        lds = fac_lds_bytes(A.N, band_width(A));
    }
    hipLaunchKernelGGL(k_mg_coarse_solve, dim3(1), dim3(1024), lds, s, A, b, x, tmp, sweeps, w, fac);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_relax_field(hipStream_t s, double* x, const double* prev, double alpha, size_t n) {
    if (n == 0) return FY_OK;
    hipLaunchKernelGGL(k_relax_field, dim3(div_up(n, 256)), dim3(256), 0, s, x, prev, alpha, n);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_copy_f64(hipStream_t s, double* dst, const double* src, size_t n) {
    if (!n) return FY_OK;
    hipLaunchKernelGGL(k_copy, dim3(div_up(n, 256)), dim3(256), 0, s, dst, src, n);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_p_ghost_uz(hipStream_t s, FvGeo g, CFace3 rAUf, CFace3 alphaf, PMat A) {
    hipLaunchKernelGGL(k_p_ghost_uz, dim3(div_up((size_t)g.nx * g.ny, 256)), dim3(256), 0, s, g, rAUf, alphaf, A);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_mg_coarsen_ghost(hipStream_t s, PMat F, PMat C) {
    hipLaunchKernelGGL(k_mg_coarsen_ghost, dim3(div_up((size_t)C.nx * C.ny, 256)), dim3(256), 0, s, F, C);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_add_f64(hipStream_t s, double* y, const double* x, size_t n) {
    if (!n) return FY_OK;
    hipLaunchKernelGGL(k_add, dim3(div_up(n, 256)), dim3(256), 0, s, y, x, n);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

#if FY_FVK_GRADED
}  // namespace gr
#endif
}  // namespace fy
