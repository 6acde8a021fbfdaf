// ldu_amg.hpp: the ELL form of fy_ldu_solver's pressure matrix, the agglomeration hierarchy (host, once per mesh) and the V-cycle's kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <numeric>

#include "fv_kernels.hpp"
#include "ldu_amg.hpp"

namespace fy {
namespace {

constexpr double kWa = 1.7318685872766142, kWb = 0.5695012757370842;      // fv_solver.hpp: the degree-2 Chebyshev pair of D^-1 A on [1/3, 2]

#define FY_LAUNCH_CHECK()                                                                                     \
    do {                                                                                                      \
        hipError_t _e = hipGetLastError();                                                                    \
        if (_e != hipSuccess) return fail(FY_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

__device__ __forceinline__ double ell_offdiag(const EllMat& A, const double* __restrict__ x, int c) {
    // (a row's entries come first, its padding -- the cell itself with coefficient 0 -- after them: the loop ends at the first pad, so a mesh with a few many-faced cells
    // pays their width in memory only, not in every row's traffic)
    double s = 0.0;
    for (int k = 0; k < A.W; ++k) {
        const size_t e = (size_t)k * A.n + c;
        const int nb = A.nbr[e];
        if (nb == c) break;
        s += A.coef[e] * x[nb];
    }
    return s;
}

__global__ __launch_bounds__(256) void k_ell_fill(int n, int W, const int32_t* __restrict__ ell_face, const double* __restrict__ pcoef, double* __restrict__ coef) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)n * W) return;
    const int f = ell_face[e];
    coef[e] = f >= 0 ? pcoef[f] : 0.0;
}
__global__ __launch_bounds__(256) void k_ell_jacobi(int n, const double* __restrict__ diag, const double* __restrict__ r, double* __restrict__ u) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) u[c] = r[c] / diag[c];
}
__global__ __launch_bounds__(256) void k_ell_apply_dot(EllMat A, const double* __restrict__ u, const double* __restrict__ r, double* __restrict__ w, double* __restrict__ partials) {
    __shared__ double sh[4][2];
    double v0 = 0.0, v1 = 0.0;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < A.n) {
        const double uc = u[c];
        const double wc = A.diag[c] * uc - ell_offdiag(A, u, c);
        w[c] = wc;
        v0 = uc * r[c]; v1 = uc * wc;
    }
    for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_down(v0, o, 64); v1 += __shfl_down(v1, o, 64); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sh[wv][0] = v0; sh[wv][1] = v1; }
    __syncthreads();
    if (threadIdx.x < 2) partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
}

// ---- the coarse operators: entry e = (slot k, coarse cell I) of the next level sums the fine entries its list names; the diagonal the children's
// diagonals minus the entries that became internal to the aggregate (each internal face appears from both sides: - 2 a)
__global__ __launch_bounds__(256) void k_amg_galerkin_coef(size_t n_ent, const int32_t* __restrict__ ent_off, const int32_t* __restrict__ ent_idx, const double* __restrict__ fcoef,
                                                           double* __restrict__ ccoef, double scale) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_ent) return;
    double s = 0.0;
    for (int q = ent_off[e]; q < ent_off[e + 1]; ++q) s += fcoef[ent_idx[q]];
    ccoef[e] = scale * s;
}
__global__ __launch_bounds__(256) void k_amg_galerkin_diag(int nc, const int32_t* __restrict__ child_off, const int32_t* __restrict__ child, const int32_t* __restrict__ din_off,
                                                           const int32_t* __restrict__ din_idx, const double* __restrict__ fdiag, const double* __restrict__ fcoef,
                                                           const double* __restrict__ diag0, int ref_cell0, int ref_here, double* __restrict__ cdiag, double* __restrict__ cinvd, double scale) {
    const int I = blockIdx.x * 256 + threadIdx.x;
    if (I >= nc) return;
    double d = 0.0, in = 0.0;
    for (int q = child_off[I]; q < child_off[I + 1]; ++q) d += fdiag[child[q]];
    for (int q = din_off[I]; q < din_off[I + 1]; ++q) in += fcoef[din_idx[q]];
    double v = scale * (d - in);
    if (I == ref_here) v += (1.0 - scale) * (0.5 * diag0[ref_cell0]);      // setReference doubled the fine diagonal: the point term is half of it
    cdiag[I] = v; cinvd[I] = 1.0 / v;
}
__global__ __launch_bounds__(256) void k_amg_invd(int n, const double* __restrict__ diag, double* __restrict__ invd) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) invd[c] = 1.0 / diag[c];
}

// ---- smoothing, one row each (shared by the one-launch-per-sweep kernels of the large levels and the one-workgroup tail of the small ones).  From a zero
// guess the first sweep is x = wa b / d; the pair's second sweep forms the neighbours' first iterate inline
__device__ __forceinline__ void row_two_from_zero(const EllMat& A, const double* __restrict__ invd, const double* __restrict__ b, double* __restrict__ xn, double wa, double wb, int c) {
    double s = 0.0;
    for (int k = 0; k < A.W; ++k) {
        const size_t e = (size_t)k * A.n + c;
        const int nb = A.nbr[e];
        if (nb == c) break;
        s += A.coef[e] * (wa * b[nb] * invd[nb]);
    }
    const double xc = wa * b[c] * invd[c];
    xn[c] = xc + wb * invd[c] * (b[c] - (A.diag[c] * xc - s));
}
__device__ __forceinline__ void row_smooth(const EllMat& A, const double* __restrict__ invd, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ xn, double w, int c) {
    const double xc = x[c];
    xn[c] = xc + w * invd[c] * (b[c] - (A.diag[c] * xc - ell_offdiag(A, x, c)));
}
// the first post-smoothing sweep with the prolongation folded in: x' = x + e[agg]
__device__ __forceinline__ void row_prolong_smooth(const EllMat& A, const double* __restrict__ invd, const double* __restrict__ b, const double* __restrict__ x,
                                                   const int32_t* __restrict__ agg, const double* __restrict__ e, double* __restrict__ xn, double w, int c) {
    double s = 0.0;
    for (int k = 0; k < A.W; ++k) {
        const size_t q = (size_t)k * A.n + c;
        const int nb = A.nbr[q];
        if (nb == c) break;
        s += A.coef[q] * (x[nb] + e[agg[nb]]);
    }
    const double xc = x[c] + e[agg[c]];
    xn[c] = xc + w * invd[c] * (b[c] - (A.diag[c] * xc - s));
}
__device__ __forceinline__ void row_residual(const EllMat& A, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ r, int c) {
    r[c] = b[c] - (A.diag[c] * x[c] - ell_offdiag(A, x, c));
}
__device__ __forceinline__ void row_restrict(const int32_t* __restrict__ child_off, const int32_t* __restrict__ child, const double* __restrict__ r, double* __restrict__ bc, int I) {
    double s = 0.0;
    for (int q = child_off[I]; q < child_off[I + 1]; ++q) s += r[child[q]];
    bc[I] = s;
}
__global__ __launch_bounds__(256) void k_amg_two_from_zero(EllMat A, const double* __restrict__ invd, const double* __restrict__ b, double* __restrict__ xn, double wa, double wb) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < A.n) row_two_from_zero(A, invd, b, xn, wa, wb, c);
}
__global__ __launch_bounds__(256) void k_amg_smooth(EllMat A, const double* __restrict__ invd, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ xn, double w) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < A.n) row_smooth(A, invd, b, x, xn, w, c);
}
__global__ __launch_bounds__(256) void k_amg_prolong_smooth(EllMat A, const double* __restrict__ invd, const double* __restrict__ b, const double* __restrict__ x,
                                                            const int32_t* __restrict__ agg, const double* __restrict__ e, double* __restrict__ xn, double w) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < A.n) row_prolong_smooth(A, invd, b, x, agg, e, xn, w, c);
}
__global__ __launch_bounds__(256) void k_amg_residual(EllMat A, const double* __restrict__ b, const double* __restrict__ x, double* __restrict__ r) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < A.n) row_residual(A, b, x, r, c);
}
__global__ __launch_bounds__(256) void k_amg_restrict(int nc, const int32_t* __restrict__ child_off, const int32_t* __restrict__ child, const double* __restrict__ r, double* __restrict__ bc) {
    const int I = blockIdx.x * 256 + threadIdx.x;
    if (I < nc) row_restrict(child_off, child, r, bc, I);
}
// the coarsest level (<= kAmgCoarsest cells) is solved exactly: its matrix is inverted once per assembly -- in-place Gauss-Jordan in LDS, no pivoting
// (the matrix is symmetric positive definite: the reference cell's point term, or a fixed-value patch, removes the constant) -- and the solve is the
// product with the inverse.  Sweeps of damped Jacobi in its place left the constant-like mode in (160 sweeps at 0.8: asymptotic contraction of the cycle
// 0.97 on 64^3 cubes) and cost more than the rest of the coarse levels together (~1.3 us per sweep: every sweep is a barrier)
__global__ __launch_bounds__(1024) void k_amg_coarsest_invert(EllMat A, double* __restrict__ inv) {
    __shared__ double a[kAmgCoarsest * kAmgCoarsest];
    __shared__ double col[kAmgCoarsest];
    const int n = A.n, t = threadIdx.x;
    for (int q = t; q < n * n; q += 1024) a[q] = 0.0;
    __syncthreads();
    if (t < n) {
        a[t * n + t] = A.diag[t];
        for (int k = 0; k < A.W; ++k) {
            const size_t e = (size_t)k * n + t;
            const int nb = A.nbr[e];
            if (nb != t) a[t * n + nb] -= A.coef[e];
        }
    }
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const double p = 1.0 / a[k * n + k];
        if (t < n) col[t] = a[t * n + k];
        __syncthreads();
        if (t < n) a[k * n + t] = t == k ? p : a[k * n + t] * p;
        __syncthreads();
        for (int q = t; q < n * n; q += 1024) {
            const int i = q / n, j = q - i * n;
            if (i == k) continue;
            a[q] = j == k ? -col[i] * p : a[q] - col[i] * a[k * n + j];
        }
        __syncthreads();
    }
    for (int q = t; q < n * n; q += 1024) inv[q] = a[q];
}
// The small levels (<= kAmgTailCells cells and everything below) in ONE workgroup: a launch per sweep costs ~7 us of dispatch for a few hundred rows of work, and a
// V-cycle has seven of them per level; here a sweep ends at a workgroup barrier instead (the iterates stay in global memory: L2).  Same rows, same order, same bits
struct TailLevel {
    EllMat A;
    const double *invd;
    const double *b;                      // the level's right-hand side (the first level: the caller's)
    double *x0, *x1, *bnext;              // scratch / answer, the next level's right-hand side
    const int32_t *agg, *child_off, *child;
    int n_next;
};
struct TailArgs { TailLevel L[kAmgTailMax]; int n; const double* inv; double* xlast; };
__global__ __launch_bounds__(1024) void k_amg_tail(TailArgs T, double wa, double wb) {
    __shared__ double bs[kAmgCoarsest];
    const int t = threadIdx.x;
    for (int l = 0; l + 1 < T.n; ++l) {
        const TailLevel& L = T.L[l];
        for (int c = t; c < L.A.n; c += 1024) row_two_from_zero(L.A, L.invd, L.b, L.x1, wa, wb, c);
        __syncthreads();
        for (int c = t; c < L.A.n; c += 1024) row_residual(L.A, L.b, L.x1, L.x0, c);
        __syncthreads();
        for (int I = t; I < L.n_next; I += 1024) row_restrict(L.child_off, L.child, L.x0, L.bnext, I);
        __syncthreads();
    }
    {
        const TailLevel& L = T.L[T.n - 1];
        const int n = L.A.n;
        if (t < n) bs[t] = L.b[t];
        __syncthreads();
        if (t < n) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += T.inv[(size_t)j * n + t] * bs[j];
            T.xlast[t] = s;
        }
        __syncthreads();
    }
    for (int l = T.n - 2; l >= 0; --l) {
        const TailLevel& L = T.L[l];
        const double* e = l + 1 == T.n - 1 ? T.xlast : T.L[l + 1].x1;
        for (int c = t; c < L.A.n; c += 1024) row_prolong_smooth(L.A, L.invd, L.b, L.x1, L.agg, e, L.x0, wb, c);
        __syncthreads();
        for (int c = t; c < L.A.n; c += 1024) row_smooth(L.A, L.invd, L.b, L.x0, L.x1, wa, c);
        __syncthreads();
    }
}

inline dim3 grid_of(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---- host: the graphs of the hierarchy
struct Graph {
    int n = 0;
    std::vector<int32_t> off, nbr;
    std::vector<double> w;
};

// one pairwise matching pass [OF-6 pairGAMGAgglomeration::agglomerate]: a free cell pairs with its free neighbour across the heaviest face; cells left
// alone join the cluster of their heaviest neighbour.  Faces within kPairTie of the heaviest count as equally heavy and the FIRST of them (the lowest
// neighbour) is taken: on a lattice, and on one that is mildly distorted, the face areas differ by rounding or by a few per cent, and following those
// differences would zig-zag the pairs where taking the first in order stacks them into 2 x 2 x 2 boxes after three passes (measured on 64^3 cubes:
// 510 aggregates of 16 neighbours and 23 PCG iterations per step with the strict maximum).  Clusters are numbered in the order of their first cell
constexpr double kPairTie = 0.8;
int pick(const Graph& G, const std::vector<int32_t>& cl, int c, bool want_free) {
    double bw = -1.0;
    for (int q = G.off[(size_t)c]; q < G.off[(size_t)c + 1]; ++q) {
        const int nb = G.nbr[(size_t)q];
        if (nb != c && (want_free ? cl[(size_t)nb] < 0 : cl[(size_t)nb] >= 0)) bw = std::max(bw, G.w[(size_t)q]);
    }
    if (bw < 0.0) return -1;
    for (int q = G.off[(size_t)c]; q < G.off[(size_t)c + 1]; ++q) {
        const int nb = G.nbr[(size_t)q];
        if (nb != c && (want_free ? cl[(size_t)nb] < 0 : cl[(size_t)nb] >= 0) && G.w[(size_t)q] >= kPairTie * bw) return nb;
    }
    return -1;
}
int pair_pass(const Graph& G, std::vector<int32_t>* cl) {
    cl->assign((size_t)G.n, -1);
    int nc = 0;
    for (int c = 0; c < G.n; ++c) {
        if ((*cl)[(size_t)c] >= 0) continue;
        const int best = pick(G, *cl, c, true);
        if (best >= 0) { (*cl)[(size_t)c] = (*cl)[(size_t)best] = nc++; }
        else (*cl)[(size_t)c] = -2;                       // alone for now
    }
    for (int c = 0; c < G.n; ++c) {
        if ((*cl)[(size_t)c] != -2) continue;
        const int best = pick(G, *cl, c, false);
        (*cl)[(size_t)c] = best >= 0 ? (*cl)[(size_t)best] : nc++;
    }
    // number the clusters by their first cell (the joins above may have left the order intact already; pairs are numbered at their first cell)
    std::vector<int32_t> renum((size_t)nc, -1);
    int next = 0;
    for (int c = 0; c < G.n; ++c) { int32_t& v = (*cl)[(size_t)c]; if (renum[(size_t)v] < 0) renum[(size_t)v] = next++; v = renum[(size_t)v]; }
    return nc;
}

// the graph of the clusters: an edge between two clusters weighs the sum of the edges between their cells; neighbours ascending
Graph coarse_graph(const Graph& G, const std::vector<int32_t>& cl, int nc) {
    Graph C;
    C.n = nc;
    std::vector<int32_t> coff((size_t)nc + 1, 0), child((size_t)G.n);
    for (int c = 0; c < G.n; ++c) ++coff[(size_t)cl[(size_t)c] + 1];
    for (int I = 0; I < nc; ++I) coff[(size_t)I + 1] += coff[(size_t)I];
    { std::vector<int32_t> fill(coff.begin(), coff.end() - 1); for (int c = 0; c < G.n; ++c) child[(size_t)fill[(size_t)cl[(size_t)c]]++] = c; }
    C.off.assign((size_t)nc + 1, 0);
    std::vector<int32_t> pos((size_t)nc, -1);
    std::vector<std::pair<int32_t, double> > row;
    for (int I = 0; I < nc; ++I) {
        row.clear();
        for (int q = coff[(size_t)I]; q < coff[(size_t)I + 1]; ++q) {
            const int c = child[(size_t)q];
            for (int e = G.off[(size_t)c]; e < G.off[(size_t)c + 1]; ++e) {
                const int J = cl[(size_t)G.nbr[(size_t)e]];
                if (J == I) continue;
                if (pos[(size_t)J] < 0) { pos[(size_t)J] = (int32_t)row.size(); row.emplace_back(J, 0.0); }
                row[(size_t)pos[(size_t)J]].second += G.w[(size_t)e];
            }
        }
        for (auto& r : row) pos[(size_t)r.first] = -1;
        std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
        for (auto& r : row) { C.nbr.push_back(r.first); C.w.push_back(r.second); }
        C.off[(size_t)I + 1] = (int32_t)C.nbr.size();
    }
    return C;
}

template <class T>
int upload(hipStream_t s, DevBuf<T>& d, const std::vector<T>& h) {
    FY_TRY(d.alloc_exact(std::max<size_t>(h.size(), 1)));
    if (!h.empty()) {
        FY_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
        FY_HIP(hipStreamSynchronize(s));               // (the callers hand over vectors that die with their scope; the hierarchy is built once)
    }
    return FY_OK;
}

// the ELL pattern of a graph: slot-major neighbour table, padded with the cell itself
std::vector<int32_t> ell_pattern(const Graph& G, int* W) {
    int w = 1;
    for (int c = 0; c < G.n; ++c) w = std::max(w, G.off[(size_t)c + 1] - G.off[(size_t)c]);
    std::vector<int32_t> nb((size_t)w * G.n);
    for (int c = 0; c < G.n; ++c)
        for (int k = 0; k < w; ++k) {
            const int q = G.off[(size_t)c] + k;
            nb[(size_t)k * G.n + c] = q < G.off[(size_t)c + 1] ? G.nbr[(size_t)q] : c;
        }
    *W = w;
    return nb;
}

}  // namespace

int LduAmg::build(hipStream_t s, int n_cells, int n_internal, const int32_t* own, const int32_t* nei, const std::vector<int32_t>& cf_off, const std::vector<int32_t>& cf_face,
                  const double* face_weight, int ref_cell, bool with_hierarchy) {
    lev.clear();
    ref_cell0 = ref_cell; hier = with_hierarchy;
    // the mesh's graph: per cell its internal faces in ascending face order (the order the face-addressed kernels sum in)
    Graph G;
    G.n = n_cells;
    G.off.assign((size_t)n_cells + 1, 0);
    std::vector<int32_t> face_of;
    for (int c = 0; c < n_cells; ++c) {
        for (int q = cf_off[(size_t)c]; q < cf_off[(size_t)c + 1]; ++q) {
            const int f = cf_face[(size_t)q];
            if (f >= n_internal) continue;
            G.nbr.push_back(own[f] == c ? nei[f] : own[f]);
            G.w.push_back(face_weight[f]);
            face_of.push_back(f);
        }
        G.off[(size_t)c + 1] = (int32_t)G.nbr.size();
    }
    auto add_level = [&](const Graph& g, int ref) -> int {
        lev.emplace_back(new AmgLevel());
        AmgLevel& L = *lev.back();
        L.n = g.n; L.ref_cell = ref;
        std::vector<int32_t> nb = ell_pattern(g, &L.W);
        FY_TRY(upload(s, L.nbr, nb));
        return FY_OK;
    };
    FY_TRY(add_level(G, ref_cell));
    {
        AmgLevel& L0 = *lev[0];
        std::vector<int32_t> ef((size_t)L0.W * G.n, -1);
        for (int c = 0; c < G.n; ++c) for (int q = G.off[(size_t)c]; q < G.off[(size_t)c + 1]; ++q) ef[(size_t)(q - G.off[(size_t)c]) * G.n + c] = face_of[(size_t)q];
        FY_TRY(upload(s, ell_face, ef));
        FY_TRY(L0.coef.alloc_exact((size_t)L0.W * L0.n));
    }
    if (with_hierarchy) {
        int ref = ref_cell;
        while (G.n > kAmgCoarsest && lev.size() < 24) {
            // three pairwise passes, composed
            std::vector<int32_t> agg((size_t)G.n);
            std::iota(agg.begin(), agg.end(), 0);
            Graph cur = G;                                       // (a copy per level: the passes shrink it)
            int nc = G.n;
            for (int pass = 0; pass < passes && nc > kAmgCoarsest / 2; ++pass) {
                std::vector<int32_t> cl;
                const int n2 = pair_pass(cur, &cl);
                if (n2 >= nc) break;
                for (int32_t& a : agg) a = cl[(size_t)a];
                cur = coarse_graph(cur, cl, n2);
                nc = n2;
            }
            if (nc >= G.n || (double)nc > 0.9 * G.n) break;     // (a graph that does not coarsen: keep what there is)
            // `cur` is the next level's graph.  The lists that build its matrix from this level's ELL entries
            AmgLevel& F = *lev.back();
            std::vector<int32_t> child_off((size_t)nc + 1, 0), child((size_t)G.n);
            for (int c = 0; c < G.n; ++c) ++child_off[(size_t)agg[(size_t)c] + 1];
            for (int I = 0; I < nc; ++I) child_off[(size_t)I + 1] += child_off[(size_t)I];
            { std::vector<int32_t> fill(child_off.begin(), child_off.end() - 1); for (int c = 0; c < G.n; ++c) child[(size_t)fill[(size_t)agg[(size_t)c]]++] = c; }
            int Wc = 1;
            for (int I = 0; I < nc; ++I) Wc = std::max(Wc, cur.off[(size_t)I + 1] - cur.off[(size_t)I]);
            std::vector<std::vector<int32_t> > ent((size_t)Wc * nc);          // (small vectors; the level is built once)
            std::vector<int32_t> din_off((size_t)nc + 1, 0), din_idx;
            std::vector<int32_t> pos((size_t)nc, -1);
            for (int I = 0; I < nc; ++I) {
                for (int q = cur.off[(size_t)I]; q < cur.off[(size_t)I + 1]; ++q) pos[(size_t)cur.nbr[(size_t)q]] = q - cur.off[(size_t)I];
                for (int q = child_off[(size_t)I]; q < child_off[(size_t)I + 1]; ++q) {
                    const int c = child[(size_t)q];
                    for (int e = G.off[(size_t)c]; e < G.off[(size_t)c + 1]; ++e) {
                        const int J = agg[(size_t)G.nbr[(size_t)e]];
                        const int32_t idx = (int32_t)((size_t)(e - G.off[(size_t)c]) * G.n + c);       // this level's ELL entry
                        if (J == I) din_idx.push_back(idx);
                        else ent[(size_t)pos[(size_t)J] * nc + I].push_back(idx);
                    }
                }
                din_off[(size_t)I + 1] = (int32_t)din_idx.size();
                for (int q = cur.off[(size_t)I]; q < cur.off[(size_t)I + 1]; ++q) pos[(size_t)cur.nbr[(size_t)q]] = -1;
            }
            std::vector<int32_t> ent_off((size_t)Wc * nc + 1, 0), ent_idx;
            for (size_t e = 0; e < ent.size(); ++e) { ent_idx.insert(ent_idx.end(), ent[e].begin(), ent[e].end()); ent_off[e + 1] = (int32_t)ent_idx.size(); }
            if ((size_t)F.W * (size_t)F.n > (size_t)INT32_MAX) return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: the mesh is too large for 32-bit matrix entry numbers");
            FY_TRY(upload(s, F.agg, agg)); FY_TRY(upload(s, F.child_off, child_off)); FY_TRY(upload(s, F.child, child));
            FY_TRY(upload(s, F.ent_off, ent_off)); FY_TRY(upload(s, F.ent_idx, ent_idx)); FY_TRY(upload(s, F.din_off, din_off)); FY_TRY(upload(s, F.din_idx, din_idx));
            ref = ref >= 0 ? agg[(size_t)ref] : -1;
            G = std::move(cur);
            FY_TRY(add_level(G, ref));
            AmgLevel& L = *lev.back();
            FY_TRY(L.coef.alloc_exact((size_t)L.W * L.n)); FY_TRY(L.diag.alloc_exact((size_t)L.n));
            FY_TRY(L.b.alloc_exact((size_t)L.n));
        }
        for (auto& lp : lev) { FY_TRY(lp->invd.alloc_exact((size_t)lp->n)); FY_TRY(lp->x0.alloc_exact((size_t)lp->n)); FY_TRY(lp->x1.alloc_exact((size_t)lp->n)); }
        FY_TRY(coarse_inv.alloc_exact((size_t)kAmgCoarsest * kAmgCoarsest));
        if (lev.back()->n > kAmgCoarsest) return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: the agglomeration stalled at %d cells (more than %d): use the diagonal preconditioner", lev.back()->n, kAmgCoarsest);
    }
    FY_HIP(hipStreamSynchronize(s));
    return FY_OK;
}

int LduAmg::setup(hipStream_t s, const double* pcoef, const double* pdiag) {
    AmgLevel& L0 = *lev[0];
    diag0_ = pdiag;
    hipLaunchKernelGGL(k_ell_fill, grid_of((size_t)L0.n * L0.W), dim3(256), 0, s, L0.n, L0.W, ell_face.p, pcoef, L0.coef.p);
    FY_LAUNCH_CHECK();
    if (!has_hierarchy()) return FY_OK;
    hipLaunchKernelGGL(k_amg_invd, grid_of((size_t)L0.n), dim3(256), 0, s, L0.n, pdiag, L0.invd.p);
    FY_LAUNCH_CHECK();
    for (size_t l = 0; l + 1 < lev.size(); ++l) {
        AmgLevel& F = *lev[l];
        AmgLevel& C = *lev[l + 1];
        const double* fdiag = l == 0 ? pdiag : F.diag.p;
        const size_t n_ent = (size_t)C.W * C.n;
        // the over-correction of piecewise-constant transfer: 1 / (cells per aggregate)^(1/3), i.e. 1/2 for the 2 x 2 x 2 of three clean pairwise passes
        const double scale = std::pow((double)F.n / (double)C.n, -1.0 / 3.0);
        hipLaunchKernelGGL(k_amg_galerkin_coef, grid_of(n_ent), dim3(256), 0, s, n_ent, F.ent_off.p, F.ent_idx.p, F.coef.p, C.coef.p, scale);
        FY_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_amg_galerkin_diag, grid_of((size_t)C.n), dim3(256), 0, s, C.n, F.child_off.p, F.child.p, F.din_off.p, F.din_idx.p, fdiag, F.coef.p, pdiag,
                           ref_cell0 >= 0 ? ref_cell0 : 0, ref_cell0 >= 0 ? C.ref_cell : -1, C.diag.p, C.invd.p, scale);
        FY_LAUNCH_CHECK();
    }
    {
        EllMat A = lev.back()->mat();
        if (lev.size() == 1) A.diag = pdiag;
        hipLaunchKernelGGL(k_amg_coarsest_invert, dim3(1), dim3(1024), 0, s, A, coarse_inv.p);
        FY_LAUNCH_CHECK();
    }
    return FY_OK;
}

int LduAmg::vcycle(hipStream_t s, const double* r, double* u) {
    // down: pre-smoothing pair from zero (into x1), residual (into x0), restriction (the next level's b)
    const size_t nl = lev.size();
    auto mat_of = [&](size_t l, const double* pdiag0) { EllMat A = lev[l]->mat(); if (l == 0) A.diag = pdiag0; return A; };
    const double* diag0 = diag0_;
    // the first level of the one-workgroup tail: the largest level of at most kAmgTailCells cells, at most kAmgTailMax levels from the end, never level 0 unless it is alone
    size_t tail = nl - 1;
    while (tail > 0 && nl - (tail - 1) <= (size_t)kAmgTailMax && lev[tail - 1]->n <= kAmgTailCells && (tail - 1 > 0 || nl == 1)) --tail;
    for (size_t l = 0; l < tail; ++l) {
        AmgLevel& L = *lev[l];
        const EllMat A = mat_of(l, diag0);
        const double* b = l == 0 ? r : L.b.p;
        hipLaunchKernelGGL(k_amg_two_from_zero, grid_of((size_t)L.n), dim3(256), 0, s, A, L.invd.p, b, L.x1.p, kWa, kWb);
        FY_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_amg_residual, grid_of((size_t)L.n), dim3(256), 0, s, A, b, L.x1.p, L.x0.p);
        FY_LAUNCH_CHECK();
        AmgLevel& C = *lev[l + 1];
        hipLaunchKernelGGL(k_amg_restrict, grid_of((size_t)C.n), dim3(256), 0, s, C.n, L.child_off.p, L.child.p, L.x0.p, C.b.p);
        FY_LAUNCH_CHECK();
    }
    {
        TailArgs T{};
        T.n = (int)(nl - tail);
        for (size_t l = tail; l < nl; ++l) {
            AmgLevel& L = *lev[l];
            TailLevel& q = T.L[l - tail];
            q.A = mat_of(l, diag0); q.invd = L.invd.p; q.b = l == 0 ? r : L.b.p; q.x0 = L.x0.p; q.x1 = L.x1.p;
            q.agg = L.agg.p; q.child_off = L.child_off.p; q.child = L.child.p;
            q.bnext = l + 1 < nl ? lev[l + 1]->b.p : nullptr; q.n_next = l + 1 < nl ? lev[l + 1]->n : 0;
        }
        T.inv = coarse_inv.p;
        T.xlast = nl == 1 ? u : lev[nl - 1]->x1.p;
        hipLaunchKernelGGL(k_amg_tail, dim3(1), dim3(1024), 0, s, T, kWa, kWb);
        FY_LAUNCH_CHECK();
    }
    // up: prolongation folded into the first post-smoothing sweep (weights in reverse order), the second sweep leaves the level's answer in x1 (level 0: u)
    for (size_t l = tail; l-- > 0;) {
        AmgLevel& L = *lev[l];
        const EllMat A = mat_of(l, diag0);
        const double* b = l == 0 ? r : L.b.p;
        hipLaunchKernelGGL(k_amg_prolong_smooth, grid_of((size_t)L.n), dim3(256), 0, s, A, L.invd.p, b, L.x1.p, L.agg.p, lev[l + 1]->x1.p, L.x0.p, kWb);
        FY_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_amg_smooth, grid_of((size_t)L.n), dim3(256), 0, s, A, L.invd.p, b, L.x0.p, l == 0 ? u : L.x1.p, kWa);
        FY_LAUNCH_CHECK();
    }
    return FY_OK;
}

int launch_ell_fill(hipStream_t s, int n, int W, const int32_t* ell_face, const double* pcoef, double* coef) {
    hipLaunchKernelGGL(k_ell_fill, grid_of((size_t)n * W), dim3(256), 0, s, n, W, ell_face, pcoef, coef);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ell_jacobi(hipStream_t s, int n, const double* diag, const double* r, double* u) {
    hipLaunchKernelGGL(k_ell_jacobi, grid_of((size_t)n), dim3(256), 0, s, n, diag, r, u);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ell_apply_dot(hipStream_t s, EllMat A, const double* u, const double* r, double* w, double* partials) {
    hipLaunchKernelGGL(k_ell_apply_dot, dim3(red_blocks(A.n)), dim3(256), 0, s, A, u, r, w, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

}  // namespace fy
