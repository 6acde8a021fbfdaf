// HIP kernels of fy_ldu_solver (ldu.hpp): icoFoamYade's operators (icoFoamYade.C:65-149) on a general polyhedral mesh with owner / neighbour
// addressing.  Two shapes only: one lane per FACE (coefficients, fluxes) and one lane per CELL gathering over the cell's faces through the
// cell -> face lists (sums, gradients, matrix rows) -- no scatter, hence no atomics, and every sum has a fixed order.  FP64, bandwidth / latency
// bound like the structured kernels (no MFMA: nothing here is a dense contraction); the indirection through own / nei / cf_face costs what a
// general mesh costs.  The arithmetic is OpenFOAM-6's, restated [OF-6]; oracle/ldu_oracle.cpp is the CPU restatement the tests compare with.
#include <hip/hip_runtime.h>

#include "fv_kernels.hpp"
#include "ldu.hpp"

namespace fy {
namespace {

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 ld3(const double* p, int q) { return D3{p[3 * (size_t)q], p[3 * (size_t)q + 1], p[3 * (size_t)q + 2]}; }
__device__ __forceinline__ void st3(double* p, int q, D3 a) { p[3 * (size_t)q] = a.x; p[3 * (size_t)q + 1] = a.y; p[3 * (size_t)q + 2] = a.z; }
__device__ __forceinline__ double dot3(D3 a, D3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ D3 lerp3(double w, D3 a, D3 b) { return D3{w * a.x + (1.0 - w) * b.x, w * a.y + (1.0 - w) * b.y, w * a.z + (1.0 - w) * b.z}; }

// boundary value of a velocity-like field on boundary face f: the patch's value (fixedValue), the cell's (zeroGradient), or the cell's without its component along the
// face's normal (symmetry / symmetryPlane / slip [OF-6 basicSymmetryFvPatchField::evaluate]: (U_P + transform(I - 2 n n, U_P)) / 2)
__device__ __forceinline__ D3 Ub(const LduGeo& g, const double* F, int f) {
    const int pa = g.patch_of[f - g.nInt];
    if (g.u_bc[pa] == FY_BC_U_FIXED_VALUE) return ld3(g.u_val, pa);
    const D3 uc = ld3(F, g.own[f]);
    if (g.u_bc[pa] == FY_BC_U_SLIP) {
        const double rm = 1.0 / g.magSf[f];
        const D3 S = ld3(g.Sf, f), n = D3{rm * S.x, rm * S.y, rm * S.z};
        const double un = dot3(n, uc);
        return D3{uc.x - un * n.x, uc.y - un * n.y, uc.z - un * n.z};
    }
    return uc;
}
// a symmetry face in the momentum matrix [OF-6 transformFvPatchField: gradientInternalCoeffs = -deltaCoeffs snGradTransformDiag, gradientBoundaryCoeffs = snGrad -
// gradientInternalCoeffs U_P; basicSymmetryFvPatchField: snGradTransformDiag = (|n_x|, |n_y|, |n_z|), snGrad = -n (n & U_P) deltaCoeffs]: bd += the per-component
// diagonal, b += the explicit remainder around the U the matrix is assembled with, bmax / bmin += cmptMax / cmptMin of the coefficient (fvMatrix::relax).  The flux
// through the face is U_b & Sf = 0 up to rounding: the convection term sees it like a zeroGradient face
__device__ __forceinline__ void slip_face(const LduGeo& g, int f, double gm, D3 uc, double (&bd)[3], double (&b)[3], double& bmax, double& bmin) {
    const double rm = 1.0 / g.magSf[f];
    const D3 S = ld3(g.Sf, f);
    const double nn[3] = {rm * S.x, rm * S.y, rm * S.z}, an[3] = {fabs(nn[0]), fabs(nn[1]), fabs(nn[2])}, u3[3] = {uc.x, uc.y, uc.z};
    const double un = dot3(D3{nn[0], nn[1], nn[2]}, uc);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bd[q] += gm * an[q];
        b[q] += gm * (an[q] * u3[q] - nn[q] * un);
    }
    bmax += gm * fmax(an[0], fmax(an[1], an[2]));
    bmin += gm * fmin(an[0], fmin(an[1], an[2]));
}
__device__ __forceinline__ double pbv(const LduGeo& g, const double* p, int f) {
    const int pa = g.patch_of[f - g.nInt];
    if (g.p_bc[pa] == FY_BC_P_FIXED_VALUE) return g.p_val[pa];
    // fixedFluxPressure is a fixed-gradient condition: p_b = p_P + snGrad / deltaCoeffs, the gradient as constrainPressure left it (pEqn.H:21)
    if (g.p_bc[pa] == FY_BC_P_FIXED_FLUX && g.psn) return p[g.own[f]] + g.psn[f - g.nInt] / g.dcNO[f];
    return p[g.own[f]];
}

// the faces of cell c in ascending face order through the slot tables (LduGeo::ef / en, slot-major like the ELL matrix: the lanes of a wave read consecutive
// words, and the neighbour comes with the slot instead of through owner / neighbour): f = the face, nb = the cell across it (-1: boundary face).  Internal
// faces have owner < neighbour, so c owns face f exactly when nb > c
#define FY_CELL_FACES(g, c, f, nb)                                                                                                     \
    for (int _k = 0, f = 0, nb = 0; _k < (g).Wall && (f = (g).ef[(size_t)_k * (g).nCells + (c)]) >= 0 && ((nb = (g).en[(size_t)_k * (g).nCells + (c)]), true); ++_k)

template <int N>
__device__ __forceinline__ void block_reduce_store(double (&v)[N], const int (&is_max)[N], double* partials) {
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
        partials[(size_t)q * gridDim.x + blockIdx.x] = x;
    }
}

// fvc::flux(F) = linearInterpolate(F) & Sf (createPhi; boundary: the patch value)
__global__ __launch_bounds__(256) void k_ldu_flux_of(LduGeo g, const double* __restrict__ F, double* __restrict__ phi) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    const D3 uf = f < g.nInt ? lerp3(g.w[f], ld3(F, g.own[f]), ld3(F, g.nei[f])) : Ub(g, F, f);
    phi[f] = dot3(uf, ld3(g.Sf, f));
}

// CourantNo.H [OF-6] (icoFoamYade.C:68): sumPhi = fvc::surfaceSum(mag(phi)); slot 0 = max sumPhi / V, slot 1 = sum sumPhi
__global__ __launch_bounds__(256) void k_ldu_courant(LduGeo g, const double* __restrict__ phi, double* __restrict__ partials) {
    double v[2] = {0, 0};
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.nCells) {
        double s = 0.0;
        FY_CELL_FACES(g, c, f, nb) s += fabs(phi[f]);
        v[0] = s / g.V[c]; v[1] = s;
    }
    const int mx[2] = {1, 0};
    block_reduce_store<2>(v, mx, partials);
}

// fvc::grad(F), Gauss linear: T[3 i + j] = (1/V) sum_f (+-Sf_i) F_f,j -- through the slot coefficients: gG0_i F_c,j + sum_k gB_k,i F_(neighbour or patch value),j
__global__ __launch_bounds__(256) void k_ldu_grad_vec(LduGeo g, const double* __restrict__ F, double* __restrict__ T) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const size_t n = (size_t)g.nCells, wn = (size_t)g.Wall * n;
    const D3 fc = ld3(F, c);
    const double g0[3] = {g.gG0[c], g.gG0[n + c], g.gG0[2 * n + c]}, u0[3] = {fc.x, fc.y, fc.z};
    double t[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) t[3 * i + j] = g0[i] * u0[j];
    for (int k = 0; k < g.Wall; ++k) {
        const size_t e = (size_t)k * n + c;
        const int nb = g.en[e];
        if (nb == -2) break;
        const D3 uf = nb >= 0 ? ld3(F, nb) : Ub(g, F, g.ef[e]);
        const double s[3] = {g.gB[e], g.gB[wn + e], g.gB[2 * wn + e]}, u[3] = {uf.x, uf.y, uf.z};
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) t[3 * i + j] += s[i] * u[j];
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) T[9 * (size_t)c + e] = t[e];
}

// the per-slot coefficients (LduGeo::gB, gG0, rT): one lane per cell, once at set-up
__global__ __launch_bounds__(256) void k_ldu_slot_coefs(LduGeo g, double* __restrict__ gB, double* __restrict__ gG0, double* __restrict__ rT) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const size_t n = (size_t)g.nCells, wn = (size_t)g.Wall * n;
    const double rV = 1.0 / g.V[c];
    const double* R = g.recon + 9 * (size_t)c;
    double g0[3] = {0, 0, 0};
    for (int k = 0; k < g.Wall; ++k) {
        const size_t e = (size_t)k * n + c;
        const int f = g.ef[e];
        double b[3] = {0, 0, 0}, t[3] = {0, 0, 0};
        if (f >= 0) {
            const int nb = g.en[e];
            const D3 S = ld3(g.Sf, f);
            const double s3[3] = {S.x, S.y, S.z}, rm = 1.0 / g.magSf[f];
            double sg = 1.0, wnb = 1.0, wc = 0.0;
            if (f < g.nInt) { const bool o = nb > c; sg = o ? 1.0 : -1.0; wc = o ? g.w[f] : 1.0 - g.w[f]; wnb = o ? 1.0 - g.w[f] : g.w[f]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) { b[q] = sg * s3[q] * wnb * rV; g0[q] += sg * s3[q] * wc * rV; }
#pragma unroll
            for (int q = 0; q < 3; ++q) t[q] = (R[3 * q] * s3[0] + R[3 * q + 1] * s3[1] + R[3 * q + 2] * s3[2]) * rm;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) { gB[q * wn + e] = b[q]; rT[q * wn + e] = t[q]; }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) gG0[q * n + c] = g0[q];
}
// Gauss-linear gradient of a scalar through the slot coefficients: grad_c = gG0_c p_c + sum_k gB_k p_(neighbour or patch value)
__device__ __forceinline__ D3 grad_scalar_at(const LduGeo& g, const double* __restrict__ p, int c) {
    const size_t n = (size_t)g.nCells, wn = (size_t)g.Wall * n;
    const double pc = p[c];
    D3 a{g.gG0[c] * pc, g.gG0[n + c] * pc, g.gG0[2 * n + c] * pc};
    for (int k = 0; k < g.Wall; ++k) {
        const size_t e = (size_t)k * n + c;
        const int nb = g.en[e];
        if (nb == -2) break;
        const double v = nb >= 0 ? p[nb] : pbv(g, p, g.ef[e]);
        a.x += g.gB[e] * v; a.y += g.gB[wn + e] * v; a.z += g.gB[2 * wn + e] * v;
    }
    return a;
}
__global__ __launch_bounds__(256) void k_ldu_grad_scalar(LduGeo g, const double* __restrict__ p, double* __restrict__ gp) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.nCells) st3(gp, c, grad_scalar_at(g, p, c));
}

// fvc::grad(magSqr(F)) for the limited schemes' r (boundary value magSqr(F_b))
__device__ __forceinline__ double msq(D3 a) { return (a.x * a.x + a.y * a.y) + a.z * a.z; }
__global__ __launch_bounds__(256) void k_ldu_grad_magsqr(LduGeo g, const double* __restrict__ F, double* __restrict__ gradL) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const double lc = msq(ld3(F, c));
    D3 a{0, 0, 0};
    FY_CELL_FACES(g, c, f, nb) {
        double lf, sg = 1.0;
        if (f < g.nInt) { const bool o = nb > c; const double ln = msq(ld3(F, nb)); lf = o ? g.w[f] * lc + (1.0 - g.w[f]) * ln : g.w[f] * ln + (1.0 - g.w[f]) * lc; sg = o ? 1.0 : -1.0; }
        else lf = msq(Ub(g, F, f));
        const D3 S = ld3(g.Sf, f);
        a.x += sg * S.x * lf; a.y += sg * S.y * lf; a.z += sg * S.z * lf;
    }
    const double rV = 1.0 / g.V[c];
    st3(gradL, c, D3{a.x * rV, a.y * rV, a.z * rV});
}
// NVD / TVD limited schemes [OF-6 LimitedScheme<vector, Limiter<NVDTVD>, limitFuncs::magSqr>, NVDTVD.H], as fv_kernels.hip's limiter_fn / limited weight: ONE limiter per
// face from lPhi = magSqr(U): r = 2 (d . grad(lPhi)_C) / (lPhi_N - lPhi_P) - 1, C the upwind cell of the face flux, d = C_N - C_P; the owner's weight is
// limiter w_linear + (1 - limiter) pos0(flux) [limitedSurfaceInterpolationScheme::weights], used implicitly
__device__ __forceinline__ double ldu_limiter_fn(int scheme, double twoByk, double r) {
    switch (scheme) {
        case FY_CONVECTION_LIMITED_LINEAR: return fmax(fmin(twoByk * r, 1.0), 0.0);
        case FY_CONVECTION_VAN_LEER: return (r + fabs(r)) / (1.0 + fabs(r));
        case FY_CONVECTION_MUSCL: return fmax(fmin(fmin(2.0 * r, 0.5 * r + 0.5), 2.0), 0.0);
        case FY_CONVECTION_MINMOD: return fmax(fmin(fmin(r, 1.0), 2.0), 0.0);
        case FY_CONVECTION_SUPERBEE: return fmax(fmax(fmin(2.0 * r, 1.0), fmin(r, 2.0)), 0.0);
        default: return fmax(fmin((3.0 + r) / 4.0, 2.0), 0.0);          // QUICK
    }
}
// the owner's weight of the convected value on internal face f for the face flux fl, by the scheme
__device__ __forceinline__ double ldu_conv_weight(const LduGeo& g, int f, double fl, const double* __restrict__ U) {
    if (g.upwind == 0) return g.w[f];
    const double up = fl >= 0.0 ? 1.0 : 0.0;
    if (g.upwind <= 2) return up;
    const int o = g.own[f], n = g.nei[f];
    const double gradf = msq(ld3(U, n)) - msq(ld3(U, o));
    const D3 co = ld3(g.C, o), gl = ld3(g.gradL, fl > 0.0 ? o : n);
    D3 cn = ld3(g.C, n);
    if (g.sep) { const D3 sp = ld3(g.sep, f); cn = D3{cn.x + sp.x, cn.y + sp.y, cn.z + sp.z}; }      // (a folded cyclic face: the neighbour's image)
    const double gradcf = ((cn.x - co.x) * gl.x + (cn.y - co.y) * gl.y) + (cn.z - co.z) * gl.z;
    double r;
    if (fabs(gradcf) >= 1000.0 * fabs(gradf)) r = 2.0 * 1000.0 * (gradcf >= 0 ? 1.0 : -1.0) * (gradf >= 0 ? 1.0 : -1.0) - 1.0;
    else r = 2.0 * (gradcf / gradf) - 1.0;
    const double lim = ldu_limiter_fn(g.upwind, g.lim_two_by_k, r);
    return lim * g.w[f] + (1.0 - lim) * up;
}

// Gauss linearUpwind grad(U) [OF-6 linearUpwind::correction]: the face value is the upwind cell's plus (C_f - C_upwind) . grad(U)_upwind, the second term explicit
// (deferred correction) with the Gauss-linear gradient of the iterate the matrix is assembled from; returned as the flux of it, fl (d . grad U)_j
__device__ __forceinline__ void linear_upwind_flux(const LduGeo& g, int f, double fl, const double* __restrict__ gradU, double (&lu)[3]) {
    const int up = fl >= 0.0 ? g.own[f] : g.nei[f];
    const D3 cf = ld3(g.Cf, f);
    D3 cu = ld3(g.C, up);
    if (g.sep && fl < 0.0) { const D3 sp = ld3(g.sep, f); cu = D3{cu.x + sp.x, cu.y + sp.y, cu.z + sp.z}; }      // (a folded cyclic face: the neighbour's image)
    const double d[3] = {cf.x - cu.x, cf.y - cu.y, cf.z - cu.z};
    const double* T = gradU + 9 * (size_t)up;
#pragma unroll
    for (int j = 0; j < 3; ++j) lu[j] = fl * ((d[0] * T[j] + d[1] * T[3 + j]) + d[2] * T[6 + j]);
}

// UEqn, face part: gaussConvectionScheme<linear>::fvmDiv (lower = -w phi, upper = lower + phi) minus gaussLaplacianScheme::fvmLaplacianUncorrected
// (gamma |Sf| nonOrthDeltaCoeffs on both), and the corrected scheme's explicit flux nu |Sf| (k & linearInterpolate(grad U)) per internal face
__global__ __launch_bounds__(256) void k_ldu_mom_faces(LduGeo g, const double* __restrict__ phi, const double* __restrict__ U, const double* __restrict__ gradU, LduMom M, double* __restrict__ corr) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nInt) return;
    const double gm = g.nu * g.magSf[f];
    double lo = -ldu_conv_weight(g, f, phi[f], U) * phi[f];      // (Gauss linear: the linear weight; upwind [OF-6 upwind::weights]: pos0(faceFlux); limited: in between)
    double up = lo + phi[f];
    lo -= gm * g.dcNO[f]; up -= gm * g.dcNO[f];
    M.lower[f] = lo; M.upper[f] = up;
    const D3 k = ld3(g.kvec, f);
    const double kk[3] = {k.x, k.y, k.z};
    const double* To = gradU + 9 * (size_t)g.own[f];
    const double* Tn = gradU + 9 * (size_t)g.nei[f];
    const double w = g.w[f];
    double lu[3] = {0, 0, 0};
    if (g.upwind == 2) linear_upwind_flux(g, f, phi[f], gradU, lu);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double cj = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) cj += kk[i] * (w * To[3 * i + j] + (1.0 - w) * Tn[3 * i + j]);
        corr[3 * (size_t)f + j] = gm * cj - lu[j];
    }
}
// ... cell part: EulerDdtScheme::fvmDdt, negSumDiag, the patches' coefficients (fixedValue: value* 0 / U_b, gradient* -+ deltaCoeffs; zeroGradient: value* 1),
// == uSource, and the divergence of the explicit flux on the right-hand side
__global__ __launch_bounds__(256) void k_ldu_mom_cells(LduGeo g, const double* __restrict__ phi, const double* __restrict__ Uold, const double* __restrict__ uSource, LduMom M,
                                                       const double* __restrict__ corr) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const double Vc = g.V[c], rdt = Vc / g.dt;
    double dg = rdt;
    const D3 uo = ld3(Uold, c), us = ld3(uSource, c);
    double b[3] = {rdt * uo.x + Vc * us.x, rdt * uo.y + Vc * us.y, rdt * uo.z + Vc * us.z};
    double bd[3] = {0, 0, 0}, bmax = 0.0, bmin = 0.0;
    FY_CELL_FACES(g, c, f, nb) {
        if (f < g.nInt) {
            const D3 cr = ld3(corr, f);
            if (nb > c) { dg -= M.lower[f]; b[0] += cr.x; b[1] += cr.y; b[2] += cr.z; }
            else { dg -= M.upper[f]; b[0] -= cr.x; b[1] -= cr.y; b[2] -= cr.z; }
        } else {
            const int pa = g.patch_of[f - g.nInt];
            if (g.u_bc[pa] == FY_BC_U_FIXED_VALUE) {
                const double gm = g.nu * g.magSf[f] * g.dcNO[f];
                const D3 ub = ld3(g.u_val, pa);
                dg += gm;
                b[0] += (-phi[f] + gm) * ub.x; b[1] += (-phi[f] + gm) * ub.y; b[2] += (-phi[f] + gm) * ub.z;
            } else {
                if (g.u_bc[pa] == FY_BC_U_SLIP) slip_face(g, f, g.nu * g.magSf[f] * g.dcNO[f], uo, bd, b, bmax, bmin);      // (UEqn is assembled around U = U.oldTime())
                dg += phi[f];
            }
        }
    }
    M.diag[c] = dg;
    st3(M.b, c, D3{b[0], b[1], b[2]});
    if (M.bdiag) st3(M.bdiag, c, D3{bd[0], bd[1], bd[2]});
}

// row of the momentum matrix applied to x without its diagonal: sum offdiag x_nb; also the row's off-diagonal sum (for A xbar)
__device__ __forceinline__ void mom_offdiag(const LduGeo& g, const LduMom& M, const double* __restrict__ x, int c, double (&s)[3], double* offsum) {
    s[0] = s[1] = s[2] = 0.0;
    double os = 0.0;
    FY_CELL_FACES(g, c, f, nb) {
        if (f >= g.nInt) continue;
        const bool o = nb > c;
        const double a = o ? M.upper[f] : M.lower[f];
        const D3 xn = ld3(x, nb);
        s[0] += a * xn.x; s[1] += a * xn.y; s[2] += a * xn.z;
        os += a;
    }
    if (offsum) *offsum = os;
}
// the momentum predictor's Jacobi pass (solve(UEqn == -fvc::grad(p)), icoFoamYade.C:91-94): residual sums of x and the next iterate in one pass
__global__ __launch_bounds__(256) void k_ldu_mom_pass(LduGeo g, LduMom M, const double* __restrict__ rhs, const double* __restrict__ gradp, const double* __restrict__ x,
                                                      double* __restrict__ xn, const double* __restrict__ xsum3, double* __restrict__ partials) {
    double v[6] = {0, 0, 0, 0, 0, 0};
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.nCells) {
        double s[3], os;
        mom_offdiag(g, M, x, c, s, &os);
        const double dgs = M.diag[c], Vc = g.V[c];
        const D3 b0 = ld3(rhs, c), gp = gradp ? ld3(gradp, c) : D3{0, 0, 0}, xc = ld3(x, c), bdv = M.bdiag ? ld3(M.bdiag, c) : D3{0, 0, 0};
        const double bd[3] = {bdv.x, bdv.y, bdv.z};                  // fvMatrix::solveSegregated: addBoundaryDiag per component
        const double b[3] = {gradp ? b0.x - Vc * gp.x : b0.x, gradp ? b0.y - Vc * gp.y : b0.y, gradp ? b0.z - Vc * gp.z : b0.z}, xx[3] = {xc.x, xc.y, xc.z};
        double o[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double dg = dgs + bd[q];
            const double Ax = dg * xx[q] + s[q];
            const double Aref = (dg + os) * (xsum3[q] / (double)g.nCells);
            v[q] = fabs(b[q] - Ax);
            v[3 + q] = fabs(Ax - Aref) + fabs(b[q] - Aref);
            o[q] = (b[q] - s[q]) / dg;
        }
        st3(xn, c, D3{o[0], o[1], o[2]});
    }
    const int mx[6] = {0, 0, 0, 0, 0, 0};
    block_reduce_store<6>(v, mx, partials);
}

// rAU = 1 / A, HbyA = rAU H (icoFoamYade.C:99-100; constrainHbyA acts on the boundary values, formed where they are used)
__global__ __launch_bounds__(256) void k_ldu_HbyA(LduGeo g, LduMom M, const double* __restrict__ U, double* __restrict__ rAU, double* __restrict__ HbyA) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    double s[3];
    mom_offdiag(g, M, U, c, s, nullptr);
    const double Vc = g.V[c];
    D3 b = ld3(M.b, c);
    double av = 0.0;
    if (M.bdiag) {                       // fvMatrix::A(): addCmptAvBoundaryDiag; fvMatrix::H(): what a component's boundary diagonal has over the average stays with H
        const D3 bd = ld3(M.bdiag, c), uc = ld3(U, c);
        av = (bd.x + bd.y + bd.z) / 3.0;
        b.x += (av - bd.x) * uc.x; b.y += (av - bd.y) * uc.y; b.z += (av - bd.z) * uc.z;
    }
    const double r = 1.0 / ((M.diag[c] + av) / Vc);
    rAU[c] = r;
    st3(HbyA, c, D3{r * ((b.x - s[0]) / Vc), r * ((b.y - s[1]) / Vc), r * ((b.z - s[2]) / Vc)});
}

// phiHbyA = fvc::flux(HbyA) + fvc::interpolate(rAU) fvc::ddtCorr(U, phi) (icoFoamYade.C:101-106) [OF-6 EulerDdtScheme::fvcDdtPhiCorr / fvcDdtPhiCoeff]
__global__ __launch_bounds__(256) void k_ldu_phiHbyA(LduGeo g, const double* __restrict__ HbyA, const double* __restrict__ rAU, const double* __restrict__ Uold,
                                                     const double* __restrict__ phiOld, const double* __restrict__ alphaf, double* __restrict__ rAUf, double* __restrict__ phiHbyA) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    const D3 S = ld3(g.Sf, f);
    double rf, fl, uf;
    bool fixes = false;
    if (f < g.nInt) {
        const int o = g.own[f], n = g.nei[f];
        const double w = g.w[f];
        rf = w * rAU[o] + (1.0 - w) * rAU[n];
        fl = dot3(lerp3(w, ld3(HbyA, o), ld3(HbyA, n)), S);
        uf = dot3(lerp3(w, ld3(Uold, o), ld3(Uold, n)), S);
    } else {
        const int pa = g.patch_of[f - g.nInt];
        fixes = g.u_bc[pa] == FY_BC_U_FIXED_VALUE;
        rf = rAU[g.own[f]];
        fl = dot3(Ub(g, HbyA, f), S);       // constrainHbyA: U's value where the patch fixes it; a symmetry patch keeps its type on HbyA
        uf = dot3(Ub(g, Uold, f), S);
    }
    const double phiCorr = phiOld[f] - uf;
    const double coef = fixes ? 0.0 : 1.0 - fmin(fabs(phiCorr) / (fabs(phiOld[f]) + 1e-15), 1.0);
    rAUf[f] = rf;
    double add = rf * (coef * (1.0 / g.dt) * phiCorr);
    if (alphaf) add *= alphaf[f];                         // pEqn.H:9: alphacf rAUcf ddtCorr(Uc, phic)
    phiHbyA[f] = fl + add;
}

// adjustPhi [OF-6 adjustPhi.C] (icoFoamYade.C:108), only when no patch fixes the pressure: the four sums as block partials (folded by k_reduce_finalize),
// then the boundary faces are scaled
__global__ __launch_bounds__(256) void k_ldu_adjust_sums(LduGeo g, const double* __restrict__ phiHbyA, double* __restrict__ partials) {
    double v[4] = {0, 0, 0, 0};              // massIn, fixedMassOut, adjustableMassOut, sum |internal flux|
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < g.nFaces) {
        const double fl = phiHbyA[f];
        if (f < g.nInt) v[3] = fabs(fl);
        else if (fl < 0.0) v[0] = -fl;
        else if (g.u_bc[g.patch_of[f - g.nInt]] == FY_BC_U_FIXED_VALUE) v[1] = fl;
        else v[2] = fl;
    }
    const int mx[4] = {0, 0, 0, 0};
    block_reduce_store<4>(v, mx, partials);
}
__global__ __launch_bounds__(256) void k_ldu_adjust_apply(LduGeo g, const double* __restrict__ sums, double* __restrict__ phiHbyA, int* __restrict__ err) {
    const int f = g.nInt + blockIdx.x * 256 + threadIdx.x;
    const double massIn = sums[0], fixedOut = sums[1], adjOut = sums[2], total = sums[3] + 1e-300;
    double massCorr = 1.0;
    if (fabs(adjOut) > 1e-300 && fabs(adjOut) / total > 1e-15) massCorr = (massIn - fixedOut) / adjOut;
    else if (fabs(fixedOut - massIn) / total > 1e-8) { if (f == g.nInt) *err = 1; }
    if (f >= g.nFaces || massCorr == 1.0) return;
    if (g.u_bc[g.patch_of[f - g.nInt]] != FY_BC_U_FIXED_VALUE && phiHbyA[f] > 0.0) phiHbyA[f] *= massCorr;
}

// pEqn, face part (icoFoamYade.C:118-121): c_f = rAUf |Sf| nonOrthDeltaCoeffs, and the corrected scheme's explicit flux rAUf |Sf| (k & interpolate(grad p))
// from the pressure as it stands -- what each pass of the correctNonOrthogonal loop (icoFoamYade.C:114-131) renews
// (pt: the face's contribution to the right-hand side, -phiHbyA + the explicit flux, for the cell part: one gather there instead of two)
__global__ __launch_bounds__(256) void k_ldu_p_faces(LduGeo g, const double* __restrict__ rAUf, const double* __restrict__ gradp, const double* __restrict__ phiHbyA,
                                                     double* __restrict__ pcoef, double* __restrict__ pcorr, double* __restrict__ pt) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    const double gm = rAUf[f] * g.magSf[f];
    pcoef[f] = gm * g.dcNO[f];
    double t = -phiHbyA[f];
    if (f < g.nInt) {
        const double pc = gm * dot3(ld3(g.kvec, f), lerp3(g.w[f], ld3(gradp, g.own[f]), ld3(gradp, g.nei[f])));
        pcorr[f] = pc;
        t += pc;
    }
    pt[f] = t;
}
// ... cell part, in the positive form  sum_f c_f (p_P - p_N) + sum_b c_b (p_P - p_b) = -div(phiHbyA) + div(explicit flux);  fvMatrix::setReference
__global__ __launch_bounds__(256) void k_ldu_p_cells(LduGeo g, const double* __restrict__ pt, const double* __restrict__ pcoef,
                                                     double* __restrict__ pdiag, double* __restrict__ prhs) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    double dg = 0.0, b = 0.0;
    FY_CELL_FACES(g, c, f, nb) {
        const double t = pt[f];
        if (f < g.nInt) {
            dg += pcoef[f];
            b += nb > c ? t : -t;
        } else {
            const int pa = g.patch_of[f - g.nInt];
            b += t;
            if (g.p_bc[pa] == FY_BC_P_FIXED_VALUE) { dg += pcoef[f]; b += pcoef[f] * g.p_val[pa]; }
        }
    }
    if (g.need_ref && c == g.p_ref_cell) { b += dg * g.p_ref_value; dg += dg; }
    pdiag[c] = dg; prhs[c] = b;
}

// r = b - A x; slot 0 = sum |r|, slot 1 = sum (|A x - A xbar| + |b - A xbar|)  [OF-6 lduMatrix::solver::normFactor]
// (the row through the ELL form of the matrix -- ell_nbr / ell_coef [W nCells], the same coefficients in the same order, coalesced -- instead of a gather by face number)
__global__ __launch_bounds__(256) void k_ldu_p_init(LduGeo g, const double* __restrict__ pdiag, int ellW, const int32_t* __restrict__ ell_nbr, const double* __restrict__ ell_coef,
                                                    const double* __restrict__ b, const double* __restrict__ x, const double* __restrict__ xsum, double inv_n, double* __restrict__ r,
                                                    double* __restrict__ partials) {
    double v[2] = {0, 0};
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.nCells) {
        double cs = 0.0, off = 0.0;
        for (int k = 0; k < ellW; ++k) {
            const size_t e = (size_t)k * g.nCells + c;
            const int nb = ell_nbr[e];
            if (nb == c) break;
            const double a = ell_coef[e];
            off += a * x[nb];
            cs += a;
        }
        const double Ax = pdiag[c] * x[c] - off, Aref = (pdiag[c] - cs) * (xsum[0] * inv_n);
        const double rr = b[c] - Ax;
        r[c] = rr;
        v[0] = fabs(rr); v[1] = fabs(Ax - Aref) + fabs(b[c] - Aref);
    }
    const int mx[2] = {0, 0};
    block_reduce_store<2>(v, mx, partials);
}
// phi = phiHbyA - pEqn.flux() (icoFoamYade.C:127-130): the matrix's flux c_f (p_N - p_P) and the explicit non-orthogonal flux it was assembled with
__global__ __launch_bounds__(256) void k_ldu_flux_correct(LduGeo g, const double* __restrict__ p, const double* __restrict__ phiHbyA, const double* __restrict__ pcoef,
                                                          const double* __restrict__ pcorr, double* __restrict__ phi) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    if (f < g.nInt) phi[f] = phiHbyA[f] - (pcoef[f] * (p[g.nei[f]] - p[g.own[f]]) + pcorr[f]);
    else {
        const int pa = g.patch_of[f - g.nInt];
        phi[f] = phiHbyA[f] - (g.p_bc[pa] == FY_BC_P_FIXED_VALUE ? pcoef[f] * (g.p_val[pa] - p[g.own[f]]) : 0.0);
    }
}

// U = HbyA - rAU fvc::grad(p) (icoFoamYade.C:136-137) and continuityErrs.H (:134): slot 0 = sum |div phi|, slot 1 = sum div phi
__global__ __launch_bounds__(256) void k_ldu_U_correct(LduGeo g, const double* __restrict__ HbyA, const double* __restrict__ rAU, const double* __restrict__ p,
                                                       const double* __restrict__ phi, double* __restrict__ U, double* __restrict__ partials) {
    double v[2] = {0, 0};
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.nCells) {
        const D3 gp = grad_scalar_at(g, p, c), h = ld3(HbyA, c);
        const double r = rAU[c];
        st3(U, c, D3{h.x - r * gp.x, h.y - r * gp.y, h.z - r * gp.z});
        double dv = 0.0;
        FY_CELL_FACES(g, c, f, nb) {
            dv += (f >= g.nInt || nb > c) ? phi[f] : -phi[f];
        }
        v[0] = fabs(dv); v[1] = dv;
    }
    const int mx[2] = {0, 0};
    block_reduce_store<2>(v, mx, partials);
}


// =====================================================================================================================================
// pimpleFoamYade on the general mesh (pimpleFoamYade.C:60-114, UcEqn.H, pEqn.H): the alpha-weighted equations.  alphac's boundary value is 1
// (FoamYade.C:68), uSource's 0; laminar Stokes stress [OF-6 linearViscousStress::divDevRhoReff]; what restates it on the CPU: oracle/ldu_oracle.cpp
__global__ __launch_bounds__(256) void k_ldu_alphaf(LduGeo g, const double* __restrict__ alpha, double* __restrict__ alphaf) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    alphaf[f] = f < g.nInt ? g.w[f] * alpha[g.own[f]] + (1.0 - g.w[f]) * alpha[g.nei[f]] : 1.0;
}

// the coupling's input fields (pimpleFoamYade.C:73-76): ddtU_f = fvc::ddt(Uc) + fvc::div(phic, Uc) -- the ddt is zero where it is evaluated (DESIGN.md
// section 4, quirk F-Q1's mechanism) --, divT = 2 nu fvc::laplacian(alphac, Uc) with the corrected scheme; gradP and vGrad by the gradient kernels
__global__ __launch_bounds__(256) void k_ldu_pre_coupling(LduGeo g, const double* __restrict__ phi, const double* __restrict__ U, const double* __restrict__ vGrad,
                                                          const double* __restrict__ alphaf, double* __restrict__ ddtU, double* __restrict__ divT) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const D3 uc = ld3(U, c);
    double cv[3] = {0, 0, 0}, lp[3] = {0, 0, 0};
    // this cell's own tensor once, not once per face (the kernel is bound by the number of lines its gathers touch): the face's interpolate
    // w T_owner + (1 - w) T_neighbour is the sum of the same two products whichever of the two this cell is, and a sum does not depend on the order of its terms
    double Tc[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Tc[q] = vGrad[9 * (size_t)c + q];
    FY_CELL_FACES(g, c, f, nb) {
        const double af = alphaf[f], gm = af * g.magSf[f];
        if (f < g.nInt) {
            const bool o = nb > c;
            const D3 un = ld3(U, nb);
            const double w = g.w[f], wc = o ? w : 1.0 - w;                  // the weight of THIS cell's value
            const double fl = o ? phi[f] : -phi[f];
            const double uf[3] = {wc * uc.x + (1.0 - wc) * un.x, wc * uc.y + (1.0 - wc) * un.y, wc * uc.z + (1.0 - wc) * un.z};
            const D3 k = ld3(g.kvec, f);
            const double* Tb = vGrad + 9 * (size_t)nb;
            const double wn = o ? 1.0 - w : w;                              // ... and of the neighbour's
            const double sg = o ? 1.0 : -1.0, du[3] = {un.x - uc.x, un.y - uc.y, un.z - uc.z}, kk[3] = {k.x, k.y, k.z};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double cj = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) cj += kk[i] * (wc * Tc[3 * i + j] + wn * Tb[3 * i + j]);
                cv[j] += fl * uf[j];
                lp[j] += gm * (g.dcNO[f] * du[j] + sg * cj);               // outward normal gradient: the correction vector points owner -> neighbour
            }
        } else {
            const D3 ub = Ub(g, U, f);
            cv[0] += phi[f] * ub.x; cv[1] += phi[f] * ub.y; cv[2] += phi[f] * ub.z;
            lp[0] += gm * g.dcNO[f] * (ub.x - uc.x); lp[1] += gm * g.dcNO[f] * (ub.y - uc.y); lp[2] += gm * g.dcNO[f] * (ub.z - uc.z);
        }
    }
    const double rV = 1.0 / g.V[c];
    st3(ddtU, c, D3{cv[0] * rV, cv[1] * rV, cv[2] * rV});
    st3(divT, c, D3{2 * g.nu * (lp[0] * rV), 2 * g.nu * (lp[1] * rV), 2 * g.nu * (lp[2] * rV)});
}

// UcEqn.H:3-10, face part: fvm::div(alphaPhic, Uc) and - fvm::laplacian(alpha nu, Uc) as lower / upper, the corrected laplacian's explicit flux, and the flux
// of the explicit stress Sf . (alpha nu dev2(T(grad U)))_f (the cell tensor interpolated linearly; a boundary face takes its cell's)
// nut != null (LES Smagorinsky, DPMTurbulenceModels.C:73-74): nuEff = nu + nut in both parts of divDevRhoReff -- the face diffusivity is the linear interpolate of the CELL
// field alpha (nu + nut) [OF-6 gaussLaplacianScheme::fvmLaplacian(vol gamma)]; on the boundary alpha_b (nu + nut_b), nut_b by the patch of 0/nut (ldu_nut_b)
__device__ __forceinline__ double ldu_k_b(const LduGeo& g, const LduPim& P, const double* __restrict__ k, int f) {
    const int pa = g.patch_of[f - g.nInt];
    return P.k_bc[pa] == FY_BC_NUT_FIXED_VALUE ? P.k_val[pa] : k[g.own[f]];
}
__device__ __forceinline__ double ldu_nut_b(const LduGeo& g, const LduPim& P, int f) {
    if (!P.nut) return 0.0;
    const int pa = g.patch_of[f - g.nInt], c = g.own[f];
    const int t = P.nut_bc[pa];
    if (t == FY_BC_NUT_FIXED_VALUE || (t == FY_BC_NUT_CALCULATED && !(P.k && P.nut_live))) return P.nut_val[pa];
    // calculated [OF-6 GeometricField::operator=: nut_ = Ck sqrt(k_) delta | Cmu sqr(k_) / epsilon_ assigns the patches too]
    if (t == FY_BC_NUT_CALCULATED && P.eps) {
        const double kb = ldu_k_b(g, P, P.k, f), eb = P.eps_bc[pa] == FY_BC_NUT_FIXED_VALUE ? P.eps_val[pa] : P.eps[c];
        return P.cmu * (kb * kb) / eb;
    }
    if (t == FY_BC_NUT_CALCULATED) return P.ck * sqrt(ldu_k_b(g, P, P.k, f)) * (P.delta_coeff * cbrt(g.V[c]));
    return P.nut[c];
}
__global__ __launch_bounds__(256) void k_ldu_pmom_faces(LduGeo g, LduPim P, const double* __restrict__ phi, const double* __restrict__ U, const double* __restrict__ alpha, const double* __restrict__ alphaf,
                                                        const double* __restrict__ gradU, LduMom M, double* __restrict__ aphi, double* __restrict__ fstress) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    const D3 S = ld3(g.Sf, f);
    const double ss[3] = {S.x, S.y, S.z};
    const int o = g.own[f];
    const double* To = gradU + 9 * (size_t)o;
    const double tro = To[0] + To[4] + To[8], ao = alpha[o] * (g.nu + (P.nut ? P.nut[o] : 0.0));
    if (f >= g.nInt) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            double t = 0.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) t += ss[a] * (ao * (To[3 * b + a] - (a == b ? (2.0 / 3.0) * tro : 0.0)));
            fstress[3 * (size_t)f + b] = t;
        }
        aphi[f] = phi[f];                                    // (alphac's boundary value is 1)
        return;
    }
    const int n = g.nei[f];
    const double* Tn = gradU + 9 * (size_t)n;
    const double trn = Tn[0] + Tn[4] + Tn[8], an = alpha[n] * (g.nu + (P.nut ? P.nut[n] : 0.0)), w = g.w[f];
    const double af = alphaf[f], fl = af * phi[f], gm = (P.nut ? w * ao + (1.0 - w) * an : g.nu * af) * g.magSf[f];
    aphi[f] = fl;                                            // alphaPhic: the cell part gathers this ONE face array instead of alphaf and phi
    double lo = -ldu_conv_weight(g, f, fl, U) * fl;
    double up = lo + fl;
    lo -= gm * g.dcNO[f]; up -= gm * g.dcNO[f];
    M.lower[f] = lo; M.upper[f] = up;
    const D3 k = ld3(g.kvec, f);
    const double kk[3] = {k.x, k.y, k.z};
    double lu[3] = {0, 0, 0};
    if (g.upwind == 2) linear_upwind_flux(g, f, fl, gradU, lu);
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        double cj = 0.0, t = 0.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            cj += kk[a] * (w * To[3 * a + b] + (1.0 - w) * Tn[3 * a + b]);
            const double go = ao * (To[3 * b + a] - (a == b ? (2.0 / 3.0) * tro : 0.0)), gn = an * (Tn[3 * b + a] - (a == b ? (2.0 / 3.0) * trn : 0.0));
            t += ss[a] * (w * go + (1.0 - w) * gn);
        }
        fstress[3 * (size_t)f + b] = (gm * cj - lu[b]) + t;      // the two explicit fluxes of a face enter every sum together: one vector
    }
}
// ... cell part: fvm::ddt(alphac, Uc), negSumDiag, the patches, - fvm::Sp(fvc::ddt(alphac) + fvc::div(alphaPhic)), == fvm::Sp(uSourceDrag), the explicit fluxes'
// divergences on the right-hand side, UcEqn.relax() [OF-6 fvMatrix::relax: D = max(|D|, sum |offdiag|) / factor, source += (D_new - D) psi; no factor: nothing]
__global__ __launch_bounds__(256) void k_ldu_pmom_cells(LduGeo g, LduPim P, const double* __restrict__ phi, const double* __restrict__ alpha, const double* __restrict__ alphaOld,
                                                        const double* __restrict__ alphaf, const double* __restrict__ Uold, const double* __restrict__ U,
                                                        const double* __restrict__ uSourceDrag, LduMom M, const double* __restrict__ aphi, const double* __restrict__ fstress,
                                                        double u_relax, double* __restrict__ rAU) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const double Vc = g.V[c], rdt = Vc / g.dt;
    double dg = alpha[c] * rdt;
    const D3 uo = ld3(Uold, c);
    double b[3] = {alphaOld[c] * rdt * uo.x, alphaOld[c] * rdt * uo.y, alphaOld[c] * rdt * uo.z};
    double divAPhi = 0.0, offsum = 0.0;
    double bd[3] = {0, 0, 0}, bmax = 0.0, bmin = 0.0;
    const D3 uc = ld3(U, c);
    FY_CELL_FACES(g, c, f, nb) {
        const D3 st = ld3(fstress, f);                          // (internal faces: the corrected laplacian's explicit flux, linearUpwind's and the explicit stress's, summed by the face kernel)
        const double fl = aphi[f];
        if (f < g.nInt) {
            if (nb > c) { dg -= M.lower[f]; offsum += fabs(M.upper[f]); divAPhi += fl; b[0] += st.x; b[1] += st.y; b[2] += st.z; }
            else { dg -= M.upper[f]; offsum += fabs(M.lower[f]); divAPhi -= fl; b[0] -= st.x; b[1] -= st.y; b[2] -= st.z; }
        } else {
            const int pa = g.patch_of[f - g.nInt];
            divAPhi += fl;                                     // (= phi on a boundary face)
            b[0] += st.x; b[1] += st.y; b[2] += st.z;
            if (g.u_bc[pa] == FY_BC_U_FIXED_VALUE) {
                const double gm = (g.nu + ldu_nut_b(g, P, f)) * g.magSf[f] * g.dcNO[f];
                const D3 ub = ld3(g.u_val, pa);
                dg += gm;
                b[0] += (-fl + gm) * ub.x; b[1] += (-fl + gm) * ub.y; b[2] += (-fl + gm) * ub.z;
            } else {
                if (g.u_bc[pa] == FY_BC_U_SLIP) slip_face(g, f, (g.nu + ldu_nut_b(g, P, f)) * g.magSf[f] * g.dcNO[f], uc, bd, b, bmax, bmin);
                dg += fl;
            }
        }
    }
    const double S = (alpha[c] - alphaOld[c]) / g.dt + divAPhi / Vc;
    dg -= Vc * S;
    dg -= Vc * uSourceDrag[c];
    if (u_relax > 0) {
        // (a symmetry face's coefficient differs by component: relax() adds cmptMax(cmptMag(internalCoeffs)) before the dominance test and takes cmptMin off afterwards)
        const double dn = fmax(fabs(dg + bmax), offsum) / u_relax - bmin;
        b[0] += (dn - dg) * uc.x; b[1] += (dn - dg) * uc.y; b[2] += (dn - dg) * uc.z;
        dg = dn;
    }
    M.diag[c] = dg;
    st3(M.b, c, D3{b[0], b[1], b[2]});
    if (M.bdiag) st3(M.bdiag, c, D3{bd[0], bd[1], bd[2]});
    rAU[c] = 1.0 / ((dg + (bd[0] + bd[1] + bd[2]) / 3.0) / Vc);
}

// continuousPhaseTurbulence->correct() (pimpleFoamYade.C:101-104) for LESModel Smagorinsky [OF-6 Smagorinsky.C: k(gradU), correctNut()], as fv_kernels.hip's
// k_smagorinsky_nut: D = symm(grad U); a = Ce / delta; b = 2/3 tr D; c = 2 Ck delta (dev D && D); k = sqr((-b + sqrt(b^2 + 4 a c)) / 2a); nut = Ck delta sqrt(k);
// delta = deltaCoeff cbrt(V) per cell (cubeRootVolDelta)
__global__ __launch_bounds__(256) void k_ldu_smagorinsky_nut(LduGeo g, const double* __restrict__ vGrad, double ck, double ce, double delta_coeff, double* __restrict__ nut) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const double delta = delta_coeff * cbrt(g.V[c]);
    const double* T = vGrad + 9 * (size_t)c;
    const double Dxx = T[0], Dyy = T[4], Dzz = T[8];
    const double Dxy = 0.5 * (T[1] + T[3]), Dxz = 0.5 * (T[2] + T[6]), Dyz = 0.5 * (T[5] + T[7]);
    const double trD = Dxx + Dyy + Dzz;
    const double a = ce / delta, b = (2.0 / 3.0) * trD, third = (1.0 / 3.0) * trD;
    const double dd = (Dxx - third) * Dxx + (Dyy - third) * Dyy + (Dzz - third) * Dzz + 2.0 * (Dxy * Dxy) + 2.0 * (Dxz * Dxz) + 2.0 * (Dyz * Dyz);
    const double cc = 2.0 * ck * delta * dd;
    const double r = (-b + sqrt(b * b + 4.0 * a * cc)) / (2.0 * a);
    nut[c] = ck * delta * sqrt(r * r);
}

// ---- LESModel kEqn [OF-6 LES/kEqn/kEqn.C correct()] (DPMTurbulenceModels.C:76-77), as fv_kernels.hip's transport kernel with the faces of a general mesh:
//   fvm::ddt(alpha, k) + fvm::div(alphaPhi, k) - fvm::laplacian(alpha DkEff, k) == alpha G - fvm::SuSp(2/3 alpha divU, k) - fvm::Sp(Ce alpha sqrt(k) / delta, k)
//   DkEff = nut + nu; G = nut (gradU && dev(twoSymm(gradU))); divU = fvc::div(phi); Gauss linear corrected laplacian: its explicit part (alpha DkEff)_f |Sf| (k & interpolate(grad k))
// The matrix goes into the momentum matrix's arrays (free once the correctors are done) and is solved as component 0 of a three-component system by the momentum passes
__device__ __forceinline__ double ldu_x_b(const LduGeo& g, const LduKEqn& K, const double* __restrict__ x, int f) {
    const int pa = g.patch_of[f - g.nInt];
    return K.x_bc[pa] == FY_BC_NUT_FIXED_VALUE ? K.x_val[pa] : x[g.own[f]];
}
__global__ __launch_bounds__(256) void k_ldu_grad_k(LduGeo g, LduPim P, LduKEqn K, double* __restrict__ gk) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const double kc = K.X[c];
    D3 a{0, 0, 0};
    FY_CELL_FACES(g, c, f, nb) {
        double kf, sg = 1.0;
        if (f < g.nInt) { const bool o = nb > c; const double kn = K.X[nb]; kf = o ? g.w[f] * kc + (1.0 - g.w[f]) * kn : g.w[f] * kn + (1.0 - g.w[f]) * kc; sg = o ? 1.0 : -1.0; }
        else kf = ldu_x_b(g, K, K.X, f);
        const D3 S = ld3(g.Sf, f);
        a.x += sg * S.x * kf; a.y += sg * S.y * kf; a.z += sg * S.z * kf;
    }
    const double rV = 1.0 / g.V[c];
    st3(gk, c, D3{a.x * rV, a.y * rV, a.z * rV});
}
__global__ __launch_bounds__(256) void k_ldu_k_faces(LduGeo g, LduPim P, LduKEqn K, const double* __restrict__ phi, const double* __restrict__ gk, LduMom M, double* __restrict__ corr) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nInt) return;
    const int o = g.own[f], n = g.nei[f];
    const double w = g.w[f], fl = P.alphaf[f] * phi[f];
    const double gam = (w * (P.alpha[o] * (g.nu + P.nut[o] / K.sigma)) + (1.0 - w) * (P.alpha[n] * (g.nu + P.nut[n] / K.sigma))) * g.magSf[f];
    const double wc = K.upwind ? (fl >= 0.0 ? 1.0 : 0.0) : w;
    double lo = -wc * fl, up = lo + fl;
    lo -= gam * g.dcNO[f]; up -= gam * g.dcNO[f];
    M.lower[f] = lo; M.upper[f] = up;
    const D3 kv = ld3(g.kvec, f), go = ld3(gk, o), gn = ld3(gk, n);
    corr[f] = gam * dot3(kv, lerp3(w, go, gn));
}
__global__ __launch_bounds__(256) void k_ldu_k_cells(LduGeo g, LduPim P, LduKEqn K, const double* __restrict__ phi, const double* __restrict__ vGrad, LduMom M,
                                                     const double* __restrict__ corr, double* __restrict__ x3) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const double Vc = g.V[c], ac = P.alpha[c], xc = K.X[c];
    double dg = ac * Vc / g.dt, b = ac * Vc / g.dt * xc, sumPhi = 0.0, offsum = 0.0;
    FY_CELL_FACES(g, c, f, nb) {
        if (f < g.nInt) {
            if (nb > c) { dg -= M.lower[f]; offsum += fabs(M.upper[f]); sumPhi += phi[f]; b += corr[f]; }
            else { dg -= M.upper[f]; offsum += fabs(M.lower[f]); sumPhi -= phi[f]; b -= corr[f]; }
        } else {
            const int pa = g.patch_of[f - g.nInt];
            const double fl = P.alphaf[f] * phi[f];
            sumPhi += phi[f];
            if (K.x_bc[pa] == FY_BC_NUT_FIXED_VALUE) {
                const double gb = (g.nu + ldu_nut_b(g, P, f) / K.sigma) * g.magSf[f] * g.dcNO[f];
                dg += gb; b += (-fl + gb) * K.x_val[pa];
            } else dg += fl;
        }
    }
    const double* T = vGrad + 9 * (size_t)c;
    const double tr2 = 2.0 * (T[0] + T[4] + T[8]);
    double GG = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 3; ++q) GG += T[3 * a + q] * ((T[3 * a + q] + T[3 * q + a]) - (a == q ? (1.0 / 3.0) * tr2 : 0.0));
    const double G = P.nut[c] * GG, divU = sumPhi / Vc;
    // mode 0 (kEqn's k): Su = alpha G, c1 = 2/3 alpha divU, c2 = Ce alpha sqrt(k) / delta; mode 1 (epsilon): Su = C1 alpha G eps / k, c1 = (2/3 C1 - C3) alpha divU, c2 = C2 alpha eps / k;
    // mode 2 (kEpsilon's k): Su = alpha G, c1 = 2/3 alpha divU, c2 = alpha eps / k
    double Su, c1, c2;
    if (K.mode == 0) { Su = ac * G; c1 = (2.0 / 3.0) * ac * divU; c2 = K.ce * ac * sqrt(xc) / (P.delta_coeff * cbrt(Vc)); }
    else if (K.mode == 1) { const double kc = P.k[c]; Su = K.c1 * ac * G * xc / kc; c1 = ((2.0 / 3.0) * K.c1 - K.c3) * ac * divU; c2 = K.c2 * ac * xc / kc; }
    else { Su = ac * G; c1 = (2.0 / 3.0) * ac * divU; c2 = ac * P.eps[c] / xc; }
    dg += Vc * (fmax(c1, 0.0) + c2);                        // fvm::SuSp: the positive part implicit, the negative part on the source; fvm::Sp
    b += Vc * Su - Vc * fmin(c1, 0.0) * xc;
    if (K.relax > 0) {                                      // fvMatrix::relax
        const double dn = fmax(fabs(dg), offsum) / K.relax;
        b += (dn - dg) * xc;
        dg = dn;
    }
    M.diag[c] = dg;
    st3(M.b, c, D3{b, 0.0, 0.0});
    st3(x3, c, D3{xc, 0.0, 0.0});
}
// bound(k, kMin) [OF-6 bound.C: k = max(max(k, fvc::average(max(k, kMin)) pos0(-k)), kMin), fvc::average = sum |Sf| k_f / sum |Sf|], then correctNut(): nut = Ck sqrt(k) delta
__global__ __launch_bounds__(256) void k_ldu_k_bound_nut(LduGeo g, LduPim P, LduKEqn K, const double* __restrict__ x3, double* __restrict__ X, double* __restrict__ nut) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const double kMin = 1e-15, xc = x3[3 * (size_t)c];
    double xb = xc;
    if (!(xc > 0.0)) {
        double av = 0.0, asum = 0.0;
        FY_CELL_FACES(g, c, f, nb) {
            double xf;
            if (f < g.nInt) {
                const double mo = fmax(xc, kMin), mn = fmax(x3[3 * (size_t)nb], kMin);
                xf = nb > c ? g.w[f] * mo + (1.0 - g.w[f]) * mn : g.w[f] * mn + (1.0 - g.w[f]) * mo;
            } else {
                const int pa = g.patch_of[f - g.nInt];
                xf = fmax(K.x_bc[pa] == FY_BC_NUT_FIXED_VALUE ? K.x_val[pa] : xc, kMin);
            }
            av += g.magSf[f] * xf; asum += g.magSf[f];
        }
        xb = fmax(xc, av / asum);
    }
    const double xn = fmax(xb, kMin);
    X[c] = xn;
    if (K.mode == 0) nut[c] = P.ck * sqrt(xn) * (P.delta_coeff * cbrt(g.V[c]));
    else if (K.mode == 2) nut[c] = P.cmu * (xn * xn) / P.eps[c];            // correctNut() after the k equation, with the epsilon of this correct()
}

// rAUcf = interpolate(rAUc) (boundary: the cell's), phicForces = fvc::flux(rAUc uSource) + rAUcf (g & Sf) (UcEqn.H:15-20)
__global__ __launch_bounds__(256) void k_ldu_forces(LduGeo g, const double* __restrict__ rAU, const double* __restrict__ uSource, double gx, double gy, double gz,
                                                    double* __restrict__ rAUf, double* __restrict__ phiForces) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    const D3 S = ld3(g.Sf, f);
    double rf, fl = 0.0;
    if (f < g.nInt) {
        const int o = g.own[f], n = g.nei[f];
        const double w = g.w[f], ro = rAU[o], rn = rAU[n];
        const D3 so = ld3(uSource, o), sn = ld3(uSource, n);
        rf = w * ro + (1.0 - w) * rn;
        fl = dot3(D3{w * (ro * so.x) + (1.0 - w) * (rn * sn.x), w * (ro * so.y) + (1.0 - w) * (rn * sn.y), w * (ro * so.z) + (1.0 - w) * (rn * sn.z)}, S);
    } else rf = rAU[g.own[f]];
    rAUf[f] = rf;
    phiForces[f] = fl + rf * dot3(D3{gx, gy, gz}, S);
}

// the face field the momentum predictor reconstructs (UcEqn.H:26-31): phicForces / rAUcf - snGrad(p) |Sf|, snGrad with the non-orthogonal correction
__global__ __launch_bounds__(256) void k_ldu_ssf_predictor(LduGeo g, const double* __restrict__ phiForces, const double* __restrict__ rAUf, const double* __restrict__ p,
                                                           const double* __restrict__ gradp, double* __restrict__ ssf) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    double sn;
    if (f < g.nInt) {
        const int o = g.own[f], n = g.nei[f];
        sn = g.dcNO[f] * (p[n] - p[o]) + dot3(ld3(g.kvec, f), lerp3(g.w[f], ld3(gradp, o), ld3(gradp, n)));
    } else sn = g.dcNO[f] * (pbv(g, p, f) - p[g.own[f]]);
    ssf[f] = phiForces[f] / rAUf[f] - sn * g.magSf[f];
}

// out = base + scale fvc::reconstruct(ssf) [OF-6 fvcReconstruct.C: inv(surfaceSum(Sf Sf / |Sf|)) & surfaceSum(Sf / |Sf| ssf)]; the inverse tensors are the mesh's
// (LduGeo::recon).  scale = V (the predictor's right-hand side, base = the matrix source) or rAUc (pEqn.H:43-45, base = HbyA)
__global__ __launch_bounds__(256) void k_ldu_reconstruct(LduGeo g, const double* __restrict__ ssf, const double* __restrict__ base, const double* __restrict__ scale,
                                                         double* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= g.nCells) return;
    const size_t n = (size_t)g.nCells, wn = (size_t)g.Wall * n;
    double a[3] = {0, 0, 0};                                   // recon_c . sum_f (Sf / |Sf|) ssf_f, the tensor folded into the slot coefficients (LduGeo::rT)
    for (int k = 0; k < g.Wall; ++k) {
        const size_t e = (size_t)k * n + c;
        const int f = g.ef[e];
        if (f < 0) break;
        const double t = ssf[f];
        a[0] += g.rT[e] * t; a[1] += g.rT[wn + e] * t; a[2] += g.rT[2 * wn + e] * t;
    }
    const D3 b = ld3(base, c);
    const double sc = scale[c];
    st3(out, c, D3{b.x + sc * a[0], b.y + sc * a[1], b.z + sc * a[2]});
}

// pEqn.H:4-21 after adjustPhi: phiHbyA += phicForces; constrainPressure: snGrad(p) = (phiHbyA - Sf . U_b) / (|Sf| rAUcf) on the fixedFluxPressure faces
__global__ __launch_bounds__(256) void k_ldu_add_forces_constrain(LduGeo g, const double* __restrict__ phiForces, const double* __restrict__ rAUf, const double* __restrict__ U,
                                                                  double* __restrict__ phiHbyA, double* __restrict__ psn) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    const double v = phiHbyA[f] + phiForces[f];
    phiHbyA[f] = v;
    if (f >= g.nInt) {
        const int pa = g.patch_of[f - g.nInt];
        psn[f - g.nInt] = g.p_bc[pa] == FY_BC_P_FIXED_FLUX ? (v - dot3(Ub(g, U, f), ld3(g.Sf, f))) / (g.magSf[f] * rAUf[f]) : 0.0;
    }
}

// what the pressure assembly kernels of the point-force solver take: the diffusivity alphacf rAUcf, and the flux alphacf phiHbyA whose divergence is the source
// (a fixedFluxPressure face contributes its fixed-gradient flux: what is left is Sf . U_b)
__global__ __launch_bounds__(256) void k_ldu_pim_pfaces(LduGeo g, const double* __restrict__ alphaf, const double* __restrict__ rAUf, const double* __restrict__ phiHbyA,
                                                        const double* __restrict__ psn, double* __restrict__ arAUf, double* __restrict__ phiA) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    const double af = alphaf[f];
    arAUf[f] = af * rAUf[f];
    double ph = af * phiHbyA[f];
    if (f >= g.nInt && g.p_bc[g.patch_of[f - g.nInt]] == FY_BC_P_FIXED_FLUX) ph = af * (phiHbyA[f] - rAUf[f] * g.magSf[f] * psn[f - g.nInt]);
    phiA[f] = ph;
}
// fvc::ddt(alphac) on the right-hand side of pEqn (pEqn.H:30)
__global__ __launch_bounds__(256) void k_ldu_prhs_ddt_alpha(LduGeo g, const double* __restrict__ alpha, const double* __restrict__ alphaOld, double* __restrict__ prhs) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.nCells) prhs[c] -= g.V[c] * (alpha[c] - alphaOld[c]) / g.dt;
}

// phic = phiHbyA - pEqn.flux() / alphacf (pEqn.H:39) and the face field of the velocity correction (phicForces - pEqn.flux() / alphacf) / rAUcf (pEqn.H:43-45)
__global__ __launch_bounds__(256) void k_ldu_pim_flux(LduGeo g, const double* __restrict__ p, const double* __restrict__ phiHbyA, const double* __restrict__ pcoef,
                                                      const double* __restrict__ pcorr, const double* __restrict__ alphaf, const double* __restrict__ rAUf,
                                                      const double* __restrict__ phiForces, const double* __restrict__ psn, double* __restrict__ phi, double* __restrict__ ssf,
                                                      double* __restrict__ aphi) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= g.nFaces) return;
    double pf;                                         // pEqn.flux() in the sign of the positive-definite form: c_f (p_N - p_P) + the explicit part
    if (f < g.nInt) pf = pcoef[f] * (p[g.nei[f]] - p[g.own[f]]) + pcorr[f];
    else {
        const int pa = g.patch_of[f - g.nInt];
        pf = g.p_bc[pa] == FY_BC_P_FIXED_VALUE ? pcoef[f] * (g.p_val[pa] - p[g.own[f]]) : (g.p_bc[pa] == FY_BC_P_FIXED_FLUX ? alphaf[f] * rAUf[f] * g.magSf[f] * psn[f - g.nInt] : 0.0);
    }
    const double q = pf / alphaf[f];
    const double ph = phiHbyA[f] - q;
    phi[f] = ph;
    aphi[f] = alphaf[f] * ph;                          // alphacf phic, for continuityErrs.H's cell sums (one gather there)
    ssf[f] = (phiForces[f] - q) / rAUf[f];
}

// continuityErrs.H of pimpleFoamYade: fvc::ddt(alphac) + fvc::div(alphacf phic); slot 0 sum |.| V, slot 1 sum . V
__global__ __launch_bounds__(256) void k_ldu_pim_continuity(LduGeo g, const double* __restrict__ aphi, const double* __restrict__ alpha,
                                                            const double* __restrict__ alphaOld, double* __restrict__ partials) {
    double v[2] = {0, 0};
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < g.nCells) {
        double dv = 0.0;
        FY_CELL_FACES(g, c, f, nb) dv += (f >= g.nInt || nb > c) ? aphi[f] : -aphi[f];
        const double ce = dv / g.V[c] + (alpha[c] - alphaOld[c]) / g.dt;
        v[0] = fabs(ce) * g.V[c]; v[1] = ce * g.V[c];
    }
    const int mx[2] = {0, 0};
    block_reduce_store<2>(v, mx, partials);
}

__global__ __launch_bounds__(256) void k_ldu_sum(const double* __restrict__ x, int n, int ncomp, double* __restrict__ partials) {
    double v[3] = {0, 0, 0};
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < n) for (int q = 0; q < ncomp; ++q) v[q] = x[(size_t)ncomp * c + q];
    const int mx[3] = {0, 0, 0};
    block_reduce_store<3>(v, mx, partials);
}

// mesh.findCell (FoamYade.C:251) on a general mesh of convex cells: from the cell whose centre is nearest, step across the face the point lies
// furthest outside of until no face has it outside (tolerance 1e-10 of the cell's size); a boundary face on the way = outside the mesh
__global__ __launch_bounds__(256) void k_ldu_find_cell(LduGeo g, const double* __restrict__ rec, int rec_len, int64_t n, const int32_t* __restrict__ hint, int32_t* __restrict__ cell_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = rec + (size_t)rec_len * (size_t)i;
    const D3 x{r[0], r[1], r[2]};
    int c = hint[i];
    if (!(x.x == x.x) || !(x.y == x.y) || !(x.z == x.z)) c = -1;
    for (int hop = 0; c >= 0 && hop < 64; ++hop) {
        // step across the INTERNAL face the point is furthest outside of; the point is outside the mesh only when every face it violates is a
        // boundary (or cyclic) face -- on a curved or concave boundary the extended plane of the hint cell's wall face can be the worst offender
        // while an internal face still leads to the containing cell
        double worst = 0.0, worst_b = 0.0;
        int wf = -1, wn = -1;
        const double tol = 1e-10 * cbrt(g.V[c]);
        FY_CELL_FACES(g, c, f, nb) {
            const D3 S = ld3(g.Sf, f);
            D3 cf = ld3(g.Cf, f);
            const double sg = (f >= g.nInt || nb > c) ? 1.0 : -1.0;
            const bool folded = g.sep && f >= g.nIntReal && f < g.nInt;
            if (folded && sg < 0) { const D3 sp = ld3(g.sep, f); cf = D3{cf.x - sp.x, cf.y - sp.y, cf.z - sp.z}; }      // (seen from the neighbour, the face lies at that cell's own half)
            const double s = sg * dot3(D3{x.x - cf.x, x.y - cf.y, x.z - cf.z}, S) / g.magSf[f];
            const int to = folded ? -1 : nb;                                  // (beyond a cyclic half = outside the mesh, as mesh.findCell has it)
            if (to >= 0) { if (s > worst) { worst = s; wf = f; wn = to; } }
            else if (s > worst_b) worst_b = s;
        }
        if (wf < 0 || worst <= tol) {               // no internal face violated: inside this cell, or beyond a boundary face of it
            if (worst_b > tol) c = -1;
            break;
        }
        c = wn;
        if (hop == 63) c = -1;
    }
    cell_out[i] = c;
}

__global__ __launch_bounds__(256) void k_ldu_positions(const double* __restrict__ rec, int rec_len, int64_t n, double* __restrict__ pos3) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = rec + (size_t)rec_len * (size_t)i;
    pos3[3 * (size_t)i] = r[0]; pos3[3 * (size_t)i + 1] = r[1]; pos3[3 * (size_t)i + 2] = r[2];
}

#define FY_LAUNCH_CHECK()                                                                                     \
    do {                                                                                                      \
        hipError_t _e = hipGetLastError();                                                                    \
        if (_e != hipSuccess) return fail(FY_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

}  // namespace

int ldu_red_blocks(int n) { return red_blocks(n); }

int launch_ldu_flux_of(hipStream_t s, LduGeo g, const double* F, double* phi) {
    hipLaunchKernelGGL(k_ldu_flux_of, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, F, phi);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_courant(hipStream_t s, LduGeo g, const double* phi, double* partials) {
    hipLaunchKernelGGL(k_ldu_courant, dim3(red_blocks(g.nCells)), dim3(256), 0, s, g, phi, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_grad_vec(hipStream_t s, LduGeo g, const double* F, double* T) {
    hipLaunchKernelGGL(k_ldu_grad_vec, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, F, T);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_slot_coefs(hipStream_t s, LduGeo g, double* gB, double* gG0, double* rT) {
    hipLaunchKernelGGL(k_ldu_slot_coefs, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, gB, gG0, rT);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_grad_scalar(hipStream_t s, LduGeo g, const double* p, double* gp) {
    hipLaunchKernelGGL(k_ldu_grad_scalar, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, p, gp);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_grad_magsqr(hipStream_t s, LduGeo g, const double* U, double* gradL) {
    hipLaunchKernelGGL(k_ldu_grad_magsqr, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, U, gradL);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_assemble_momentum(hipStream_t s, LduGeo g, const double* phi, const double* Uold, const double* uSource, const double* gradU, LduMom M, double* face_corr) {
    if (g.nInt > 0) hipLaunchKernelGGL(k_ldu_mom_faces, dim3(div_up(g.nInt, 256)), dim3(256), 0, s, g, phi, Uold, gradU, M, face_corr);      // (U has not been written since runTime++: the limiter sees the current field)
    hipLaunchKernelGGL(k_ldu_mom_cells, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, phi, Uold, uSource, M, face_corr);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_mom_pass(hipStream_t s, LduGeo g, LduMom M, const double* rhs, const double* gradp, const double* x, double* xn, const double* xsum3, double* partials) {
    hipLaunchKernelGGL(k_ldu_mom_pass, dim3(red_blocks(g.nCells)), dim3(256), 0, s, g, M, rhs, gradp, x, xn, xsum3, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_HbyA(hipStream_t s, LduGeo g, LduMom M, const double* U, double* rAU, double* HbyA) {
    hipLaunchKernelGGL(k_ldu_HbyA, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, M, U, rAU, HbyA);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_phiHbyA(hipStream_t s, LduGeo g, const double* HbyA, const double* rAU, const double* Uold, const double* phiOld, const double* alphaf, double* rAUf, double* phiHbyA) {
    hipLaunchKernelGGL(k_ldu_phiHbyA, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, HbyA, rAU, Uold, phiOld, alphaf, rAUf, phiHbyA);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_adjust_phi(hipStream_t s, LduGeo g, double* phiHbyA, double* sums4, int* err, double* partials) {
    hipLaunchKernelGGL(k_ldu_adjust_sums, dim3(red_blocks(g.nFaces)), dim3(256), 0, s, g, phiHbyA, partials);
    FY_LAUNCH_CHECK();
    FY_TRY(launch_reduce_finalize(s, partials, g.nFaces, 4, nullptr, sums4, nullptr, 0));
    if (g.nFaces == g.nInt) return FY_OK;                 // (every side cyclic: no boundary face to adjust)
    hipLaunchKernelGGL(k_ldu_adjust_apply, dim3(div_up(g.nFaces - g.nInt, 256)), dim3(256), 0, s, g, sums4, phiHbyA, err);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_assemble_pressure(hipStream_t s, LduGeo g, const double* rAUf, const double* phiHbyA, const double* gradp, double* pcoef, double* pcorr, double* pt, double* pdiag, double* prhs) {
    hipLaunchKernelGGL(k_ldu_p_faces, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, rAUf, gradp, phiHbyA, pcoef, pcorr, pt);
    hipLaunchKernelGGL(k_ldu_p_cells, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, pt, pcoef, pdiag, prhs);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_p_init(hipStream_t s, LduGeo g, const double* pdiag, int ellW, const int32_t* ell_nbr, const double* ell_coef, const double* b, const double* x, const double* xsum, double inv_n,
                      double* r, double* partials) {
    hipLaunchKernelGGL(k_ldu_p_init, dim3(red_blocks(g.nCells)), dim3(256), 0, s, g, pdiag, ellW, ell_nbr, ell_coef, b, x, xsum, inv_n, r, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_flux_correct(hipStream_t s, LduGeo g, const double* p, const double* phiHbyA, const double* pcoef, const double* pcorr, double* phi) {
    hipLaunchKernelGGL(k_ldu_flux_correct, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, p, phiHbyA, pcoef, pcorr, phi);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_U_correct(hipStream_t s, LduGeo g, const double* HbyA, const double* rAU, const double* p, const double* phi, double* U, double* partials) {
    hipLaunchKernelGGL(k_ldu_U_correct, dim3(red_blocks(g.nCells)), dim3(256), 0, s, g, HbyA, rAU, p, phi, U, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

// ---- pimpleFoamYade
int launch_ldu_alphaf(hipStream_t s, LduGeo g, const double* alpha, double* alphaf) {
    hipLaunchKernelGGL(k_ldu_alphaf, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, alpha, alphaf);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_pre_coupling(hipStream_t s, LduGeo g, const double* phi, const double* U, const double* vGrad, const double* alphaf, double* ddtU, double* divT) {
    hipLaunchKernelGGL(k_ldu_pre_coupling, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, phi, U, vGrad, alphaf, ddtU, divT);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_assemble_momentum_pimple(hipStream_t s, LduGeo g, LduPim P, const double* phi, const double* Uold, const double* U, const double* gradU, LduMom M, double* aphi /* [nF] scratch: alphaPhic */,
                                        double* fstress, double u_relax, double* rAU) {
    hipLaunchKernelGGL(k_ldu_pmom_faces, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, P, phi, U, P.alpha, P.alphaf, gradU, M, aphi, fstress);
    hipLaunchKernelGGL(k_ldu_pmom_cells, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, P, phi, P.alpha, P.alphaOld, P.alphaf, Uold, U, P.uSourceDrag, M, aphi, fstress, u_relax, rAU);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_grad_k(hipStream_t s, LduGeo g, LduPim P, LduKEqn K, double* gk) {
    hipLaunchKernelGGL(k_ldu_grad_k, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, P, K, gk);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_k_assemble(hipStream_t s, LduGeo g, LduPim P, LduKEqn K, const double* phi, const double* vGrad, const double* gk, LduMom M, double* face_corr, double* x3) {
    if (g.nInt > 0) hipLaunchKernelGGL(k_ldu_k_faces, dim3(div_up(g.nInt, 256)), dim3(256), 0, s, g, P, K, phi, gk, M, face_corr);
    hipLaunchKernelGGL(k_ldu_k_cells, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, P, K, phi, vGrad, M, face_corr, x3);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_k_bound_nut(hipStream_t s, LduGeo g, LduPim P, LduKEqn K, const double* x3, double* X, double* nut) {
    hipLaunchKernelGGL(k_ldu_k_bound_nut, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, P, K, x3, X, nut);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_smagorinsky_nut(hipStream_t s, LduGeo g, const double* vGrad, double ck, double ce, double delta_coeff, double* nut) {
    hipLaunchKernelGGL(k_ldu_smagorinsky_nut, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, vGrad, ck, ce, delta_coeff, nut);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_forces(hipStream_t s, LduGeo g, LduPim P, const double* rAU, double* rAUf, double* phiForces) {
    hipLaunchKernelGGL(k_ldu_forces, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, rAU, P.uSource, P.g[0], P.g[1], P.g[2], rAUf, phiForces);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_ssf_predictor(hipStream_t s, LduGeo g, const double* phiForces, const double* rAUf, const double* p, const double* gradp, double* ssf) {
    hipLaunchKernelGGL(k_ldu_ssf_predictor, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, phiForces, rAUf, p, gradp, ssf);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_reconstruct(hipStream_t s, LduGeo g, const double* ssf, const double* base, const double* scale, double* out) {
    hipLaunchKernelGGL(k_ldu_reconstruct, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, ssf, base, scale, out);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_add_forces_constrain(hipStream_t s, LduGeo g, const double* phiForces, const double* rAUf, const double* U, double* phiHbyA, double* psn) {
    hipLaunchKernelGGL(k_ldu_add_forces_constrain, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, phiForces, rAUf, U, phiHbyA, psn);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_pim_pfaces(hipStream_t s, LduGeo g, const double* alphaf, const double* rAUf, const double* phiHbyA, const double* psn, double* arAUf, double* phiA) {
    hipLaunchKernelGGL(k_ldu_pim_pfaces, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, alphaf, rAUf, phiHbyA, psn, arAUf, phiA);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_prhs_ddt_alpha(hipStream_t s, LduGeo g, const double* alpha, const double* alphaOld, double* prhs) {
    hipLaunchKernelGGL(k_ldu_prhs_ddt_alpha, dim3(div_up(g.nCells, 256)), dim3(256), 0, s, g, alpha, alphaOld, prhs);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_pim_flux(hipStream_t s, LduGeo g, const double* p, const double* phiHbyA, const double* pcoef, const double* pcorr, const double* alphaf, const double* rAUf,
                        const double* phiForces, const double* psn, double* phi, double* ssf, double* aphi /* [nF]: alphacf phic */) {
    hipLaunchKernelGGL(k_ldu_pim_flux, dim3(div_up(g.nFaces, 256)), dim3(256), 0, s, g, p, phiHbyA, pcoef, pcorr, alphaf, rAUf, phiForces, psn, phi, ssf, aphi);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_pim_continuity(hipStream_t s, LduGeo g, const double* aphi, const double* alpha, const double* alphaOld, double* partials) {
    hipLaunchKernelGGL(k_ldu_pim_continuity, dim3(red_blocks(g.nCells)), dim3(256), 0, s, g, aphi, alpha, alphaOld, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_sum(hipStream_t s, const double* x, int n, int ncomp, double* partials) {
    if (ncomp < 1 || ncomp > 3) return fail(FY_ERR_INVALID, "launch_ldu_sum: 1 .. 3 components");
    hipLaunchKernelGGL(k_ldu_sum, dim3(red_blocks(n)), dim3(256), 0, s, x, n, ncomp, partials);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_positions(hipStream_t s, const double* rec, int rec_len, int64_t n, double* pos3) {
    if (n > 0) hipLaunchKernelGGL(k_ldu_positions, dim3(div_up(n, 256)), dim3(256), 0, s, rec, rec_len, n, pos3);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_ldu_find_cell(hipStream_t s, LduGeo g, const double* rec, int rec_len, int64_t n, const int32_t* hint, int32_t* cell_out) {
    if (n > 0) hipLaunchKernelGGL(k_ldu_find_cell, dim3(div_up(n, 256)), dim3(256), 0, s, g, rec, rec_len, n, hint, cell_out);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

}  // namespace fy
