// Launchers of the FV-half HIP kernels: the operators that icoFoamYade.C:65-149 / pimpleFoamYade.C:60-114 (+UcEqn.H, pEqn.H,
// CourantNo.H, continuityErrs.H) invoke, restated for a uniform hex block (blockMesh order) as coalesced FP64 stencil kernels.
// No MFMA anywhere: every kernel is bandwidth bound.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fy {

// geometry + boundary conditions, passed by value to every kernel
// z-slab decomposition: a rank owns nz consecutive z-planes of a global block of nzglob planes; every CELL array
// carries gz ghost planes below and above them (storage index c = i + nx*(j + ny*(k + gz)), owned cells are the contiguous range
// [c0, c0 + Nc)).  FACE arrays have no ghosts: the interface face planes are computed redundantly by both neighbours.
// Single domain: gz = 0, c0 = 0, kglob0 = 0, nzglob = nz.
struct FvGeo {
    int nx, ny, nz, Nc;     // local owned extent
    int gz, c0;             // ghost planes per side, storage index of the first owned cell (= gz*nx*ny)
    int kglob0, nzglob;     // global k of the first owned plane, global number of planes
    double dx, Af, V;
    double rdx, rhdx, rV;   // 1/dx, 1/(dx/2), 1/V: the uniform block's deltaCoeffs and reciprocal volume (one FP64 multiply instead of a ~60-cycle divide per use)
    int u_bc[6];            // FY_BC_U_*
    double u_val[6][3];
    int p_bc[6];            // FY_BC_P_*
    double p_val[6];
    int pimple;
    int upwind;             // div(phi,U): 0 Gauss linear, 1 Gauss upwind (first order, bounded), 2 Gauss linearUpwind (upwind + explicit gradient correction)
    double dt, nu;
    // continuousPhaseTurbulence: nut = nullptr is the laminar (Stokes) model -- every kernel then computes exactly what it did without one
    const double* nut;      // [storage cells] eddy viscosity of the Smagorinsky model (k_smagorinsky_nut), ghost planes refreshed by the solver
    int nut_bc[6];          // FY_BC_NUT_*
    double nut_val[6];
    // wall functions: nut_bc == 2 is nutkWallFunction [OF-6 nutkWallFunctionFvPatchScalarField.C]: nut_w = nu (y+ kappa / ln(E y+) - 1) where
    // y+ = Cmu^1/4 y sqrt(k_P) / nu exceeds yPlusLam, else 0 (y = dx/2: the wall cell's centre distance).  The boundary values OpenFOAM keeps
    // are those of the last correctNut(); k does not change in between, so they are evaluated where they are used -- except before the
    // first correct(), when the 0/nut file's value stands (nut_wall_live = 0)
    // nut_bc == 3 is a `calculated` patch: after correctNut() it holds the model's expression evaluated with the boundary values of k (and epsilon):
    // Ck sqrt(k_b) delta (turb_model 2, kEqn) or Cmu k_b^2 / eps_b (3, kEpsilon); k_b / eps_b = the patch value, or the cell's for zeroGradient
    int turb_model; double turb_ck, turb_cmu, turb_delta;
    const double* epsturb;  // [storage cells] epsilon (kEpsilon)
    int k_bc[6], eps_bc[6]; double k_val[6], eps_val[6];
    const double* kturb;    // [storage cells] k of the kEqn / kEpsilon models (nullptr: none)
    int nut_wall_live;
    double wf_cmu25, wf_kappa, wf_E, wf_yPlusLam;
    double u_relax;         // fvMatrix::relax factor of UcEqn for the current outer iteration (<= 0: no relaxationFactors entry, relax() is a no-op)
    double g[3];
    int need_ref, p_ref_cell;
    double p_ref_value;
};

struct Face3 { double* a[3]; };           // +axis oriented face arrays (x: (nx+1)*ny*nz, y: nx*(ny+1)*nz, z: nx*ny*(nz+1))
struct CFace3 { const double* a[3]; };
struct Mom7 { double* diag; double* an[6]; };   // momentum matrix: diag + neighbour coefficient across face 2*d+s

// symmetric 7-point pressure matrix of one multigrid level: (A x)_c = diag_c x_c - sum u_f x_nb, u_* stored at the owner (low) cell
struct PMat {
    int nx, ny, nz, N;      // owned extent of this level
    int c0, ntot;           // storage index of the first owned cell, storage size (owned + ghost planes)
    double *diag, *ux, *uy, *uz;
};

// reducing kernels emit one partial per 256-cell block; the count is rounded up to a multiple of 8 for the XCD-aware block order
inline int red_blocks(int n) { return (((n + 255) / 256) + 7) & ~7; }

inline size_t fv_fsize(const FvGeo& g, int d) {
    return d == 0 ? (size_t)(g.nx + 1) * g.ny * g.nz : d == 1 ? (size_t)g.nx * (g.ny + 1) * g.nz : (size_t)g.nx * g.ny * (g.nz + 1);
}

// ---- reductions: kernels over n cells write per-block partials to scratch[slot*red_blocks(n) + block]; finalize folds them in fixed order
// flag/seq (mapped host memory, optional): flag[slot] = seq is stored, system scope, after out[slot] -- the host may spin on it
int launch_reduce_finalize(hipStream_t s, const double* partials, int n_cells, int nslots, const int* ops /*0 sum,1 max (device)*/, double* out,
                           unsigned long long* flag = nullptr, unsigned long long seq = 0);

// ---- field operators
int launch_flux_of(hipStream_t s, FvGeo g, const double* F, Face3 out);                                   // fvc::flux(F), createPhi
int launch_courant(hipStream_t s, FvGeo g, CFace3 phi, double* partials);                                  // slots 0 (max) 1 (sum)
// Gout != nullptr: also emit the laminar stress tensor alpha nu dev2(T(grad U)); write_vgrad = 0 skips the vGrad store,
// write_pfields = 0 skips gradP / divT (pimple only); ddtU != nullptr: also ddtU_f = fvc::div(phi, U) (pimpleFoamYade.C:73)
int launch_pre_coupling(hipStream_t s, FvGeo g, const double* U, const double* p, const double* alpha, CFace3 psn,
                        double* vGrad, double* gradP, double* divT, double* Gout, int write_vgrad, int write_pfields,
                        CFace3 phi = CFace3{}, double* ddtU = nullptr, double* Uold_out = nullptr);
int launch_interp_alpha(hipStream_t s, FvGeo g, const double* alpha, Face3 alphaf);
// G: three vec3 fields (rows of the tensor) over the whole storage, as k_pre_coupling writes them
int launch_div_G(hipStream_t s, FvGeo g, const double* G, double* divG);
// The transport equations of the two-equation / one-equation closures (DPMTurbulenceModels.C:70-71 RAS kEpsilon, :76-77 LES kEqn):
//   fvm::ddt(alpha, X) + fvm::div(alphaPhi, X) - fvm::laplacian(alpha (nut / sigma + nu), X) == Su - fvm::SuSp(c1, X) - fvm::Sp(c2, X)
// mode 0  kEqn      X = k:    Su = alpha G,            c1 = 2/3 alpha divU,            c2 = Ce alpha sqrt(k) / delta          (sigma = 1)
// mode 1  kEpsilon  X = eps:  Su = C1 alpha G eps / k,  c1 = (2/3 C1 - C3) alpha divU,  c2 = C2 alpha eps / k                  (sigma = sigmaEps)
// mode 2  kEpsilon  X = k:    Su = alpha G,            c1 = 2/3 alpha divU,            c2 = alpha eps / k  (eps already new)  (sigma = sigmak)
// with G = nut (gradU && dev(twoSymm(gradU))), divU = fvc::div(phi).  bc / val: boundary conditions of X (0 zeroGradient, 1 fixedValue);
// upwind: convection scheme of X (0 Gauss linear, 1 Gauss upwind); relax: relaxation factor of the equation (<= 0: none); xmin: bound()
// wall[p] = 1: patch p carries epsilonWallFunction [OF-6 epsilonWallFunctionFvPatchScalarField.C]: in the cells next to it
//   eps = Cmu^3/4 k^3/2 / (kappa y) is IMPOSED (fvMatrix::setValues: the row becomes diag x = diag value, the neighbours' coefficients towards the
//   cell move to their sources) and the production G of both equations is replaced by the corner-weighted wall value
//   (1/W) sum_faces (nut_w + nu) |snGrad U| Cmu^1/4 sqrt(k) / (kappa y)
struct TurbEqn { int mode; double ck, ce, delta, c1, c2, c3, sigma, xmin, relax; int upwind; int bc[6]; double val[6]; int wall[6]; double cmu75, cmu25, kappa; };
// assembles the equation into the momentum matrix's storage (free after the correctors) as a 3-component system whose components 1, 2 are
// identically zero, so that the momentum solver's Jacobi pass solves it: x3 = {X, 0, 0}, b3 = {source, 0, 0}
int launch_assemble_turb(hipStream_t s, FvGeo g, TurbEqn e, const double* k, const double* eps, const double* alpha, CFace3 alphaf, CFace3 phi,
                         const double* vGrad, const double* U, Mom7 M, double* b3, double* x3);
// bound(X, xmin) [OF-6 bound.C] on the solved component 0 of x3 -> X; nut_mode 1: nut = Ck sqrt(k) delta (kEqn::correctNut, X = k),
// 2: nut = Cmu k^2 / eps (kEpsilon::correctNut, X = k, eps given), 0: leave nut alone (the epsilon equation)
int launch_turb_finish(hipStream_t s, FvGeo g, TurbEqn e, const double* x3, double* X, int nut_mode, double cmu, const double* eps, double* nut);
int launch_smagorinsky_nut(hipStream_t s, FvGeo g, const double* vGrad, double ck, double ce, double delta, double* nut);
int launch_assemble_momentum(hipStream_t s, FvGeo g, const double* U, const double* Uold, const double* alpha, const double* alphaOld,
                             CFace3 alphaf, CFace3 phi, const double* uSource, const double* uSourceDrag, const double* divG,
                             const double* vGrad /* grad(U) of the current iterate: linearUpwind only */, Mom7 M, double* src, double* rAU);
int launch_interp_rAU(hipStream_t s, FvGeo g, const double* rAU, Face3 rAUf);
// rAUf = interpolate(rAU) and phiForces (UcEqn.H:15-20) as one cell-centred sweep
// adjustPhi (icoFoamYade.C:108, pEqn.H:13-16): partial sums {massIn, fixedMassOut, adjustableMassOut, sum |internal flux|}; apply scales the
// outflow of the patches that do not fix U (and refreshes snGrad(p) of fixedFluxPressure patches); *err = 1 where OpenFOAM would stop
int launch_adjust_phi_sums(hipStream_t s, FvGeo g, CFace3 phiHbyA, CFace3 phiForces, double* partials);
int launch_adjust_phi_apply(hipStream_t s, FvGeo g, const double* sums, Face3 phiHbyA, CFace3 phiForces, CFace3 rAUf, const double* U, Face3 psn, int* err);
int launch_rAUf_phi_forces(hipStream_t s, FvGeo g, const double* rAU, const double* uSource, Face3 rf, Face3 out);
int launch_bmom(hipStream_t s, FvGeo g, const double* src, const double* p, CFace3 psn, CFace3 phiForces, CFace3 rAUf, double* bmom);
// one fused Jacobi pass: residual sums of x (slots 0..2), norm-factor sums (slots 3..5, uses xbar[3]) and xn = next iterate
int launch_mom_pass(hipStream_t s, FvGeo g, Mom7 M, const double* b, const double* x, double* xn, const double* xsum /* [3] component sums of x (device) */,
                    double n_glob, double* partials);
int launch_sum3(hipStream_t s, const double* x, int n, double* partials);                                  // slots 0..2 = component sums
int launch_HbyA(hipStream_t s, FvGeo g, Mom7 M, const double* src, const double* U, const double* rAU, double* HbyA);
int launch_phiHbyA(hipStream_t s, FvGeo g, const double* HbyA, const double* U, const double* Uold, CFace3 phiOld, CFace3 rAUf,
                   CFace3 alphaf, CFace3 phiForces, Face3 phiHbyA, Face3 psn);
int launch_assemble_pressure(hipStream_t s, FvGeo g, CFace3 phiHbyA, CFace3 rAUf, CFace3 alphaf, CFace3 psn, const double* alpha,
                             const double* alphaOld, PMat A, double* rhs);
int launch_flux_correct(hipStream_t s, FvGeo g, const double* p, CFace3 phiHbyA, CFace3 rAUf, CFace3 alphaf, CFace3 psn, Face3 pflux, Face3 phi);
int launch_cont_err(hipStream_t s, FvGeo g, CFace3 phi, CFace3 alphaf, const double* alpha, const double* alphaOld, double* partials);   // slots 0,1
int launch_U_correct(hipStream_t s, FvGeo g, const double* HbyA, const double* rAU, const double* p, CFace3 psn, CFace3 phiForces,
                     CFace3 pflux, CFace3 alphaf, CFace3 rAUf, double* U);
// the same sweep + continuity errors (slots 0, 1) + next step's Courant sums (slots 2 max, 3 sum) of `phi`
int launch_U_correct_diag(hipStream_t s, FvGeo g, const double* HbyA, const double* rAU, const double* p, CFace3 psn, CFace3 phiForces,
                          CFace3 pflux, CFace3 alphaf, CFace3 rAUf, double* U, CFace3 phi, const double* alpha, const double* alphaOld, double* partials);

// ---- pressure solver building blocks
int launch_p_apply(hipStream_t s, PMat A, const double* x, double* y);                                     // y = A x (the roofline kernel)
int launch_p_apply_dot(hipStream_t s, PMat A, const double* x, double* y, double* partials);              // + slot 0 = x.y
// r = b - A x; slots 0 |r|, 1 norm factor; xbar = xsum_dev[0] * inv_n stays on the device (it is an all-reduced sum)
int launch_p_init(hipStream_t s, PMat A, const double* b, const double* x, const double* xsum_dev, double inv_n, double* r, double* partials);
int launch_dot(hipStream_t s, int n, int c0, const double* a, const double* b /* nullptr: sum(a) */, double* partials);   // slot 0, over [c0, c0+n)
int launch_pcg_update_p(hipStream_t s, int n, int c0, const double* z, double* p, const double* sc, int first);    // p = z + (sc[0]/sc[1]) p
int launch_pcg_update_xr(hipStream_t s, int n, int c0, double* x, double* r, const double* p, const double* w, double* sc /* sc[1] = sc[0] on the way out */, double* partials);   // alpha = sc[0]/sc[2]; slot 0 = sum|r|
int launch_jacobi_precond(hipStream_t s, PMat A, const double* r, double* z);
// ref_term != nullptr: the coarse cell ref_c (local index of C, -1: not in C) holds the pressure reference cell and keeps its point term unscaled
// (see k_mg_coarsen); launch_mg_ref_term leaves that term (level 0's, 0 where ref_local < 0) in out[0]
int launch_mg_coarsen(hipStream_t s, PMat F, PMat C, int ref_c = -1, const double* ref_term = nullptr);
int launch_mg_ref_term(hipStream_t s, PMat A0, int ref_local, double* out);
int launch_mg_smooth_first(hipStream_t s, PMat A, const double* b, double* x, double w);                   // x = w b / diag
int launch_mg_smooth_two_from_zero(hipStream_t s, PMat A, const double* b, double* xn, double w, double w2);      // smooth_first(w) + smooth(w2) fused (bit-identical)
int launch_mg_smooth(hipStream_t s, PMat A, const double* b, const double* x, double* xn, double w);
// the same sweep + the block partials of xn . b (slot 0), what launch_dot(xn, b) would leave there
int launch_mg_smooth_dot(hipStream_t s, PMat A, const double* b, const double* x, double* xn, double w, double* partials);      // xn = x + w (b - A x)/diag
int launch_mg_residual_restrict(hipStream_t s, PMat A, const double* b, const double* x, PMat C, double* bc);   // bc = P^T (b - A x)
int launch_mg_prolong_add(hipStream_t s, PMat A, double* x, PMat C, const double* xc);
int launch_mg_smooth_prolong(hipStream_t s, PMat A, const double* b, const double* x, PMat C, const double* xc, double* xn, double w);   // x += P xc, then one sweep (fused)
// the coarsest level: x = A^-1 b from the banded Cholesky factor `fac` (launch_mg_coarse_factor: mg_coarse_factor_doubles(A) doubles; for
// levels with mg_coarse_direct_ok(A): no ghost planes, <= kMgDirectMax cells, band <= kMgDirectBand); fac == nullptr or a failed
// factorisation: `sweeps` damped-Jacobi sweeps from a zero guess
constexpr int kMgDirectMax = 128, kMgDirectBand = 64;
bool mg_coarse_direct_ok(PMat A);
int mg_coarse_factor_doubles(PMat A);
int launch_mg_coarse_factor(hipStream_t s, PMat A, double* fac);
int launch_mg_coarse_solve(hipStream_t s, PMat A, const double* b, double* x, double* tmp, int sweeps, double w, const double* fac = nullptr);
// the whole V-cycle below a size threshold in one workgroup; level l result: x1[l] (x0 for the coarsest / a single-level tail)
constexpr int kMgTailMax = 6;
constexpr int kMgTailCells = 1024;   // measured: at 8000 cells one workgroup (137 us) is SLOWER than the ~20 separate launches it replaces
struct MgWeights { int n; double w[4]; };      // the smoother's Jacobi weights per sweep (pre-smoothing order; post-smoothing runs them backwards)
int launch_mg_tail(hipStream_t s, const PMat* A, double* const* x0, double* const* x1, double* const* b, int n, double w, int coarse_sweeps, MgWeights W,
                   const double* fac = nullptr);

int launch_copy_f64(hipStream_t s, double* dst, const double* src, size_t n);
int launch_relax_field(hipStream_t s, double* x, const double* prev, double alpha, size_t n);   // x = prev + alpha (x - prev)
// slab interfaces: coefficient of the z-face below the first owned plane, stored at the ghost cell under it (what p_row reads as uz[c - sz])
int launch_p_ghost_uz(hipStream_t s, FvGeo g, CFace3 rAUf, CFace3 alphaf, PMat A);
int launch_mg_coarsen_ghost(hipStream_t s, PMat F, PMat C);
// y += x on a contiguous range (reverse-halo accumulation)
int launch_add_f64(hipStream_t s, double* y, const double* x, size_t n);

}  // namespace fy
