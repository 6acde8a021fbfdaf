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
    int upwind;             // div(phi,U): 0 Gauss linear, 1 Gauss upwind (first order, bounded), 2 Gauss linearUpwind (upwind + explicit gradient correction),
                            // 3 .. 8 the NVD / TVD limited schemes limitedLinear k | vanLeer | MUSCL | Minmod | SuperBee | QUICK (FY_CONVECTION_*)
    double lim_twoByk;      // limitedLinear: 2 / max(k, small)
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
    double turb_dcoeff;     // LESdelta deltaCoeff (the graded block's kernels form deltaCoeff * cbrt(V) per cell; turb_delta is the uniform block's)
    const double* epsturb;  // [storage cells] epsilon (kEpsilon)
    int k_bc[6], eps_bc[6]; double k_val[6], eps_val[6];
    const double* kturb;    // [storage cells] k of the kEqn / kEpsilon models (nullptr: none)
    int nut_wall_live;
    double wf_cmu25, wf_kappa, wf_E, wf_yPlusLam;
    double u_relax;         // fvMatrix::relax factor of UcEqn for the current outer iteration (<= 0: no relaxationFactors entry, relax() is a no-op)
    double g[3];
    int need_ref, p_ref_cell;
    double p_ref_value;
    // graded (rectilinear) single block: cell sizes along x, y, z (device arrays of nx, ny, nz doubles; blockMesh simpleGrading).  Only the
    // kernels of namespace fy::gr read them (the same source compiled with the general geometry, fv_kernels_graded.hip); the uniform block's
    // kernels keep their constants dx / Af / V and are untouched by it
    int graded;
    const double* h[3];
    // strip order of the cell sweeps' blocks (fv_block in fv_kernels.hip): strip_B blocks per strip (0: off -- the plain XCD-contiguous order),
    // strip_bp = 256-cell blocks per z-plane, strip_nzx = planes per XCD
    int strip_B, strip_bp, strip_nzx;
    // block window (0: the whole owned range): the sweep covers the 256-cell blocks [win_blk0, win_blk0 + win_nblk) only -- whole z-planes, chosen by
    // the host so that a sweep's interior planes run beside the halo exchange its end planes wait for; red_stride = blocks of the whole sweep (where
    // a reducing kernel's partial sums of slot q start: the fold sees the same partials whatever the windows were)
    int win_blk0, win_nblk, red_stride;
};

struct Face3 { double* a[3]; };           // +axis oriented face arrays (x: (nx+1)*ny*nz, y: nx*(ny+1)*nz, z: nx*ny*(nz+1))
struct CFace3 { const double* a[3]; };
// momentum matrix: diag + neighbour coefficient across face 2*d+s.  bd ([3 x storage cells], nullptr without a slip patch): the boundary
// diagonal that differs by component -- a symmetryPlane / slip patch puts the wall coefficient on the NORMAL component only (fvMatrix keeps
// such internalCoeffs apart from the scalar lduMatrix diagonal: solve adds them per component, A() their component average, H() the rest)
struct Mom7 { double* diag; double* an[6]; double* bd; };

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

struct TurbEqn { int mode; double ck, ce, delta, c1, c2, c3, sigma, xmin, relax; int upwind; int bc[6]; double val[6]; int wall[6]; double cmu75, cmu25, kappa; };
constexpr int kMgDirectMax = 128, kMgDirectBand = 64;
constexpr int kMgTailMax = 6;
constexpr int kMgTailCells = 1024;   // measured: at 8000 cells one workgroup (137 us) is SLOWER than the ~20 separate launches it replaces
struct MgWeights { int n; double w[4]; };      // the smoother's Jacobi weights per sweep (pre-smoothing order; post-smoothing runs them backwards)

// The launchers exist twice: fy::launch_* for the uniform block (dx, Af, V constants in every kernel) and fy::gr::launch_* for a graded block
// (per-axis cell sizes, linear-interpolation weights, |Sf| / |d| per face) -- ONE source, fv_kernels.hip, compiled once per geometry model
// (fv_kernels_graded.hip), so that the uniform block's kernels carry no trace of the general one.  The geometry-free launchers (linear
// algebra, multigrid) are the same code in both.
#include "fv_kernels_api.inc"
namespace gr {
#include "fv_kernels_api.inc"
}  // namespace gr

}  // namespace fy
