// fy_ldu_solver: icoFoamYade's loop body (icoFoamYade/icoFoamYade.C:65-149, with the correctNonOrthogonal loop of :114-131) on a general
// polyhedral mesh in OpenFOAM's addressing (constant/polyMesh, what createMesh.H hands the solver: icoFoamYade.C:42).  See include/foamyade_hip.h.
//   ldu_mesh.cpp     host: OpenFOAM's geometry from points / faces / owner / neighbour [OF-6 primitiveMesh*, surfaceInterpolation], cell -> face lists
//   ldu_kernels.hip  the operators as HIP kernels: one lane per cell GATHERING over the cell's faces, or one lane per face (no scatter, no atomics)
//   ldu_solver.cpp   sequencing + the C entry points
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "../../include/foamyade_hip.h"
#include "common.hpp"

namespace fy {

// host-side geometry in OpenFOAM's conventions (face area vectors point owner -> neighbour / outwards)
struct LduHostMesh {
    int nPoints = 0, nFaces = 0, nInt = 0, nCells = 0, nPatches = 0;
    std::vector<int32_t> own, nei, patch_of;            // patch_of[f - nInt]
    std::vector<int32_t> cf_off, cf_face;               // per cell: its faces (ascending face number)
    int Wall = 0;                                       // the most faces a cell has; slot tables [Wall nCells], slot-major: face (-1: no more) and the cell across (-1: boundary)
    std::vector<int32_t> ef, en;
    std::vector<double> Cf, Sf, magSf, C, V, w, dcNO, kvec;      // [3 nF] [3 nF] [nF] [3 nc] [nc] [nInt] [nF] [3 nInt]
    std::vector<double> recon;                          // [9 nc] inv(sum_f Sf Sf / |Sf|): fvc::reconstruct's tensor [OF-6 fvcReconstruct.C]
    double bbox_min[3], bbox_max[3];
    // cyclic patches folded into internal faces (ldu_mesh.cpp): faces [n_real_internal, nInt) are the folded pairs; sep [3 nInt] (empty without a cyclic pair):
    // the neighbour cell's image is C_N + sep_f; orig_face[f] = the caller's face the solver's face f was made from; the folded mesh's own arrays
    int n_real_internal = 0;
    std::vector<double> sep;
    std::vector<int32_t> orig_face, f_off, f_pts, f_own, f_nei, f_pstart, f_psize;
    int fold_cyclics(fy_poly_mesh* m);
    int build(const fy_poly_mesh* m);                   // FY_OK or an error (malformed addressing)
};

// what the kernels see (device pointers), passed by value
struct LduGeo {
    int nCells, nFaces, nInt, nPatches;
    const int32_t *own, *nei, *patch_of, *cf_off, *cf_face;
    int Wall;
    const int32_t *ef, *en;                              // cell -> (face, cell across) slot tables (LduHostMesh)
    const double *Cf, *Sf, *magSf, *C, *V, *w, *dcNO, *kvec;
    const int32_t *u_bc, *p_bc;                          // per patch
    const double *u_val, *p_val;
    const double* recon;                                 // [9 nc]
    const double* psn;                                   // [nF - nInt] snGrad(p) of the fixedFluxPressure faces (pimpleFoamYade; else null)
    double dt, nu;
    int upwind;                                          // div(phi,U), FY_CONVECTION_*: 0 Gauss linear; 1 Gauss upwind (the owner's weight is pos0(flux)); 2 Gauss linearUpwind grad(U) (upwind + an
                                                         // explicit gradient correction); 3 .. 8 the NVD / TVD limited schemes (limitedLinear k, vanLeer, MUSCL, Minmod, SuperBee, QUICK)
    double lim_two_by_k;                                 // limitedLinear: 2 / max(k, small)
    const double* gradL;                                 // [3 nc] Gauss-linear gradient of |U|^2 of the iterate the matrix is assembled from (limited schemes; else null)
    int need_ref, p_ref_cell;
    double p_ref_value;
    // per-slot coefficients of the two widest gathers (built once on the device, k_ldu_slot_coefs), component-major and slot-major like the slot tables:
    //   gB [3][Wall][nCells]: internal slot k of cell c: +-Sf w_nb / V (the neighbour's share of the face value in the cell's Gauss gradient); boundary slot: Sf / V
    //   gG0 [3][nCells]:      the cell's own share, sum over its internal faces of +-Sf w_c / V
    //   rT [3][Wall][nCells]: fvc::reconstruct's recon_c . Sf / |Sf| per slot
    const double *gB, *gG0, *rT;
    const double* sep;                                   // [3 nInt] or null: folded cyclic faces -- the neighbour cell's image is C_N + sep_f (zero on the mesh's own internal faces)
    int nIntReal;                                        // faces [nIntReal, nInt) are folded cyclic pairs
};

// momentum matrix in LDU form: diag [nc] (boundary diagonal included), lower / upper per internal face, b [3 nc] (boundary sources included)
struct LduMom {
    double *diag, *lower, *upper, *b;
    double* bdiag;                   // [3 nCells] or null: the per-component part of the boundary diagonal (symmetry patches), kept apart from the scalar diagonal
};
// pimpleFoamYade's extra fields: the void fraction (cells, old time, faces), the coupling's implicit and explicit momentum sources, gravity
struct LduPim {
    const double *alpha, *alphaOld, *alphaf, *uSourceDrag, *uSource;
    double g[3];
    const double* nut;               // LES: the eddy viscosity per cell (null: laminar), its patch conditions
    const int32_t* nut_bc;
    const double* nut_val;
    // LES kEqn: k per cell (null: another model) and its patch conditions (FY_BC_NUT_ZERO_GRADIENT | _FIXED_VALUE); a FY_BC_NUT_CALCULATED nut patch carries
    // Ck sqrt(k_b) delta once correctNut() has run (nut_live), the file's value before
    const double* k;
    const int32_t* k_bc;
    const double* k_val;
    int nut_live;
    double ck, delta_coeff;
    // RAS kEpsilon: epsilon per cell (null: another model) and its patch conditions; a calculated nut patch then carries Cmu k_b^2 / eps_b
    const double* eps;
    const int32_t* eps_bc;
    const double* eps_val;
    double cmu;
};
// one transport equation of a closure (fv_kernels.hip's modes): 0 kEqn's k, 1 kEpsilon's epsilon, 2 kEpsilon's k; X the transported field with its patches
struct LduKEqn {
    int mode;
    double ce, relax;
    int upwind;
    double sigma, c1, c2, c3;
    const double* X;
    const int32_t* x_bc;
    const double* x_val;
};

int ldu_red_blocks(int n);      // partials per slot of the reducing kernels (= red_blocks(n) of fv_kernels.hpp: the folds are shared)

int launch_ldu_flux_of(hipStream_t s, LduGeo g, const double* F, double* phi);
int launch_ldu_courant(hipStream_t s, LduGeo g, const double* phi, double* partials);                    // slot 0 = max sumPhi / V, slot 1 = sum sumPhi
int launch_ldu_grad_vec(hipStream_t s, LduGeo g, const double* F, double* T);                           // T[9 c + 3 i + j] = d_i F_j
int launch_ldu_slot_coefs(hipStream_t s, LduGeo g, double* gB, double* gG0, double* rT);
int launch_ldu_grad_scalar(hipStream_t s, LduGeo g, const double* p, double* gp);
int launch_ldu_grad_magsqr(hipStream_t s, LduGeo g, const double* U, double* gradL);                    // fvc::grad(magSqr(U)), boundary value magSqr(U_b): the limiters' gradient
// UEqn (icoFoamYade.C:79-85): face part (lower / upper, the explicit non-orthogonal flux of the laplacian), then the cell part (diag, b)
int launch_ldu_assemble_momentum(hipStream_t s, LduGeo g, const double* phi, const double* Uold, const double* uSource, const double* gradU, LduMom M,
                                 double* face_corr /* [3 nInt] scratch */);
// one Jacobi pass: xn = (b - V gradp - offdiag x) / diag and the L1 residual / normFactor sums of x (slots 0-2 |b - A x|, 3-5 normFactor terms)
int launch_ldu_mom_pass(hipStream_t s, LduGeo g, LduMom M, const double* rhs /* M.b, or the predictor's full right-hand side */, const double* gradp /* or null */, const double* x, double* xn,
                        const double* xsum3, double* partials);
int launch_ldu_HbyA(hipStream_t s, LduGeo g, LduMom M, const double* U, double* rAU, double* HbyA);
int launch_ldu_phiHbyA(hipStream_t s, LduGeo g, const double* HbyA, const double* rAU, const double* Uold, const double* phiOld, const double* alphaf /* or null */, double* rAUf, double* phiHbyA);
int launch_ldu_adjust_phi(hipStream_t s, LduGeo g, double* phiHbyA, double* sums4 /* device scratch */, int* err, double* partials);
// pEqn (icoFoamYade.C:118-123): face part (coefficients, the explicit non-orthogonal flux from grad p), cell part (diag, right-hand side, setReference)
int launch_ldu_assemble_pressure(hipStream_t s, LduGeo g, const double* rAUf, const double* phiHbyA, const double* gradp, double* pcoef, double* pcorr, double* pt, double* pdiag, double* prhs);
// r = b - A x with slot 0 = sum |r|, slot 1 = normFactor terms (xbar from xsum); the iterations run on the ELL form (ldu_amg.hpp)
int launch_ldu_p_init(hipStream_t s, LduGeo g, const double* pdiag, int ellW, const int32_t* ell_nbr, const double* ell_coef, const double* b, const double* x, const double* xsum, double inv_n,
                      double* r, double* partials);
int launch_ldu_flux_correct(hipStream_t s, LduGeo g, const double* p, const double* phiHbyA, const double* pcoef, const double* pcorr, double* phi);
// U = HbyA - rAU grad(p) (gradient formed inline) + continuity sums (slot 0 sum |div phi|, slot 1 sum div phi)
int launch_ldu_U_correct(hipStream_t s, LduGeo g, const double* HbyA, const double* rAU, const double* p, const double* phi, double* U, double* partials);
// ---- pimpleFoamYade (pimpleFoamYade.C:60-114, UcEqn.H, pEqn.H); see the kernels' comments
int launch_ldu_alphaf(hipStream_t s, LduGeo g, const double* alpha, double* alphaf);
int launch_ldu_pre_coupling(hipStream_t s, LduGeo g, const double* phi, const double* U, const double* vGrad, const double* alphaf, double* ddtU, double* divT);
int launch_ldu_assemble_momentum_pimple(hipStream_t s, LduGeo g, LduPim P, const double* phi, const double* Uold, const double* U, const double* gradU, LduMom M, double* aphi /* [nF] scratch: alphaPhic */,
                                        double* fstress /* [3 nF] */, double u_relax, double* rAU);
int launch_ldu_smagorinsky_nut(hipStream_t s, LduGeo g, const double* vGrad, double ck, double ce, double delta_coeff, double* nut);
// LES kEqn: the k equation's matrix into M (face part, then cells: diag, b[3 c] = the source, b[3 c + 1 .. 2] = 0), x3 = (k, 0, 0); after the solve bound() and nut
int launch_ldu_grad_k(hipStream_t s, LduGeo g, LduPim P, LduKEqn K, double* gk);
int launch_ldu_k_assemble(hipStream_t s, LduGeo g, LduPim P, LduKEqn K, const double* phi, const double* vGrad, const double* gk, LduMom M, double* face_corr, double* x3);
int launch_ldu_k_bound_nut(hipStream_t s, LduGeo g, LduPim P, LduKEqn K, const double* x3, double* X, double* nut);
int launch_ldu_forces(hipStream_t s, LduGeo g, LduPim P, const double* rAU, double* rAUf, double* phiForces);
int launch_ldu_ssf_predictor(hipStream_t s, LduGeo g, const double* phiForces, const double* rAUf, const double* p, const double* gradp, double* ssf);
int launch_ldu_reconstruct(hipStream_t s, LduGeo g, const double* ssf, const double* base, const double* scale, double* out);
int launch_ldu_add_forces_constrain(hipStream_t s, LduGeo g, const double* phiForces, const double* rAUf, const double* U, double* phiHbyA, double* psn);
int launch_ldu_pim_pfaces(hipStream_t s, LduGeo g, const double* alphaf, const double* rAUf, const double* phiHbyA, const double* psn, double* arAUf, double* phiA);
int launch_ldu_prhs_ddt_alpha(hipStream_t s, LduGeo g, const double* alpha, const double* alphaOld, double* prhs);
int launch_ldu_pim_flux(hipStream_t s, LduGeo g, const double* p, const double* phiHbyA, const double* pcoef, const double* pcorr, const double* alphaf, const double* rAUf,
                        const double* phiForces, const double* psn, double* phi, double* ssf, double* aphi /* [nF]: alphacf phic */);
int launch_ldu_pim_continuity(hipStream_t s, LduGeo g, const double* aphi, const double* alpha, const double* alphaOld, double* partials);
int launch_ldu_sum(hipStream_t s, const double* x, int n, int ncomp, double* partials);                  // slot q = sum of component q
// mesh.findCell stand-in for the point-force locate: from the nearest centre (hint[i], or -1: not located) walk across the face the point lies
// furthest outside of until it lies inside every face of a cell; cell_out[i] = that cell or -1 (outside the mesh)
int launch_ldu_find_cell(hipStream_t s, LduGeo g, const double* rec, int rec_len, int64_t n, const int32_t* hint, int32_t* cell_out);
int launch_ldu_positions(hipStream_t s, const double* rec, int rec_len, int64_t n, double* pos3);      // pos3[i] = rec[i][0..2]

}  // namespace fy
