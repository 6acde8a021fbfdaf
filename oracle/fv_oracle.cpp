// ORACLE / TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// CPU restatement of the FV half of the hot path: the loop bodies of icoFoamYade (icoFoamYade/icoFoamYade.C:65-149)
// and pimpleFoamYade (pimpleFoamYade/pimpleFoamYade.C:60-114, UcEqn.H:3-33, pEqn.H:1-50, CourantNo.H:32-49,
// continuityErrs.H:32-46) on a uniform hex block (blockMesh order, cell = i + nx*(j + ny*k)).
//
// PARITY UNPINNED.  The control flow above is the reference's, but every operator it calls (fvm::ddt/div/laplacian/Sp,
// fvc::grad/div/flux/interpolate/reconstruct/ddtCorr, fvMatrix::A/H/flux/relax/setReference, the linear solvers) lives
// in OpenFOAM-6 (libfiniteVolume/libOpenFOAM; README.md:17, icoFoamYade/Make/options:12-19), which is not vendored in
// /root/reference, not installed here, and the reference ships no case, test or golden data for it.  The operators are
// restated from OpenFOAM-6's published semantics (Euler ddt, Gauss linear grad/div/laplacian on an orthogonal uniform
// mesh, EulerDdtScheme::fvcDdtPhiCorr, fvMatrix::H/A/relax/setReference, lduMatrix L1-normalised residuals, PCG) and
// validated by known-answer flows in tests/test_fv_oracle.py, never by comparison with OpenFOAM output.
// Deliberate departures, all documented in DESIGN.md: momentum equations are solved with Jacobi sweeps instead of
// symGaussSeidel, pressure with PCG + geometric multigrid (or Jacobi) instead of DIC/GAMG -- sequential sweeps do not map
// to a GPU, and the same algorithm on both sides lets tests compare CPU and GPU fields tightly.
//
// Conventions: face arrays are +axis oriented; phi_x has (nx+1)*ny*nz entries (face i between cells i-1 and i), etc.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
// must match oracle.py FvCase
struct orc_fv_case {
    int solver;                  // 0 ico, 1 pimple
    int nx, ny, nz;
    double dx;
    double origin[3];
    double dt, nu, rho_fluid, rho_particle;
    double g[3];
    int u_bc[6];                 // 0 fixedValue, 1 zeroGradient, 2 slip / symmetryPlane (normal component 0, tangential zeroGradient)
    double u_value[6][3];
    int p_bc[6];                 // 0 zeroGradient, 1 fixedValue, 2 fixedFluxPressure
    double p_value[6];
    int n_outer, n_corr, n_non_orth, momentum_predictor;
    int p_ref_cell; double p_ref_value;
    int p_solver;                // 0 PCG+Jacobi, 1 PCG+MG
    double p_tol, p_rel_tol, p_final_tol, p_final_rel_tol; int p_max_iter;
    double u_tol, u_rel_tol; int u_max_iter;
    int convection_scheme;      // 0 Gauss linear, 1 Gauss upwind, 2 Gauss linearUpwind (unlimited, Gauss-linear gradient), 3.. the NVD / TVD limited schemes (limited_weight)
    // controlDict adjustTimeStep / maxCo / maxDeltaT (readTimeControls.H + setDeltaT.H, pimpleFoamYade.C:62-64)
    int adjust_time_step; double max_co, max_delta_t;
    // fvSolution relaxationFactors: equations { Uc; UcFinal } (UcEqn.relax(), UcEqn.H:12), fields { p; pFinal } (p.relax(), pEqn.H:41); <= 0: no entry
    double u_relax, u_relax_final, p_relax, p_relax_final;
    // constant/turbulenceProperties of pimpleFoamYade: 0 laminar (Stokes), 1 LES Smagorinsky (DPMTurbulenceModels.C:67-74), delta cubeRootVol;
    // 0/nut: per-side boundary type (0 zeroGradient, 1 fixedValue) and value, uniform initial value
    int turbulence_model; double les_ck, les_ce, les_delta_coeff;
    int nut_bc[6]; double nut_value[6]; double nut_initial;
    // LES kEqn (turbulence_model 2): the 0/k file, div(alphaPhic,k) scheme (0 linear, 1 upwind), solvers.k, relaxationFactors equations k
    int k_bc[6]; double k_value[6]; double k_initial; int k_convection_scheme; double k_tol, k_rel_tol; int k_max_iter; double k_relax;
    // RAS kEpsilon (turbulence_model 3): model constants, the 0/epsilon file, div(alphaPhic,epsilon) scheme, solvers.epsilon, relaxation
    double ras_cmu, ras_c1, ras_c2, ras_c3, ras_sigmak, ras_sigmaeps;
    int eps_bc[6]; double eps_value[6]; double eps_initial; int eps_convection_scheme; double eps_tol, eps_rel_tol; int eps_max_iter; double eps_relax;
    // wall functions: nut_bc == 2 nutkWallFunction, eps_bc == 2 epsilonWallFunction [OF-6]; kappa, E (Cmu = ras_cmu)
    double wf_kappa, wf_E;
    // graded (rectilinear) block: cell sizes along x, y, z (nx, ny, nz doubles; blockMesh simpleGrading); all null = uniform cubes of edge dx
    const double* hx; const double* hy; const double* hz;
    double convection_limiter_k;     // limitedLinear's coefficient (Gauss limitedLinear k)
};
struct orc_fv_stats {
    double courant_mean, courant_max, cont_sum_local, cont_global, cont_cumulative;
    int p_iters_total, p_solves, u_iters_total;
    double p_initial_residual, p_final_residual;
    double delta_t;              // the time step this pass of the loop used (setDeltaT.H)
};
}

namespace {

typedef std::vector<double> vec;
const double SMALL = 1e-15;      // OpenFOAM `small` for double
const double VSMALL = 1e-300;
const int kMgCoarsest = 128, kMgCoarsestEdge = 8, kMgCoarseSweeps = 120;    // coarsest multigrid level (same rule as the product so iteration counts are comparable):
                                                                            // solved exactly out of its dense inverse; the sweeps only stand in if that inverse cannot be formed

struct MgLevel {
    int nx, ny, nz, N;
    vec diag, ux, uy, uz;        // symmetric 7-point: (A x)_c = diag_c x_c - sum_f u_f x_nb ; u_* stored at the owner (low) cell
    vec x, b, r;
};

struct Fv {
    orc_fv_case cs;
    int nx, ny, nz, Nc, n[3], stride[3];
    double dx, Af, V;
    bool pimple;
    // ---- geometry of a graded (rectilinear, orthogonal) block [OF-6 surfaceInterpolation::weights / deltaCoeffs on a hex mesh whose cell sizes
    // vary along the axes]: the uniform block keeps its constants (dx, Af, V) and its original expressions, `graded` switches every operator to
    // the general form -- linear-interpolation weight of the low-side cell at an internal face = distance from the face to the high-side
    // centre over the centre distance; |Sf| / |d| from the face area and the centre distance (a boundary face: centre to face)
    bool graded = false;
    vec h[3];
    double total_volume = 0.0;
    inline int idx_of(int d, int c) const { return (c / stride[d]) % n[d]; }
    inline double vol(int i, int j, int k) const { return graded ? (h[0][i] * h[1][j]) * h[2][k] : V; }
    inline double volc(int c) const { return graded ? vol(c % nx, (c / nx) % ny, c / (nx * ny)) : V; }
    // area of a face normal to d whose transverse indices are those of (i, j, k) (cell or face indices: the normal one is not used)
    inline double area(int d, int i, int j, int k) const { return graded ? (d == 0 ? h[1][j] * h[2][k] : d == 1 ? h[0][i] * h[2][k] : h[0][i] * h[1][j]) : Af; }
    inline double wlow(int d, int q) const { return graded ? h[d][q] / (h[d][q - 1] + h[d][q]) : 0.5; }
    inline double lerp(int d, int q, double lo, double hi) const { if (!graded) return 0.5 * (lo + hi); const double w = wlow(d, q); return w * lo + (1.0 - w) * hi; }
    // centre distance across face q of axis d (q = 0 or n[d]: cell centre to the boundary face)
    inline double delta(int d, int q) const {
        if (!graded) return (q == 0 || q == n[d]) ? 0.5 * dx : dx;
        if (q == 0) return 0.5 * h[d][0];
        if (q == n[d]) return 0.5 * h[d][n[d] - 1];
        return 0.5 * (h[d][q - 1] + h[d][q]);
    }
    // face index along d of face (d, s) of cell (i, j, k)
    inline int fq(int d, int s, int i, int j, int k) const { return (d == 0 ? i : d == 1 ? j : k) + s; }
    // |Sf| / |d| of face (d, s) of cell (i, j, k).  Uniform block: dx for EVERY face -- the boundary factor 2 stays where the uniform code has it
    inline double sfd(int d, int s, int i, int j, int k) const { return graded ? area(d, i, j, k) / delta(d, fq(d, s, i, j, k)) : dx; }
    inline double bfac() const { return graded ? 1.0 : 2.0; }          // (gb = bfac * gamma: see sfd)
    inline double hcell(int d, int c) const { return graded ? h[d][idx_of(d, c)] : dx; }
    // LESdelta cubeRootVol [OF-6 cubeRootVolDelta.C]: deltaCoeff * cbrt(V) of the cell; nearWallDist: half the cell's extent along the wall normal
    inline double les_delta(int c) const { return graded ? cs.les_delta_coeff * std::cbrt(volc(c)) : cs.les_delta_coeff * std::pow(V, 1.0 / 3.0); }
    inline double ywall_of(int d, int c) const { return 0.5 * hcell(d, c); }
    // linear face interpolate of a cell pair across face (d, s) of the cell with index qc along d: own = this cell's value, nb = the neighbour's
    inline double lerp_side(int d, int s, int qc, double own, double nb) const { return !graded ? 0.5 * (own + nb) : (s ? lerp(d, qc + 1, own, nb) : lerp(d, qc, nb, own)); }
    int threads = 1;
    // state
    vec U, Uold, p, alpha, alphaOld, uSource, uSourceDrag, uParticle, gradP, divT, vGrad, ddtU;
    vec nut;                                 // eddy viscosity (Smagorinsky, kEqn); empty = laminar
    vec kturb, epsturb;                      // turbulent kinetic energy (kEqn, kEpsilon), dissipation rate (kEpsilon)
    int k_iters = 0;
    vec phi[3], phiOld[3], psn[3];           // psn: d p / d axis on fixedFluxPressure boundary faces
    // work
    vec bdiag;                   // [3 Nc] per-component boundary diagonal of the momentum matrix (slip patches); empty without one
    vec diag, an[6], src, rAU, HbyA, alphaf[3], phiHbyA[3], phiForces[3], rAUf[3], pflux[3], bmom, Sc, divG;
    std::vector<MgLevel> mg;
    vec pb, pr, pw, pp, pz;
    orc_fv_stats st{};
    double cumulativeContErr = 0.0;

    int fsize(int d) const { return d == 0 ? (nx + 1) * ny * nz : d == 1 ? nx * (ny + 1) * nz : nx * ny * (nz + 1); }
    inline int cid(int i, int j, int k) const { return i + nx * (j + ny * k); }
    inline int fid(int d, int i, int j, int k) const {
        return d == 0 ? i + (nx + 1) * (j + ny * k) : d == 1 ? i + nx * (j + (ny + 1) * k) : i + nx * (j + ny * k);
    }
    // face of cell (i,j,k) in direction d, side s (0 low, 1 high)
    inline int cface(int d, int s, int i, int j, int k) const {
        return fid(d, i + (d == 0 ? s : 0), j + (d == 1 ? s : 0), k + (d == 2 ? s : 0));
    }
    inline bool onb(int d, int s, int i, int j, int k) const {
        const int q = d == 0 ? i : d == 1 ? j : k;
        return s ? q == n[d] - 1 : q == 0;
    }

    void init(const orc_fv_case& c) {
        cs = c; nx = c.nx; ny = c.ny; nz = c.nz; Nc = nx * ny * nz; dx = c.dx; Af = dx * dx; V = dx * dx * dx;
        n[0] = nx; n[1] = ny; n[2] = nz; stride[0] = 1; stride[1] = nx; stride[2] = nx * ny;
        pimple = c.solver == 1;
        graded = c.hx != nullptr && c.hy != nullptr && c.hz != nullptr;
        total_volume = V * Nc;
        if (graded) {
            h[0].assign(c.hx, c.hx + nx); h[1].assign(c.hy, c.hy + ny); h[2].assign(c.hz, c.hz + nz);
            total_volume = 0.0;
            for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) total_volume += vol(i, j, k);
        }
        U.assign(3 * (size_t)Nc, 0.0); Uold = U; p.assign(Nc, 0.0); alpha.assign(Nc, 1.0); alphaOld = alpha;
        uSource.assign(3 * (size_t)Nc, 0.0); uSourceDrag.assign(Nc, 0.0); uParticle.assign(3 * (size_t)Nc, 0.0);
        gradP.assign(3 * (size_t)Nc, 0.0); divT.assign(3 * (size_t)Nc, 0.0); vGrad.assign(9 * (size_t)Nc, 0.0); ddtU.assign(3 * (size_t)Nc, 0.0);
        for (int d = 0; d < 3; ++d) {
            phi[d].assign(fsize(d), 0.0); phiOld[d] = phi[d]; psn[d].assign(fsize(d), 0.0);
            alphaf[d].assign(fsize(d), 1.0); phiHbyA[d].assign(fsize(d), 0.0); phiForces[d].assign(fsize(d), 0.0);
            rAUf[d].assign(fsize(d), 0.0); pflux[d].assign(fsize(d), 0.0);
        }
        diag.assign(Nc, 0.0); for (auto& a : an) a.assign(Nc, 0.0);
        for (int q = 0; q < 6; ++q) if (cs.u_bc[q] == 2) bdiag.assign(3 * (size_t)Nc, 0.0);
        src.assign(3 * (size_t)Nc, 0.0); rAU.assign(Nc, 0.0); HbyA.assign(3 * (size_t)Nc, 0.0); bmom.assign(3 * (size_t)Nc, 0.0);
        Sc.assign(Nc, 0.0); divG.assign(3 * (size_t)Nc, 0.0);
        pb.assign(Nc, 0.0); pr = pb; pw = pb; pp = pb; pz = pb;
        if (pimple && c.turbulence_model >= 1) nut.assign(Nc, c.nut_initial);      // the 0/nut file (eddyViscosity: MUST_READ)
        if (pimple && c.turbulence_model >= 2) kturb.assign(Nc, c.k_initial);      // the 0/k file
        if (pimple && c.turbulence_model == 3) epsturb.assign(Nc, c.eps_initial);  // the 0/epsilon file
        build_mg_shapes();
        // createPhi: phi = linearInterpolate(U) & Sf (icoFoamYade/createFields.H:151, pimpleFoamYade/createFields.H:70-81)
        flux_of(U, phi);
    }

    // ---- boundary values ----------------------------------------------------------------------------------
    inline void Ub(const vec& F, int c, int patch, double* out) const {          // velocity-like field with U's BCs
        if (cs.u_bc[patch] == 0) { out[0] = cs.u_value[patch][0]; out[1] = cs.u_value[patch][1]; out[2] = cs.u_value[patch][2]; }
        else { out[0] = F[3 * (size_t)c]; out[1] = F[3 * (size_t)c + 1]; out[2] = F[3 * (size_t)c + 2]; }
        // symmetryPlane / slip on a planar, axis-aligned patch [OF-6 basicSymmetryFvPatchField::evaluate]: the cell value with its normal
        // component removed, (pif + transform(I - 2 nn, pif)) / 2
        if (cs.u_bc[patch] == 2) out[patch / 2] = 0.0;
    }
    inline double pbv(int c, int d, int s, int face) const {                     // boundary value of p
        const int patch = 2 * d + s;
        if (cs.p_bc[patch] == 1) return cs.p_value[patch];
        if (cs.p_bc[patch] == 2) return p[c] + (s ? 0.5 : -0.5) * hcell(d, c) * psn[d][face];
        return p[c];
    }
    bool need_reference() const { for (int q = 0; q < 6; ++q) if (cs.p_bc[q] == 1) return false; return true; }

    // fvc::flux(F) = linearInterpolate(F) & Sf, boundary value from U's BCs (used for U and HbyA)
    void flux_of(const vec& F, vec* out) const {
        for (int d = 0; d < 3; ++d)
            for (int k = 0; k < nz + (d == 2); ++k) for (int j = 0; j < ny + (d == 1); ++j) for (int i = 0; i < nx + (d == 0); ++i) {
                const int q = d == 0 ? i : d == 1 ? j : k;
                const int f = fid(d, i, j, k);
                double v;
                if (q == 0) { double b[3]; Ub(F, cid(i, j, k), 2 * d, b); v = b[d]; }
                else if (q == n[d]) { double b[3]; const int c = cid(i - (d == 0), j - (d == 1), k - (d == 2)); Ub(F, c, 2 * d + 1, b); v = b[d]; }
                else { const int c = cid(i, j, k); v = lerp(d, q, F[3 * (size_t)(c - stride[d]) + d], F[3 * (size_t)c + d]); }
                out[d][f] = v * area(d, i, j, k);
            }
    }

    // fvc::grad(p), Gauss linear
    void grad_p(vec& G) const {
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            for (int d = 0; d < 3; ++d) {
                double fv[2];
                for (int s = 0; s < 2; ++s) {
                    if (onb(d, s, i, j, k)) fv[s] = pbv(c, d, s, cface(d, s, i, j, k));
                    else if (!graded) fv[s] = 0.5 * (p[c] + p[c + (s ? stride[d] : -stride[d])]);
                    else fv[s] = s ? lerp(d, fq(d, 1, i, j, k), p[c], p[c + stride[d]]) : lerp(d, fq(d, 0, i, j, k), p[c - stride[d]], p[c]);
                }
                G[3 * (size_t)c + d] = (fv[1] - fv[0]) / hcell(d, c);
            }
        }
    }

    // ---- NVD / TVD limited convection schemes for div(phi,U) [OF-6 LimitedScheme<vector, Limiter<NVDTVD>, limitFuncs::magSqr>, NVDTVD.H] ----------
    // One limiter per face from the scalar lPhi = magSqr(U): r = 2 (d . grad(lPhi)_C) / (lPhi_N - lPhi_P) - 1 with C the upwind cell of the face
    // flux and the Gauss-linear gradient of lPhi (boundary value magSqr(U_b)); the face weight of the OWNER's value is
    // limiter * w_linear + (1 - limiter) * pos0(faceFlux) [limitedSurfaceInterpolationScheme::weights], used implicitly.
    vec lphi, gradL;
    void limiter_gradient(const vec& F) {
        lphi.assign(Nc, 0.0); gradL.assign(3 * (size_t)Nc, 0.0);
        for (int c = 0; c < Nc; ++c) lphi[c] = (F[3 * (size_t)c] * F[3 * (size_t)c] + F[3 * (size_t)c + 1] * F[3 * (size_t)c + 1]) + F[3 * (size_t)c + 2] * F[3 * (size_t)c + 2];
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            for (int d = 0; d < 3; ++d) {
                double fv[2];
                for (int s = 0; s < 2; ++s) {
                    if (onb(d, s, i, j, k)) { double b[3]; Ub(F, c, 2 * d + s, b); fv[s] = (b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]; }
                    else fv[s] = lerp_side(d, s, d == 0 ? i : d == 1 ? j : k, lphi[c], lphi[c + (s ? stride[d] : -stride[d])]);
                }
                gradL[3 * (size_t)c + d] = (fv[1] - fv[0]) / hcell(d, c);
            }
        }
    }
    static double limiter_fn(int scheme, double twoByk, double r) {
        switch (scheme) {
            case 3: return std::max(std::min(twoByk * r, 1.0), 0.0);                                             // limitedLinear
            case 4: return (r + std::fabs(r)) / (1.0 + std::fabs(r));                                            // vanLeer
            case 5: return std::max(std::min(std::min(2.0 * r, 0.5 * r + 0.5), 2.0), 0.0);                       // MUSCL
            case 6: return std::max(std::min(std::min(r, 1.0), 2.0), 0.0);                                       // Minmod
            case 7: return std::max(std::max(std::min(2.0 * r, 1.0), std::min(r, 2.0)), 0.0);                    // SuperBee
            default: return std::max(std::min((3.0 + r) / 4.0, 2.0), 0.0);        // QUICK (8) [OF-6 QUICK.H]: (phif - phiU) / (phiCD - phiU) = (3 + r) / 4, limited to [0, 2] only
        }
    }
    // weight of the owner's (low cell's) value on interior face q of axis d between cells own and nei, for the face flux `flux` (owner -> neighbour)
    double limited_weight(int d, int q, int own, int nei, double flux) const {
        const double gradf = lphi[nei] - lphi[own];
        const double gradcf = delta(d, q) * gradL[3 * (size_t)(flux > 0 ? own : nei) + d];
        double r;
        if (std::fabs(gradcf) >= 1000.0 * std::fabs(gradf)) r = 2.0 * 1000.0 * (gradcf >= 0 ? 1.0 : -1.0) * (gradf >= 0 ? 1.0 : -1.0) - 1.0;
        else r = 2.0 * (gradcf / gradf) - 1.0;
        const double kk = std::max(cs.convection_limiter_k, SMALL);            // limitedLinearLimiter: twoByk_ = 2 / max(k, small) (k = 1: min(2 r, 1), TVD conforming)
        const double lim = limiter_fn(cs.convection_scheme, 2.0 / kk, r);
        const double wl = graded ? wlow(d, q) : 0.5;
        return lim * wl + (1.0 - lim) * (flux >= 0 ? 1.0 : 0.0);
    }

    // fvc::grad(U): T[3*i + j] = d_i U_j  (row-major xx xy xz ...), Gauss linear
    void grad_U(const vec& F, vec& T) const {
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            for (int d = 0; d < 3; ++d) {
                double fv[2][3];
                for (int s = 0; s < 2; ++s) {
                    if (onb(d, s, i, j, k)) Ub(F, c, 2 * d + s, fv[s]);
                    else {
                        const int nb = c + (s ? stride[d] : -stride[d]);
                        for (int q = 0; q < 3; ++q) fv[s][q] = !graded ? 0.5 * (F[3 * (size_t)c + q] + F[3 * (size_t)nb + q])
                                                             : (s ? lerp(d, fq(d, 1, i, j, k), F[3 * (size_t)c + q], F[3 * (size_t)nb + q]) : lerp(d, fq(d, 0, i, j, k), F[3 * (size_t)nb + q], F[3 * (size_t)c + q]));
                    }
                }
                for (int q = 0; q < 3; ++q) T[9 * (size_t)c + 3 * d + q] = (fv[1][q] - fv[0][q]) / hcell(d, c);
            }
        }
    }

    // alphacf = fvc::interpolate(alphac) (pimpleFoamYade.C:84); boundary value 1 (calculated patch, set by `alpha = 1.0`, FoamYade.C:68)
    void interp_alpha() {
        for (int d = 0; d < 3; ++d)
            for (int k = 0; k < nz + (d == 2); ++k) for (int j = 0; j < ny + (d == 1); ++j) for (int i = 0; i < nx + (d == 0); ++i) {
                const int q = d == 0 ? i : d == 1 ? j : k;
                const int f = fid(d, i, j, k);
                if (q == 0 || q == n[d]) alphaf[d][f] = 1.0;
                else { const int c = cid(i, j, k); alphaf[d][f] = lerp(d, q, alpha[c - stride[d]], alpha[c]); }
            }
    }

    // CourantNo.H:32-49 (pimple) / OpenFOAM CourantNo.H (ico, icoFoamYade.C:68)
    void courant() {
        double mx = 0.0, sum = 0.0;
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            double s = 0.0;
            for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) s += std::fabs(phi[d][cface(d, sd, i, j, k)]);
            mx = std::max(mx, s / vol(i, j, k)); sum += s;
        }
        st.courant_max = 0.5 * mx * cs.dt;
        st.courant_mean = 0.5 * (sum / (graded ? total_volume : V * Nc)) * cs.dt;
    }

    // pre-coupling fields, pimpleFoamYade.C:73-76: gradP = grad(p); divT = 2 nu laplacian(alphac, Uc); vGrad = grad(Uc)
    void pre_coupling_fields() {
        grad_U(U, vGrad);                                   // icoFoamYade.C:71 / pimpleFoamYade.C:76
        if (!pimple) return;
        // pimpleFoamYade.C:73  ddtU_f = fvc::ddt(Uc) + fvc::div(phic, Uc).  At this point of the loop Uc has not been written since
        // runTime++, so Uc.oldTime() (stored on this very access) equals Uc and the Euler fvc::ddt term is exactly zero [OF-6
        // GeometricField::storeOldTimes]; what remains is the Gauss-linear convective term.  (icoFoamYade never computes ddtU_f.)
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            double acc[3] = {0, 0, 0};
            for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) {
                const double flux = (s ? 1.0 : -1.0) * phi[d][cface(d, s, i, j, k)];
                double uf[3];
                if (onb(d, s, i, j, k)) Ub(U, c, 2 * d + s, uf);
                else {
                    const int nb = c + (s ? stride[d] : -stride[d]);
                    for (int q = 0; q < 3; ++q) uf[q] = !graded ? 0.5 * (U[3 * (size_t)c + q] + U[3 * (size_t)nb + q])
                                                      : (s ? lerp(d, fq(d, 1, i, j, k), U[3 * (size_t)c + q], U[3 * (size_t)nb + q]) : lerp(d, fq(d, 0, i, j, k), U[3 * (size_t)nb + q], U[3 * (size_t)c + q]));
                }
                for (int q = 0; q < 3; ++q) acc[q] += flux * uf[q];
            }
            for (int q = 0; q < 3; ++q) ddtU[3 * (size_t)c + q] = acc[q] / vol(i, j, k);
        }
        grad_p(gradP);
        interp_alpha();                                     // alphac is 1 here (reset by setSourceZero), kept general
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            double acc[3] = {0, 0, 0};
            for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) {
                const double af = alphaf[d][cface(d, s, i, j, k)];
                if (onb(d, s, i, j, k)) {
                    double b[3]; Ub(U, c, 2 * d + s, b);
                    for (int q = 0; q < 3; ++q) acc[q] += graded ? af * sfd(d, s, i, j, k) * (b[q] - U[3 * (size_t)c + q]) : af * Af * (b[q] - U[3 * (size_t)c + q]) / (0.5 * dx);
                } else {
                    const int nb = c + (s ? stride[d] : -stride[d]);
                    for (int q = 0; q < 3; ++q) acc[q] += graded ? af * sfd(d, s, i, j, k) * (U[3 * (size_t)nb + q] - U[3 * (size_t)c + q]) : af * Af * (U[3 * (size_t)nb + q] - U[3 * (size_t)c + q]) / dx;
                }
            }
            for (int q = 0; q < 3; ++q) divT[3 * (size_t)c + q] = 2 * cs.nu * (acc[q] / vol(i, j, k));
        }
    }

    // ---- momentum matrix ------------------------------------------------------------------------------------
    // ico:    fvm::ddt(U) + fvm::div(phi,U) - fvm::laplacian(nu,U) == uSource                      (icoFoamYade.C:79-85)
    // pimple: fvm::ddt(a,Uc) + fvm::div(aPhi,Uc) - fvm::Sp(fvc::ddt(a)+fvc::div(aPhi),Uc) + divDevRhoReff(Uc) == fvm::Sp(uSourceDrag,Uc)
    //                                                                                              (UcEqn.H:3-10), then relax() (UcEqn.H:12)
    // an[2*d+s] = coefficient of the neighbour across face (d,s); src excludes any pressure term.
    double u_relax_now = 1.0;      // the factor fvMatrix::relax() finds for this outer iteration (Uc / UcFinal)
    vec pPrev;                     // p.prevIter(): stored by pimple.loop() at the start of every outer iteration when p has a relaxation factor
    // boundary value of nut on `patch` next to cell c: 0 zeroGradient, 1 fixedValue, 2 nutkWallFunction [OF-6 nutkWallFunctionFvPatchScalarField::nut():
    // yPlus = Cmu^0.25 y sqrt(k_P)/nu_w; nut_w = nu_w (yPlus kappa/log(E yPlus) - 1) if yPlus > yPlusLam else 0; yPlusLam by ten fixed-point sweeps of
    // ypl = log(max(E ypl, 1))/kappa from 11].  OpenFOAM keeps the values of the last correctNut(); before the first correct() the file's value stands
    bool nut_wall_live = false;
    double nut_boundary(int patch, int c) const {
        const int t = cs.nut_bc[patch];
        if (t == 1 || ((t == 2 || t == 3) && !nut_wall_live)) return cs.nut_value[patch];
        if (t == 3) {       // calculated: nut_ = <model expression> assigns the boundary too [OF-6 GeometricField::operator=]: Ck sqrt(k_b) delta | Cmu k_b^2/eps_b
            const double kb = cs.k_bc[patch] == 1 ? cs.k_value[patch] : kturb[c];
            if (cs.turbulence_model == 2) return cs.les_ck * std::sqrt(kb) * les_delta(c);
            const double eb = cs.eps_bc[patch] == 1 ? cs.eps_value[patch] : epsturb[c];
            return cs.ras_cmu * (kb * kb) / eb;
        }
        if (t == 2) {
            double ypl = 11.0;
            for (int it = 0; it < 10; ++it) ypl = std::log(std::max(cs.wf_E * ypl, 1.0)) / cs.wf_kappa;
            const double y = ywall_of(patch / 2, c);
            const double yPlus = std::pow(cs.ras_cmu, 0.25) * y * std::sqrt(kturb[c]) / cs.nu;
            return yPlus > ypl ? cs.nu * (yPlus * cs.wf_kappa / std::log(cs.wf_E * yPlus) - 1.0) : 0.0;
        }
        return nut[c];
    }
    void assemble_momentum() {
        const double nu = cs.nu, dt = cs.dt;
        if (cs.convection_scheme >= 3) limiter_gradient(U);          // the limiter sees the CURRENT U (the scheme is built when UEqn is assembled)
        if (pimple) {
            // explicit part of divDevRhoReff: + fvc::div(alpha nu dev2(T(grad U))) on the RHS (laminar Stokes model)
            grad_U(U, vGrad);
            vec G(9 * (size_t)Nc);
            for (int c = 0; c < Nc; ++c) {
                const double* T = &vGrad[9 * (size_t)c];
                const double tr = T[0] + T[4] + T[8];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
                    G[9 * (size_t)c + 3 * a + b] = (nut.empty() ? alpha[c] * nu : alpha[c] * (nu + nut[c])) * (T[3 * b + a] - (a == b ? (2.0 / 3.0) * tr : 0.0));   // alpha nuEff dev2(T(grad U)) [OF-6 linearViscousStress::divDevRhoReff]
            }
#pragma omp parallel for num_threads(threads) collapse(2)
            for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
                const int c = cid(i, j, k);
                double acc[3] = {0, 0, 0};
                for (int d = 0; d < 3; ++d) {
                    double fv[2][3];
                    for (int s = 0; s < 2; ++s) {
                        if (onb(d, s, i, j, k)) for (int q = 0; q < 3; ++q) fv[s][q] = G[9 * (size_t)c + 3 * d + q];
                        else {
                            const int nb = c + (s ? stride[d] : -stride[d]);
                            for (int q = 0; q < 3; ++q) {
                                const double gc = G[9 * (size_t)c + 3 * d + q], gn = G[9 * (size_t)nb + 3 * d + q];
                                fv[s][q] = !graded ? 0.5 * (gc + gn) : (s ? lerp(d, fq(d, 1, i, j, k), gc, gn) : lerp(d, fq(d, 0, i, j, k), gn, gc));
                            }
                        }
                    }
                    for (int q = 0; q < 3; ++q) acc[q] += (fv[1][q] - fv[0][q]) / hcell(d, c);
                }
                for (int q = 0; q < 3; ++q) divG[3 * (size_t)c + q] = acc[q];
            }
        }
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            const double aP = pimple ? alpha[c] : 1.0, aP0 = pimple ? alphaOld[c] : 1.0;
            const double Vc = vol(i, j, k);
            double dg = aP * Vc / dt;                                        // fvm::ddt
            if (!bdiag.empty()) for (int q = 0; q < 3; ++q) bdiag[3 * (size_t)c + q] = 0.0;
            double s3[3];
            for (int q = 0; q < 3; ++q) s3[q] = aP0 * Vc * Uold[3 * (size_t)c + q] / dt;
            double divAPhi = 0.0;
            for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) {
                const int f = cface(d, s, i, j, k);
                const double af = pimple ? alphaf[d][f] : 1.0;
                const double phio = (s ? 1.0 : -1.0) * af * phi[d][f];       // outward (alpha-weighted) flux
                divAPhi += phio;
                const double geo = sfd(d, s, i, j, k);                       // |Sf| / |d| (uniform block: dx, the boundary's factor 2 is bfac())
                double gam = nu * af * geo;                                  // (alpha nu)_f |Sf| / |d|
                if (!nut.empty()) {
                    // - fvm::laplacian(alpha nuEff, U) [OF-6 linearViscousStress::divDevRhoReff]: the cell field alpha (nu + nut) is interpolated
                    // linearly to the faces (gaussLaplacianScheme::fvmLaplacian(vol gamma)); boundary value alpha_b (nu + nut_b) with nut_b by 0/nut
                    if (onb(d, s, i, j, k)) {
                        const double nb = nut_boundary(2 * d + s, c);
                        gam = (af * (nu + nb)) * geo;
                    } else {
                        const int nbc = c + (s ? stride[d] : -stride[d]);
                        gam = lerp_side(d, s, d == 0 ? i : d == 1 ? j : k, aP * (nu + nut[c]), alpha[nbc] * (nu + nut[nbc])) * geo;
                    }
                }
                if (onb(d, s, i, j, k)) {
                    an[2 * d + s][c] = 0.0;
                    const int patch = 2 * d + s;
                    if (cs.u_bc[patch] == 0) {                               // fixedValue
                        const double gb = bfac() * gam;
                        dg += gb;
                        for (int q = 0; q < 3; ++q) s3[q] += (-phio + gb) * cs.u_value[patch][q];
                    } else if (cs.u_bc[patch] == 2) {
                        // symmetryPlane / slip [OF-6 transformFvPatchField::gradientInternalCoeffs = -deltaCoeffs * snGradTransformDiag, with
                        // basicSymmetry's snGradTransformDiag = the squared normal's components]: on an axis-aligned patch the NORMAL component
                        // sees a fixed value 0 (implicit coefficient (alpha nu)_f |Sf| deltaCoeffs, boundary source 0), the tangential ones a
                        // zero gradient; the flux through the patch is zero, so convection adds nothing.  A per-component boundary diagonal:
                        // kept apart from the scalar diagonal, as fvMatrix keeps internalCoeffs apart from the lduMatrix
                        bdiag[3 * (size_t)c + d] += bfac() * gam;
                        dg += phio;
                    } else {                                                 // zeroGradient
                        dg += phio;
                    }
                } else {
                    const bool up = cs.convection_scheme == 1 || cs.convection_scheme == 2;      // 1 upwind, 2 linearUpwind (implicit part = upwind)
                    // Gauss linear: the face value is w_P U_P + (1 - w_P) U_N with the linear weights of the (graded) block
                    const double wP = !graded ? 0.5 : (s ? wlow(d, fq(d, 1, i, j, k)) : 1.0 - wlow(d, fq(d, 0, i, j, k)));
                    double cP = up ? std::max(phio, 0.0) : wP * phio;
                    double cN = up ? std::min(phio, 0.0) : (!graded ? 0.5 * phio : (1.0 - wP) * phio);
                    if (cs.convection_scheme >= 3) {
                        const int nb = c + (s ? stride[d] : -stride[d]);
                        const double w = limited_weight(d, fq(d, s, i, j, k), s ? c : nb, s ? nb : c, af * phi[d][f]);
                        cP = s ? phio * w : phio * (1.0 - w);
                        cN = s ? phio * (1.0 - w) : phio * w;
                    }
                    dg += cP + gam;
                    an[2 * d + s][c] = cN - gam;
                    if (cs.convection_scheme == 2) {
                        // linearUpwind [OF-6 linearUpwind::correction]: face value = upwind cell value + (C_f - C_upwind) . grad(U)_upwind,
                        // the second term explicit (deferred correction) with the Gauss-linear gradient of the current U
                        const int nb = c + (s ? stride[d] : -stride[d]);
                        const int uw = phio > 0.0 ? c : nb;
                        const double half = (phio > 0.0 ? (s ? 0.5 : -0.5) : (s ? -0.5 : 0.5)) * hcell(d, uw);      // half the UPWIND cell's extent, towards the face
                        for (int q = 0; q < 3; ++q) s3[q] -= phio * (half * vGrad[9 * (size_t)uw + 3 * d + q]);
                    }
                }
            }
            if (pimple) {
                const double S = (alpha[c] - alphaOld[c]) / dt + divAPhi / Vc;  // fvc::ddt(alphac) + fvc::div(alphaPhic)
                Sc[c] = S;
                dg -= Vc * S;                                                // - fvm::Sp(S, Uc)
                dg -= Vc * uSourceDrag[c];                                   // == fvm::Sp(uSourceDrag, Uc)
                for (int q = 0; q < 3; ++q) s3[q] += Vc * divG[3 * (size_t)c + q];
            } else {
                for (int q = 0; q < 3; ++q) s3[q] += Vc * uSource[3 * (size_t)c + q];  // == uSource
            }
            if (pimple && u_relax_now > 0) {
                // fvMatrix::relax(alpha) [OF-6 fvMatrix.C]: the boundary coefficients join the diagonal for the dominance test (they are
                // part of dg here all along), D = max(|D|, sum|offdiag|) / alpha, source += (D_new - D_old) psi.  No relaxationFactors
                // entry for the equation (alpha <= 0): relax() does nothing at all, not even the dominance step.
                double so = 0.0; for (int q = 0; q < 6; ++q) so += std::fabs(an[q][c]);
                // a slip patch's coefficient differs by component: relax() adds cmptMax(cmptMag(internalCoeffs)) before the test and
                // takes cmptMin(internalCoeffs) (= 0 there) off afterwards, so the whole of it takes part in the dominance test and the
                // scalar diagonal keeps what the test gave
                const double bsum = bdiag.empty() ? 0.0 : bdiag[3 * (size_t)c] + bdiag[3 * (size_t)c + 1] + bdiag[3 * (size_t)c + 2];
                const double dn = std::max(std::fabs(dg + bsum), so) / u_relax_now;
                for (int q = 0; q < 3; ++q) s3[q] += (dn - dg) * U[3 * (size_t)c + q];
                dg = dn;
            }
            diag[c] = dg;
            for (int q = 0; q < 3; ++q) src[3 * (size_t)c + q] = s3[q];
            // 1/UEqn.A(): fvMatrix::A() adds the COMPONENT AVERAGE of the boundary diagonal (addCmptAvBoundaryDiag)
            const double bav = bdiag.empty() ? 0.0 : (bdiag[3 * (size_t)c] + bdiag[3 * (size_t)c + 1] + bdiag[3 * (size_t)c + 2]) / 3.0;
            rAU[c] = 1.0 / ((dg + bav) / Vc);
        }
    }

    // lduMatrix residual normalisation: sum(|A psi - A xbar| + |b - A xbar|) + 1e-20
    // Jacobi solve of diag*x + sum an*x_nb = b for the 3 components (stand-in for smoothSolver symGaussSeidel)
    // diagonal of component q: the scalar one + that component's boundary diagonal (slip patches)
    bool bdiag_on = false;       // (solve_vec3 also solves the k / epsilon equations, which have none)
    inline double dgc(int c, int q) const { return bdiag_on ? diag[c] + bdiag[3 * (size_t)c + q] : diag[c]; }
    int solve_momentum(const vec& b) {
        bdiag_on = !bdiag.empty();
        const int it = solve_vec3(U, b, cs.u_tol, cs.u_rel_tol, cs.u_max_iter);
        bdiag_on = false;
        return it;
    }
    // Jacobi sweeps on (diag, an) for a 3-component field X (updated in place), lduMatrix-style L1 residual control per component
    int solve_vec3(vec& X, const vec& b, double tol, double rel_tol, int max_iter) {
        vec x = X, xn(3 * (size_t)Nc), Ax(3 * (size_t)Nc);
        auto apply = [&](const vec& v, vec& out) {
#pragma omp parallel for num_threads(threads) collapse(2)
            for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
                const int c = cid(i, j, k);
                double acc[3];
                for (int q = 0; q < 3; ++q) acc[q] = dgc(c, q) * v[3 * (size_t)c + q];      // (fvMatrix::solveSegregated: addBoundaryDiag per component)
                for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) if (!onb(d, s, i, j, k)) {
                    const int nb = c + (s ? stride[d] : -stride[d]); const double a = an[2 * d + s][c];
                    for (int q = 0; q < 3; ++q) acc[q] += a * v[3 * (size_t)nb + q];
                }
                for (int q = 0; q < 3; ++q) out[3 * (size_t)c + q] = acc[q];
            }
        };
        double norm[3], res0[3] = {0, 0, 0};
        {
            double xbar[3] = {0, 0, 0};
            for (int c = 0; c < Nc; ++c) for (int q = 0; q < 3; ++q) xbar[q] += x[3 * (size_t)c + q];
            for (int q = 0; q < 3; ++q) xbar[q] /= Nc;
            apply(x, Ax);
            vec ones(3 * (size_t)Nc); for (int c = 0; c < Nc; ++c) for (int q = 0; q < 3; ++q) ones[3 * (size_t)c + q] = xbar[q];
            vec Aref(3 * (size_t)Nc); apply(ones, Aref);
            for (int q = 0; q < 3; ++q) { norm[q] = 0; }
            for (int c = 0; c < Nc; ++c) for (int q = 0; q < 3; ++q) {
                const size_t e = 3 * (size_t)c + q;
                norm[q] += std::fabs(Ax[e] - Aref[e]) + std::fabs(b[e] - Aref[e]);
                res0[q] += std::fabs(b[e] - Ax[e]);
            }
            for (int q = 0; q < 3; ++q) { norm[q] += 1e-20; res0[q] /= norm[q]; }
        }
        auto conv = [&](const double* r) {
            for (int q = 0; q < 3; ++q) if (!(r[q] < tol || (rel_tol > 0 && r[q] < rel_tol * res0[q]))) return false;
            return true;
        };
        int it = 0;
        double res[3] = {res0[0], res0[1], res0[2]};
        while (!conv(res) && it < max_iter) {
#pragma omp parallel for num_threads(threads) collapse(2)
            for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
                const int c = cid(i, j, k);
                double acc[3] = {b[3 * (size_t)c], b[3 * (size_t)c + 1], b[3 * (size_t)c + 2]};
                for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) if (!onb(d, s, i, j, k)) {
                    const int nb = c + (s ? stride[d] : -stride[d]); const double a = an[2 * d + s][c];
                    for (int q = 0; q < 3; ++q) acc[q] -= a * x[3 * (size_t)nb + q];
                }
                for (int q = 0; q < 3; ++q) xn[3 * (size_t)c + q] = acc[q] / dgc(c, q);
            }
            x.swap(xn);
            ++it;
            apply(x, Ax);
            for (int q = 0; q < 3; ++q) res[q] = 0;
            for (int c = 0; c < Nc; ++c) for (int q = 0; q < 3; ++q) res[q] += std::fabs(b[3 * (size_t)c + q] - Ax[3 * (size_t)c + q]);
            for (int q = 0; q < 3; ++q) res[q] /= norm[q];
        }
        X = x;
        return it;
    }

    // H() / V and HbyA = rAU * H with constrainHbyA (icoFoamYade.C:99-100, pEqn.H:2)
    void compute_HbyA() {
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            double acc[3] = {src[3 * (size_t)c], src[3 * (size_t)c + 1], src[3 * (size_t)c + 2]};
            for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) if (!onb(d, s, i, j, k)) {
                const int nb = c + (s ? stride[d] : -stride[d]); const double a = an[2 * d + s][c];
                for (int q = 0; q < 3; ++q) acc[q] -= a * U[3 * (size_t)nb + q];
            }
            if (!bdiag.empty()) {
                // fvMatrix::H(): per component (component-averaged boundary diagonal - that component's boundary diagonal) * psi -- what
                // A() holds too much or too little of for this component goes to H
                const double bav = (bdiag[3 * (size_t)c] + bdiag[3 * (size_t)c + 1] + bdiag[3 * (size_t)c + 2]) / 3.0;
                for (int q = 0; q < 3; ++q) acc[q] += (bav - bdiag[3 * (size_t)c + q]) * U[3 * (size_t)c + q];
            }
            for (int q = 0; q < 3; ++q) HbyA[3 * (size_t)c + q] = rAU[c] * (acc[q] / vol(i, j, k));
        }
    }

    // rAUf = interpolate(rAU) (boundary: extrapolated cell value); pimple keeps rAUcf separately from alphacf
    void interp_rAU() {
        for (int d = 0; d < 3; ++d)
            for (int k = 0; k < nz + (d == 2); ++k) for (int j = 0; j < ny + (d == 1); ++j) for (int i = 0; i < nx + (d == 0); ++i) {
                const int q = d == 0 ? i : d == 1 ? j : k;
                const int f = fid(d, i, j, k);
                if (q == 0) rAUf[d][f] = rAU[cid(i, j, k)];
                else if (q == n[d]) rAUf[d][f] = rAU[cid(i - (d == 0), j - (d == 1), k - (d == 2))];
                else { const int c = cid(i, j, k); rAUf[d][f] = lerp(d, q, rAU[c - stride[d]], rAU[c]); }
            }
    }

    // phicForces = fvc::flux(rAUc*uSource) + rAUcf*(g & Sf)   (UcEqn.H:17-20); uSource's calculated boundary value is 0
    void compute_phi_forces() {
        for (int d = 0; d < 3; ++d)
            for (int k = 0; k < nz + (d == 2); ++k) for (int j = 0; j < ny + (d == 1); ++j) for (int i = 0; i < nx + (d == 0); ++i) {
                const int q = d == 0 ? i : d == 1 ? j : k;
                const int f = fid(d, i, j, k);
                double fl = 0.0;
                const double Afc = area(d, i, j, k);
                if (q != 0 && q != n[d]) { const int c = cid(i, j, k), cm = c - stride[d]; fl = lerp(d, q, rAU[cm] * uSource[3 * (size_t)cm + d], rAU[c] * uSource[3 * (size_t)c + d]) * Afc; }
                phiForces[d][f] = fl + rAUf[d][f] * (cs.g[d] * Afc);
            }
    }

    // phiHbyA = fvc::flux(HbyA) + [alphacf*]rAUf*fvc::ddtCorr(U, phi) [+ phicForces]   (icoFoamYade.C:101-106, pEqn.H:4-18)
    bool adjust_phi_failed = false;
    void compute_phiHbyA() {
        flux_of(HbyA, phiHbyA);
        const double rDt = 1.0 / cs.dt;
        for (int d = 0; d < 3; ++d)
            for (int k = 0; k < nz + (d == 2); ++k) for (int j = 0; j < ny + (d == 1); ++j) for (int i = 0; i < nx + (d == 0); ++i) {
                const int q = d == 0 ? i : d == 1 ? j : k;
                const int f = fid(d, i, j, k);
                double uf;   // (Sf & U.oldTime()_f)
                bool fixes = false;
                const double Afc = area(d, i, j, k);
                if (q == 0) { const int patch = 2 * d; fixes = cs.u_bc[patch] == 0; double b[3]; Ub(Uold, cid(i, j, k), patch, b); uf = b[d] * Afc; }
                else if (q == n[d]) { const int patch = 2 * d + 1; fixes = cs.u_bc[patch] == 0; double b[3]; Ub(Uold, cid(i - (d == 0), j - (d == 1), k - (d == 2)), patch, b); uf = b[d] * Afc; }
                else { const int c = cid(i, j, k); uf = lerp(d, q, Uold[3 * (size_t)(c - stride[d]) + d], Uold[3 * (size_t)c + d]) * Afc; }
                const double phiCorr = phiOld[d][f] - uf;
                // EulerDdtScheme::fvcDdtPhiCoeff: 1 - min(|phiCorr| / (|phi| + small), 1); 0 where U fixes the value
                double coef = fixes ? 0.0 : 1.0 - std::min(std::fabs(phiCorr) / (std::fabs(phiOld[d][f]) + SMALL), 1.0);
                double add = rAUf[d][f] * (coef * rDt * phiCorr);
                if (pimple) add *= alphaf[d][f];
                phiHbyA[d][f] += add;
                if (pimple) phiHbyA[d][f] += phiForces[d][f];
            }
        // adjustPhi(phiHbyA, U, p) (icoFoamYade.C:108, pEqn.H:13-16) [OF-6 adjustPhi.C]: with no fixed-value pressure patch the outflow through
        // the patches that do not fix U is scaled to balance the inflow.  pimpleFoamYade applies it before phicForces are added.
        bool need_ref = true;
        for (int q = 0; q < 6; ++q) if (cs.p_bc[q] == 1) need_ref = false;
        if (need_ref) {
            double massIn = 0, fixedOut = 0, adjOut = 0, total = VSMALL;
            for (int d = 0; d < 3; ++d)
                for (int k = 0; k < nz + (d == 2); ++k) for (int j = 0; j < ny + (d == 1); ++j) for (int i = 0; i < nx + (d == 0); ++i) {
                    const int q = d == 0 ? i : d == 1 ? j : k;
                    const int f = fid(d, i, j, k);
                    const double fl = phiHbyA[d][f] - (pimple ? phiForces[d][f] : 0.0);
                    if (q == 0 || q == n[d]) {
                        const int s = q == 0 ? 0 : 1;
                        const double outw = s ? fl : -fl;
                        if (outw < 0.0) massIn -= outw;
                        else if (cs.u_bc[2 * d + s] == 0) fixedOut += outw;
                        else adjOut += outw;
                    } else total += std::fabs(fl);
                }
            double massCorr = 1.0;
            if (std::fabs(adjOut) > VSMALL && std::fabs(adjOut) / total > SMALL) massCorr = (massIn - fixedOut) / adjOut;
            else if (std::fabs(fixedOut - massIn) / total > 1e-8) adjust_phi_failed = true;
            if (massCorr != 1.0)
                for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) {
                    if (cs.u_bc[2 * d + s] == 0) continue;
                    const int q = s ? n[d] : 0;
                    const int e1 = d == 0 ? ny : nx, e2 = d == 2 ? ny : nz;
                    for (int b2 = 0; b2 < e2; ++b2) for (int b1 = 0; b1 < e1; ++b1) {
                        int i, j, k;
                        if (d == 0) { i = q; j = b1; k = b2; } else if (d == 1) { i = b1; j = q; k = b2; } else { i = b1; j = b2; k = q; }
                        const int f = fid(d, i, j, k);
                        const double pf = pimple ? phiForces[d][f] : 0.0;
                        const double fl = phiHbyA[d][f] - pf;
                        if ((s ? fl : -fl) > 0.0) phiHbyA[d][f] = fl * massCorr + pf;
                    }
                }
        }
        // constrainPressure for fixedFluxPressure patches: snGrad(p) = (phiHbyA - (Sf & U_b)) / (magSf * rAUf), so that the
        // corrected boundary flux phiHbyA - rAUf |Sf| snGrad(p) equals Sf & U_b.
        for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) {
            const int patch = 2 * d + s;
            if (cs.p_bc[patch] != 2) continue;
            const int q = s ? n[d] : 0;
            const int e1 = d == 0 ? ny : nx, e2 = d == 2 ? ny : nz;
            for (int b2 = 0; b2 < e2; ++b2) for (int b1 = 0; b1 < e1; ++b1) {
                int i, j, k;
                if (d == 0) { i = q; j = b1; k = b2; } else if (d == 1) { i = b1; j = q; k = b2; } else { i = b1; j = b2; k = q; }
                const int f = fid(d, i, j, k);
                const int c = cid(i - (d == 0 && s), j - (d == 1 && s), k - (d == 2 && s));
                double ub[3]; Ub(U, c, patch, ub);
                const double Afc = area(d, i, j, k);
                const double target = ub[d] * Afc;
                psn[d][f] = (phiHbyA[d][f] - target) / (rAUf[d][f] * Afc);    // d p / d axis at the face
            }
        }
    }

    // ---- pressure matrix (SPD form): sum_f g_f (p_P - p_nb) [+ g_b (p_P - p_b)] = -(div term) -------------------
    void assemble_pressure(MgLevel& L) {
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            double dg = 0.0, rhs = 0.0;
            for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) {
                const int f = cface(d, s, i, j, k);
                const double af = pimple ? alphaf[d][f] : 1.0;
                double ph = (s ? 1.0 : -1.0) * af * phiHbyA[d][f];
                if (onb(d, s, i, j, k) && cs.p_bc[2 * d + s] == 2)           // fixed-gradient source of the laplacian
                    ph = (s ? 1.0 : -1.0) * af * (phiHbyA[d][f] - rAUf[d][f] * area(d, i, j, k) * psn[d][f]);
                rhs -= ph;
                if (onb(d, s, i, j, k)) {
                    const int patch = 2 * d + s;
                    if (cs.p_bc[patch] == 1) { const double gb = bfac() * af * rAUf[d][f] * sfd(d, s, i, j, k); dg += gb; rhs += gb * cs.p_value[patch]; }
                    if (s) (d == 0 ? L.ux : d == 1 ? L.uy : L.uz)[c] = 0.0;
                } else {
                    const double g = af * rAUf[d][f] * sfd(d, s, i, j, k);
                    dg += g;
                    if (s) (d == 0 ? L.ux : d == 1 ? L.uy : L.uz)[c] = g;
                }
            }
            if (pimple) rhs -= vol(i, j, k) * (alpha[c] - alphaOld[c]) / cs.dt;  // fvc::ddt(alphac), pEqn.H:30
            L.diag[c] = dg; pb[c] = rhs;
        }
        if (need_reference()) {                                              // fvMatrix::setReference
            const int c = cs.p_ref_cell;
            pb[c] += L.diag[c] * cs.p_ref_value;
            L.diag[c] += L.diag[c];
        }
    }

    static void apply(const MgLevel& L, const vec& x, vec& y, int threads) {
        const int nx = L.nx, ny = L.ny, nz = L.nz, sx = 1, sy = nx, sz = nx * ny;
#pragma omp parallel for num_threads(threads) collapse(2)
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = i + nx * (j + ny * k);
            double a = L.diag[c] * x[c];
            if (i > 0) a -= L.ux[c - sx] * x[c - sx];
            if (i < nx - 1) a -= L.ux[c] * x[c + sx];
            if (j > 0) a -= L.uy[c - sy] * x[c - sy];
            if (j < ny - 1) a -= L.uy[c] * x[c + sy];
            if (k > 0) a -= L.uz[c - sz] * x[c - sz];
            if (k < nz - 1) a -= L.uz[c] * x[c + sz];
            y[c] = a;
        }
    }

    void build_mg_shapes() {
        mg.clear();
        int ax = nx, ay = ny, az = nz;
        for (;;) {
            MgLevel L; L.nx = ax; L.ny = ay; L.nz = az; L.N = ax * ay * az;
            L.diag.assign(L.N, 0.0); L.ux = L.diag; L.uy = L.diag; L.uz = L.diag; L.x = L.diag; L.b = L.diag; L.r = L.diag;
            mg.push_back(L);
            if (cs.p_solver != 1) break;
            if ((L.N <= kMgCoarsest && std::max(std::max(ax, ay), az) <= kMgCoarsestEdge) || (ax <= 2 && ay <= 2 && az <= 2)) break;   // (no elongated coarsest grid: kMgCoarseSweeps Jacobi sweeps must solve it)
            ax = (ax + 1) / 2; ay = (ay + 1) / 2; az = (az + 1) / 2;
        }
    }

    // A_coarse = 1/2 P^T A P with piecewise-constant P over 2x2x2 aggregates (the 1/2 makes it equal to re-discretisation)
    void coarsen_operators() {
        for (size_t l = 0; l + 1 < mg.size(); ++l) {
            const MgLevel& F = mg[l]; MgLevel& Cc = mg[l + 1];
            std::fill(Cc.diag.begin(), Cc.diag.end(), 0.0); std::fill(Cc.ux.begin(), Cc.ux.end(), 0.0);
            std::fill(Cc.uy.begin(), Cc.uy.end(), 0.0); std::fill(Cc.uz.begin(), Cc.uz.end(), 0.0);
            for (int k = 0; k < F.nz; ++k) for (int j = 0; j < F.ny; ++j) for (int i = 0; i < F.nx; ++i) {
                const int c = i + F.nx * (j + F.ny * k);
                const int I = i >> 1, J = j >> 1, K = k >> 1;
                const int C = I + Cc.nx * (J + Cc.ny * K);
                Cc.diag[C] += 0.5 * F.diag[c];
                if (i < F.nx - 1) { if (((i + 1) >> 1) == I) Cc.diag[C] -= F.ux[c]; else Cc.ux[C] += 0.5 * F.ux[c]; }   // -2 * 0.5 * g for internal faces
                if (j < F.ny - 1) { if (((j + 1) >> 1) == J) Cc.diag[C] -= F.uy[c]; else Cc.uy[C] += 0.5 * F.uy[c]; }
                if (k < F.nz - 1) { if (((k + 1) >> 1) == K) Cc.diag[C] -= F.uz[c]; else Cc.uz[C] += 0.5 * F.uz[c]; }
            }
            // setReference's point term a_ref (half of the doubled level-0 diagonal of the reference cell) is what makes 1^T A 1 = a_ref;
            // the 1/2 of the Galerkin product would halve it on every level and the exact coarse solve would over-correct the constant mode
            // by 2^levels: the aggregate that holds the reference cell gets the missing half back (the product: k_mg_coarsen)
            if (need_reference()) {
                const int sh = (int)l + 1;
                const int I = (cs.p_ref_cell % nx) >> sh, J = ((cs.p_ref_cell / nx) % ny) >> sh, K = (cs.p_ref_cell / (nx * ny)) >> sh;
                Cc.diag[I + Cc.nx * (J + Cc.ny * K)] += 0.5 * (0.5 * mg[0].diag[cs.p_ref_cell]);
            }
        }
        invert_coarsest();
    }

    // explicit inverse of the coarsest operator, rebuilt with the operators (the product: k_mg_coarse_invert): dense, in-place Gauss-Jordan
    // without pivoting (symmetric positive definite M-matrix); coarse_inv_ok = false if a pivot is not positive and finite
    vec coarse_inv;
    bool coarse_inv_ok = false;
    void invert_coarsest() {
        const MgLevel& L = mg.back();
        const int N = L.N, sy = L.nx, sz = L.nx * L.ny;
        coarse_inv_ok = false;
        if (N > kMgCoarsest) return;
        vec& M = coarse_inv;
        M.assign((size_t)N * N, 0.0);
        for (int k = 0; k < L.nz; ++k) for (int j = 0; j < L.ny; ++j) for (int i = 0; i < L.nx; ++i) {      // the rows of apply()
            const int c = i + L.nx * (j + L.ny * k);
            double* row = &M[(size_t)c * N];
            row[c] = L.diag[c];
            if (i > 0) row[c - 1] = -L.ux[c - 1];
            if (i < L.nx - 1) row[c + 1] = -L.ux[c];
            if (j > 0) row[c - sy] = -L.uy[c - sy];
            if (j < L.ny - 1) row[c + sy] = -L.uy[c];
            if (k > 0) row[c - sz] = -L.uz[c - sz];
            if (k < L.nz - 1) row[c + sz] = -L.uz[c];
        }
        vec colp(N), rowp(N);
        for (int p = 0; p < N; ++p) {
            const double piv = M[(size_t)p * N + p];
            if (!(piv > 0.0) || !(piv < 1e300)) return;
            const double ip = 1.0 / piv;
            for (int q = 0; q < N; ++q) { colp[q] = M[(size_t)q * N + p]; rowp[q] = M[(size_t)p * N + q]; }
            for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) {
                double v;
                if (i == p) v = (j == p) ? ip : rowp[j] * ip;
                else if (j == p) v = -colp[i] * ip;
                else v = M[(size_t)i * N + j] - colp[i] * (rowp[j] * ip);
                M[(size_t)i * N + j] = v;
            }
        }
        coarse_inv_ok = true;
    }

    // The smoother.  Two sweeps as a pair are the degree-2 Chebyshev polynomial of D^-1 A on [1/3, 2] -- the high-frequency band of the
    // 7-point operator under 2 x 2 x 2 coarsening, 2 bounding every diagonally dominant level -- i.e. Jacobi with the weights
    // 1 / (7/6 -+ (5/6) cos(pi/4)): the pair damps that band by 0.34 where two sweeps at the fixed weight 0.8 reach 0.54, for the same
    // work.  Pre-smoothing applies (kMgWa, kMgWb), post-smoothing the reverse order (the adjoint; the V-cycle stays a symmetric
    // positive definite preconditioner).  The kMgCoarseSweeps (120) sweeps of the coarsest level keep the fixed weight.  Same constants as the product.
    static constexpr double kMgWa = 1.7318685872766142, kMgWb = 0.5695012757370842;
    static void jacobi(const MgLevel& L, vec& x, const vec& b, vec& tmp, int sweeps, bool zero_guess, int threads) {
        for (int s = 0; s < sweeps; ++s) {
            double w = 0.8;
            if (sweeps == 2) w = zero_guess ? (s == 0 ? kMgWa : kMgWb) : (s == 0 ? kMgWb : kMgWa);
            if (s == 0 && zero_guess) { for (int c = 0; c < L.N; ++c) x[c] = w * b[c] / L.diag[c]; continue; }
            apply(L, x, tmp, threads);
            for (int c = 0; c < L.N; ++c) x[c] += w * (b[c] - tmp[c]) / L.diag[c];
        }
    }

    void vcycle(size_t l) {
        MgLevel& L = mg[l];
        if (l + 1 == mg.size()) {
            if (coarse_inv_ok) {
                for (int r = 0; r < L.N; ++r) {
                    double acc = 0.0;
                    for (int c = 0; c < L.N; ++c) acc += coarse_inv[(size_t)r * L.N + c] * L.b[c];
                    L.x[r] = acc;
                }
            } else {
                jacobi(L, L.x, L.b, L.r, kMgCoarseSweeps, true, 1);
            }
            return;
        }
        jacobi(L, L.x, L.b, L.r, 2, true, threads);
        vec& r = L.r;
        apply(L, L.x, r, threads);
        for (int c = 0; c < L.N; ++c) r[c] = L.b[c] - r[c];
        MgLevel& Cc = mg[l + 1];
        std::fill(Cc.b.begin(), Cc.b.end(), 0.0);
        for (int k = 0; k < L.nz; ++k) for (int j = 0; j < L.ny; ++j) for (int i = 0; i < L.nx; ++i)
            Cc.b[(i >> 1) + Cc.nx * ((j >> 1) + Cc.ny * (k >> 1))] += r[i + L.nx * (j + L.ny * k)];
        vcycle(l + 1);
        for (int k = 0; k < L.nz; ++k) for (int j = 0; j < L.ny; ++j) for (int i = 0; i < L.nx; ++i)
            L.x[i + L.nx * (j + L.ny * k)] += Cc.x[(i >> 1) + Cc.nx * ((j >> 1) + Cc.ny * (k >> 1))];
        jacobi(L, L.x, L.b, L.r, 2, false, threads);
    }

    void precondition(const vec& r, vec& z) {
        MgLevel& L = mg[0];
        if (cs.p_solver == 1) { L.b = r; vcycle(0); z = L.x; }
        else for (int c = 0; c < Nc; ++c) z[c] = r[c] / L.diag[c];
    }

    // OpenFOAM PCG.C, with lduMatrix::solver normFactor; returns iterations
    int solve_pressure(bool final_iter) {
        MgLevel& L = mg[0];
        const double tol = final_iter ? cs.p_final_tol : cs.p_tol, rel = final_iter ? cs.p_final_rel_tol : cs.p_rel_tol;
        vec& wA = pw; vec& rA = pr; vec& pA = pp; vec& zA = pz;
        apply(L, p, wA, threads);
        double xbar = 0; for (int c = 0; c < Nc; ++c) xbar += p[c]; xbar /= Nc;
        { vec ones(Nc, xbar); apply(L, ones, pA, threads); }
        double norm = 0, res = 0;
        for (int c = 0; c < Nc; ++c) { norm += std::fabs(wA[c] - pA[c]) + std::fabs(pb[c] - pA[c]); rA[c] = pb[c] - wA[c]; res += std::fabs(rA[c]); }
        norm += 1e-20; res /= norm;
        const double res0 = res;
        st.p_initial_residual = res0;
        int it = 0;
        double wArAold = 1.0;
        auto converged = [&](double r) { return r < tol || (rel > 0 && r < rel * res0); };
        if (!converged(res)) {
            do {
                precondition(rA, zA);
                double wArA = 0; for (int c = 0; c < Nc; ++c) wArA += zA[c] * rA[c];
                if (it == 0) for (int c = 0; c < Nc; ++c) pA[c] = zA[c];
                else { const double beta = wArA / wArAold; for (int c = 0; c < Nc; ++c) pA[c] = zA[c] + beta * pA[c]; }
                apply(L, pA, wA, threads);
                double wApA = 0; for (int c = 0; c < Nc; ++c) wApA += wA[c] * pA[c];
                const double al = wArA / wApA;
                res = 0;
                for (int c = 0; c < Nc; ++c) { p[c] += al * pA[c]; rA[c] -= al * wA[c]; res += std::fabs(rA[c]); }
                res /= norm;
                wArAold = wArA;
            } while (++it < cs.p_max_iter && !converged(res));
        }
        st.p_final_residual = res;
        st.p_iters_total += it; st.p_solves += 1;
        return it;
    }

    // pEqn.flux(): g_f (p_hi - p_lo), +axis oriented, incl. fixedValue boundaries; fixedFlux boundaries carry what removes the forcing
    void pressure_flux(const MgLevel& L) {
        for (int d = 0; d < 3; ++d)
            for (int k = 0; k < nz + (d == 2); ++k) for (int j = 0; j < ny + (d == 1); ++j) for (int i = 0; i < nx + (d == 0); ++i) {
                const int q = d == 0 ? i : d == 1 ? j : k;
                const int f = fid(d, i, j, k);
                const double af = pimple ? alphaf[d][f] : 1.0;
                double fl = 0.0;
                if (q == 0 || q == n[d]) {
                    const int s = q == 0 ? 0 : 1, patch = 2 * d + s;
                    const int c = cid(i - (d == 0 && s), j - (d == 1 && s), k - (d == 2 && s));
                    const double geo = graded ? area(d, i, j, k) / delta(d, q) : dx;
                    if (cs.p_bc[patch] == 1) { const double gb = bfac() * af * rAUf[d][f] * geo; fl = s ? gb * (cs.p_value[patch] - p[c]) : gb * (p[c] - cs.p_value[patch]); }
                    else if (cs.p_bc[patch] == 2) fl = af * rAUf[d][f] * area(d, i, j, k) * psn[d][f];
                } else {
                    const int c = cid(i, j, k);
                    const double geo = graded ? area(d, i, j, k) / delta(d, q) : dx;
                    fl = af * rAUf[d][f] * geo * (p[c] - p[c - stride[d]]);
                }
                pflux[d][f] = fl;
            }
    }

    void continuity_errors() {
        double sl = 0, gl = 0;
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            double dv = 0;
            for (int d = 0; d < 3; ++d) for (int s = 0; s < 2; ++s) { const int f = cface(d, s, i, j, k); dv += (s ? 1.0 : -1.0) * (pimple ? alphaf[d][f] : 1.0) * phi[d][f]; }
            const double Vc = vol(i, j, k);
            double ce = dv / Vc;
            if (pimple) ce += (alpha[c] - alphaOld[c]) / cs.dt;
            sl += std::fabs(ce) * Vc; gl += ce * Vc;
        }
        const double tv = graded ? total_volume : V * Nc;
        st.cont_sum_local = cs.dt * sl / tv; st.cont_global = cs.dt * gl / tv;
        cumulativeContErr += st.cont_global; st.cont_cumulative = cumulativeContErr;
    }

    // fvc::reconstruct(s_f) on the uniform block: per axis (s_{f+} + s_{f-}) / (2 |Sf|)
    // ---- one PISO/PIMPLE corrector (icoFoamYade.C:97-140 / pEqn.H) -------------------------------------------
    double p_relax_now = 0.0;
    void corrector(bool final_inner) {
        compute_HbyA();
        if (!pimple) interp_rAU();
        compute_phiHbyA();
        MgLevel& L = mg[0];
        for (int no = 0; no <= cs.n_non_orth; ++no) {
            assemble_pressure(L);
            if (cs.p_solver == 1) coarsen_operators();
            solve_pressure(final_inner && no == cs.n_non_orth);
            if (no == cs.n_non_orth) {
                pressure_flux(L);
                for (int d = 0; d < 3; ++d) for (size_t f = 0; f < phi[d].size(); ++f)
                    phi[d][f] = phiHbyA[d][f] - pflux[d][f] / (pimple ? alphaf[d][f] : 1.0);      // icoFoamYade.C:129, pEqn.H:39
                if (pimple && p_relax_now > 0 && p_relax_now < 1)                                     // p.relax(), pEqn.H:41 (after the flux: the flux keeps the unrelaxed solution)
                    for (int c = 0; c < Nc; ++c) p[c] = pPrev[c] + p_relax_now * (p[c] - pPrev[c]);
            }
        }
        continuity_errors();                                                 // icoFoamYade.C:134, pEqn.H:50
        if (!pimple) {
            grad_p(gradP);                                                   // U = HbyA - rAU*fvc::grad(p), icoFoamYade.C:136
            for (int c = 0; c < Nc; ++c) for (int q = 0; q < 3; ++q) U[3 * (size_t)c + q] = HbyA[3 * (size_t)c + q] - rAU[c] * gradP[3 * (size_t)c + q];
        } else {
            // Uc = HbyA + rAUc*fvc::reconstruct((phicForces - pEqn.flux()/alphacf)/rAUcf), pEqn.H:43-45
#pragma omp parallel for num_threads(threads) collapse(2)
            for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
                const int c = cid(i, j, k);
                for (int d = 0; d < 3; ++d) {
                    double sm = 0;
                    for (int s = 0; s < 2; ++s) { const int f = cface(d, s, i, j, k); sm += (phiForces[d][f] - pflux[d][f] / alphaf[d][f]) / rAUf[d][f]; }
                    U[3 * (size_t)c + d] = HbyA[3 * (size_t)c + d] + rAU[c] * (sm / (2.0 * area(d, i, j, k)));
                }
            }
        }
    }

    // ---- one pass of the while(runTime.loop()) body, with the coupling call injected by the caller --------------
    void step_begin() {            // up to (excluding) yadeCoupling.setParticleAction
        st = orc_fv_stats{}; st.cont_cumulative = cumulativeContErr;
        courant();                                                           // icoFoamYade.C:68, pimpleFoamYade.C:63
        if (cs.adjust_time_step) {                                           // setDeltaT.H [OF-6], pimpleFoamYade.C:64
            const double maxDeltaTFact = cs.max_co / (st.courant_max + SMALL);
            const double deltaTFact = std::min(std::min(maxDeltaTFact, 1.0 + 0.1 * maxDeltaTFact), 1.2);
            cs.dt = std::min(deltaTFact * cs.dt, cs.max_delta_t);
        }
        st.delta_t = cs.dt;
        // runTime++ : old-time fields (U.oldTime(), phi.oldTime(), alphac.oldTime())
        Uold = U; for (int d = 0; d < 3; ++d) phiOld[d] = phi[d];
        pre_coupling_fields();                                               // icoFoamYade.C:71, pimpleFoamYade.C:73-76
    }
    void step_end() {              // everything after setParticleAction, up to (excluding) setSourceZero
        // alphac.oldTime(): OpenFOAM captures old-time values lazily, at the first *tracked* access of a field in a new time
        // step (GeometricField::storeOldTimes, called from oldTime() and from the non-const accessors).  FoamYade writes alpha
        // through UList::operator[] (FoamYade.C:324,560), which is untracked, so the first tracked access of alphac in a step is
        // alphac.correctBoundaryConditions() at pimpleFoamYade.C:83 -- AFTER setParticleAction.  Hence alphac.oldTime() equals
        // this step's alpha and fvc::ddt(alphac) == 0 in the reference.  [OF-6 semantics restated from memory; parity unpinned]
        alphaOld = alpha;
        if (pimple) interp_alpha();                                          // pimpleFoamYade.C:83-85 (alphaPhic is formed on the fly)
        const int nOuter = pimple ? std::max(cs.n_outer, 1) : 1;
        for (int outer = 0; outer < nOuter; ++outer) {
            // pimple.loop(): "finalIteration" is set on the last outer corrector; fvMatrix::relax() / GeometricField::relax() then look
            // for the <name>Final factor first [OF-6 solutionControl / fvMatrix.C / GeometricField.C]
            const bool final_outer = outer == nOuter - 1;
            u_relax_now = (final_outer && cs.u_relax_final > 0) ? cs.u_relax_final : cs.u_relax;
            p_relax_now = (final_outer && cs.p_relax_final > 0) ? cs.p_relax_final : cs.p_relax;
            if (pimple && p_relax_now > 0 && p_relax_now < 1) pPrev = p;          // storePrevIterFields()
            assemble_momentum();
            if (pimple) { interp_rAU(); compute_phi_forces(); }
            if (cs.momentum_predictor) {
                if (!pimple) {                                               // solve(UEqn == -fvc::grad(p)), icoFoamYade.C:91-94
                    grad_p(gradP);
                    for (int c = 0; c < Nc; ++c) for (int q = 0; q < 3; ++q) bmom[3 * (size_t)c + q] = src[3 * (size_t)c + q] - volc(c) * gradP[3 * (size_t)c + q];
                } else {                                                     // UcEqn.H:22-33
                    for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
                        const int c = cid(i, j, k);
                        for (int d = 0; d < 3; ++d) {
                            double sm = 0;
                            for (int s = 0; s < 2; ++s) {
                                const int f = cface(d, s, i, j, k);
                                double sng;   // snGrad(p) along +axis
                                const double dl = delta(d, fq(d, s, i, j, k));
                                if (onb(d, s, i, j, k)) { const double pbd = pbv(c, d, s, f); sng = s ? (pbd - p[c]) / dl : (p[c] - pbd) / dl; }
                                else sng = s ? (p[c + stride[d]] - p[c]) / dl : (p[c] - p[c - stride[d]]) / dl;
                                sm += phiForces[d][f] / rAUf[d][f] - sng * area(d, i, j, k);
                            }
                            bmom[3 * (size_t)c + d] = src[3 * (size_t)c + d] + vol(i, j, k) * (sm / (2.0 * area(d, i, j, k)));
                        }
                    }
                }
                st.u_iters_total += solve_momentum(bmom);
            }
            for (int corr = 0; corr < cs.n_corr; ++corr) corrector(outer == nOuter - 1 && corr == cs.n_corr - 1);
            if (!nut.empty() && final_outer) turbulence_correct();             // pimple.turbCorr() (final outer iteration), pimpleFoamYade.C:101-104
        }
    }
    // The transport equations of LESModel kEqn (DPMTurbulenceModels.C:76-77) [OF-6 LES/kEqn/kEqn.C] and RASModel kEpsilon
    // (DPMTurbulenceModels.C:70-71) [OF-6 RAS/kEpsilon/kEpsilon.C]:
    //   divU = fvc::div(fvc::absolute(phi, U)); G = nut (gradU && dev(twoSymm(gradU)));
    //   fvm::ddt(alpha, rho, X) + fvm::div(alphaRhoPhi, X) - fvm::laplacian(alpha rho DXEff(), X) == Su - fvm::SuSp(c1, X) - fvm::Sp(c2, X)
    //   mode 0 (kEqn, X = k):       Su = alpha G,           c1 = 2/3 alpha divU,           c2 = Ce alpha sqrt(k)/delta, DkEff = nut + nu
    //   mode 1 (kEpsilon, X = eps): Su = C1 alpha G eps/k,  c1 = (2/3 C1 - C3) alpha divU, c2 = C2 alpha eps/k,         DepsilonEff = nut/sigmaEps + nu
    //   mode 2 (kEpsilon, X = k):   Su = alpha G,           c1 = 2/3 alpha divU,           c2 = alpha eps/k,            DkEff = nut/sigmak + nu
    //   X.relax(); solve; bound(X, XMin); correctNut(): kEqn nut = Ck sqrt(k) delta, kEpsilon (after the k equation) nut = Cmu k^2/eps
    // [OF-6 fvm::SuSp: diag += V max(susp, 0), source -= V min(susp, 0) psi; fvMatrix == volField: source += V field;
    //  bound.C: X = max(max(X, fvc::average(max(X, XMin)) pos0(-X)), XMin)].  alpha.oldTime() == alpha (quirk F-Q1).
    void turb_eqn(int mode) {
        const double nu = cs.nu, dt = cs.dt, xMin = 1e-15;
        const double sigma = mode == 0 ? 1.0 : mode == 1 ? cs.ras_sigmaeps : cs.ras_sigmak;
        int bcv[6];
        for (int q = 0; q < 6; ++q) bcv[q] = mode == 1 ? (cs.eps_bc[q] == 2 ? 0 : cs.eps_bc[q]) : cs.k_bc[q];      // an epsilonWallFunction patch is never asked for a face value: the wall cells are imposed
        const int* bc = bcv;
        auto wallp = [&](int patch) { return mode != 0 && cs.eps_bc[patch] == 2; };
        auto wall_count = [&](int i, int j, int k) { int w = 0; for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) if (wallp(2 * d + sd) && onb(d, sd, i, j, k)) ++w; return w; };
        const double cmu75 = std::pow(cs.ras_cmu, 0.75), cmu25 = std::pow(cs.ras_cmu, 0.25);
        // the value epsilonWallFunction imposes on a cell with wall faces: the average over them of Cmu^3/4 k^3/2 / (kappa y_w)
        auto wall_eps = [&](int c, int i, int j, int k) {
            if (!graded) return cmu75 * std::pow(kturb[c], 1.5) / (cs.wf_kappa * ywall_of(0, c));
            double sum = 0.0; int W = 0;
            for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) if (wallp(2 * d + sd) && onb(d, sd, i, j, k)) { sum += cmu75 * std::pow(kturb[c], 1.5) / (cs.wf_kappa * ywall_of(d, c)); ++W; }
            return sum / (double)W;
        };
        const double* bval = mode == 1 ? cs.eps_value : cs.k_value;
        const int scheme = mode == 1 ? cs.eps_convection_scheme : cs.k_convection_scheme;
        const double relax = mode == 1 ? cs.eps_relax : cs.k_relax;
        vec& Xf = mode == 1 ? epsturb : kturb;
        vec b3(3 * (size_t)Nc, 0.0), x3(3 * (size_t)Nc, 0.0);
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            const double aP = alpha[c], xc = Xf[c], nutc = nut[c];
            const double V = vol(i, j, k), delta = les_delta(c);
            double dg = aP * V / dt, s = aP * V * xc / dt, sumPhi = 0.0;
            for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) {
                const int f = cface(d, sd, i, j, k);
                const double af = alphaf[d][f];
                const double pv = (sd ? 1.0 : -1.0) * phi[d][f];
                const double phio = af * pv;
                sumPhi += pv;
                if (onb(d, sd, i, j, k)) {
                    an[2 * d + sd][c] = 0.0;
                    const int patch = 2 * d + sd;
                    const double nb = nut_boundary(patch, c);
                    const double gam = (af * (nu + nb / sigma)) * sfd(d, sd, i, j, k);
                    if (bc[patch] == 1) { const double gb = bfac() * gam; dg += gb; s += (-phio + gb) * bval[patch]; }
                    else dg += phio;
                } else {
                    const int nbc = c + (sd ? stride[d] : -stride[d]);
                    const int qc = d == 0 ? i : d == 1 ? j : k;
                    const double gam = lerp_side(d, sd, qc, aP * (nu + nutc / sigma), alpha[nbc] * (nu + nut[nbc] / sigma)) * sfd(d, sd, i, j, k);
                    const bool up = scheme != 0;
                    const double wP = !graded ? 0.5 : (sd ? wlow(d, qc + 1) : 1.0 - wlow(d, qc));
                    const double cP = up ? std::max(phio, 0.0) : wP * phio, cN = up ? std::min(phio, 0.0) : (!graded ? 0.5 * phio : (1.0 - wP) * phio);
                    dg += cP + gam;
                    an[2 * d + sd][c] = cN - gam;
                }
            }
            const double* T = &vGrad[9 * (size_t)c];
            const double tr2 = 2.0 * (T[0] + T[4] + T[8]);
            double GG = 0.0;
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) GG += T[3 * a + b] * ((T[3 * a + b] + T[3 * b + a]) - (a == b ? (1.0 / 3.0) * tr2 : 0.0));
            double G = nutc * GG;
            // epsilonWallFunction [OF-6 epsilonWallFunctionFvPatchScalarField::calculate / updateCoeffs]: in the wall cells
            //   G = sum_f w (nut_w + nu_w) |snGrad U_w| Cmu^0.25 sqrt(k)/(kappa y),  eps = sum_f w Cmu^0.75 k^1.5/(kappa y),  w = 1/(wall faces of the cell)
            const int Wc = wall_count(i, j, k);
            if (Wc) {
                double Gw = 0.0;
                for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) if (wallp(2 * d + sd) && onb(d, sd, i, j, k)) {
                    const int patch = 2 * d + sd;
                    double ub[3]; Ub(U, c, patch, ub);
                    const double ywall = ywall_of(d, c);
                    const double d0 = (ub[0] - U[3 * (size_t)c]) / ywall, d1 = (ub[1] - U[3 * (size_t)c + 1]) / ywall, d2 = (ub[2] - U[3 * (size_t)c + 2]) / ywall;
                    Gw += (nut_boundary(patch, c) + nu) * std::sqrt(d0 * d0 + d1 * d1 + d2 * d2) * cmu25 * std::sqrt(kturb[c]) / (cs.wf_kappa * ywall);
                }
                G = Gw / (double)Wc;
            }
            const double divU = sumPhi / V;
            double Su, c1, c2;
            if (mode == 0) { Su = aP * G; c1 = (2.0 / 3.0) * aP * divU; c2 = cs.les_ce * aP * std::sqrt(xc) / delta; }
            else if (mode == 1) { const double kc = kturb[c]; Su = cs.ras_c1 * aP * G * xc / kc; c1 = ((2.0 / 3.0) * cs.ras_c1 - cs.ras_c3) * aP * divU; c2 = cs.ras_c2 * aP * xc / kc; }
            else { Su = aP * G; c1 = (2.0 / 3.0) * aP * divU; c2 = aP * epsturb[c] / xc; }
            dg += V * (std::max(c1, 0.0) + c2);
            s += V * Su - V * std::min(c1, 0.0) * xc;
            if (relax > 0) {
                double so = 0.0;
                for (int q = 0; q < 6; ++q) so += std::fabs(an[q][c]);
                const double dn = std::max(std::fabs(dg), so) / relax;
                s += (dn - dg) * xc;
                dg = dn;
            }
            double x0 = xc;
            if (mode == 1) {
                // epsEqn.boundaryManipulate -> fvMatrix::setValues(faceCells, value) [OF-6 fvMatrix.C setValuesFromList]: the wall cell's row becomes
                // diag x = diag value, its neighbours' coefficients towards it move to their sources
                if (Wc) {
                    const double v = wall_eps(c, i, j, k);
                    for (int q = 0; q < 6; ++q) an[q][c] = 0.0;
                    s = dg * v; x0 = v;
                } else {
                    for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) if (!onb(d, sd, i, j, k)) {
                        const int ni = i + (d == 0 ? (sd ? 1 : -1) : 0), nj = j + (d == 1 ? (sd ? 1 : -1) : 0), nk = k + (d == 2 ? (sd ? 1 : -1) : 0);
                        if (wall_count(ni, nj, nk)) {
                            const int nbc = c + (sd ? stride[d] : -stride[d]);
                            s -= an[2 * d + sd][c] * wall_eps(nbc, ni, nj, nk);
                            an[2 * d + sd][c] = 0.0;
                        }
                    }
                }
            }
            diag[c] = dg;
            b3[3 * (size_t)c] = s;
            x3[3 * (size_t)c] = x0;
        }
        k_iters += mode == 1 ? solve_vec3(x3, b3, cs.eps_tol, cs.eps_rel_tol, cs.eps_max_iter) : solve_vec3(x3, b3, cs.k_tol, cs.k_rel_tol, cs.k_max_iter);
        vec xn(Nc);
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            const int c = cid(i, j, k);
            const double xc = x3[3 * (size_t)c];
            double xb = xc;
            if (!(xc > 0.0)) {
                const double mP = std::max(xc, xMin);
                double av = 0.0, asum = 0.0;                      // fvc::average: sum |Sf| x_f / sum |Sf| (six equal faces on the uniform block)
                for (int d = 0; d < 3; ++d) for (int sd = 0; sd < 2; ++sd) {
                    double xf;
                    if (onb(d, sd, i, j, k)) { const int patch = 2 * d + sd; xf = std::max(bc[patch] == 1 ? bval[patch] : xc, xMin); }
                    else { const int nbc = c + (sd ? stride[d] : -stride[d]); xf = lerp_side(d, sd, d == 0 ? i : d == 1 ? j : k, mP, std::max(x3[3 * (size_t)nbc], xMin)); }
                    if (graded) { av += area(d, i, j, k) * xf; asum += area(d, i, j, k); } else av += xf;
                }
                xb = std::max(xc, graded ? av / asum : av / 6.0);
            }
            xn[c] = std::max(xb, xMin);
        }
        Xf = xn;
        if (mode == 0) for (int c = 0; c < Nc; ++c) nut[c] = cs.les_ck * std::sqrt(kturb[c]) * les_delta(c);
        else if (mode == 2) for (int c = 0; c < Nc; ++c) nut[c] = cs.ras_cmu * (kturb[c] * kturb[c]) / epsturb[c];
        if (mode != 1) nut_wall_live = true;          // correctNut(): the wall-function patches now carry nut_w(k)
    }
    // continuousPhaseTurbulence->correct() for LESModel Smagorinsky [OF-6 LES/Smagorinsky/Smagorinsky.C: correct() -> correctNut();
    // k(gradU): D = symm(gradU), a = Ce/delta, b = (2/3) tr(D), c = 2 Ck delta (dev(D) && D), k = sqr((-b + sqrt(sqr(b) + 4 a c))/(2 a));
    // nut = Ck delta sqrt(k)]; delta = deltaCoeff * cbrt(V) [OF-6 LES/LESdeltas/cubeRootVolDelta]; gradU = fvc::grad(U) (Gauss linear)
    void turbulence_correct() {
        grad_U(U, vGrad);
        if (cs.turbulence_model == 2) { turb_eqn(0); return; }
        if (cs.turbulence_model == 3) { turb_eqn(1); turb_eqn(2); return; }      // kEpsilon::correct(): epsilon first, then k with the new epsilon
        const double Ck = cs.les_ck, Ce = cs.les_ce;
        for (int c = 0; c < Nc; ++c) {
            const double delta = les_delta(c);
            const double* T = &vGrad[9 * (size_t)c];
            double D[3][3];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) D[a][b] = 0.5 * (T[3 * a + b] + T[3 * b + a]);
            const double trD = D[0][0] + D[1][1] + D[2][2];
            const double a = Ce / delta, b = (2.0 / 3.0) * trD;
            const double third = (1.0 / 3.0) * trD;
            const double dd = (D[0][0] - third) * D[0][0] + (D[1][1] - third) * D[1][1] + (D[2][2] - third) * D[2][2]
                            + 2.0 * (D[0][1] * D[0][1]) + 2.0 * (D[0][2] * D[0][2]) + 2.0 * (D[1][2] * D[1][2]);
            const double cc = 2.0 * Ck * delta * dd;
            const double r = (-b + std::sqrt(b * b + 4.0 * a * cc)) / (2.0 * a);
            nut[c] = Ck * delta * std::sqrt(r * r);
        }
    }
};

}  // namespace

extern "C" {

// the limiter function of an NVD / TVD scheme at gradient ratio r (tests: Sweby's region)
double orc_fv_limiter(int scheme, double k, double r) { return Fv::limiter_fn(scheme, 2.0 / std::max(k, SMALL), r); }
void* orc_fv_create(const orc_fv_case* c) {
    // a graded block carries the laminar operators with Gauss linear / upwind convection only (the closures' delta, wall distance and the
    // linearUpwind correction assume uniform cubes)
    Fv* f = new Fv(); f->init(*c); return f;
}
void orc_fv_destroy(void* h) { delete (Fv*)h; }
void orc_fv_turbulence_correct(void* h) { Fv* f = (Fv*)h; if (!f->nut.empty()) f->turbulence_correct(); }
void orc_fv_set_threads(void* h, int t) { ((Fv*)h)->threads = t < 1 ? 1 : t; }

static vec* fv_field(Fv* f, const char* name) {
    const std::string s = name;
    const struct { const char* nm; vec* v; } tab[] = {
        {"U", &f->U},
        {"p", &f->p},
        {"alpha", &f->alpha},
        {"uSource", &f->uSource},
        {"uSourceDrag", &f->uSourceDrag},
        {"uParticle", &f->uParticle},
        {"gradP", &f->gradP},
        {"divT", &f->divT},
        {"vGrad", &f->vGrad},
        {"nut", &f->nut},
        {"k", &f->kturb},
        {"epsilon", &f->epsturb},
        {"ddtU", &f->ddtU},
        {"phi_x", &f->phi[0]},
        {"phi_y", &f->phi[1]},
        {"phi_z", &f->phi[2]},
        {"rAU", &f->rAU},
        {"HbyA", &f->HbyA},
        {"p_rhs", &f->pb},
        {"p_diag", &f->mg[0].diag},
        {"p_ux", &f->mg[0].ux},
        {"p_uy", &f->mg[0].uy},
        {"p_uz", &f->mg[0].uz},
        {"mom_diag", &f->diag},
        {"mom_src", &f->src},
    };
    for (const auto& e : tab) if (s == e.nm) return e.v;
    return nullptr;
}
int orc_fv_field_size(void* h, const char* name) { vec* v = fv_field((Fv*)h, name); return v ? (int)v->size() : -1; }
int orc_fv_get(void* h, const char* name, double* out) { vec* v = fv_field((Fv*)h, name); if (!v) return 1; std::memcpy(out, v->data(), v->size() * sizeof(double)); return 0; }
int orc_fv_set(void* h, const char* name, const double* in) {
    Fv* f = (Fv*)h; vec* v = fv_field(f, name); if (!v) return 1;
    std::memcpy(v->data(), in, v->size() * sizeof(double));
    if (std::string(name) == "U") f->flux_of(f->U, f->phi);   // createPhi
    return 0;
}
double* orc_fv_ptr(void* h, const char* name) { vec* v = fv_field((Fv*)h, name); return v ? v->data() : nullptr; }
void orc_fv_step_begin(void* h) { ((Fv*)h)->step_begin(); }
void orc_fv_step_end(void* h) { ((Fv*)h)->step_end(); }
void orc_fv_get_stats(void* h, orc_fv_stats* out) { *out = ((Fv*)h)->st; }
// A x = rhs with the oracle's pressure solver (pFinal tolerances) and the matrix of the last step; returns the iteration count
int orc_fv_solve_p(void* h, const double* rhs, double* x) {
    Fv* f = (Fv*)h;
    std::memcpy(f->pb.data(), rhs, (size_t)f->Nc * sizeof(double));
    std::memcpy(f->p.data(), x, (size_t)f->Nc * sizeof(double));
    const int it = f->solve_pressure(true);
    std::memcpy(x, f->p.data(), (size_t)f->Nc * sizeof(double));
    return it;
}
// y = A x with the current pressure matrix (level 0)
void orc_fv_apply_p(void* h, const double* x, double* y) {
    Fv* f = (Fv*)h; vec xv(x, x + f->Nc), yv(f->Nc);
    Fv::apply(f->mg[0], xv, yv, f->threads);
    std::memcpy(y, yv.data(), yv.size() * sizeof(double));
}

}  // extern "C"
