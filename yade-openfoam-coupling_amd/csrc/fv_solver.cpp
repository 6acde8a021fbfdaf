// fy_solver: the time-loop bodies of icoFoamYade (icoFoamYade/icoFoamYade.C:65-149) and pimpleFoamYade
// (pimpleFoamYade/pimpleFoamYade.C:60-114 + UcEqn.H + pEqn.H) driving the HIP kernels of fv_kernels.hip, with the coupling
// engine (fy_ctx) sharing the same device-resident fields and stream.  Host code only sequences kernels and reads back the
// handful of scalars the control flow needs (residuals, Courant number, continuity errors).
//
// Multi-GPU: the block is cut into z-slabs, one per rank (fy::Comm, comm.hpp).  Cell arrays carry gz ghost planes per side; a
// halo exchange precedes every kernel that reads a z-neighbour of a field that changed; reductions are all-reduced on the device;
// multigrid levels whose 2x2x2 aggregates stay inside a slab are distributed, the coarse remainder is all-gathered and solved
// redundantly (bit-identically) on every rank.  With one rank every communication call is a no-op and gz = 0.
#include "fv_solver.hpp"

namespace fy {

int Solver::create(const fy_case_desc* c, const fy_transport* tr, int dev, Comm* cm) {
    if (c && (c->convection_scheme < FY_CONVECTION_LINEAR || c->convection_scheme > FY_CONVECTION_QUICK)) return fail(FY_ERR_INVALID, "fy_solver_create: unknown convection_scheme");
    if (c && c->convection_scheme == FY_CONVECTION_LIMITED_LINEAR && !(c->convection_limiter_k >= 0 && c->convection_limiter_k <= 1)) return fail(FY_ERR_INVALID, "fy_solver_create: limitedLinear's coefficient must lie in [0, 1]");
    const bool gradedc = c && (c->hx || c->hy || c->hz);
    if (!c || c->nx <= 0 || c->ny <= 0 || c->nz <= 0 || (!gradedc && !(c->dx > 0)) || !(c->dt > 0)) return fail(FY_ERR_INVALID, "fy_solver_create: bad case");
    if (gradedc) {
        // a graded (rectilinear) single block, one domain
        if (!(c->hx && c->hy && c->hz)) return fail(FY_ERR_INVALID, "fy_solver_create: a graded block needs hx, hy AND hz");
        if (cm && cm->size > 1) return fail(FY_ERR_UNSUPPORTED, "fy_solver_create: z-slabs need the uniform block (a graded block runs on one domain)");
        const double* hh[3] = {c->hx, c->hy, c->hz};
        const int nn[3] = {c->nx, c->ny, c->nz};
        for (int a = 0; a < 3; ++a) for (int q = 0; q < nn[a]; ++q) if (!(hh[a][q] > 0)) return fail(FY_ERR_INVALID, "fy_solver_create: graded block with a non-positive cell size");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FY_ERR_NO_DEVICE, "no HIP device visible: libfoamyade_hip has no CPU path");
    if (dev < 0 || dev >= ndev) return fail(FY_ERR_INVALID, "device ordinal out of range");
    cs = *c; device = dev; pimple = c->solver == FY_SOLVER_PIMPLE;
    cs.hx = cs.hy = cs.hz = nullptr;        // (the caller's arrays are copied below, not kept)
    comm = cm ? cm : &self_comm;
    FY_HIP(hipSetDevice(device));
    FY_HIP(hipStreamCreate(&stream));
    FY_HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
    FY_HIP(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&ev_halo, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&ev_assembled, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&ev_usum0, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&ev_usum1, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&ev_coarse, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&ev_factor, hipEventDisableTiming));
    overlap_halos = !options().no_halo_overlap && options().halo_overlap;
    overlap_sweeps = options().halo_overlap;
    fused_corrector = !options().no_fused_corrector;
    faces_from_cells = !options().faces_from_arrays;
    comm->set_aux_stream(comm_stream);
    // ---- slab extents: the case describes the GLOBAL block; rank r owns planes [r*nz, (r+1)*nz)
    const int S = comm->size;
    if (c->nz % S != 0) return fail(FY_ERR_INVALID, "nz (%d) must be divisible by the number of slabs (%d)", c->nz, S);
    const int nzl = c->nz / S;
    if (S > 1 && (nzl % 2 != 0 || nzl < 2)) return fail(FY_ERR_INVALID, "each slab needs an even number (>= 2) of z-planes, got %d", nzl);
    // ghost width: 1 plane for the FV stencils; the Gaussian stencil reaches sqrt(1.25)*4 dx = 4.47 dx => 5 planes
    const int gz = S == 1 ? 0 : (pimple ? 5 : 1);
    if (S > 1 && nzl < gz) return fail(FY_ERR_INVALID, "slab thinner (%d planes) than the particle halo (%d)", nzl, gz);
    plane = (size_t)c->nx * c->ny;
    Nc = (int)(plane * nzl);
    nstore = plane * (size_t)(nzl + 2 * gz);
    Nglob = (int64_t)plane * c->nz;
    g.nx = c->nx; g.ny = c->ny; g.nz = nzl; g.Nc = Nc; g.gz = gz; g.c0 = (int)(plane * gz); g.kglob0 = comm->rank * nzl; g.nzglob = c->nz;
    g.dx = c->dx; g.Af = c->dx * c->dx; g.V = c->dx * c->dx * c->dx;
    g.graded = 0; g.h[0] = g.h[1] = g.h[2] = nullptr;
    total_volume = g.V * (double)Nglob;
    if (gradedc) {
        const double* hh[3] = {c->hx, c->hy, c->hz};
        const int nn[3] = {c->nx, c->ny, c->nz};
        double len[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a) {
            h_host[a].assign(hh[a], hh[a] + nn[a]);
            for (double v : h_host[a]) len[a] += v;
            FY_TRY(d_h[a].alloc_exact((size_t)nn[a]));
            FY_HIP(hipMemcpyAsync(d_h[a].p, hh[a], (size_t)nn[a] * sizeof(double), hipMemcpyHostToDevice, stream));
            g.h[a] = d_h[a].p;
        }
        FY_HIP(hipStreamSynchronize(stream));
        g.graded = 1;
        g.dx = std::cbrt((h_host[0][0] * h_host[1][0]) * h_host[2][0]);       // (only what still assumes cubes reads it: nothing on this path)
        g.Af = g.dx * g.dx; g.V = g.dx * g.dx * g.dx;
        total_volume = (len[0] * len[1]) * len[2];
    }
    // strip order of the cell sweeps (fv_kernels.hip, fv_block): only where a plane is a whole number of 256-cell blocks and every XCD gets whole planes
    g.strip_B = g.strip_bp = g.strip_nzx = 0;
    {
        const int want = options().strip_blocks;
        if (want != 0 && plane % 256 == 0 && nzl % 8 == 0) {
            const int bp = (int)(plane / 256);
            const int target = want > 0 ? want : std::max(1, (int)std::lround(8.0 * c->nx / 256.0));      // ~8 rows of cells
            int best = 1;
            for (int d = 1; d <= bp; ++d) if (bp % d == 0 && std::abs(d - target) < std::abs(best - target)) best = d;
            g.strip_B = best; g.strip_bp = bp; g.strip_nzx = nzl / 8;
        }
    }
    g.upwind = c->convection_scheme;          // 0 linear, 1 upwind, 2 linearUpwind, 3 .. 8 limited
    g.lim_twoByk = 2.0 / std::max(c->convection_limiter_k, 1e-15);
    g.rdx = 1.0 / g.dx; g.rhdx = 1.0 / (0.5 * g.dx); g.rV = 1.0 / g.V;
    g.pimple = pimple ? 1 : 0; g.dt = c->dt; g.nu = c->nu;
    bool need_ref = true;
    for (int q = 0; q < 6; ++q) {
        if (c->u_bc[q] != FY_BC_U_FIXED_VALUE && c->u_bc[q] != FY_BC_U_ZERO_GRADIENT && c->u_bc[q] != FY_BC_U_SLIP) return fail(FY_ERR_INVALID, "fy_solver_create: unknown velocity boundary type %d on side %d", c->u_bc[q], q);
        if (c->p_bc[q] != FY_BC_P_ZERO_GRADIENT && c->p_bc[q] != FY_BC_P_FIXED_VALUE && c->p_bc[q] != FY_BC_P_FIXED_FLUX) return fail(FY_ERR_INVALID, "fy_solver_create: unknown pressure boundary type %d on side %d", c->p_bc[q], q);
        g.u_bc[q] = c->u_bc[q]; g.p_bc[q] = c->p_bc[q]; g.p_val[q] = c->p_value[q];
        for (int a = 0; a < 3; ++a) g.u_val[q][a] = c->u_value[q][a];
        if (c->p_bc[q] == FY_BC_P_FIXED_VALUE) need_ref = false;
    }
    for (int a = 0; a < 3; ++a) g.g[a] = c->g[a];
    g.need_ref = need_ref ? 1 : 0; g.p_ref_cell = c->p_ref_cell; g.p_ref_value = c->p_ref_value;
    g.u_relax = c->u_relax;
    if (c->turbulence_model != FY_TURBULENCE_LAMINAR && c->turbulence_model != FY_TURBULENCE_SMAGORINSKY && c->turbulence_model != FY_TURBULENCE_KEQN && c->turbulence_model != FY_TURBULENCE_KEPSILON) return fail(FY_ERR_UNSUPPORTED, "fy_solver_create: unknown turbulence_model %d (DPMTurbulenceModels.C:67-77: laminar Stokes, RAS kEpsilon, LES Smagorinsky, LES kEqn)", c->turbulence_model);
    if (c->turbulence_model != FY_TURBULENCE_LAMINAR) {
        if (!pimple) return fail(FY_ERR_INVALID, "fy_solver_create: icoFoamYade has no turbulence model (icoFoamYade.C:79-85 is laplacian(nu, U))");
        if (!(c->les_ck > 0 && c->les_ce > 0 && c->les_delta_coeff > 0) || c->nut_initial < 0) return fail(FY_ERR_INVALID, "fy_solver_create: Smagorinsky needs Ck, Ce, deltaCoeff > 0 and nut >= 0");
        for (int q = 0; q < 6; ++q) {
            const bool has_k = c->turbulence_model == FY_TURBULENCE_KEQN || c->turbulence_model == FY_TURBULENCE_KEPSILON;
            if (c->nut_bc[q] != FY_BC_NUT_ZERO_GRADIENT && c->nut_bc[q] != FY_BC_NUT_FIXED_VALUE && !((c->nut_bc[q] == FY_BC_WALL_FUNCTION || c->nut_bc[q] == FY_BC_NUT_CALCULATED) && has_k))
                return fail(FY_ERR_INVALID, "fy_solver_create: unknown nut boundary type (nutkWallFunction / calculated need a model with a k equation)");
            g.nut_bc[q] = c->nut_bc[q]; g.nut_val[q] = c->nut_value[q];
        }
        les_delta = c->les_delta_coeff * std::pow(g.V, 1.0 / 3.0);
        {   // nutWallFunction::yPlusLam [OF-6 nutWallFunctionFvPatchScalarField.C]
            if (!(c->wf_kappa > 0 && c->wf_E > 0)) return fail(FY_ERR_INVALID, "fy_solver_create: wall-function constants kappa, E must be positive");
            double ypl = 11.0;
            for (int it = 0; it < 10; ++it) ypl = std::log(std::max(c->wf_E * ypl, 1.0)) / c->wf_kappa;
            g.wf_yPlusLam = ypl; g.wf_kappa = c->wf_kappa; g.wf_E = c->wf_E; g.wf_cmu25 = std::pow(c->ras_cmu, 0.25);
            g.nut_wall_live = 0;
            g.turb_model = c->turbulence_model; g.turb_ck = c->les_ck; g.turb_cmu = c->ras_cmu; g.turb_delta = les_delta; g.turb_dcoeff = c->les_delta_coeff;
            for (int q = 0; q < 6; ++q) {
                g.k_bc[q] = c->k_bc[q]; g.k_val[q] = c->k_value[q];
                g.eps_bc[q] = c->eps_bc[q] == FY_BC_NUT_FIXED_VALUE ? 1 : 0; g.eps_val[q] = c->eps_value[q];
            }
        }
        const bool keqn = c->turbulence_model == FY_TURBULENCE_KEQN, keps = c->turbulence_model == FY_TURBULENCE_KEPSILON;
        if (keqn || keps) {
            if (!(c->k_initial >= 0) || (keps && !(c->k_initial > 0)) || !(c->k_tol >= 0) || c->k_max_iter < 0 || c->k_relax > 1) return fail(FY_ERR_INVALID, "fy_solver_create: the k equation needs k >= 0 (> 0 for kEpsilon), a solver tolerance and a relaxation factor in (0, 1]");
            if (c->k_convection_scheme != FY_CONVECTION_LINEAR && c->k_convection_scheme != FY_CONVECTION_UPWIND) return fail(FY_ERR_UNSUPPORTED, "fy_solver_create: div(alphaPhic,k) must be Gauss linear or Gauss upwind");
            eq_k.mode = keqn ? 0 : 2;
            eq_k.ck = c->les_ck; eq_k.ce = c->les_ce; eq_k.delta = les_delta; eq_k.xmin = 1e-15;        // kMin_ = small [OF-6 LESModel.C, RASModel.C]
            eq_k.c1 = c->ras_c1; eq_k.c2 = c->ras_c2; eq_k.c3 = c->ras_c3; eq_k.sigma = keqn ? 1.0 : c->ras_sigmak;
            eq_k.relax = c->k_relax; eq_k.upwind = c->k_convection_scheme == FY_CONVECTION_UPWIND ? 1 : 0;
            for (int q = 0; q < 6; ++q) {
                if (c->k_bc[q] != FY_BC_NUT_ZERO_GRADIENT && c->k_bc[q] != FY_BC_NUT_FIXED_VALUE) return fail(FY_ERR_INVALID, "fy_solver_create: unknown k boundary type");
                eq_k.bc[q] = c->k_bc[q]; eq_k.val[q] = c->k_value[q];
            }
        }
        if (keps) {
            if (!(c->eps_initial > 0) || !(c->eps_tol >= 0) || c->eps_max_iter < 0 || c->eps_relax > 1 || !(c->ras_cmu > 0 && c->ras_sigmak > 0 && c->ras_sigmaeps > 0))
                return fail(FY_ERR_INVALID, "fy_solver_create: kEpsilon needs epsilon > 0, Cmu / sigmak / sigmaEps > 0, a solver tolerance and a relaxation factor in (0, 1]");
            if (c->eps_convection_scheme != FY_CONVECTION_LINEAR && c->eps_convection_scheme != FY_CONVECTION_UPWIND) return fail(FY_ERR_UNSUPPORTED, "fy_solver_create: div(alphaPhic,epsilon) must be Gauss linear or Gauss upwind");
            eq_eps = eq_k;
            eq_eps.mode = 1; eq_eps.sigma = c->ras_sigmaeps; eq_eps.xmin = 1e-15;                        // epsilonMin_ = small [OF-6 RASModel.C]
            eq_eps.relax = c->eps_relax; eq_eps.upwind = c->eps_convection_scheme == FY_CONVECTION_UPWIND ? 1 : 0;
            for (int q = 0; q < 6; ++q) {
                if (c->eps_bc[q] != FY_BC_NUT_ZERO_GRADIENT && c->eps_bc[q] != FY_BC_NUT_FIXED_VALUE && c->eps_bc[q] != FY_BC_WALL_FUNCTION) return fail(FY_ERR_INVALID, "fy_solver_create: unknown epsilon boundary type");
                // an epsilonWallFunction patch behaves like a zero-gradient one wherever its face value would be asked for: the wall cells' rows are imposed
                eq_eps.bc[q] = c->eps_bc[q] == FY_BC_WALL_FUNCTION ? 0 : c->eps_bc[q]; eq_eps.val[q] = c->eps_value[q];
                eq_eps.wall[q] = eq_k.wall[q] = c->eps_bc[q] == FY_BC_WALL_FUNCTION ? 1 : 0;
            }
            eq_eps.cmu75 = eq_k.cmu75 = std::pow(c->ras_cmu, 0.75); eq_eps.cmu25 = eq_k.cmu25 = std::pow(c->ras_cmu, 0.25); eq_eps.kappa = eq_k.kappa = c->wf_kappa;
        }
    }
    if (c->adjust_time_step && !(c->max_co > 0 && c->max_delta_t > 0)) return fail(FY_ERR_INVALID, "adjustTimeStep needs maxCo > 0 and maxDeltaT > 0");
    if (c->u_relax > 1 || c->u_relax_final > 1 || c->p_relax > 1 || c->p_relax_final > 1) return fail(FY_ERR_INVALID, "relaxation factors lie in (0, 1]");
    if (need_ref) {
        // adjustPhi (icoFoamYade.C:108, pEqn.H:13-16) acts when no patch fixes the pressure.  It is the identity when every patch fixes U
        // and the prescribed normal velocities balance (closed boxes, cavities): nothing is launched then.  With fixed-value patches
        // that do NOT balance and no patch to adjust, OpenFOAM stops at the first corrector; say so here.
        double net = 0.0, mag = 0.0;
        bool adjustable = false;
        const double area[3] = {(double)c->ny * c->nz, (double)c->nx * c->nz, (double)c->nx * c->ny};
        for (int q = 0; q < 6; ++q) {
            if (c->u_bc[q] == FY_BC_U_SLIP) continue;                       // (carries no flux, and none to adjust)
            if (c->u_bc[q] != FY_BC_U_FIXED_VALUE) { adjustable = true; continue; }
            const double un = c->u_value[q][q / 2] * ((q & 1) ? 1.0 : -1.0) * area[q / 2];
            net += un; mag += std::fabs(un);
        }
        if (!adjustable && std::fabs(net) > 1e-8 * (mag + 1e-300))
            return fail(FY_ERR_UNSUPPORTED, "no patch fixes the pressure and the fixed-value velocity patches do not balance (net flux %g of %g): OpenFOAM's adjustPhi "
                                            "ends such a run with 'Continuity error cannot be removed by adjusting the outflow'", net, mag);
        adjust_phi = adjustable || mag > 0.0;
    }

    const size_t n = nstore;
    DevBuf<double>* v3[] = {&U, &Uold, &uSource, &uParticle, &gradP, &divT, &ddtU, &src, &HbyA, &bmom, &divG, &xscr};
    for (auto* b : v3) { FY_TRY(b->alloc_exact(3 * n)); FY_TRY(zero(*b)); }
    DevBuf<double>* v1[] = {&p, &alpha, &uSourceDrag, &mdiag, &rAU, &prhs, &pr, &pw, &pp, &ps, &pzj};
    for (auto* b : v1) { FY_TRY(b->alloc_exact(n)); FY_TRY(zero(*b)); }
    for (auto& b : man) { FY_TRY(b.alloc_exact(n)); FY_TRY(zero(b)); }
    for (int q = 0; q < 6; ++q) if (c->u_bc[q] == FY_BC_U_SLIP && !mbd.p) { FY_TRY(mbd.alloc_exact(3 * n)); FY_TRY(zero(mbd)); }
    FY_TRY(vGrad.alloc_exact(9 * n)); FY_TRY(zero(vGrad));
    if (c->convection_scheme >= FY_CONVECTION_LIMITED_LINEAR) { FY_TRY(gradL.alloc_exact(3 * n)); FY_TRY(zero(gradL)); }
    if (c->turbulence_model != FY_TURBULENCE_LAMINAR) { FY_TRY(nut.alloc_exact(n)); FY_TRY(launch_fill_f64(stream, nut.p, n, c->nut_initial)); g.nut = nut.p; }
    if (c->turbulence_model == FY_TURBULENCE_KEQN || c->turbulence_model == FY_TURBULENCE_KEPSILON) { FY_TRY(kturb.alloc_exact(n)); FY_TRY(launch_fill_f64(stream, kturb.p, n, c->k_initial)); g.kturb = kturb.p; }
    if (c->turbulence_model == FY_TURBULENCE_KEPSILON) { FY_TRY(epsturb.alloc_exact(n)); FY_TRY(launch_fill_f64(stream, epsturb.p, n, c->eps_initial)); g.epsturb = epsturb.p; }
    FY_TRY(Gt.alloc_exact(9 * n)); FY_TRY(zero(Gt));
    for (int d = 0; d < 3; ++d) {
        DevBuf<double>* fs[] = {&phi[d], &phiOld[d], &psn[d], &alphaf[d], &phiHbyA[d], &phiForces[d], &rAUf[d], &pflux[d], &ddtc[d], &dcorr[d]};
        for (auto* b : fs) { FY_TRY(b->alloc_exact(fv_fsize(g, d))); FY_TRY(zero(*b)); }
        FY_TRY(launch_fill_f64(stream, alphaf[d].p, alphaf[d].n, 1.0));
    }
    FY_TRY(launch_fill_f64(stream, alpha.p, n, 1.0));
    FY_TRY(launch_fill_f64(stream, rAU.p, n, 1.0));            // ghost planes must hold finite values before the first exchange
    FY_TRY(partials.alloc_exact(8 * (size_t)red_blocks(Nc))); FY_TRY(red_out.alloc_exact(8)); FY_TRY(sc.alloc_exact(8)); FY_TRY(xbar3.alloc_exact(3));
    FY_TRY(zero(partials)); FY_TRY(zero(sc)); FY_TRY(zero(red_out));
    if (adjust_phi) { FY_TRY(adj_sums.alloc_exact(4)); FY_TRY(adj_err.alloc_exact(1)); FY_HIP(hipMemsetAsync(adj_err.p, 0, sizeof(int), stream)); }
    if (hipHostMalloc((void**)&red_host, (kDeferBase + kDeferMax) * sizeof(double), hipHostMallocMapped) == hipSuccess) {
        if (hipHostGetDevicePointer((void**)&red_host_dev, red_host, 0) != hipSuccess) { (void)hipHostFree(red_host); red_host = nullptr; }
    } else {
        red_host = nullptr;              // fall back to the copy path
    }
    if (red_host && hipHostMalloc((void**)&red_flag, 8 * sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
        for (int q = 0; q < 8; ++q) red_flag[q] = 0;
        if (hipHostGetDevicePointer((void**)&red_flag_dev, red_flag, 0) != hipSuccess) { (void)hipHostFree(red_flag); red_flag = nullptr; }
    } else {
        red_flag = nullptr;
    }
    FY_TRY(ops_courant.alloc_exact(2));
    FY_TRY(ops_diag.alloc_exact(4));
    { const int h4[4] = {0, 0, 1, 0}; FY_HIP(hipMemcpyAsync(ops_diag.p, h4, sizeof(h4), hipMemcpyHostToDevice, stream)); }
    { const int h[2] = {1, 0}; FY_HIP(hipMemcpyAsync(ops_courant.p, h, sizeof(h), hipMemcpyHostToDevice, stream)); FY_HIP(hipStreamSynchronize(stream)); }

    // ---- multigrid hierarchy: 2x2x2 aggregation down to <= kMgCoarsest cells.  With several slabs the levels whose aggregates
    // stay inside a slab keep the slab layout (+1 ghost plane); from the first level with <= kMgReplicateBelow global cells (or
    // whose parent has an odd plane count) on, the hierarchy is replicated on every rank.
    {
        int ax = g.nx, ay = g.ny, az_loc = g.nz, az_glob = c->nz;
        bool dist = S > 1;
        size_t lvl = 0;
        mg_rep = (size_t)-1;
        for (;;) {
            std::unique_ptr<MgLev> L(new MgLev());
            L->distributed = dist;
            // level 0 shares the layout of the solver's cell vectors (p, r, ...); the distributed levels under it carry the ghost depth of the
            // communication-avoiding V-cycle where their slabs are thick enough for it
            L->gz = dist ? (lvl == 0 ? g.gz : (az_loc >= kMgDeepGhost ? kMgDeepGhost : 1)) : 0;
            L->plane = (size_t)ax * ay;
            const int nzv = dist ? az_loc : az_glob;
            L->A.nx = ax; L->A.ny = ay; L->A.nz = nzv; L->A.N = (int)(L->plane * nzv);
            L->A.c0 = (int)(L->plane * L->gz); L->A.ntot = (int)(L->plane * (nzv + 2 * L->gz));
            const size_t m = (size_t)L->A.ntot;
            DevBuf<double>* bs[] = {&L->diag, &L->ux, &L->uy, &L->uz, &L->x0, &L->x1, &L->b};
            for (auto* b : bs) { FY_TRY(b->alloc_exact(m)); FY_TRY(zero(*b)); }
            FY_TRY(launch_fill_f64(stream, L->diag.p, m, 1.0));          // never divide by an unset ghost diagonal
            L->A.diag = L->diag.p; L->A.ux = L->ux.p; L->A.uy = L->uy.p; L->A.uz = L->uz.p;
            L->xcur = L->x0.p; L->xalt = L->x1.p; L->bptr = L->b.p;
            const int64_t Ng = (int64_t)L->plane * az_glob;
            if (!dist && mg_rep == (size_t)-1 && S > 1) mg_rep = lvl;
            mg.push_back(std::move(L));
            if (cs.p_solver != FY_PSOLVER_PCG_MG) break;
            // the coarsest level is "solved" by 40 damped-Jacobi sweeps, which settles modes up to a few cells long: an elongated coarse grid
            // (3 x 3 x 20 under a 160 x 160 x 1280 column) would keep its longest modes and the V-cycle would lose its grip on tall
            // domains (PCG iterations per step 2.6 / 4.4 / 4.6 / 6.4 for 1 / 2 / 4 / 8 stacked C3 boxes) -- so coarsening also goes on
            // while any edge is longer than kMgCoarsestEdge cells
            if ((Ng <= kMgCoarsest && std::max(std::max(ax, ay), az_glob) <= kMgCoarsestEdge) || (ax <= 2 && ay <= 2 && az_glob <= 2)) break;
            // next level
            const int nax = (ax + 1) / 2, nay = (ay + 1) / 2, naz_glob = (az_glob + 1) / 2;
            if (dist) {
                const int64_t nNg = (int64_t)nax * nay * naz_glob;
                if (az_loc % 2 != 0) return fail(FY_ERR_UNSUPPORTED, "slab plane count %d cannot be aggregated", az_loc);
                az_loc /= 2;                                         // the slice of the next level this rank's cells aggregate to
                if (az_loc % 2 != 0 || az_loc < 2 || nNg <= kMgReplicateBelow) dist = false;   // next level: replicated
            }
            ax = nax; ay = nay; az_glob = naz_glob;
            ++lvl;
        }
        if (mg_rep == (size_t)-1) mg_rep = mg.size();
        if (S > 1 && cs.p_solver == FY_PSOLVER_PCG_MG && mg_rep >= mg.size()) return fail(FY_ERR_UNSUPPORTED, "multigrid hierarchy never became replicable");
        if (mg.back()->A.N > 1024 && cs.p_solver == FY_PSOLVER_PCG_MG) return fail(FY_ERR_UNSUPPORTED, "coarsest multigrid level too large");
        if (mg_rep < mg.size()) FY_TRY(rep_stage.alloc_exact(4 * ((size_t)mg[mg_rep]->A.N / S + 8)));
        mg_deep = S > 1 && cs.p_solver == FY_PSOLVER_PCG_MG && mgw.n == 2 && !options().no_deep_vcycle && mg_rep < mg.size();
        for (auto& L : mg) if (L->distributed && (L->gz < kMgDeepGhost || L->A.nz < kMgDeepGhost)) mg_deep = false;
    }
    for (auto& t : tim) FY_TRY(t.init());
    // who still streams the face interpolates rAUf / alphacf from their arrays: the separate corrector sweeps (switched on, or taken by adjustPhi / the
    // non-orthogonal correctors), the interface coefficient of a slab without deep ghost planes, the turbulence transport equations, the stand-alone
    // continuity-error sweep; otherwise the arrays are never filled
    face_arrays = !fused_corrector || !faces_from_cells || adjust_phi || cs.n_non_orth_correctors > 0 || (S > 1 && !mg_deep) ||
                  c->turbulence_model != FY_TURBULENCE_LAMINAR || !fuse_diag || !red_host || (pimple && !cs.momentum_predictor);

    // the coupling object shares the solver's device fields and stream (icoFoamYade.C:54, pimpleFoamYade.C:54); its tree spans
    // the GLOBAL block (the improvement chain depends on the whole tree), its cell arrays are this slab's storage
    {
        const size_t ng = (size_t)Nglob;
        std::vector<double> C(3 * ng), V(ng, g.V);
        std::vector<double> fc[3];           // graded block: face planes per axis
        fy_mesh_desc md{};
        if (!g.graded) {
            for (int k = 0; k < c->nz; ++k) for (int j = 0; j < g.ny; ++j) for (int i = 0; i < g.nx; ++i) {
                const size_t cc = (size_t)i + (size_t)g.nx * (j + (size_t)g.ny * k);
                C[3 * cc] = c->origin[0] + (i + 0.5) * c->dx; C[3 * cc + 1] = c->origin[1] + (j + 0.5) * c->dx; C[3 * cc + 2] = c->origin[2] + (k + 0.5) * c->dx;
            }
            md.bbox_max[0] = c->origin[0] + g.nx * c->dx; md.bbox_max[1] = c->origin[1] + g.ny * c->dx; md.bbox_max[2] = c->origin[2] + c->nz * c->dx;
        } else {
            for (int a = 0; a < 3; ++a) {
                fc[a].resize(h_host[a].size() + 1);
                fc[a][0] = c->origin[a];
                for (size_t q = 0; q < h_host[a].size(); ++q) fc[a][q + 1] = fc[a][q] + h_host[a][q];
                md.bbox_max[a] = fc[a].back();
            }
            for (int k = 0; k < c->nz; ++k) for (int j = 0; j < g.ny; ++j) for (int i = 0; i < g.nx; ++i) {
                const size_t cc = (size_t)i + (size_t)g.nx * (j + (size_t)g.ny * k);
                C[3 * cc] = 0.5 * (fc[0][i] + fc[0][i + 1]); C[3 * cc + 1] = 0.5 * (fc[1][j] + fc[1][j + 1]); C[3 * cc + 2] = 0.5 * (fc[2][k] + fc[2][k + 1]);
                V[cc] = (h_host[0][i] * h_host[1][j]) * h_host[2][k];
            }
            md.xf = fc[0].data(); md.yf = fc[1].data(); md.zf = fc[2].data();
        }
        md.n_cells = (int32_t)Nglob; md.centres = C.data(); md.volumes = V.data();
        md.nx = g.nx; md.ny = g.ny; md.nz = c->nz; md.dx = g.dx;
        for (int a = 0; a < 3; ++a) { md.origin[a] = c->origin[a]; md.bbox_min[a] = c->origin[a]; }
        fy_field_ptrs fp{};
        fp.location = FY_MEM_DEVICE;
        fp.U = U.p; fp.gradP = gradP.p; fp.vGrad = vGrad.p; fp.divT = divT.p; fp.ddtU = ddtU.p;
        for (int a = 0; a < 3; ++a) fp.g[a] = c->g[a];
        fp.uSourceDrag = uSourceDrag.p; fp.alpha = alpha.p; fp.uSource = uSource.p; fp.uParticle = uParticle.p;
        cpl = new (std::nothrow) fy_ctx();
        if (!cpl) return fail(FY_ERR_INVALID, "out of host memory");
        cpl->c.ext_stream = stream;
        if (S > 1) {
            cpl->c.slab.active = true; cpl->c.slab.comm = comm; cpl->c.slab.gz = gz; cpl->c.slab.nz = nzl; cpl->c.slab.plane = plane;
            cpl->c.slab.n_store = nstore; cpl->c.slab.base = ((int64_t)g.kglob0 - gz) * (int64_t)plane;
            cpl->c.slab.kglob0 = g.kglob0; cpl->c.slab.nzglob = g.nzglob;
            if (overlap_halos && overlap_sweeps) cpl->c.slab.aux = comm_stream;       // the particle phase's exchanges beside independent work (coupling.cpp)
        }
        FY_TRY(cpl->c.create(&md, &fp, pimple ? 1 : 0, tr, device));      // gaussianInterp: false for ico, true for pimple (icoFoamYade.C:53, pimpleFoamYade.C:53)
        cpl->c.rhoP = c->rho_particle; cpl->c.rhoF = c->rho_fluid; cpl->c.nu = c->nu;   // setScalarProperties (icoFoamYade.C:55)
    }
    FY_TRY(halo_U());
    FY_TRY(FVK(launch_flux_of, stream, g, U.p, F3(phi)));                     // createPhi
    FY_HIP(hipStreamSynchronize(stream));
    return FY_OK;
}

// fold the block partials, all-reduce over the slabs, read back
int Solver::reduce_read(int nslots, bool courant, double* h) {
    if (cpl) FY_TRY(cpl->c.poll_results());
    if (comm->size == 1 && red_host) {
        // single domain: the fold writes straight into mapped pinned host memory -- no device-to-host blit per read-back
        if (red_flag && nslots <= 8) {
            // spin on the flags the fold stores behind its results; everything enqueued before it has completed by then (in-order stream)
            const unsigned long long seq = ++red_seq;
            FY_TRY(launch_reduce_finalize(stream, partials.p, Nc, nslots, courant ? ops_courant.p : nullptr, red_host_dev, red_flag_dev, seq));
            for (int q = 0; q < nslots; ++q) {
                unsigned long spins = 0;
                while (__atomic_load_n(&red_flag[q], __ATOMIC_ACQUIRE) != seq) {
                    if ((++spins & 0x3ffu) == 0 && cpl) FY_TRY(cpl->c.poll_results());      // (answers that have landed meanwhile go out: Coupling::poll_results)
                    if ((spins & 0xfffu) == 0) {                        // every 4096 polls: is the stream still alive?
                        const hipError_t e = hipStreamQuery(stream);
                        if (e == hipSuccess) { if (__atomic_load_n(&red_flag[q], __ATOMIC_ACQUIRE) == seq) break; return fail(FY_ERR_HIP, "reduction flag never arrived"); }
                        if (e != hipErrorNotReady) return fail(FY_ERR_HIP, "stream failed while waiting for a reduction: %s", hipGetErrorString(e));
                    }
                }
                h[q] = red_host[q];
            }
            return FY_OK;
        }
        FY_TRY(launch_reduce_finalize(stream, partials.p, Nc, nslots, courant ? ops_courant.p : nullptr, red_host_dev));
        FY_HIP(hipStreamSynchronize(stream));
        for (int q = 0; q < nslots; ++q) h[q] = red_host[q];
        return FY_OK;
    }
    FY_TRY(launch_reduce_finalize(stream, partials.p, Nc, nslots, courant ? ops_courant.p : nullptr, red_out.p));
    FY_TRY(comm->allreduce_ops(stream, red_out.p, nslots, courant ? 1u : 0u));         // Courant pair: {max, sum} in one collective
    double* land = (red_host && nslots <= 8) ? red_host : h;          // pinned landing zone: a pageable destination makes the copy a staged, blocking one
    FY_HIP(hipMemcpyAsync(land, red_out.p, nslots * sizeof(double), hipMemcpyDeviceToHost, stream));
    FY_HIP(hipStreamSynchronize(stream));
    if (land != h) for (int q = 0; q < nslots; ++q) h[q] = land[q];
    return FY_OK;
}
bool Solver::reduce_deferred(int nslots, bool courant, int* slot, int* rc, const int* ops) {      // ops: ops_diag.p (sum, sum, max, sum) or none
    *rc = FY_OK;
    if (!red_host || n_deferred + nslots > kDeferMax) return false;
    *slot = kDeferBase + n_deferred;
    n_deferred += nslots;
    if (comm->size == 1) {
        *rc = launch_reduce_finalize(stream, partials.p, Nc, nslots, ops ? ops : (courant ? ops_courant.p : nullptr), red_host_dev + *slot);
        return true;
    }
    *rc = launch_reduce_finalize(stream, partials.p, Nc, nslots, ops ? ops : (courant ? ops_courant.p : nullptr), red_out.p);
    if (*rc == FY_OK) {
        // one collective per group: the four diagnostics of k_U_correct<true> are {sum, sum, max, sum}, the Courant pair {max, sum}
        *rc = comm->allreduce_ops(stream, red_out.p, nslots, ops ? 4u : (courant ? 1u : 0u));
    }
    if (*rc == FY_OK && hipMemcpyAsync(red_host + *slot, red_out.p, nslots * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess)
        *rc = fail(FY_ERR_HIP, "deferred read-back failed");
    return true;
}
int Solver::reduce_to_device(double* dst) {          // one slot, stays on the device (PCG scalars)
    FY_TRY(launch_reduce_finalize(stream, partials.p, Nc, 1, nullptr, dst));
    return comm->allreduce(stream, dst, 1, false);
}

// Jacobi sweeps on the 7-point matrix in M7() for a 3-component field X (in place; xscr is the other buffer), lduMatrix-style L1
// residual control per component
int Solver::solve_vec3(DevBuf<double>& X, const double* rhs, double tol, double rel_tol, int max_iter, int* iters, bool momentum) {
    Comm::Tag tag(comm, momentum ? "momentum_solve" : "turbulence_solve");
    double h[6];
    // sum(X) per component for xbar = average(X): folded (and all-reduced) on the device, divided where it is used
    if (momentum && usum_pending) {                          // (formed beside the particle phase: step())
        usum_pending = false;
        FY_HIP(hipStreamWaitEvent(stream, ev_usum1, 0));
    } else {
        FY_TRY(launch_sum3(stream, X.p + 3 * (size_t)g.c0, Nc, partials.p));
        FY_TRY(launch_reduce_finalize(stream, partials.p, Nc, 3, nullptr, xbar3.p));
        FY_TRY(comm->allreduce(stream, xbar3.p, 3, false));
    }
    double* xc = X.p; double* xn = xscr.p;
    double norm[3] = {1, 1, 1}, res0[3] = {0, 0, 0}, res[3];
    int it = 0;
    for (;;) {
        // (the predictor's first pass reads the ghost planes the step's opening exchange left in U)
        const bool need_x = !(it == 0 && &X == &U && U_ghosts_fresh);
        kc[KC_MOM_PASS].begin(stream);
        // from the second pass on the momentum predictor's pass also leaves HbyA of its iterate: the pass that finds it converged has then done the
        // first corrector's H-operator sweep (corrector(): hbya_ready)
        const bool with_h = momentum && fused_corrector && cs.n_correctors > 0 && it >= 1;
        FY_TRY(overlapped(XW_MOMENTUM, need_x, [&](hipStream_t on) { return halo(xc, 3, plane, g.nz, g.gz, 1, on); }, [&](const FvGeo& gw) {
            return FVK(launch_mom_pass, stream, gw, M7(momentum), rhs, xc, xn, xbar3.p, (double)Nglob, partials.p, with_h ? src.p : nullptr, with_h ? rAU.p : nullptr,
                       with_h ? HbyA.p : nullptr);
        }));
        if (momentum) hbya_ready = with_h;
        kc[KC_MOM_PASS].end(stream);
        FY_TRY(reduce_read(6, false, h));
        if (it == 0) for (int q = 0; q < 3; ++q) { norm[q] = h[3 + q] + 1e-20; res0[q] = h[q] / norm[q]; }
        bool conv = true;
        for (int q = 0; q < 3; ++q) {
            res[q] = h[q] / norm[q];
            if (!(res[q] < tol || (rel_tol > 0 && res[q] < rel_tol * res0[q]))) conv = false;
        }
        if (conv || it >= max_iter) break;
        std::swap(xc, xn);
        ++it;
    }
    // the converged iterate may sit in the scratch buffer: the two arrays (same size, both whole-storage) trade places instead of 3 Nc doubles being
    // copied (36 us at 160^3).  Ghost planes of the new X are whatever the scratch held: every reader across a slab face exchanges first.
    if (xc != X.p) {
        std::swap(X.p, xscr.p);
        if (&X == &U && cpl) cpl->c.dU = U.p;             // (the coupling gathers U through its own pointer)
    }
    // the pass that found the iterate converged had just exchanged that iterate's ghost planes (or they were fresh): they still are
    if (&X == &U) U_ghosts_fresh = true;
    *iters = it;
    return FY_OK;
}

// ---- one PISO / PIMPLE corrector (icoFoamYade.C:97-140, pEqn.H) ----------------------------------------------------
int Solver::corrector(bool final_inner) {
    Comm::Tag tag(comm, "corrector");
    if (hbya_ready) hbya_ready = false;                      // (the predictor's last pass wrote HbyA of the U it accepted: solve_vec3)
    else FY_TRY(overlapped(XW_CORRECTOR, !U_ghosts_fresh, [&](hipStream_t on) { return halo_U(on); },
                           [&](const FvGeo& gw) { return FVK(launch_HbyA, stream, gw, M7(), src.p, U.p, rAU.p, HbyA.p); }));
    // rAU (hence rAUf and the pressure matrix rAUf*alphaf) belongs to the momentum matrix: it only changes when that is assembled,
    // not between the PISO correctors of one assembly
    if (!pimple && rAU_new) { FY_TRY(halo_cells(rAU, 1, 1)); if (face_arrays) FY_TRY(FVK(launch_interp_rAU, stream, g, rAU.p, F3(rAUf))); }
    MgLev& L = *mg[0];
    // the two fused sweeps (fv_kernels.hip "fused corrector sweeps") stand for phiHbyA + assembly + PCG's first residual and for the flux + velocity
    // corrections; adjustPhi, which needs global sums of phiHbyA between the first two, keeps the separate sweeps
    const bool fused_front = fused_corrector && !adjust_phi;
    const bool fused_back = fused_corrector;
    // the fused sweeps re-form rAUf / alphacf from rAU / alpha; both cell fields carry fresh ghost planes here (rAU: exchanged after the assembly below /
    // above; alpha: by the coupling)
    const bool ffc = faces_from_cells;
    if (fused_front) {
        // the ddtCorr term is the same in every corrector of one momentum assembly: stored by the first, read back by the others
        clk_pres.begin(stream);
        FY_TRY(prepare_p_init(false));
        // HbyA's and p's ghost planes travel in one exchange, beside the sweep's interior planes
        FY_TRY(overlapped(XW_CORRECTOR, true, [&](hipStream_t on) {
            comm->group_begin();
            FY_TRY(halo_cells(HbyA, 3, 1, on));
            FY_TRY(halo_p(on));
            return comm->group_end(on);
        }, [&](const FvGeo& gw) {
            return FVK(launch_corr_front, stream, gw, HbyA.p, U.p, C3(dcorr), C3(rAUf), C3(alphaf), C3(phiForces), F3(phiHbyA), F3(psn), rAU.p, alpha.p, /* alphaOld */ alpha.p,
                       L.A, prhs.p, rAU_new, p.p, p_sum_valid ? nullptr : sc.p + 6, p_sum, 1.0 / (double)Nglob, pr.p, partials.p, ffc);
        }));
    } else {
        FY_TRY(halo_cells(HbyA, 3, 1));
        FY_TRY(FVK(launch_phiHbyA, stream, g, HbyA.p, U.p, Uold.p, C3(phiOld), C3(rAUf), C3(alphaf), C3(phiForces), F3(phiHbyA), F3(psn), F3(ddtc), rAU_new ? 1 : 2));
        if (adjust_phi) {                                       // icoFoamYade.C:108, pEqn.H:13-16
            FY_TRY(FVK(launch_adjust_phi_sums, stream, g, C3(phiHbyA), C3(phiForces), partials.p));
            FY_TRY(launch_reduce_finalize(stream, partials.p, Nc, 4, nullptr, adj_sums.p));
            FY_TRY(comm->allreduce(stream, adj_sums.p, 4, false));
            FY_TRY(FVK(launch_adjust_phi_apply, stream, g, adj_sums.p, F3(phiHbyA), C3(phiForces), C3(rAUf), U.p, F3(psn), adj_err.p));
        }
        clk_pres.begin(stream);
    }
    for (int no = 0; no <= cs.n_non_orth_correctors; ++no) {
        const bool front_done = fused_front && no == 0;
        if (!front_done) FY_TRY(FVK(launch_assemble_pressure, stream, g, C3(phiHbyA), C3(rAUf), C3(alphaf), C3(psn), alpha.p, /* alphaOld */ alpha.p, L.A, prhs.p, rAU_new));
        if (rAU_new && L.distributed && comm->has_down() && !mg_deep) FY_TRY(FVK(launch_p_ghost_uz, stream, g, C3(rAUf), C3(alphaf), L.A));
        if (rAU_new) {                                        // same matrix as in the previous corrector otherwise: only the right-hand side moved
            if (comm->size == 1 && cs.p_solver == FY_PSOLVER_PCG_MG && mg.size() > 1) {
                FY_HIP(hipEventRecord(ev_assembled, stream));
                FY_HIP(hipStreamWaitEvent(comm_stream, ev_assembled, 0));
                std::swap(stream, comm_stream);               // (build_coarse_operators launches on `stream`)
                coarse_on_side = true; coarse_marked = false;
                const int rc = build_coarse_operators();      // (records ev_coarse behind the last operator, before the coarsest level's factorisation)
                coarse_on_side = false;
                std::swap(stream, comm_stream);
                FY_TRY(rc);
                if (coarse_marked) { FY_HIP(hipEventRecord(ev_factor, comm_stream)); factor_pending = true; }
                else FY_HIP(hipEventRecord(ev_coarse, comm_stream));
                coarse_pending = true;
            } else {
                FY_TRY(build_coarse_operators());
            }
        }
        rAU_new = false;
        FY_TRY(solve_pressure(final_inner && no == cs.n_non_orth_correctors, front_done));
        FY_TRY(wait_factor());                                // (a solve that never left level 0)
        if (no == cs.n_non_orth_correctors && !fused_back) {
            FY_TRY(halo_p());
            FY_TRY(FVK(launch_flux_correct, stream, g, p.p, C3(phiHbyA), C3(rAUf), C3(alphaf), C3(psn), C3(phiForces), F3(pflux), F3(phi)));
            phi_fresh = true;
            // p.relax() (pEqn.H:41): after the flux, which keeps the unrelaxed solution; the velocity correction below works with
            // pEqn.flux() (pflux), not with grad(p), so only the carried pressure field is relaxed
            if (pimple && p_relax_now > 0 && p_relax_now < 1) { FY_TRY(launch_relax_field(stream, p.p, pPrev.p, p_relax_now, nstore)); p_sum_valid = false; p_ghosts_fresh = false; }
        }
    }
    clk_pres.end(stream);
    double h[2];
    int slot = 0, rc = FY_OK;
    const bool diag_fused = fuse_diag && red_host && n_deferred + 4 <= kDeferMax;
    if (fused_back) {
        // flux correction + velocity correction [+ the continuity errors and the NEXT step's Courant sums] in one sweep; p.relax() (pEqn.H:41) after it:
        // the sweep works with the unrelaxed solution, as pEqn.flux() and the reconstruction do (ico reads p itself, but has no relaxation)
        FY_TRY(overlapped(XW_CORRECTOR, !p_ghosts_fresh, [&](hipStream_t on) { return halo_p(on); }, [&](const FvGeo& gw) {
            return FVK(launch_corr_back, stream, gw, p.p, C3(phiHbyA), C3(rAUf), C3(alphaf), C3(psn), C3(phiForces), F3(phi), HbyA.p, rAU.p, U.p, alpha.p,
                       /* alphaOld */ alpha.p, diag_fused ? partials.p : nullptr, ffc);
        }));
        phi_fresh = true;
        U_ghosts_fresh = false;
        if (pimple && p_relax_now > 0 && p_relax_now < 1) { FY_TRY(launch_relax_field(stream, p.p, pPrev.p, p_relax_now, nstore)); p_sum_valid = false; p_ghosts_fresh = false; }
        if (diag_fused) {
            if (!reduce_deferred(4, false, &slot, &rc, ops_diag.p)) return fail(FY_ERR_INVALID, "no room for the deferred diagnostics");
            FY_TRY(rc);
            cont_slots.push_back(slot);
            carry_slot = slot + 2;
            return FY_OK;
        }
        carry_slot = -1;
        // (the deferred-slot pool is full: the separate continuity sweep streams alphacf from its face array, which the fused sweeps do not keep filled)
        if (pimple && !face_arrays) FY_TRY(FVK(launch_interp_alpha, stream, g, alpha.p, F3(alphaf)));
        FY_TRY(FVK(launch_cont_err, stream, g, C3(phi), C3(alphaf), alpha.p, /* alphaOld */ alpha.p, partials.p));
        if (reduce_deferred(2, false, &slot, &rc)) { FY_TRY(rc); cont_slots.push_back(slot); }
        else { FY_TRY(reduce_read(2, false, h)); note_cont_err(h); }
        return FY_OK;
    }
    if (diag_fused) {
        // the continuity errors and the NEXT step's Courant sums ride on the velocity correction's sweep
        // (k_U_correct<true>; same values as k_cont_err / k_courant) instead of being two sweeps of their own
        FY_TRY(FVK(launch_U_correct_diag, stream, g, HbyA.p, rAU.p, p.p, C3(psn), C3(phiForces), C3(pflux), C3(alphaf), C3(rAUf), U.p, C3(phi), alpha.p,
                                     /* alphaOld */ alpha.p, partials.p));
        U_ghosts_fresh = false;
        if (!reduce_deferred(4, false, &slot, &rc, ops_diag.p)) return fail(FY_ERR_INVALID, "no room for the deferred diagnostics");
        FY_TRY(rc);
        cont_slots.push_back(slot);
        carry_slot = slot + 2;
        return FY_OK;
    }
    carry_slot = -1;
    FY_TRY(FVK(launch_cont_err, stream, g, C3(phi), C3(alphaf), alpha.p, /* alphaOld */ alpha.p, partials.p));
    if (reduce_deferred(2, false, &slot, &rc)) { FY_TRY(rc); cont_slots.push_back(slot); }
    else { FY_TRY(reduce_read(2, false, h)); note_cont_err(h); }
    FY_TRY(FVK(launch_U_correct, stream, g, HbyA.p, rAU.p, p.p, C3(psn), C3(phiForces), C3(pflux), C3(alphaf), C3(rAUf), U.p));
    U_ghosts_fresh = false;
    return FY_OK;
}

// continuousPhaseTurbulence->correct() for LES Smagorinsky: nut from the Gauss-linear gradient of the corrected velocity
int Solver::turbulence_correct() {
    Comm::Tag tag(comm, "turbulence");
    FY_TRY(halo_U());
    FY_TRY(FVK(launch_pre_coupling, stream, g, U.p, p.p, alpha.p, C3(psn), vGrad.p, gradP.p, divT.p, nullptr, 1, 0));
    if (cs.turbulence_model == FY_TURBULENCE_KEQN || cs.turbulence_model == FY_TURBULENCE_KEPSILON) {
        // kEqn::correct() / kEpsilon::correct(): each transport equation is assembled into the momentum matrix's storage and solved by the
        // momentum solver's pass as a 3-component system whose last two components are identically zero (HbyA / bmom / xscr are free
        // after the correctors).  kEpsilon: the dissipation equation first, then k with the new epsilon, then nut = Cmu k^2 / epsilon
        const bool keps = cs.turbulence_model == FY_TURBULENCE_KEPSILON;
        int it = 0;
        FY_TRY(halo_cells(kturb, 1, 1));
        if (keps) {
            FY_TRY(halo_cells(epsturb, 1, 1));
            FY_TRY(FVK(launch_assemble_turb, stream, g, eq_eps, kturb.p, epsturb.p, alpha.p, C3(alphaf), C3(phi), vGrad.p, U.p, M7(), bmom.p, HbyA.p));
            FY_TRY(solve_vec3(HbyA, bmom.p, cs.eps_tol, cs.eps_rel_tol, cs.eps_max_iter, &it));
            st_k_iters += it;
            FY_TRY(halo_cells(HbyA, 3, 1));
            FY_TRY(FVK(launch_turb_finish, stream, g, eq_eps, HbyA.p, epsturb.p, 0, 0.0, nullptr, nullptr));
        }
        FY_TRY(FVK(launch_assemble_turb, stream, g, eq_k, kturb.p, epsturb.p, alpha.p, C3(alphaf), C3(phi), vGrad.p, U.p, M7(), bmom.p, HbyA.p));
        FY_TRY(solve_vec3(HbyA, bmom.p, cs.k_tol, cs.k_rel_tol, cs.k_max_iter, &it));
        st_k_iters += it;
        FY_TRY(halo_cells(HbyA, 3, 1));
        FY_TRY(FVK(launch_turb_finish, stream, g, eq_k, HbyA.p, kturb.p, keps ? 2 : 1, cs.ras_cmu, epsturb.p, nut.p));
        FY_TRY(halo_cells(kturb, 1, 1));
        g.nut_wall_live = 1;              // correctNut(): from now on the wall-function patches carry nut_w(k), not the file's value
        return halo_cells(nut, 1, 1);
    }
    FY_TRY(FVK(launch_smagorinsky_nut, stream, g, vGrad.p, cs.les_ck, cs.les_ce, les_delta, nut.p));
    return halo_cells(nut, 1, 1);
}

// ---- one pass of the while (runTime.loop()) body ---------------------------------------------------------------------
int Solver::step() {
    Comm::Tag tag(comm, "step_start");
    FY_HIP(hipSetDevice(device));
    st = fy_step_stats{}; st.cont_err_cumulative = cumulative_cont_err;
    clk_mom.on = clk_pres.on = timing; clk_mom.per_collect = clk_pres.per_collect = 1024; clk_mom.reset(); clk_pres.reset();
    if (timing) tim[3].start(stream);
    if (sources_pending) { FY_TRY(cpl->c.set_source_zero()); sources_pending = false; }   // the previous step's deferred setSourceZero
    double h[2];
    const bool carried = carry_valid;                                                      // the previous pass's last corrector already summed |phi|
    carry_valid = false; carry_slot = -1;
    if (!carried) FY_TRY(FVK(launch_courant, stream, g, C3(phi), partials.p));                 // icoFoamYade.C:68, pimpleFoamYade.C:63
    n_deferred = 0; cont_slots.clear(); courant_slot = -1;
    if (carried) note_courant(carry_h);
    if (cs.adjust_time_step) {
        // readTimeControls.H + CourantNo.H + setDeltaT.H (pimpleFoamYade.C:62-64) [OF-6 setDeltaT.H]: the step's deltaT follows from the
        // Courant number of the current flux at the OLD deltaT, so the host needs that number now
        if (!carried) { FY_TRY(reduce_read(2, true, h)); note_courant(h); }
        const double maxDeltaTFact = cs.max_co / (st.courant_max + 1e-15);
        const double deltaTFact = std::min(std::min(maxDeltaTFact, 1.0 + 0.1 * maxDeltaTFact), 1.2);
        cs.dt = std::min(deltaTFact * cs.dt, cs.max_delta_t);
        g.dt = cs.dt;
    } else if (!carried) {
        int slot = 0, rc = FY_OK;
        if (reduce_deferred(2, true, &slot, &rc)) { FY_TRY(rc); courant_slot = slot; }
        else { FY_TRY(reduce_read(2, true, h)); note_courant(h); }
    }
    st.delta_t = cs.dt;
    // runTime++ : store old-time fields.  phi.oldTime(): the flux arrays trade places instead of being copied -- what was phi is phiOld now, and until
    // the first flux correction of this step rewrites phi (every face of it) the current flux is read from phiOld (phi_now())
    if (cs.n_correctors > 0) {
        for (int d = 0; d < 3; ++d) { std::swap(phi[d].p, phiOld[d].p); std::swap(phi[d].n, phiOld[d].n); }
        phi_fresh = false;
    } else {
        for (int d = 0; d < 3; ++d) FY_TRY(launch_copy_f64(stream, phiOld[d].p, phi[d].p, phi[d].n));
    }
    // icoFoamYade.C:71, pimpleFoamYade.C:73-76.  pimple: alpha is 1 here (reset by setSourceZero), so G is re-formed after the
    // coupling call with this step's alpha; only gradP / divT are needed now.
    // the opt-in force models (fy_set_force_models on fy_solver_coupling()) read vGrad / ddtU_f, which the shipped path never does
    const unsigned fm = cpl->c.force_models;
    const bool want_vgrad = !pimple || (fm & FY_FORCE_GAUSSIAN_TORQUE), want_ddtU = pimple && (fm & FY_FORCE_ADDED_MASS);
    // single domain, Gaussian mode: the sweep also leaves the force pass's packed cell records (the coupling then skips its own pack pass)
    // (a slab keeps the coupling's own pack pass: measured in round 6, the records written from this sweep cost a slab 0.17 ms per step -- the sweep is not hidden
    //  beside the walk there -- against 0.12 for the pass; the coupling can take owned-cell records from here all the same: Coupling::cellrec_ghosts_stale)
    double* rec_out = (pimple && comm->size == 1) ? cpl->c.d_cellrec.p : nullptr;
    // On a single domain in Gaussian mode the sweep is handed to the coupling as a hook and launched right after the locate + deposit (which
    // read no fluid field): it then runs beside the side stream's tree walk of the few particles the candidate lists hand over -- ~90 us of
    // memory latency that would otherwise sit alone between the locate and the cells' finalisation (Coupling::mid_hook)
    const bool defer_sweep = comm->size == 1 && cpl->c.gaussian;
    // U.oldTime() of the owned cells is written by the pre-coupling sweep that reads U anyway
    const bool fuse_uold = true;
    auto sweep_on = [&](const FvGeo& gw) -> int {
        return FVK(launch_pre_coupling, stream, gw, U.p, p.p, alpha.p, C3(psn), vGrad.p, gradP.p, divT.p, nullptr, want_vgrad ? 1 : 0, 1, phi_now(),
                   want_ddtU ? ddtU.p : nullptr, fuse_uold ? Uold.p : nullptr, rec_out, cpl->c.nu, cpl->c.rhoF,
                   (fused_corrector && !adjust_phi) ? F3(dcorr) : Face3{});
    };
    std::function<int()> pre_sweep = [&]() -> int { return sweep_on(g); };
    // the step's opening exchange: U goes with the full particle-halo width straight away; p's ghosts from the last corrector still stand unless p was
    // written since.  On slabs the sweep that consumes it runs beside it (interior planes first)
    auto opening_exchange = [&](hipStream_t on) -> int {
        comm->group_begin();
        FY_TRY(halo_cells(U, 3, g.gz > 1 ? g.gz : 1, on));
        if (!p_ghosts_fresh) FY_TRY(halo_cells(p, 1, 1, on));
        FY_TRY(halo_cells(alpha, 1, 1, on));
        return comm->group_end(on);
    };
    if (defer_sweep) {
        FY_TRY(opening_exchange(stream));                      // (one domain: nothing travels)
        cpl->c.mid_hook = [](void* u) -> int { return (*static_cast<std::function<int()>*>(u))(); };
        cpl->c.mid_hook_user = &pre_sweep;
    } else {
        cpl->c.mid_hook = nullptr;
        FY_TRY(overlapped(XW_STEP, true, opening_exchange, sweep_on));
    }
    p_ghosts_fresh = true; U_ghosts_fresh = true;
    // a slab copies U.oldTime()'s ghost planes (the opening sweep's ddtCorr coefficient and the separate phiHbyA sweep read Uold across the slab faces),
    // which the exchange above has just refreshed in U
    if (comm->size > 1 && g.gz > 0) {
        const size_t gh = 3 * plane * (size_t)g.gz;
        FY_TRY(launch_copy_f64(stream, Uold.p, U.p, gh));
        FY_TRY(launch_copy_f64(stream, Uold.p + 3 * plane * (size_t)(g.gz + g.nz), U.p + 3 * plane * (size_t)(g.gz + g.nz), gh));
    }
    cpl->c.cellrec_external = rec_out != nullptr;

    if (timing) tim[0].start(stream);
    comm->tag = "particle";
    if (g.gz > 1) {                      // the particle gathers reach gz planes into the neighbours
        // with the second channel the planes travel while the coupling locates and deposits (which read no fluid field); it waits for ev_fields
        // where it first gathers one (Coupling::run_batch, pack_records)
        const bool side_ch = cpl->c.slab.aux != nullptr;
        hipStream_t on = side_ch ? comm_stream : stream;
        if (side_ch) {
            if (!ev_fields) FY_HIP(hipEventCreateWithFlags(&ev_fields, hipEventDisableTiming));
            FY_HIP(hipEventRecord(ev_ready, stream));
            FY_HIP(hipStreamWaitEvent(comm_stream, ev_ready, 0));
        }
        comm->group_begin();
        FY_TRY(halo_cells(gradP, 3, g.gz, on)); FY_TRY(halo_cells(divT, 3, g.gz, on));
        if (fm & FY_FORCE_GAUSSIAN_TORQUE) FY_TRY(halo_cells(vGrad, 9, g.gz, on));
        if (want_ddtU) FY_TRY(halo_cells(ddtU, 3, g.gz, on));
        FY_TRY(comm->group_end(on));
        if (side_ch) { FY_HIP(hipEventRecord(ev_fields, comm_stream)); cpl->c.slab.fields_event = ev_fields; }
    }
    usum_pending = false;
    if (comm->size == 1 && cs.momentum_predictor) {
        if (!usum_partials.p) FY_TRY(usum_partials.alloc_exact(3 * (size_t)red_blocks(Nc)));
        FY_HIP(hipEventRecord(ev_usum0, stream));
        FY_HIP(hipStreamWaitEvent(comm_stream, ev_usum0, 0));
        FY_TRY(launch_sum3(comm_stream, U.p + 3 * (size_t)g.c0, Nc, usum_partials.p));
        FY_TRY(launch_reduce_finalize(comm_stream, usum_partials.p, Nc, 3, nullptr, xbar3.p));
        FY_HIP(hipEventRecord(ev_usum1, comm_stream));
        usum_pending = true;
    }
    {
        cpl->c.async_results = true;               // (a zero-copy wire: the answers' D2H copies run under the fluid solve, handed over as they land)
        const int rc = cpl->c.set_particle_action(cs.dt);                                 // icoFoamYade.C:74, pimpleFoamYade.C:78
        cpl->c.mid_hook = nullptr; cpl->c.mid_hook_user = nullptr;                        // (pre_sweep lives in this frame only)
        FY_TRY(rc);
    }
    if (timing) { tim[0].stop(stream); }
    comm->tag = "momentum";

    // alphac.oldTime() is captured lazily by OpenFOAM at alphac.correctBoundaryConditions() (pimpleFoamYade.C:83), i.e. after
    // FoamYade wrote alpha through untracked operator[]: old == current, fvc::ddt(alphac) == 0.
    // The kernels keep their alphaOld argument (the term is written out as in UcEqn.H:5 / pEqn.H:30); it is handed the same array
    // -- what a copy taken here would hold, without the copy or a second stream of reads.
    // pimpleFoamYade.C:83-85 (alpha ghosts refreshed by the coupling).  The fused sweeps re-form alphacf from alpha where they need it: the face
    // array is only filled for the passes that still stream it (face_arrays)
    if (pimple && face_arrays) FY_TRY(FVK(launch_interp_alpha, stream, g, alpha.p, F3(alphaf)));
    const int nOuter = pimple ? std::max(cs.n_outer_correctors, 1) : 1;
    for (int outer = 0; outer < nOuter; ++outer) {
        // pimple.loop() marks the last outer corrector "finalIteration": relax() then prefers the <name>Final factors [OF-6], and
        // stores p.prevIter() at the start of every outer iteration when p carries a relaxation factor (storePrevIterFields)
        const bool final_outer = outer == nOuter - 1;
        g.u_relax = (final_outer && cs.u_relax_final > 0) ? cs.u_relax_final : cs.u_relax;
        p_relax_now = (final_outer && cs.p_relax_final > 0) ? cs.p_relax_final : cs.p_relax;
        if (pimple && p_relax_now > 0 && p_relax_now < 1) {
            if (!pPrev.p) FY_TRY(pPrev.alloc_exact(nstore));
            FY_TRY(launch_copy_f64(stream, pPrev.p, p.p, nstore));
        }
        clk_mom.begin(stream);
        if (pimple) {
            // explicit stress term of divDevRhoReff from the CURRENT U and this step's alpha (one fused stencil pass)
            if (outer > 0) FY_TRY(halo_U());
            FY_TRY(FVK(launch_pre_coupling, stream, g, U.p, p.p, alpha.p, C3(psn), vGrad.p, gradP.p, divT.p, Gt.p, g.upwind == 2 ? 1 : 0, 0));
            // G is stored by rows; only row z is read across the slab faces
            FY_TRY(overlapped(XW_MOMENTUM, true, [&](hipStream_t on) { return halo(Gt.p + 2 * 3 * nstore, 3, plane, g.nz, g.gz, 1, on); },
                              [&](const FvGeo& gw) { return FVK(launch_div_G, stream, gw, Gt.p, divG.p); }));
        }
        if (g.upwind == 2) FY_TRY(halo_cells(vGrad, 9, 1));      // linearUpwind reads grad(U) of the upwind neighbour (ico: written at step start)
        if (g.upwind >= 3) {                                     // the limited schemes: gradient ratio from grad(magSqr(U)) of the current U
            FY_TRY(halo_U());
            FY_TRY(FVK(launch_grad_magsqr, stream, g, U.p, gradL.p));
            FY_TRY(halo_cells(gradL, 3, 1));
        }
        FY_TRY(FVK(launch_assemble_momentum, stream, g, U.p, Uold.p, alpha.p, /* alphaOld */ alpha.p, face_arrays ? C3(alphaf) : CFace3{}, phi_now(), uSource.p, uSourceDrag.p,
                                        divG.p, g.upwind >= 3 ? gradL.p : vGrad.p, M7(), src.p, rAU.p));
        rAU_new = true; hbya_ready = false;
        // pimple: rAUcf, phicForces and the predictor's right-hand side in one sweep (k_bmom_faces); uSource ghosts refreshed by the coupling
        const bool bmom_fused = pimple && cs.momentum_predictor && fused_corrector;
        if (pimple) {
            // (uSource's ghost planes may still be on their way: the coupling sent them on the second channel)
            if (cpl->c.slab.tail_pending) { FY_HIP(hipStreamWaitEvent(stream, cpl->c.slab.ev_tail, 0)); cpl->c.slab.tail_pending = false; }
            if (bmom_fused) {
                FY_TRY(overlapped(XW_MOMENTUM, true, [&](hipStream_t on) { return halo_cells(rAU, 1, 1, on); }, [&](const FvGeo& gw) {
                    return FVK(launch_bmom_faces, stream, gw, rAU.p, uSource.p, src.p, p.p, C3(psn), face_arrays ? F3(rAUf) : Face3{}, F3(phiForces), bmom.p);
                }));
            } else {
                FY_TRY(halo_cells(rAU, 1, 1));
                FY_TRY(FVK(launch_rAUf_phi_forces, stream, g, rAU.p, uSource.p, F3(rAUf), F3(phiForces)));
            }
        }
        if (cs.momentum_predictor) {
            if (!bmom_fused) FY_TRY(FVK(launch_bmom, stream, g, src.p, p.p, C3(psn), C3(phiForces), C3(rAUf), bmom.p));
            int it = 0;
            FY_TRY(solve_momentum(&it));
            st.u_iters_total += it;
        }
        clk_mom.end(stream);
        for (int corr = 0; corr < cs.n_correctors; ++corr) FY_TRY(corrector(outer == nOuter - 1 && corr == cs.n_correctors - 1));
        if (g.nut && final_outer) FY_TRY(turbulence_correct());        // pimple.turbCorr(): on the final outer iteration only (the default) -- pimpleFoamYade.C:101-104
    }
    FY_TRY(cpl->c.finish_results());                                                      // the answers still on their way, then the dt handshake (FoamYade.C:537-553)
    if (hold_sources) sources_pending = true;                                              // reset deferred to the next step (fy_solver_hold_sources)
    else FY_TRY(cpl->c.set_source_zero());                                                // icoFoamYade.C:147, pimpleFoamYade.C:109
    if (timing) tim[3].stop(stream);
    FY_HIP(hipStreamSynchronize(stream));
    if (adjust_phi) {
        int e = 0;
        FY_HIP(hipMemcpy(&e, adj_err.p, sizeof(int), hipMemcpyDeviceToHost));
        if (e) return fail(FY_ERR_UNSUPPORTED, "adjustPhi: continuity error cannot be removed by adjusting the outflow (the in- and outflow through the "
                                               "fixed-value patches do not balance and no adjustable outflow is left) -- OpenFOAM stops here too");
    }
    if (courant_slot >= 0) note_courant(red_host + courant_slot);                         // the deferred diagnostics have landed
    for (int sl : cont_slots) note_cont_err(red_host + sl);
    if (carry_slot >= 0) { carry_h[0] = red_host[carry_slot]; carry_h[1] = red_host[carry_slot + 1]; carry_valid = true; }
    if (timing) {
        clk_mom.collect(); clk_pres.collect();
        st.ms_momentum = clk_mom.total_ms; st.ms_pressure = clk_pres.total_ms;
        st.ms_particle = tim[0].ms() - cpl->c.marks.ms(6, 7);      // (the sweep that ran as the coupling's hook is the solver's, not FoamYade's: it counts as "other")
        st.ms_total = tim[3].ms();
        st.ms_other = st.ms_total - st.ms_particle - st.ms_momentum - st.ms_pressure;
        for (auto& k : kc) k.collect();
    }
    if (xwait_timing) {
        for (auto& k : clk_xwait) k.collect();
    }
    return FY_OK;
}

// field lookup; cell fields are returned/accepted as the OWNED part only (ghost planes are an implementation detail)
int Solver::field(const char* name, double** ptr, size_t* count) {
    const std::string s = name ? name : "";
    const size_t n = (size_t)Nc;
    struct E { const char* nm; double* p; size_t c; int comp; };
    const E tab[] = {{"U", U.p, 3 * n, 3}, {"p", p.p, n, 1}, {"phi_x", phi[0].p, phi[0].n, 0}, {"phi_y", phi[1].p, phi[1].n, 0}, {"phi_z", phi[2].p, phi[2].n, 0},
                     {"rAU", rAU.p, n, 1}, {"HbyA", HbyA.p, 3 * n, 3}, {"p_rhs", prhs.p, n, 1}, {"mom_diag", mdiag.p, n, 1}, {"mom_src", src.p, 3 * n, 3},
                     {"alpha", alpha.p, n, 1}, {"uSource", uSource.p, 3 * n, 3}, {"uSourceDrag", uSourceDrag.p, n, 1}, {"uParticle", uParticle.p, 3 * n, 3},
                     {"gradP", gradP.p, 3 * n, 3}, {"divT", divT.p, 3 * n, 3}, {"vGrad", vGrad.p, 9 * n, 9}, {"ddtU", ddtU.p, 3 * n, 3}, {"nut", nut.p, n, 1}, {"k", kturb.p, n, 1}, {"epsilon", epsturb.p, n, 1}};
    for (const E& e : tab) if (s == e.nm) {
        if (!e.p) return fail(FY_ERR_INVALID, "solver field '%s' does not exist in this case (no turbulence model)", s.c_str());
        // kEqn / kEpsilon assemble and solve their transport equations in the momentum matrix's storage after the last corrector
        // (turbulence_correct): these three diagnostics would then return transport-equation data beside a momentum rAU
        const bool transport = cs.turbulence_model == FY_TURBULENCE_KEQN || cs.turbulence_model == FY_TURBULENCE_KEPSILON;
        if (transport && (s == "HbyA" || s == "mom_diag" || s == "mom_src"))
            return fail(FY_ERR_UNSUPPORTED, "solver field '%s' is overwritten by the turbulence transport solve in this case (kEqn / kEpsilon reuse the momentum matrix's storage)", s.c_str());
        *ptr = e.p + (size_t)e.comp * g.c0;          // skip the ghost planes below the owned range (comp = 0: face array)
        *count = e.c;
        return FY_OK;
    }
    MgLev& L = *mg[0];
    const struct { const char* nm; double* p; } pm[] = {{"p_diag", L.diag.p}, {"p_ux", L.ux.p}, {"p_uy", L.uy.p}, {"p_uz", L.uz.p}};
    for (auto& e : pm) if (s == e.nm) { *ptr = e.p + L.A.c0; *count = n; return FY_OK; }
    return fail(FY_ERR_INVALID, "unknown solver field '%s'", s.c_str());
}

}  // namespace fy
