// fy_ldu_solver: the time-loop body of icoFoamYade (icoFoamYade/icoFoamYade.C:65-149) on a general polyhedral mesh (ldu.hpp), sequencing the
// kernels of ldu_kernels.hip, with the coupling engine (fy_ctx, point force) sharing the device fields and the stream.  Control flow follows the
// reference line by line; the pressure solver is PCG.C's loop [OF-6] in the single-reduction form of fv_pressure.cpp with the diagonal
// preconditioner, the momentum predictor Jacobi sweeps with lduMatrix::solver's residual control (the stand-in for smoothSolver, as in fv_solver.cpp).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "coupling.hpp"
#include "fv_kernels.hpp"
#include "ldu.hpp"
#include "ldu_amg.hpp"

namespace fy {

struct LduSolver {
    LduHostMesh hm;
    fy_ldu_case cs{};
    LduGeo g{};
    int device = 0;
    hipStream_t stream = nullptr, side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_fields = nullptr;
    fy_ctx* cpl = nullptr;
    int nc = 0, nf = 0, ni = 0;
    // geometry + addressing on the device
    DevBuf<int32_t> d_own, d_nei, d_patch_of, d_cf_off, d_cf_face, d_ubc, d_pbc, d_ef, d_en;
    DevBuf<double> d_Cf, d_Sf, d_magSf, d_C, d_V, d_w, d_dcNO, d_kvec, d_uval, d_pval, d_recon;
    // fields
    DevBuf<double> U, Uold, p, phi, phiOld, uSource, uSourceExt, uSourceSum, vGrad, gradp, dummy3, dummy1;
    DevBuf<double> mdiag, mlower, mupper, mb, fcorr, xscr, rAU, HbyA, rAUf, phiHbyA, pcoef, pcorr, pdiag, prhs, pr, pu, pw, pp, ps;
    DevBuf<double> partials, sc, xsum, adj;
    // pimpleFoamYade: the coupling's fields and the alpha-weighted equations' face fields
    bool pimple = false, hold_sources = false, sources_pending = false;
    DevBuf<double> alpha, alphaf, uSourceDrag, uParticle, gradP, divT, ddtU, fstress, phiForces, psn, arAUf, phiA, ssf, bmom, pPrev;
    DevBuf<double> nut, d_nutval, gradL;
    DevBuf<int32_t> d_nutbc;
    bool les = false;
    bool keqn = false, nut_live = false;
    long long k_iters_total = 0;
    bool keps = false;
    DevBuf<double> kturb, d_kval, gradk, epsturb, d_epsval;
    DevBuf<int32_t> d_kbc, d_epsbc;
    LduPim P() const {
        return LduPim{alpha.p, alpha.p, alphaf.p, uSourceDrag.p, uSource.p, {cs.g[0], cs.g[1], cs.g[2]}, les ? nut.p : nullptr, d_nutbc.p, d_nutval.p,
                      (keqn || keps) ? kturb.p : nullptr, d_kbc.p, d_kval.p, nut_live ? 1 : 0, cs.les_ck, cs.les_delta_coeff,
                      keps ? epsturb.p : nullptr, d_epsbc.p, d_epsval.p, cs.ras_cmu};
    }      // alphac.oldTime() == alphac
    DevBuf<int> adj_err;
    bool need_ref = true, ext_source = false, has_slip = false;
    std::vector<double> orig_face_d;      // hm.orig_face as doubles (the field read-out's type); empty without a cyclic pair
    DevBuf<double> d_gB, d_gG0, d_rT;      // the per-slot coefficients of the scalar gradient and of fvc::reconstruct (LduGeo)
    DevBuf<double> pt;           // per face: -phiHbyA + the corrected laplacian's explicit flux (what the pressure equation's cell part sums)
    DevBuf<double> d_sep;        // folded cyclic faces: the neighbour image's offset per internal face
    DevBuf<double> mbdiag;       // symmetry patches: the momentum matrix's per-component boundary diagonal
    LduAmg amg;                  // the pressure matrix in ELL form; with p_solver = FY_PSOLVER_PCG_MG also the agglomeration hierarchy
    fy_step_stats st{};
    double cumulative = 0.0, total_volume = 0.0;
    EventTimer tim[2];

    ~LduSolver() {
        if (cpl) fy_destroy(cpl);
        if (red_host) (void)hipHostFree(red_host);
        if (red_flag) (void)hipHostFree(red_flag);
        for (auto& t : tim) t.destroy();
        if (side) (void)hipStreamDestroy(side);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_fields) (void)hipEventDestroy(ev_fields);
        if (stream) (void)hipStreamDestroy(stream);
    }
    template <class T> int up(DevBuf<T>& d, const std::vector<T>& h) {
        FY_TRY(d.alloc_exact(std::max<size_t>(h.size(), 1)));
        if (!h.empty()) { FY_HIP(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream)); FY_HIP(hipStreamSynchronize(stream)); }      // (set-up only; the source may die with its scope)
        return FY_OK;
    }
    int zero(DevBuf<double>& b) { if (b.n) FY_HIP(hipMemsetAsync(b.p, 0, b.n * sizeof(double), stream)); return FY_OK; }
    LduMom M() { return LduMom{mdiag.p, mlower.p, mupper.p, mb.p, has_slip ? mbdiag.p : nullptr}; }

    int create(const fy_poly_mesh* m, const fy_ldu_case* c, const fy_transport* tr, int dev) {
        if (!c || !(c->dt > 0) || !(c->nu >= 0) || !c->u_bc || !c->u_value || !c->p_bc || !c->p_value) return fail(FY_ERR_INVALID, "fy_ldu_solver_create: bad case (dt, nu, the per-patch arrays)");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(FY_ERR_NO_DEVICE, "no HIP device visible: libfoamyade_hip has no CPU path");
        if (dev < 0 || dev >= ndev) return fail(FY_ERR_INVALID, "device ordinal out of range");
        FY_TRY(hm.build(m));
        cs = *c; device = dev;
        pimple = c->solver == FY_SOLVER_PIMPLE;
        if (c->solver != FY_SOLVER_ICO && !pimple) return fail(FY_ERR_INVALID, "fy_ldu_solver: solver %d (FY_SOLVER_ICO, FY_SOLVER_PIMPLE)", c->solver);
        if (pimple && c->n_outer_correctors < 1) cs.n_outer_correctors = 1;
        if (c->adjust_time_step && !pimple) cs.adjust_time_step = 0;                      // (icoFoamYade's loop never includes setDeltaT.H: icoFoamYade.C:65-70)
        if (cs.adjust_time_step && !(c->max_co > 0 && c->max_delta_t > 0)) return fail(FY_ERR_INVALID, "fy_ldu_solver: adjustTimeStep needs maxCo > 0 and maxDeltaT > 0");
        if (c->turbulence_model != FY_TURBULENCE_LAMINAR && !(pimple && (c->turbulence_model == FY_TURBULENCE_SMAGORINSKY || c->turbulence_model == FY_TURBULENCE_KEQN || c->turbulence_model == FY_TURBULENCE_KEPSILON)))
            return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: turbulence model %d (laminar; LES Smagorinsky, LES kEqn or RAS kEpsilon with pimpleFoamYade)", c->turbulence_model);
        keps = pimple && c->turbulence_model == FY_TURBULENCE_KEPSILON;
        if (keps && !(c->ras_cmu > 0 && c->ras_sigmak > 0 && c->ras_sigmaeps > 0 && c->eps_initial > 0 && c->k_initial > 0 && c->eps_tol >= 0 && c->eps_max_iter >= 0 &&
                      (c->eps_convection_scheme == FY_CONVECTION_LINEAR || c->eps_convection_scheme == FY_CONVECTION_UPWIND) &&
                      (c->k_convection_scheme == FY_CONVECTION_LINEAR || c->k_convection_scheme == FY_CONVECTION_UPWIND)))
            return fail(FY_ERR_INVALID, "fy_ldu_solver: kEpsilon needs Cmu, sigmak, sigmaEps, k and epsilon of the start time positive, and Gauss linear or Gauss upwind for their convection");
        les = pimple && c->turbulence_model != FY_TURBULENCE_LAMINAR;
        keqn = pimple && c->turbulence_model == FY_TURBULENCE_KEQN;
        if (keqn && !(c->k_initial >= 0 && c->k_tol >= 0 && c->k_max_iter >= 0 && (c->k_convection_scheme == FY_CONVECTION_LINEAR || c->k_convection_scheme == FY_CONVECTION_UPWIND)))
            return fail(FY_ERR_INVALID, "fy_ldu_solver: kEqn needs k_initial >= 0, solver controls for k and Gauss linear or Gauss upwind for div(alphaPhic,k)");
        if (c->convection_scheme < FY_CONVECTION_LINEAR || c->convection_scheme > FY_CONVECTION_QUICK)
            return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: convection scheme %d (FY_CONVECTION_LINEAR .. FY_CONVECTION_QUICK)", c->convection_scheme);
        if (c->convection_scheme == FY_CONVECTION_LIMITED_LINEAR && !(c->convection_limiter_k >= 0 && c->convection_limiter_k <= 1)) return fail(FY_ERR_INVALID, "fy_ldu_solver: limitedLinear takes a coefficient in [0, 1]");
        if (les && !(c->les_ck > 0 && c->les_ce > 0 && c->les_delta_coeff > 0)) return fail(FY_ERR_INVALID, "fy_ldu_solver: Smagorinsky needs Ck, Ce and the delta coefficient positive");
        nc = hm.nCells; nf = hm.nFaces; ni = hm.nInt;
        std::vector<int32_t> ubc(c->u_bc, c->u_bc + hm.nPatches), pbc(c->p_bc, c->p_bc + hm.nPatches);
        std::vector<double> uval(c->u_value, c->u_value + 3 * (size_t)hm.nPatches), pval(c->p_value, c->p_value + hm.nPatches);
        need_ref = true;
        for (int pa = 0; pa < hm.nPatches; ++pa) {
            if (ubc[(size_t)pa] != FY_BC_U_FIXED_VALUE && ubc[(size_t)pa] != FY_BC_U_ZERO_GRADIENT && ubc[(size_t)pa] != FY_BC_U_SLIP)
                return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: velocity patch type %d (fixedValue, zeroGradient, symmetry / slip)", ubc[(size_t)pa]);
            has_slip = has_slip || ubc[(size_t)pa] == FY_BC_U_SLIP;
            if (pbc[(size_t)pa] != FY_BC_P_ZERO_GRADIENT && pbc[(size_t)pa] != FY_BC_P_FIXED_VALUE && !(pimple && pbc[(size_t)pa] == FY_BC_P_FIXED_FLUX))
                return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: pressure patch type %d (zeroGradient, fixedValue; fixedFluxPressure with pimpleFoamYade)", pbc[(size_t)pa]);
            if (pbc[(size_t)pa] == FY_BC_P_FIXED_VALUE) need_ref = false;
        }
        if (need_ref && (c->p_ref_cell < 0 || c->p_ref_cell >= nc)) return fail(FY_ERR_INVALID, "fy_ldu_solver: pRefCell out of range");
        if (c->p_solver != FY_PSOLVER_PCG_JACOBI && c->p_solver != FY_PSOLVER_PCG_MG) return fail(FY_ERR_INVALID, "fy_ldu_solver: p_solver %d (FY_PSOLVER_PCG_JACOBI, FY_PSOLVER_PCG_MG)", c->p_solver);
        cs.u_bc = nullptr; cs.u_value = nullptr; cs.p_bc = nullptr; cs.p_value = nullptr;      // (copied; the caller's arrays are not kept)
        FY_HIP(hipSetDevice(device));
        FY_HIP(hipStreamCreate(&stream));
        FY_TRY(up(d_own, hm.own)); FY_TRY(up(d_nei, hm.nei)); FY_TRY(up(d_patch_of, hm.patch_of)); FY_TRY(up(d_cf_off, hm.cf_off)); FY_TRY(up(d_cf_face, hm.cf_face)); FY_TRY(up(d_ef, hm.ef)); FY_TRY(up(d_en, hm.en));
        FY_TRY(up(d_Cf, hm.Cf)); FY_TRY(up(d_Sf, hm.Sf)); FY_TRY(up(d_magSf, hm.magSf)); FY_TRY(up(d_C, hm.C)); FY_TRY(up(d_V, hm.V)); FY_TRY(up(d_w, hm.w));
        FY_TRY(up(d_dcNO, hm.dcNO)); FY_TRY(up(d_kvec, hm.kvec)); FY_TRY(up(d_ubc, ubc)); FY_TRY(up(d_pbc, pbc)); FY_TRY(up(d_uval, uval)); FY_TRY(up(d_pval, pval));
        FY_TRY(up(d_recon, hm.recon));
        if (pimple) { FY_TRY(psn.alloc_exact(std::max<size_t>((size_t)(nf - ni), 1))); FY_TRY(zero(psn)); }
        g = LduGeo{nc, nf, ni, hm.nPatches, d_own.p, d_nei.p, d_patch_of.p, d_cf_off.p, d_cf_face.p, hm.Wall, d_ef.p, d_en.p, d_Cf.p, d_Sf.p, d_magSf.p, d_C.p, d_V.p, d_w.p, d_dcNO.p, d_kvec.p,
                   d_ubc.p, d_pbc.p, d_uval.p, d_pval.p, d_recon.p, pimple ? psn.p : nullptr, cs.dt, cs.nu, cs.convection_scheme, 2.0 / std::max(cs.convection_limiter_k, 1e-15), nullptr, need_ref ? 1 : 0, cs.p_ref_cell, cs.p_ref_value};
        FY_TRY(d_gB.alloc_exact(3 * (size_t)hm.Wall * nc)); FY_TRY(d_gG0.alloc_exact(3 * (size_t)nc)); FY_TRY(d_rT.alloc_exact(3 * (size_t)hm.Wall * nc));
        FY_TRY(launch_ldu_slot_coefs(stream, g, d_gB.p, d_gG0.p, d_rT.p));
        g.gB = d_gB.p; g.gG0 = d_gG0.p; g.rT = d_rT.p;
        if (!hm.sep.empty()) { FY_TRY(up(d_sep, hm.sep)); g.sep = d_sep.p; }
        g.nIntReal = hm.n_real_internal;
        orig_face_d.assign(hm.orig_face.begin(), hm.orig_face.end());
        total_volume = 0.0;
        for (double v : hm.V) total_volume += v;
        const size_t n = (size_t)nc;
        DevBuf<double>* v3[] = {&U, &Uold, &uSource, &uSourceExt, &uSourceSum, &gradp, &mb, &xscr, &HbyA, &dummy3};
        for (auto* b : v3) { FY_TRY(b->alloc_exact(3 * n)); FY_TRY(zero(*b)); }
        DevBuf<double>* v1[] = {&p, &mdiag, &rAU, &pdiag, &prhs, &pr, &pu, &pw, &pp, &ps, &dummy1};
        for (auto* b : v1) { FY_TRY(b->alloc_exact(n)); FY_TRY(zero(*b)); }
        FY_TRY(vGrad.alloc_exact(9 * n)); FY_TRY(zero(vGrad));
        if (has_slip) { FY_TRY(mbdiag.alloc_exact(3 * n)); FY_TRY(zero(mbdiag)); }
        DevBuf<double>* vf[] = {&phi, &phiOld, &rAUf, &phiHbyA, &pcoef, &pt};
        for (auto* b : vf) { FY_TRY(b->alloc_exact((size_t)nf)); FY_TRY(zero(*b)); }
        DevBuf<double>* vi[] = {&mlower, &mupper, &pcorr};
        for (auto* b : vi) { FY_TRY(b->alloc_exact(std::max<size_t>((size_t)ni, 1))); FY_TRY(zero(*b)); }
        FY_TRY(fcorr.alloc_exact(3 * std::max<size_t>((size_t)ni, 1))); FY_TRY(zero(fcorr));
        FY_TRY(partials.alloc_exact(8 * (size_t)ldu_red_blocks(std::max(nc, nf)))); FY_TRY(zero(partials));
        FY_TRY(sc.alloc_exact(8)); FY_TRY(zero(sc)); FY_TRY(xsum.alloc_exact(4)); FY_TRY(zero(xsum)); FY_TRY(adj.alloc_exact(4)); FY_TRY(zero(adj));
        FY_TRY(adj_err.alloc_exact(1)); FY_HIP(hipMemsetAsync(adj_err.p, 0, sizeof(int), stream));
        if (hipHostMalloc((void**)&red_host, 8 * sizeof(double), hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&red_host_dev, red_host, 0) != hipSuccess) {
            if (red_host) (void)hipHostFree(red_host);
            red_host = nullptr;                              // (fall back to the copy path)
        }
        if (red_host && (hipHostMalloc((void**)&red_flag, 8 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&red_flag_dev, red_flag, 0) != hipSuccess)) {
            if (red_flag) (void)hipHostFree(red_flag);
            red_flag = nullptr;
        }
        if (red_flag) for (int q = 0; q < 8; ++q) red_flag[q] = 0;
        if (pimple) {
            DevBuf<double>* p3[] = {&uParticle, &gradP, &divT, &ddtU, &bmom};
            for (auto* b : p3) { FY_TRY(b->alloc_exact(3 * n)); FY_TRY(zero(*b)); }
            DevBuf<double>* p1[] = {&alpha, &uSourceDrag, &pPrev};
            for (auto* b : p1) { FY_TRY(b->alloc_exact(n)); FY_TRY(zero(*b)); }
            DevBuf<double>* pf[] = {&alphaf, &phiForces, &arAUf, &phiA, &ssf};
            for (auto* b : pf) { FY_TRY(b->alloc_exact((size_t)nf)); FY_TRY(zero(*b)); }
            FY_TRY(fstress.alloc_exact(3 * (size_t)nf)); FY_TRY(zero(fstress));
            FY_TRY(launch_fill_f64(stream, alpha.p, n, 1.0));                 // alpha = 1.0 (FoamYade.C:68)
            FY_TRY(launch_fill_f64(stream, alphaf.p, (size_t)nf, 1.0));
            if (les) {                                                        // nut of the start time: the file's values (eddyViscosity: MUST_READ; no validate() in createFields.H)
                std::vector<int32_t> nb((size_t)hm.nPatches, FY_BC_NUT_ZERO_GRADIENT);
                std::vector<double> nv((size_t)hm.nPatches, 0.0);
                for (int pa = 0; pa < hm.nPatches; ++pa) {
                    if (c->nut_bc) nb[(size_t)pa] = c->nut_bc[pa];
                    if (c->nut_value) nv[(size_t)pa] = c->nut_value[pa];
                    if (nb[(size_t)pa] != FY_BC_NUT_ZERO_GRADIENT && nb[(size_t)pa] != FY_BC_NUT_FIXED_VALUE && !((keqn || keps) && nb[(size_t)pa] == FY_BC_NUT_CALCULATED))
                        return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: nut patch type %d (zeroGradient, fixedValue; calculated with kEqn / kEpsilon; the wall functions need the block solver)", nb[(size_t)pa]);
                }
                FY_TRY(up(d_nutbc, nb)); FY_TRY(up(d_nutval, nv));
                FY_TRY(nut.alloc_exact(n));
                FY_TRY(launch_fill_f64(stream, nut.p, n, c->nut_initial));
            }
            if (keps) {                                                       // epsilon of the start time, its patches (no epsilonWallFunction on a general mesh)
                std::vector<int32_t> eb((size_t)hm.nPatches, FY_BC_NUT_ZERO_GRADIENT);
                std::vector<double> ev((size_t)hm.nPatches, 0.0);
                for (int pa = 0; pa < hm.nPatches; ++pa) {
                    if (c->eps_bc) eb[(size_t)pa] = c->eps_bc[pa];
                    if (c->eps_value) ev[(size_t)pa] = c->eps_value[pa];
                    if (eb[(size_t)pa] != FY_BC_NUT_ZERO_GRADIENT && eb[(size_t)pa] != FY_BC_NUT_FIXED_VALUE)
                        return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: epsilon patch type %d (zeroGradient, fixedValue; epsilonWallFunction needs the block solver)", eb[(size_t)pa]);
                }
                FY_TRY(up(d_epsbc, eb)); FY_TRY(up(d_epsval, ev));
                FY_TRY(epsturb.alloc_exact(n));
                FY_TRY(launch_fill_f64(stream, epsturb.p, n, c->eps_initial));
                cs.eps_bc = nullptr; cs.eps_value = nullptr;
            }
            if (keqn || keps) {                                               // k of the start time (kEqn / kEpsilon: k_ is MUST_READ), its patches
                std::vector<int32_t> kb((size_t)hm.nPatches, FY_BC_NUT_ZERO_GRADIENT);
                std::vector<double> kv((size_t)hm.nPatches, 0.0);
                for (int pa = 0; pa < hm.nPatches; ++pa) {
                    if (c->k_bc) kb[(size_t)pa] = c->k_bc[pa];
                    if (c->k_value) kv[(size_t)pa] = c->k_value[pa];
                    if (kb[(size_t)pa] != FY_BC_NUT_ZERO_GRADIENT && kb[(size_t)pa] != FY_BC_NUT_FIXED_VALUE) return fail(FY_ERR_UNSUPPORTED, "fy_ldu_solver: k patch type %d (zeroGradient, fixedValue)", kb[(size_t)pa]);
                }
                FY_TRY(up(d_kbc, kb)); FY_TRY(up(d_kval, kv));
                FY_TRY(kturb.alloc_exact(n)); FY_TRY(gradk.alloc_exact(3 * n));
                FY_TRY(launch_fill_f64(stream, kturb.p, n, c->k_initial));
                cs.k_bc = nullptr; cs.k_value = nullptr;
            }
        }
        for (auto& t : tim) FY_TRY(t.init());
        if (cs.convection_scheme >= FY_CONVECTION_LIMITED_LINEAR) { FY_TRY(gradL.alloc_exact(3 * n)); FY_TRY(zero(gradL)); g.gradL = gradL.p; }
        FY_TRY(amg.build(stream, nc, ni, hm.own.data(), hm.nei.data(), hm.cf_off, hm.cf_face, hm.magSf.data(), need_ref ? cs.p_ref_cell : -1, cs.p_solver == FY_PSOLVER_PCG_MG));
        // the coupling object on this mesh (icoFoamYade.C:54: point force): its tree over the cell centres, its fields the solver's device arrays
        {
            fy_mesh_desc md{};
            md.n_cells = nc; md.centres = hm.C.data(); md.volumes = hm.V.data();
            md.nx = md.ny = md.nz = 0; md.dx = 0.0;
            for (int a = 0; a < 3; ++a) { md.bbox_min[a] = hm.bbox_min[a]; md.bbox_max[a] = hm.bbox_max[a]; md.origin[a] = hm.bbox_min[a]; }
            fy_field_ptrs fp{};
            fp.location = FY_MEM_DEVICE;
            fp.U = U.p; fp.gradP = pimple ? gradP.p : dummy3.p; fp.vGrad = vGrad.p; fp.divT = pimple ? divT.p : dummy3.p; fp.ddtU = pimple ? ddtU.p : dummy3.p;
            fp.uSourceDrag = pimple ? uSourceDrag.p : dummy1.p; fp.alpha = pimple ? alpha.p : dummy1.p; fp.uSource = uSource.p; fp.uParticle = pimple ? uParticle.p : dummy3.p;
            cpl = new (std::nothrow) fy_ctx();
            if (!cpl) return fail(FY_ERR_INVALID, "out of host memory");
            cpl->c.ext_stream = stream;
            cpl->c.ldu_geo = &g;
            FY_TRY(cpl->c.create(&md, &fp, pimple ? 1 : 0, tr, device));        // gaussianInterp: false in icoFoamYade (icoFoamYade.C:53), true in pimpleFoamYade (pimpleFoamYade.C:52)
            cpl->c.rhoP = c->rho_particle; cpl->c.rhoF = c->rho_fluid; cpl->c.nu = c->nu;      // setScalarProperties (icoFoamYade.C:55)
        }
        FY_TRY(launch_ldu_flux_of(stream, g, U.p, phi.p));                    // createPhi
        FY_HIP(hipStreamSynchronize(stream));
        return FY_OK;
    }

    // fold `nslots` of the block partials of an n-cell reduction and read them back
    // the fold writes straight into mapped pinned host memory and stores a sequence number behind each result; the host spins on those flags (in-order stream: everything
    // enqueued before the fold has completed by then) -- no device-to-host blit and no stream synchronisation per read-back, as fv_solver.cpp's reduce_read
    double* red_host = nullptr; double* red_host_dev = nullptr;
    unsigned long long* red_flag = nullptr; unsigned long long* red_flag_dev = nullptr;
    unsigned long long red_seq = 0;
    int reduce_read(int n, int nslots, const int* ops_dev, double* h, double* dev_land = nullptr) {      // dev_land: where the copy path lets the fold land (default: sc)
        if (red_flag && nslots <= 8) {
            const unsigned long long seq = ++red_seq;
            FY_TRY(launch_reduce_finalize(stream, partials.p, n, nslots, ops_dev, red_host_dev, red_flag_dev, seq));
            for (int q = 0; q < nslots; ++q) {
                unsigned long spins = 0;
                while (__atomic_load_n(&red_flag[q], __ATOMIC_ACQUIRE) != seq) {
                    if ((++spins & 0xfffu) == 0) {                      // every 4096 polls: is the stream still alive?
                        const hipError_t e = hipStreamQuery(stream);
                        if (e == hipSuccess) { if (__atomic_load_n(&red_flag[q], __ATOMIC_ACQUIRE) == seq) break; return fail(FY_ERR_HIP, "reduction flag never arrived"); }
                        if (e != hipErrorNotReady) return fail(FY_ERR_HIP, "stream failed while waiting for a reduction: %s", hipGetErrorString(e));
                    }
                }
                h[q] = red_host[q];
            }
            return FY_OK;
        }
        double* land = dev_land ? dev_land : sc.p;
        FY_TRY(launch_reduce_finalize(stream, partials.p, n, nslots, ops_dev, land, nullptr, 0));
        FY_HIP(hipMemcpyAsync(h, land, nslots * sizeof(double), hipMemcpyDeviceToHost, stream));
        FY_HIP(hipStreamSynchronize(stream));
        return FY_OK;
    }
    DevBuf<int> ops_courant;

    int solve_momentum(int* iters, const double* rhs, const double* gp) {
        FY_TRY(solve_vec3(iters, M(), rhs, gp, &U.p, &xscr.p, cs.u_tol, cs.u_rel_tol, cs.u_max_iter));
        if (cpl) cpl->c.dU = U.p;
        return FY_OK;
    }
    // Jacobi passes on a three-component system with lduMatrix::solver's L1 residual control per component; the converged iterate ends up in *x (the two buffers may trade places)
    int solve_vec3(int* iters, LduMom Mx, const double* rhs, const double* gp, double** x, double** scratch, double tol, double rel_tol, int max_iter) {
        FY_TRY(launch_ldu_sum(stream, *x, nc, 3, partials.p));
        FY_TRY(launch_reduce_finalize(stream, partials.p, nc, 3, nullptr, xsum.p, nullptr, 0));
        double* xc = *x; double* xn = *scratch;
        double norm[3] = {1, 1, 1}, res0[3] = {0, 0, 0}, res[3], h[6];
        int it = 0;
        for (;;) {
            FY_TRY(launch_ldu_mom_pass(stream, g, Mx, rhs, gp, xc, xn, xsum.p, partials.p));
            FY_TRY(reduce_read(nc, 6, nullptr, h));
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
        if (xc != *x) std::swap(*x, *scratch);            // the converged iterate sits in the scratch buffer: the two trade places
        *iters = it;
        return FY_OK;
    }

    // OpenFOAM PCG.C with the diagonal preconditioner and lduMatrix::solver::normFactor, in the single-reduction form (fv_pressure.cpp, k_pcg_cg_update)
    int solve_pressure(bool final_iter) {
        const double tol = final_iter ? cs.p_final_tol : cs.p_tol, rel = final_iter ? cs.p_final_rel_tol : cs.p_rel_tol;
        double h[2];
        FY_TRY(launch_ldu_sum(stream, p.p, nc, 1, partials.p));
        FY_TRY(launch_reduce_finalize(stream, partials.p, nc, 1, nullptr, xsum.p, nullptr, 0));
        { const EllMat A0 = amg.lev[0]->mat(); FY_TRY(launch_ldu_p_init(stream, g, pdiag.p, A0.W, A0.nbr, A0.coef, prhs.p, p.p, xsum.p, 1.0 / (double)nc, pr.p, partials.p)); }
        FY_TRY(reduce_read(nc, 2, nullptr, h));
        const double norm = h[1] + 1e-20;
        double res = h[0] / norm;
        const double res0 = res;
        st.p_initial_residual = res0;
        auto converged = [&](double r) { return r < tol || (rel > 0 && r < rel * res0); };
        int it = 0;
        if (!converged(res)) {
            do {
                if (amg.has_hierarchy()) FY_TRY(amg.vcycle(stream, pr.p, pu.p));                                  // u = M^-1 r
                else FY_TRY(launch_ell_jacobi(stream, nc, pdiag.p, pr.p, pu.p));
                EllMat A = amg.lev[0]->mat();
                A.diag = pdiag.p;
                FY_TRY(launch_ell_apply_dot(stream, A, pu.p, pr.p, pw.p, partials.p));                            // w = A u; gamma = u.r, delta = u.w
                FY_TRY(launch_reduce_finalize(stream, partials.p, nc, 2, nullptr, sc.p, nullptr, 0));
                FY_TRY(launch_pcg_cg_update(stream, nc, 0, pu.p, pw.p, pp.p, ps.p, p.p, pr.p, sc.p, it, partials.p));
                if (it == 0) { std::swap(pp.p, pu.p); std::swap(ps.p, pw.p); }      // p = u, s = w without a pass
                FY_TRY(reduce_read(nc, 2, nullptr, h, sc.p + 6));      // (the copy path must not touch the iteration's scalars in sc[0 .. 5])
                res = h[0] / norm;
            } while (++it < cs.p_max_iter && !converged(res));
        }
        st.p_final_residual = res;
        st.p_iters_total += it; st.p_solves += 1;
        return FY_OK;
    }

    int corrector(bool final_corr) {
        FY_TRY(launch_ldu_HbyA(stream, g, M(), U.p, rAU.p, HbyA.p));                                              // icoFoamYade.C:99-100
        FY_TRY(launch_ldu_phiHbyA(stream, g, HbyA.p, rAU.p, Uold.p, phiOld.p, nullptr, rAUf.p, phiHbyA.p));      // :101-106
        if (need_ref) FY_TRY(launch_ldu_adjust_phi(stream, g, phiHbyA.p, adj.p, adj_err.p, partials.p));                      // :108
        for (int no = 0; no <= cs.n_non_orth_correctors; ++no) {                                                   // :114-131
            FY_TRY(launch_ldu_grad_scalar(stream, g, p.p, gradp.p));
            FY_TRY(launch_ldu_assemble_pressure(stream, g, rAUf.p, phiHbyA.p, gradp.p, pcoef.p, pcorr.p, pt.p, pdiag.p, prhs.p));
            if (no == 0) FY_TRY(amg.setup(stream, pcoef.p, pdiag.p));          // (the non-orthogonal passes renew the right-hand side only)
            FY_TRY(solve_pressure(final_corr && no == cs.n_non_orth_correctors));
            if (no == cs.n_non_orth_correctors) FY_TRY(launch_ldu_flux_correct(stream, g, p.p, phiHbyA.p, pcoef.p, pcorr.p, phi.p));
        }
        FY_TRY(launch_ldu_U_correct(stream, g, HbyA.p, rAU.p, p.p, phi.p, U.p, partials.p));                     // :134-137
        double h[2];
        FY_TRY(reduce_read(nc, 2, nullptr, h));
        st.cont_err_sum_local = cs.dt * h[0] / total_volume; st.cont_err_global = cs.dt * h[1] / total_volume;
        cumulative += st.cont_err_global; st.cont_err_cumulative = cumulative;
        return FY_OK;
    }


    // pEqn.H, one pass of the PISO loop inside a PIMPLE outer corrector
    int corrector_pimple(bool final_corr, double p_relax_now) {
        FY_TRY(launch_ldu_HbyA(stream, g, M(), U.p, rAU.p, HbyA.p));                                              // pEqn.H:2
        FY_TRY(launch_ldu_phiHbyA(stream, g, HbyA.p, rAU.p, Uold.p, phiOld.p, alphaf.p, rAUf.p, phiHbyA.p));    // :4-11
        if (need_ref) FY_TRY(launch_ldu_adjust_phi(stream, g, phiHbyA.p, adj.p, adj_err.p, partials.p));          // :13-16 (before phicForces are added)
        FY_TRY(launch_ldu_add_forces_constrain(stream, g, phiForces.p, rAUf.p, U.p, phiHbyA.p, psn.p));          // :18-21
        FY_TRY(launch_ldu_pim_pfaces(stream, g, alphaf.p, rAUf.p, phiHbyA.p, psn.p, arAUf.p, phiA.p));
        for (int no = 0; no <= cs.n_non_orth_correctors; ++no) {                                                   // :24-47
            FY_TRY(launch_ldu_grad_scalar(stream, g, p.p, gradp.p));
            FY_TRY(launch_ldu_assemble_pressure(stream, g, arAUf.p, phiA.p, gradp.p, pcoef.p, pcorr.p, pt.p, pdiag.p, prhs.p));
            // (fvc::ddt(alphac), :30, is zero: alphac.oldTime() == alphac -- P())
            if (no == 0) FY_TRY(amg.setup(stream, pcoef.p, pdiag.p));
            FY_TRY(solve_pressure(final_corr && no == cs.n_non_orth_correctors));
            if (no == cs.n_non_orth_correctors) {
                FY_TRY(launch_ldu_pim_flux(stream, g, p.p, phiHbyA.p, pcoef.p, pcorr.p, alphaf.p, rAUf.p, phiForces.p, psn.p, phi.p, ssf.p, pt.p));      // :39
                if (p_relax_now > 0 && p_relax_now < 1) FY_TRY(launch_relax_field(stream, p.p, pPrev.p, p_relax_now, (size_t)nc));                  // :41
            }
        }
        FY_TRY(launch_ldu_reconstruct(stream, g, ssf.p, HbyA.p, rAU.p, U.p));                                     // :43-46
        FY_TRY(launch_ldu_pim_continuity(stream, g, pt.p, alpha.p, alpha.p, partials.p));             // :50
        double h[2];
        FY_TRY(reduce_read(nc, 2, nullptr, h));
        st.cont_err_sum_local = cs.dt * h[0] / total_volume; st.cont_err_global = cs.dt * h[1] / total_volume;
        cumulative += st.cont_err_global; st.cont_err_cumulative = cumulative;
        return FY_OK;
    }

    // pimpleFoamYade.C:60-114
    int step_pimple() {
        // The pre-coupling sweeps (1.3 ms at 4.1 M cells, bandwidth bound) on a side stream, beside the coupling's tree walk and deposit (4.3 + 1.2 ms, latency bound,
        // no fluid field read): the coupling waits for them where it first gathers a fluid field (Coupling::run_batch, pack_records: fields_event), as on z-slabs
        static const bool pre_beside = getenv("FOAMYADE_LDU_PRE_SERIAL") == nullptr;      // (A/B switch)
        const bool beside = pre_beside && cpl->c.gaussian;
        if (beside && !side) {
            FY_HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
            FY_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)); FY_HIP(hipEventCreateWithFlags(&ev_fields, hipEventDisableTiming));
        }
        hipStream_t ps = beside ? side : stream;
        if (beside) { FY_HIP(hipEventRecord(ev_fork, stream)); FY_HIP(hipStreamWaitEvent(side, ev_fork, 0)); }
        FY_TRY(launch_ldu_alphaf(ps, g, alpha.p, alphaf.p));
        FY_TRY(launch_ldu_pre_coupling(ps, g, phi.p, U.p, vGrad.p, alphaf.p, ddtU.p, divT.p));                  // :73, :75 (vGrad: :76, by the caller)
        FY_TRY(launch_ldu_grad_scalar(ps, g, p.p, gradP.p));                                                      // :74
        if (beside) { FY_HIP(hipEventRecord(ev_fields, side)); cpl->c.slab.fields_event = ev_fields; }
        tim[0].start(stream);
        FY_TRY(cpl->c.set_particle_action(cs.dt));                                                                // :78
        tim[0].stop(stream);
        if (beside) FY_HIP(hipStreamWaitEvent(stream, ev_fields, 0));                                             // (whatever the coupling did: the sweeps below read alphaf)
        FY_TRY(launch_ldu_alphaf(stream, g, alpha.p, alphaf.p));                                                  // :83-85
        if (ext_source) FY_TRY(launch_add_f64(stream, uSource.p, uSourceExt.p, 3 * (size_t)nc));
        const int nOuter = std::max(cs.n_outer_correctors, 1);
        for (int outer = 0; outer < nOuter; ++outer) {                                                             // :90
            const bool final_outer = outer == nOuter - 1;
            const double u_relax_now = (final_outer && cs.u_relax_final > 0) ? cs.u_relax_final : cs.u_relax;
            const double p_relax_now = (final_outer && cs.p_relax_final > 0) ? cs.p_relax_final : cs.p_relax;
            if (p_relax_now > 0 && p_relax_now < 1) FY_TRY(launch_copy_f64(stream, pPrev.p, p.p, (size_t)nc));  // storePrevIterFields()
            if (outer > 0) FY_TRY(launch_ldu_grad_vec(stream, g, U.p, vGrad.p));
            if (g.gradL) FY_TRY(launch_ldu_grad_magsqr(stream, g, U.p, gradL.p));
            FY_TRY(launch_ldu_assemble_momentum_pimple(stream, g, P(), phi.p, Uold.p, U.p, vGrad.p, M(), pt.p, fstress.p, u_relax_now, rAU.p));      // UcEqn.H:3-13
            FY_TRY(launch_ldu_forces(stream, g, P(), rAU.p, rAUf.p, phiForces.p));                               // UcEqn.H:15-20
            if (cs.momentum_predictor) {                                                                          // UcEqn.H:22-33
                // (first outer iteration: p and its patches' gradients stand as when the coupling's gradP was formed, pimpleFoamYade.C:74 -- the same field, not formed again)
                if (outer > 0) FY_TRY(launch_ldu_grad_scalar(stream, g, p.p, gradp.p));
                FY_TRY(launch_ldu_ssf_predictor(stream, g, phiForces.p, rAUf.p, p.p, outer > 0 ? gradp.p : gradP.p, ssf.p));
                FY_TRY(launch_ldu_reconstruct(stream, g, ssf.p, mb.p, d_V.p, bmom.p));
                int it = 0;
                FY_TRY(solve_momentum(&it, bmom.p, nullptr));
                st.u_iters_total += it;
            }
            for (int corr = 0; corr < cs.n_correctors; ++corr) FY_TRY(corrector_pimple(final_outer && corr == cs.n_correctors - 1, p_relax_now));
            if (les && final_outer) {                                                                             // pimple.turbCorr(): pimpleFoamYade.C:101-104
                FY_TRY(launch_ldu_grad_vec(stream, g, U.p, vGrad.p));
                if (keqn || keps) {                                                                               // kEqn::correct() / kEpsilon::correct(): epsilon first, then k with the new epsilon
                    LduMom Mk = M();
                    Mk.bdiag = nullptr;
                    const int modes[2] = {keps ? 1 : 0, 2};                                                       // (LduKEqn::mode) kEqn: k; kEpsilon: epsilon, then k
                    for (int mi = 0; mi < (keps ? 2 : 1); ++mi) {
                        const int mode = modes[mi];
                        LduKEqn K{};
                        K.mode = mode; K.ce = cs.les_ce; K.c1 = cs.ras_c1; K.c2 = cs.ras_c2; K.c3 = cs.ras_c3;
                        double tol, rel; int maxit;
                        if (mode == 1) { K.relax = cs.eps_relax; K.upwind = cs.eps_convection_scheme == FY_CONVECTION_UPWIND; K.sigma = cs.ras_sigmaeps; K.X = epsturb.p; K.x_bc = d_epsbc.p; K.x_val = d_epsval.p; tol = cs.eps_tol; rel = cs.eps_rel_tol; maxit = cs.eps_max_iter; }
                        else { K.relax = cs.k_relax; K.upwind = cs.k_convection_scheme == FY_CONVECTION_UPWIND; K.sigma = mode == 0 ? 1.0 : cs.ras_sigmak; K.X = kturb.p; K.x_bc = d_kbc.p; K.x_val = d_kval.p; tol = cs.k_tol; rel = cs.k_rel_tol; maxit = cs.k_max_iter; }
                        FY_TRY(launch_ldu_grad_k(stream, g, P(), K, gradk.p));
                        FY_TRY(launch_ldu_k_assemble(stream, g, P(), K, phi.p, vGrad.p, gradk.p, Mk, fcorr.p, HbyA.p));
                        int it = 0;
                        FY_TRY(solve_vec3(&it, Mk, mb.p, nullptr, &HbyA.p, &xscr.p, tol, rel, maxit));
                        k_iters_total += it;
                        FY_TRY(launch_ldu_k_bound_nut(stream, g, P(), K, HbyA.p, mode == 1 ? epsturb.p : kturb.p, nut.p));
                    }
                    nut_live = true;
                } else
                FY_TRY(launch_ldu_smagorinsky_nut(stream, g, vGrad.p, cs.les_ck, cs.les_ce, cs.les_delta_coeff, nut.p));
            }
        }
        return FY_OK;
    }

    int step() {
        FY_HIP(hipSetDevice(device));
        st = fy_step_stats{}; st.cont_err_cumulative = cumulative; st.delta_t = cs.dt;
        if (sources_pending) { FY_TRY(cpl->c.set_source_zero()); sources_pending = false; }      // the previous step's deferred setSourceZero
        tim[1].start(stream);
        if (!ops_courant.p) { FY_TRY(ops_courant.alloc_exact(2)); const int o[2] = {1, 0}; FY_HIP(hipMemcpyAsync(ops_courant.p, o, sizeof(o), hipMemcpyHostToDevice, stream)); FY_HIP(hipStreamSynchronize(stream)); }
        double h[2];
        FY_TRY(launch_ldu_courant(stream, g, phi.p, partials.p));                                                   // icoFoamYade.C:68
        FY_TRY(reduce_read(nc, 2, ops_courant.p, h));
        st.courant_max = 0.5 * h[0] * cs.dt; st.courant_mean = 0.5 * (h[1] / total_volume) * cs.dt;
        if (cs.adjust_time_step) {                                                                                  // setDeltaT.H [OF-6] (pimpleFoamYade.C:64): Co from the current flux at the OLD step
            const double maxDeltaTFact = cs.max_co / (st.courant_max + 1e-15);
            const double deltaTFact = std::min(std::min(maxDeltaTFact, 1.0 + 0.1 * maxDeltaTFact), 1.2);
            cs.dt = std::min(deltaTFact * cs.dt, cs.max_delta_t);
            g.dt = cs.dt;
        }
        st.delta_t = cs.dt;
        FY_HIP(hipMemcpyAsync(Uold.p, U.p, 3 * (size_t)nc * sizeof(double), hipMemcpyDeviceToDevice, stream));    // runTime++: old-time fields
        FY_HIP(hipMemcpyAsync(phiOld.p, phi.p, (size_t)nf * sizeof(double), hipMemcpyDeviceToDevice, stream));
        FY_TRY(launch_ldu_grad_vec(stream, g, U.p, vGrad.p));                                                      // :71
        if (pimple) FY_TRY(step_pimple());
        else {
            tim[0].start(stream);
            FY_TRY(cpl->c.set_particle_action(cs.dt));                                                                // :74
            tim[0].stop(stream);
            // the momentum source: what the coupling left (+ an external one, fy_ldu_solver_write_field_host("uSource", ...))
            const double* src = uSource.p;
            if (ext_source) { FY_TRY(launch_copy_f64(stream, uSourceSum.p, uSource.p, 3 * (size_t)nc)); FY_TRY(launch_add_f64(stream, uSourceSum.p, uSourceExt.p, 3 * (size_t)nc)); src = uSourceSum.p; }
            if (g.gradL) FY_TRY(launch_ldu_grad_magsqr(stream, g, Uold.p, gradL.p));
            FY_TRY(launch_ldu_assemble_momentum(stream, g, phi.p, Uold.p, src, vGrad.p, M(), fcorr.p));                // :79-85 (grad U of the iterate it is assembled from = vGrad)
            if (cs.momentum_predictor) {
                FY_TRY(launch_ldu_grad_scalar(stream, g, p.p, gradp.p));
                int it = 0;
                FY_TRY(solve_momentum(&it, mb.p, gradp.p));                                                          // :91-94
                st.u_iters_total += it;
            }
            for (int corr = 0; corr < cs.n_correctors; ++corr) FY_TRY(corrector(corr == cs.n_correctors - 1));         // :97-140
        }
        if (hold_sources) sources_pending = true;                                                                 // runTime.write() comes before setSourceZero (icoFoamYade.C:142-147)
        else FY_TRY(cpl->c.set_source_zero());                                                                    // :147
        tim[1].stop(stream);
        FY_HIP(hipStreamSynchronize(stream));
        if (need_ref) {
            int e = 0;
            FY_HIP(hipMemcpy(&e, adj_err.p, sizeof(int), hipMemcpyDeviceToHost));
            if (e) return fail(FY_ERR_UNSUPPORTED, "adjustPhi: continuity error cannot be removed by adjusting the outflow -- OpenFOAM stops here too");
        }
        st.ms_particle = tim[0].ms(); st.ms_total = tim[1].ms();
        return FY_OK;
    }

    int field(const char* name, double** ptr, size_t* count, const std::vector<double>** host) {
        const std::string s = name ? name : "";
        *host = nullptr;
        const size_t n = (size_t)nc;
        struct E { const char* nm; double* p; size_t c; };
        const E tab[] = {{"U", U.p, 3 * n}, {"p", p.p, n}, {"phi", phi.p, (size_t)nf}, {"uSource", uSourceExt.p, 3 * n}, {"rAU", rAU.p, n}, {"HbyA", HbyA.p, 3 * n},
                         {"phiHbyA", phiHbyA.p, (size_t)nf}, {"p_diag", pdiag.p, n}, {"p_coef", pcoef.p, (size_t)nf}, {"p_rhs", prhs.p, n}, {"vGrad", vGrad.p, 9 * n},
                         {"mom_diag", mdiag.p, n}, {"mom_lower", mlower.p, (size_t)ni}, {"mom_upper", mupper.p, (size_t)ni}, {"mom_b", mb.p, 3 * n},
                         {"alpha", alpha.p, pimple ? n : 0}, {"uSourceDrag", uSourceDrag.p, pimple ? n : 0}, {"uParticle", uParticle.p, pimple ? 3 * n : 0}, {"gradP", gradP.p, pimple ? 3 * n : 0},
                         {"divT", divT.p, pimple ? 3 * n : 0}, {"ddtU", ddtU.p, pimple ? 3 * n : 0}, {"phiForces", phiForces.p, pimple ? (size_t)nf : 0}, {"alphaf", alphaf.p, pimple ? (size_t)nf : 0},
                         {"rAUf", rAUf.p, (size_t)nf}, {"uSourceCoupling", uSource.p, 3 * n}, {"nut", nut.p, les ? n : 0}, {"k", kturb.p, (keqn || keps) ? n : 0}, {"epsilon", epsturb.p, keps ? n : 0}};
        for (const E& e : tab) if (s == e.nm) { *ptr = e.p; *count = e.c; return FY_OK; }
        const struct { const char* nm; const std::vector<double>* v; } geo[] = {{"C", &hm.C}, {"V", &hm.V}, {"Cf", &hm.Cf}, {"Sf", &hm.Sf}, {"magSf", &hm.magSf}, {"w", &hm.w},
                                                                                  {"dcNO", &hm.dcNO}, {"kvec", &hm.kvec}, {"sep", &hm.sep}, {"orig_face", &orig_face_d}};
        for (const auto& e : geo) if (s == e.nm) { *host = e.v; *count = e.v->size(); *ptr = nullptr; return FY_OK; }
        return fail(FY_ERR_INVALID, "unknown fy_ldu_solver field '%s'", s.c_str());
    }
};

}  // namespace fy

struct fy_ldu_solver { fy::LduSolver s; };

extern "C" {

void fy_ldu_case_defaults(fy_ldu_case* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->dt = 0.005; c->nu = 0.01; c->rho_fluid = 1000.0; c->rho_particle = 2650.0;
    c->n_correctors = 2; c->n_non_orth_correctors = 0; c->momentum_predictor = 1; c->p_ref_cell = 0; c->p_ref_value = 0.0;
    c->p_tol = 1e-6; c->p_rel_tol = 0.05; c->p_final_tol = 1e-6; c->p_final_rel_tol = 0.0; c->p_max_iter = 1000;
    c->u_tol = 1e-5; c->u_rel_tol = 0.0; c->u_max_iter = 1000;
    c->p_solver = FY_PSOLVER_PCG_JACOBI;
    c->solver = FY_SOLVER_ICO; c->n_outer_correctors = 1;
    c->adjust_time_step = 0; c->max_co = 1.0; c->max_delta_t = 1e300;
    c->turbulence_model = FY_TURBULENCE_LAMINAR; c->les_ck = 0.094; c->les_ce = 1.048; c->les_delta_coeff = 1.0; c->nut_initial = 0.0;
    c->convection_scheme = FY_CONVECTION_LINEAR; c->convection_limiter_k = 1.0;
    c->ras_cmu = 0.09; c->ras_c1 = 1.44; c->ras_c2 = 1.92; c->ras_c3 = 0.0; c->ras_sigmak = 1.0; c->ras_sigmaeps = 1.3;
    c->eps_initial = 0.0; c->eps_bc = nullptr; c->eps_value = nullptr; c->eps_convection_scheme = FY_CONVECTION_LINEAR; c->eps_tol = 1e-6; c->eps_rel_tol = 0.0; c->eps_max_iter = 1000; c->eps_relax = 0.0;
    c->k_initial = 0.0; c->k_bc = nullptr; c->k_value = nullptr; c->k_convection_scheme = FY_CONVECTION_LINEAR; c->k_tol = 1e-6; c->k_rel_tol = 0.0; c->k_max_iter = 1000; c->k_relax = 0.0;
}

int fy_ldu_solver_create(const fy_poly_mesh* m, const fy_ldu_case* c, const fy_transport* tr, int device_ordinal, fy_ldu_solver** out) {
    if (!out) return fy::fail(FY_ERR_INVALID, "null out");
    *out = nullptr;
    fy_ldu_solver* s = new (std::nothrow) fy_ldu_solver();
    if (!s) return fy::fail(FY_ERR_INVALID, "out of host memory");
    const int rc = s->s.create(m, c, tr, device_ordinal);
    if (rc != FY_OK) { delete s; return rc; }
    *out = s;
    return FY_OK;
}
#define FY_LS(s) if (!(s)) return fy::fail(FY_ERR_INVALID, "null fy_ldu_solver")
int fy_ldu_solver_step(fy_ldu_solver* s) { FY_LS(s); return s->s.step(); }
int fy_ldu_solver_get_stats(fy_ldu_solver* s, fy_step_stats* out) { FY_LS(s); if (!out) return fy::fail(FY_ERR_INVALID, "null out"); *out = s->s.st; return FY_OK; }
fy_ctx* fy_ldu_solver_coupling(fy_ldu_solver* s) { return s ? s->s.cpl : nullptr; }
int fy_ldu_solver_field_count(fy_ldu_solver* s, const char* name, int64_t* count) {
    FY_LS(s);
    if (!count) return fy::fail(FY_ERR_INVALID, "null count");
    double* p; size_t n; const std::vector<double>* h;
    FY_TRY(s->s.field(name, &p, &n, &h));
    *count = (int64_t)n;
    return FY_OK;
}
int fy_ldu_solver_read_field_host(fy_ldu_solver* s, const char* name, double* out) {
    FY_LS(s);
    double* p; size_t n; const std::vector<double>* h;
    FY_TRY(s->s.field(name, &p, &n, &h));
    if (h) { std::memcpy(out, h->data(), n * sizeof(double)); return FY_OK; }
    FY_HIP(hipSetDevice(s->s.device));
    FY_HIP(hipMemcpyAsync(out, p, n * sizeof(double), hipMemcpyDeviceToHost, s->s.stream));
    FY_HIP(hipStreamSynchronize(s->s.stream));
    return FY_OK;
}
int fy_ldu_solver_write_field_host(fy_ldu_solver* s, const char* name, const double* in) {
    FY_LS(s);
    double* p; size_t n; const std::vector<double>* h;
    FY_TRY(s->s.field(name, &p, &n, &h));
    const std::string nm = name;
    const bool pim_in = (s->s.pimple && (nm == "alpha" || nm == "uSourceDrag")) || (s->s.les && nm == "nut") || ((s->s.keqn || s->s.keps) && nm == "k") || (s->s.keps && nm == "epsilon");      // (what setParticleAction would leave: for tests that feed the equations a given void fraction)
    if (h || (nm != "U" && nm != "p" && nm != "uSource" && !pim_in)) return fy::fail(FY_ERR_INVALID, "fy_ldu_solver_write_field_host: '%s' cannot be written (U, p, uSource; alpha, uSourceDrag with pimpleFoamYade)", nm.c_str());
    FY_HIP(hipSetDevice(s->s.device));
    FY_HIP(hipMemcpyAsync(p, in, n * sizeof(double), hipMemcpyHostToDevice, s->s.stream));
    if (nm == "uSource") s->s.ext_source = true;
    if (nm == "U") FY_TRY(fy::launch_ldu_flux_of(s->s.stream, s->s.g, s->s.U.p, s->s.phi.p));      // createPhi
    FY_HIP(hipStreamSynchronize(s->s.stream));
    return FY_OK;
}
int fy_ldu_solver_apply(fy_ldu_solver* s, const char* op, const double* in, double* out) {
    FY_LS(s);
    if (!op || !in || !out) return fy::fail(FY_ERR_INVALID, "fy_ldu_solver_apply: null argument");
    fy::LduSolver& S = s->s;
    const std::string o = op;
    const bool mat = o == "p_matrix", pre = o == "p_precondition";
    if (!mat && !pre) return fy::fail(FY_ERR_INVALID, "fy_ldu_solver_apply: unknown operator '%s' (p_matrix, p_precondition)", op);
    if (!S.amg.diag0_) return fy::fail(FY_ERR_INVALID, "fy_ldu_solver_apply: no pressure matrix yet (step first)");
    FY_HIP(hipSetDevice(S.device));
    const size_t bytes = (size_t)S.nc * sizeof(double);
    FY_HIP(hipMemcpyAsync(S.pr.p, in, bytes, hipMemcpyHostToDevice, S.stream));
    if (mat) {
        fy::EllMat A = S.amg.lev[0]->mat();
        A.diag = S.pdiag.p;
        FY_TRY(fy::launch_ell_apply_dot(S.stream, A, S.pr.p, S.pr.p, S.pw.p, S.partials.p));
    } else if (S.amg.has_hierarchy()) FY_TRY(S.amg.vcycle(S.stream, S.pr.p, S.pw.p));
    else FY_TRY(fy::launch_ell_jacobi(S.stream, S.nc, S.pdiag.p, S.pr.p, S.pw.p));
    FY_HIP(hipMemcpyAsync(out, S.pw.p, bytes, hipMemcpyDeviceToHost, S.stream));
    FY_HIP(hipStreamSynchronize(S.stream));
    return FY_OK;
}
int fy_ldu_solver_hold_sources(fy_ldu_solver* s, int on) { FY_LS(s); s->s.hold_sources = on != 0; return FY_OK; }
int fy_ldu_solver_mg_levels(fy_ldu_solver* s, int cap, int32_t* cells, int32_t* slots, int* n_levels) {
    FY_LS(s);
    if (!n_levels) return fy::fail(FY_ERR_INVALID, "fy_ldu_solver_mg_levels: null n_levels");
    const fy::LduAmg& A = s->s.amg;
    *n_levels = A.has_hierarchy() ? (int)A.lev.size() : 0;
    for (int l = 0; l < *n_levels && l < cap; ++l) { if (cells) cells[l] = A.lev[(size_t)l]->n; if (slots) slots[l] = A.lev[(size_t)l]->W; }
    return FY_OK;
}
int fy_ldu_solver_destroy(fy_ldu_solver* s) { delete s; return FY_OK; }

}  // extern "C"
