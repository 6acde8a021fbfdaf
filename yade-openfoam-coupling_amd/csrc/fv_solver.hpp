// fy::Solver: the state and the member functions behind fy_solver (see fv_solver.cpp for the overview).  Definitions live in three units:
//   fv_solver.cpp    create, the reductions, the momentum predictor, one PISO / PIMPLE corrector, the step (the loop bodies of the reference)
//   fv_pressure.cpp  the pressure solver: PCG, the multigrid V-cycle that preconditions it, the coarse operators
//   fv_solver_api.cpp the extern "C" entry points of include/foamyade_hip.h
#pragma once
#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <functional>
#include <vector>

#include "comm.hpp"
#include "coupling.hpp"
#include "fv_kernels.hpp"

// geometry-dependent launchers exist per geometry model (fv_kernels.hpp): the uniform block's in fy, the graded block's in fy::gr
#define FVK(fn, ...) (g.graded ? ::fy::gr::fn(__VA_ARGS__) : ::fy::fn(__VA_ARGS__))

namespace fy {

// coarsest multigrid level: solved by damped-Jacobi sweeps inside one 1024-thread workgroup (every level below ~20^3 is
// launch-bound on the GPU; stopping at <= 1024 cells was tried and LOST: 40 sweeps no longer solve that level, +20 % PCG iterations)
constexpr int kMgCoarsest = kMgDirectMax, kMgCoarsestEdge = 8;      // the coarsest level (<= 128 cells, no edge over 8: band <= 64) is solved exactly from its banded Cholesky factor
constexpr int kMgDeepGhost = 5;               // ghost planes of a distributed level that runs the communication-avoiding V-cycle (fv_pressure.cpp: E + 3 <= 5)
constexpr int kMgReplicateBelow = 1 << 20;    // a distributed hierarchy hands over to the replicated one at <= this many GLOBAL cells: every
                                              // distributed level costs 4 neighbour exchanges per V-cycle, a replicated 1 M-cell level ~15 us per kernel

struct MgLev {
    PMat A{};
    bool distributed = false;     // owned z-slab + gz ghost planes per side (only when comm->size > 1)
    int gz = 0;
    size_t plane = 0;
    DevBuf<double> diag, ux, uy, uz, x0, x1, b;
    double* xcur = nullptr;       // holds the level's current iterate
    double* xalt = nullptr;
    const double* bptr = nullptr;
};

struct Solver {
    fy_case_desc cs{};
    std::vector<double> h_host[3];       // graded block: cell sizes per axis (host copy) and their device arrays (FvGeo::h)
    DevBuf<double> d_h[3];
    double total_volume = 0.0;
    FvGeo g{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;     // halo exchanges that overlap interior stencil work run here (slab mode)
    hipEvent_t ev_ready = nullptr, ev_halo = nullptr;
    hipEvent_t ev_fields = nullptr;      // behind the exchange of gradP / divT ghost planes when it runs on the second channel (step())
    // single domain: the coarse operators (k_mg_coarsen per level, the reference term, the coarsest level's Cholesky factor -- small, launch- and
    // latency-bound kernels, ~110 us in a row) are built on comm_stream while the main stream starts the solve on level 0, which needs none of them
    // (initial residual, pre-smoothing, restriction); the first touch of a coarse level waits for ev_coarse
    hipEvent_t ev_assembled = nullptr, ev_coarse = nullptr, ev_factor = nullptr;      // ev_factor: the coarsest level's Cholesky factor, which only the V-cycle's tail needs
    // single domain: the component sums of U that the momentum predictor's normFactor needs (k_sum3 + fold, 35 us) are formed on comm_stream at the top
    // of the step, beside the particle phase -- U does not change between there and the predictor -- into a partials buffer of their own
    hipEvent_t ev_usum0 = nullptr, ev_usum1 = nullptr;
    DevBuf<double> usum_partials;
    bool usum_pending = false;
    bool coarse_pending = false, factor_pending = false, coarse_on_side = false, coarse_marked = false;
    int wait_coarse() { if (coarse_pending) { coarse_pending = false; FY_HIP(hipStreamWaitEvent(stream, ev_coarse, 0)); } return FY_OK; }
    int wait_factor() { FY_TRY(wait_coarse()); if (factor_pending) { factor_pending = false; FY_HIP(hipStreamWaitEvent(stream, ev_factor, 0)); } return FY_OK; }
    fy_ctx* cpl = nullptr;
    bool pimple = false;
    int Nc = 0;                   // owned cells
    size_t nstore = 0, plane = 0; // storage cells (owned + ghost planes), cells per z-plane
    int64_t Nglob = 0;
    Comm* comm = nullptr;
    SelfComm self_comm;

    DevBuf<double> U, Uold, p, alpha, uSource, uSourceDrag, uParticle, gradP, divT, vGrad, ddtU;
    DevBuf<double> nut;                  // eddy viscosity (FY_TURBULENCE_SMAGORINSKY / _KEQN), storage cells; empty = laminar
    DevBuf<double> kturb, epsturb;       // turbulent kinetic energy k (FY_TURBULENCE_KEQN, _KEPSILON), dissipation rate epsilon (_KEPSILON)
    TurbEqn eq_k{}, eq_eps{};
    double les_delta = 0.0;              // LESdelta cubeRootVol: deltaCoeff * cbrt(V) [OF-6 cubeRootVolDelta.C]
    bool phi_fresh = true;      // phi holds the current flux (false between the start-of-step exchange with phiOld and the first flux correction)
    CFace3 phi_now() const { return phi_fresh ? CFace3{{phi[0].p, phi[1].p, phi[2].p}} : CFace3{{phiOld[0].p, phiOld[1].p, phiOld[2].p}}; }
    bool rAU_new = true;        // rAU was (re)assembled since the last corrector: rAUf and the coarse pressure operators are stale
    DevBuf<double> phi[3], phiOld[3], psn[3], alphaf[3], phiHbyA[3], phiForces[3], rAUf[3], pflux[3], ddtc[3];
    DevBuf<double> dcorr[3];             // old-time part of ddtCorr(U, phi) per face, written by the step's opening sweep (k_pre_coupling), read by k_corr_front
    DevBuf<double> gradL;                // [3 nstore] grad(magSqr(U)) for the limited convection schemes
    DevBuf<double> mbd;                  // [3 nstore] per-component boundary diagonal of the momentum matrix (Mom7::bd): only with a slip patch
    DevBuf<double> mdiag, man[6], src, rAU, HbyA, bmom, Gt, divG, xscr;
    std::vector<std::unique_ptr<MgLev> > mg;
    size_t mg_rep = 0;            // first replicated level (== mg.size() when nothing is replicated)
    DevBuf<double> rep_stage;     // local slice of the first replicated level before the all-gather (rhs / operator arrays)
    DevBuf<double> prhs, pr, pw, pp, ps, pzj;      // PCG: right-hand side, residual, w = A u, search direction p, its image s = A p, Jacobi-preconditioned residual
    DevBuf<double> rep_gather;   // landing zone of the coarse operators' all-gather (build_coarse_operators)
    bool mg_deep = false;        // every distributed level carries kMgDeepGhost ghost planes: one exchange per level and V-cycle (vcycle_deep)
    // p's ghost planes equal the neighbours' owned planes until somebody writes p: the exchanges in between are skipped
    bool p_ghosts_fresh = false;
    bool U_ghosts_fresh = false; // the same for U (one plane)
    int halo_U(hipStream_t on = nullptr) { if (comm->size > 1 && !U_ghosts_fresh) { FY_TRY(halo_cells(U, 3, 1, on)); U_ghosts_fresh = true; } return FY_OK; }
    int halo_p(hipStream_t on = nullptr) { if (comm->size > 1 && !p_ghosts_fresh) { FY_TRY(halo_cells(p, 1, 1, on)); p_ghosts_fresh = true; } return FY_OK; }
    DevBuf<double> pPrev;        // p.prevIter() (only with a field relaxation factor for p)
    double p_relax_now = 0.0;
    bool adjust_phi = false;     // adjustPhi can act (no fixed-pressure patch, and a patch that lets U float or prescribed through-flow)
    DevBuf<double> adj_sums;     // {massIn, fixedMassOut, adjustableMassOut, sum |internal flux|}
    DevBuf<int> adj_err;
    DevBuf<double> partials, red_out, sc, xbar3;
    bool hold_sources = false, sources_pending = false;
    bool overlap_halos = true;            // FOAMYADE_NO_HALO_OVERLAP=1: serial schedule (A/B switch, same results)
    double* red_host = nullptr;           // mapped pinned host memory (+ its device alias): reduce_read's landing zone (8 doubles) + deferred slots
    double* red_host_dev = nullptr;
    unsigned long long* red_flag = nullptr;      // 8 arrival flags (mapped pinned) the host spins on, and their device alias
    unsigned long long* red_flag_dev = nullptr;
    unsigned long long red_seq = 0;
    DevBuf<int> ops_courant, ops_diag;   // per-slot fold operations (0 sum, 1 max) of the Courant pair and of k_U_correct<true>'s four diagnostics
    fy_step_stats st{};
    double cumulative_cont_err = 0.0;
    EventTimer tim[4];      // particle, (unused), (unused), total
    KernelClock clk_mom, clk_pres;   // momentum / pressure phases: one event pair per outer iteration / corrector, read after the step's final
                                     // synchronisation (reading a phase time on the spot stalls the host until the phase has drained)
    bool timing = true;
    enum { KC_MG_SMOOTH0 = 0, KC_P_APPLY_DOT, KC_MOM_PASS, KC_COUNT };
    KernelClock kc[KC_COUNT];

    Face3 F3(DevBuf<double>* a) { Face3 f; for (int d = 0; d < 3; ++d) f.a[d] = a[d].p; return f; }
    CFace3 C3(DevBuf<double>* a) { CFace3 f; for (int d = 0; d < 3; ++d) f.a[d] = a[d].p; return f; }
    // with_bd: the momentum matrix (the k / epsilon equations re-use diag / an and have no per-component boundary diagonal)
    Mom7 M7(bool with_bd = true) { Mom7 m; m.diag = mdiag.p; for (int q = 0; q < 6; ++q) m.an[q] = man[q].p; m.bd = with_bd ? mbd.p : nullptr; return m; }

    ~Solver() {
        if (cpl) fy_destroy(cpl);
        for (auto& t : tim) t.destroy();
        for (auto& k : kc) k.destroy();
        for (auto& k : clk_xwait) k.destroy();
        clk_mom.destroy(); clk_pres.destroy();
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_halo) (void)hipEventDestroy(ev_halo);
        if (ev_fields) (void)hipEventDestroy(ev_fields);
        if (ev_assembled) (void)hipEventDestroy(ev_assembled);
        if (ev_usum0) (void)hipEventDestroy(ev_usum0);
        if (ev_usum1) (void)hipEventDestroy(ev_usum1);
        if (ev_coarse) (void)hipEventDestroy(ev_coarse);
        if (ev_factor) (void)hipEventDestroy(ev_factor);
        if (comm_stream) (void)hipStreamDestroy(comm_stream);
        if (red_host) (void)hipHostFree(red_host);
        if (red_flag) (void)hipHostFree(red_flag);
        if (stream) (void)hipStreamDestroy(stream);
    }

    int zero(DevBuf<double>& b) { if (b.n) FY_HIP(hipMemsetAsync(b.p, 0, b.n * sizeof(double), stream)); return FY_OK; }

    // ---- slab halos -----------------------------------------------------------------------------------------------------
    // refresh w ghost planes per side of a cell array with `ncomp` interleaved components (planes are contiguous in memory)
    int halo(double* f, int ncomp, size_t pl, int nzl, int gzl, int w, hipStream_t on = nullptr) {
        if (comm->size == 1) return FY_OK;
        if (!on) on = stream;
        const size_t P = pl * (size_t)ncomp;
        double* own_lo = f + (size_t)gzl * P;
        double* own_hi = f + (size_t)(gzl + nzl - w) * P;
        double* gh_lo = f + (size_t)(gzl - w) * P;
        double* gh_hi = f + (size_t)(gzl + nzl) * P;
        return comm->neighbour_exchange(on, own_hi, gh_lo, own_lo, gh_hi, (size_t)w * P);
    }
    int halo_cells(DevBuf<double>& f, int ncomp, int w, hipStream_t on = nullptr) { return halo(f.p, ncomp, plane, g.nz, g.gz, w, on); }

    // ---- a sweep beside the halo exchange it consumes (round 5; north_star: "RCCL halo exchange over xGMI overlapped with interior stencil work").
    // The sweep's interior planes -- whose stencils reach no ghost value -- start at once on `stream`, the exchange runs meanwhile on comm_stream (the
    // communicator's second channel), the two end planes follow when it has landed.  A plane window is a window of 256-cell BLOCKS (FvGeo::win_*): the
    // cells, their order inside a block and the slot a reducing kernel's partial sum goes to are those of the one-launch sweep, so fields and folds are
    // the serial schedule's bit for bit.  Needs planes that are whole numbers of blocks; otherwise (and with FOAMYADE_HALO_OVERLAP=0) exchange, then sweep.
    bool overlap_sweeps = true;
    bool can_window() const { return comm->size > 1 && overlap_halos && overlap_sweeps && plane % 256 == 0 && g.nz >= 3; }
    FvGeo window(int ka, int kb) const {
        FvGeo w = g;
        const int bp = (int)(plane / 256);
        w.red_stride = red_blocks(Nc);
        w.win_blk0 = ka * bp;
        w.win_nblk = (kb >= g.nz ? w.red_stride : kb * bp) - w.win_blk0;      // (the top window also runs the padding blocks of the reduction grid)
        return w;
    }
    // exposed wait per phase: how long `stream` sat waiting for an exchange (overlapped: from the end of the interior sweep to the ghosts' arrival;
    // serial: the exchange itself).  Sampled by event pairs when exchange timing is on (fy_solver_enable_exchange_timing): an event record idles the stream
    enum { XW_STEP = 0, XW_PARTICLE, XW_MOMENTUM, XW_CORRECTOR, XW_COUNT };
    KernelClock clk_xwait[XW_COUNT];
    bool xwait_timing = false;
    double xwait_ms[XW_COUNT] = {0, 0, 0, 0};
    template <class X, class L>
    int overlapped(int phase, bool needed, X&& xchg, L&& launch) {
        if (!needed || comm->size == 1) return launch(g);             // (ghosts still fresh, or no neighbours)
        KernelClock& ck = clk_xwait[phase];
        if (!can_window()) {
            ck.begin(stream);
            FY_TRY(xchg(stream));
            ck.end(stream);
            return launch(g);
        }
        FY_HIP(hipEventRecord(ev_ready, stream));                      // what the exchange sends is final
        FY_TRY(launch(window(1, g.nz - 1)));                           // (enqueued first: a host-synchronous back-end blocks in the exchange while this runs)
        FY_HIP(hipStreamWaitEvent(comm_stream, ev_ready, 0));
        FY_TRY(xchg(comm_stream));
        FY_HIP(hipEventRecord(ev_halo, comm_stream));
        ck.begin(stream);
        FY_HIP(hipStreamWaitEvent(stream, ev_halo, 0));
        ck.end(stream);
        FY_TRY(launch(window(0, 1)));
        return launch(window(g.nz - 1, g.nz));
    }
    int halo_level(MgLev& L, double* x) { return L.distributed ? halo(x, 1, L.plane, L.A.nz, L.gz, 1) : FY_OK; }

    int create(const fy_case_desc* c, const fy_transport* tr, int dev, Comm* cm);

    // ---- reductions: fold the block partials, all-reduce over the slabs, read back
    int reduce_read(int nslots, bool courant, double* h);
    // Diagnostics nobody branches on (Courant number, continuity errors): same fold [+ all-reduce], but the values land in their own
    // slots of the pinned buffer and are read after the step's final synchronisation instead of stalling the stream here.
    // Returns false when there is no slot left (the caller then reads at once).
    static constexpr int kDeferBase = 8, kDeferMax = 2 + 4 * 255; // red_host: 8 immediate doubles + Courant + 255 correctors' continuity errors
    int n_deferred = 0;
    bool reduce_deferred(int nslots, bool courant, int* slot, int* rc, const int* ops = nullptr);      // ops: ops_diag.p (sum, sum, max, sum) or none
    int reduce_to_device(double* dst);          // one slot, stays on the device (PCG scalars)

    // ---- momentum predictor: Jacobi sweeps with lduMatrix-style L1 residual control (stand-in for smoothSolver)
    int solve_momentum(int* iters) { return solve_vec3(U, bmom.p, cs.u_tol, cs.u_rel_tol, cs.u_max_iter, iters, true); }
    int solve_vec3(DevBuf<double>& X, const double* rhs, double tol, double rel_tol, int max_iter, int* iters, bool momentum = false);

    // ---- pressure solver (fv_pressure.cpp): multigrid V(2,2) with Chebyshev-weighted Jacobi pairs as the PCG preconditioner
    int smooth(size_t l, MgLev& L, double w, bool with_dot = false);
    PMat slice_of(const MgLev& L, const MgLev& Cc) const;
    // The smoother: two sweeps as a pair are the degree-2 Chebyshev polynomial of D^-1 A on [1/3, 2] -- the high-frequency band of the
    // 7-point operator under 2 x 2 x 2 coarsening; 2 bounds the spectrum of every diagonally dominant level -- i.e. Jacobi with the
    // weights 1 / (7/6 -+ (5/6) cos(pi/4)).  The pair damps that band by 0.34 where two sweeps at the fixed weight 0.8 reach 0.54, for
    // the same memory traffic: PCG iterations 66 -> 48 on a moving 48^3 bed, 19 -> 14 on the manufactured Poisson problem (CPU oracle).
    // Pre-smoothing applies (wa, wb), post-smoothing the reverse order (its adjoint: the V-cycle stays a symmetric positive definite
    // preconditioner).  The sweeps of the coarsest level (kMgCoarseSweeps) keep the fixed weight.  Degree 4 (four sweeps each way, weights 2.520 /
    // 1.180 / 0.673 / 0.516) was measured too: fewer iterations (C3 2.6 -> 2.1 per step, moving bed 5.75 -> 4.2, C2 7.5 -> 5.3) but every
    // one of the three cases slower in time (7.40 -> 7.56, 9.55 -> 9.73, 2.35 -> 2.51 ms per step): the four extra sweeps cost more
    // than the iterations they save.
    static constexpr double kMgWa = 1.7318685872766142, kMgWb = 0.5695012757370842;
    MgWeights mgw{2, {kMgWa, kMgWb, 0, 0}};
    int vcycle(size_t l);
    PMat planes_of(const MgLev& L, int ext, int* kofs = nullptr) const;
    int vcycle_deep(size_t l, int E);
    int exchange_operator_ghosts(MgLev& L);
    bool want_vcycle_dot = false, vcycle_dot_done = false;
    // damped-Jacobi sweeps that stand for the solve of the coarsest level (<= 256 cells, no edge over 8): 40 left its smoothest modes in and cost
    // PCG iterations -- C3 after 30 steps: 3.0 iterations per step with 40 or 80 sweeps, 2.17 with 120 / 160 / 240 (135 -> 140 / 139 / 137.6
    // steps/s); moving bed and C2 +-1 %.  The sweeps run inside the one-workgroup tail kernel, ~0.1 us each
    static constexpr int kMgCoarseSweeps = 120;
    int coarse_sweeps = kMgCoarseSweeps;
    bool fuse_prolong = true;            // prolongation folded into the first post-smoothing sweep (identical results)
    int ref_cell_at(size_t lvl, const PMat& A, int k0) const;
    int build_coarse_operators();
    DevBuf<double> mg_inv, mg_ref;
    double p_sum = 0.0;            // sum(p) over all cells as the last PCG update left it (solve_pressure)
    bool p_sum_valid = false;
    // init_done: r0 = b - A p and its two sums are already in pr / partials (the fused corrector sweep formed them: prepare_p_init + launch_corr_front)
    int solve_pressure(bool final_iter, bool init_done = false);      // OpenFOAM PCG.C with lduMatrix::solver::normFactor; preconditioner = MG V-cycle or Jacobi
    int prepare_p_init(bool with_halo = true);          // sum(p) for the norm factor's xbar where the last PCG update did not leave it, and p's ghost planes
    bool fused_corrector = true;   // the corrector as two fused sweeps (FOAMYADE_NO_FUSED_CORRECTOR=1: the five sweeps of rounds 1 - 4; identical results)
    bool hbya_ready = false;       // HbyA already holds rAU H(U) of the current U (written by the momentum predictor's last pass)
    bool face_arrays = true;       // rAUf / alphacf are kept as face arrays (somebody streams them: see create())
    bool faces_from_cells = true;  // the fused sweeps re-form rAUf / alphacf from rAU / alpha (FOAMYADE_FACES_FROM_ARRAYS=1: stream the face arrays)

    // ---- one PISO / PIMPLE corrector (icoFoamYade.C:97-140, pEqn.H)
    int corrector(bool final_inner);
    // Courant sums of the flux the last corrector left (max sumPhi/V, sum sumPhi), formed by k_U_correct<true>: what CourantNo.H at the top
    // of the next pass would compute from the same phi.  Dropped whenever a field is written from outside (fy_solver_write_field_host).
    bool fuse_diag = true;               // continuity errors and the next Courant sums ride on the velocity-correction sweep
    int carry_slot = -1;
    bool carry_valid = false;
    double carry_h[2] = {0, 0};

    std::vector<int> cont_slots;     // deferred continuity-error read-backs of this step, in corrector order
    int courant_slot = -1;
    void note_cont_err(const double* h) {      // continuityErrs.H:36-46
        const double tv = total_volume;
        st.cont_err_sum_local = cs.dt * h[0] / tv; st.cont_err_global = cs.dt * h[1] / tv;
        cumulative_cont_err += st.cont_err_global; st.cont_err_cumulative = cumulative_cont_err;
    }
    void note_courant(const double* h) { st.courant_max = 0.5 * h[0] * cs.dt; st.courant_mean = 0.5 * (h[1] / total_volume) * cs.dt; }
    int turbulence_correct();      // continuousPhaseTurbulence->correct()
    int st_k_iters = 0;

    int step();                    // one pass of the while (runTime.loop()) body
    // field lookup; cell fields are returned/accepted as the OWNED part only (ghost planes are an implementation detail)
    int field(const char* name, double** ptr, size_t* count);
};

}  // namespace fy

struct fy_solver { fy::Solver s; };
