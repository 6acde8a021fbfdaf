// The pressure solver of fy::Solver: OpenFOAM's PCG (PCG.C, lduMatrix::solver::normFactor) [OF-6] for the pEqn of icoFoamYade.C:118-125 /
// pEqn.H:26-35, preconditioned by a geometric multigrid V-cycle (2 x 2 x 2 aggregation, Chebyshev-weighted Jacobi pairs, exact coarsest solve).
#include "fv_solver.hpp"

namespace fy {

// ---- multigrid V(2,2) with damped Jacobi, used as the PCG preconditioner ------------------------------------------
int Solver::smooth(size_t l, MgLev& L, double w, bool with_dot) {
    if (L.distributed && overlap_halos && L.A.nz >= 4) {
        // Halo exchange overlapped with interior stencil work: the sweep over the planes that need no ghost values starts at
        // once on `stream`, the one-plane exchange of x runs meanwhile on comm_stream, and the two boundary planes are swept
        // when it has landed.  Same arithmetic per cell, so the result is the serial schedule's bit for bit.
        const int pl = (int)L.plane;
        FY_HIP(hipEventRecord(ev_ready, stream));                      // x is final
        FY_HIP(hipStreamWaitEvent(comm_stream, ev_ready, 0));
        PMat in = L.A; in.c0 += pl; in.N -= 2 * pl;
        FY_TRY(launch_mg_smooth(stream, in, L.bptr, L.xcur, L.xalt, w));
        FY_TRY(halo(L.xcur, 1, L.plane, L.A.nz, L.gz, 1, comm_stream));
        FY_HIP(hipEventRecord(ev_halo, comm_stream));
        FY_HIP(hipStreamWaitEvent(stream, ev_halo, 0));
        PMat lo = L.A; lo.N = pl;
        PMat hi = L.A; hi.c0 += (L.A.nz - 1) * pl; hi.N = pl;
        FY_TRY(launch_mg_smooth(stream, lo, L.bptr, L.xcur, L.xalt, w));
        FY_TRY(launch_mg_smooth(stream, hi, L.bptr, L.xcur, L.xalt, w));
        std::swap(L.xcur, L.xalt);
        return FY_OK;
    }
    FY_TRY(halo_level(L, L.xcur));
    if (with_dot) {                                    // (single domain only: the caller checks)
        if (l == 0) kc[KC_MG_SMOOTH0].begin(stream);
        FY_TRY(launch_mg_smooth_dot(stream, L.A, L.bptr, L.xcur, L.xalt, w, partials.p));
        if (l == 0) kc[KC_MG_SMOOTH0].end(stream);
        std::swap(L.xcur, L.xalt);
        return FY_OK;
    }
    if (l == 0) kc[KC_MG_SMOOTH0].begin(stream);
    FY_TRY(launch_mg_smooth(stream, L.A, L.bptr, L.xcur, L.xalt, w));
    if (l == 0) kc[KC_MG_SMOOTH0].end(stream);
    std::swap(L.xcur, L.xalt);
    return FY_OK;
}

// the slice of the (replicated) level `Cc` that this rank's distributed parent `L` aggregates to, as a stand-alone PMat
PMat Solver::slice_of(const MgLev& L, const MgLev& Cc) const {
    PMat loc = Cc.A;
    loc.nz = L.A.nz / 2; loc.N = (int)(Cc.plane * (size_t)loc.nz); loc.c0 = 0; loc.ntot = loc.N;
    return loc;
}

int Solver::vcycle(size_t l) {
    const double w = 0.8;
    const MgWeights& W = mgw;
    if (l >= 1) FY_TRY(wait_coarse());                        // (level 0's operator is the assembled one; everything below comes from build_coarse_operators)
    MgLev& L = *mg[l];
    if (!L.distributed && L.A.N <= kMgTailCells && mg.size() - l <= (size_t)kMgTailMax) {
        // the rest of the hierarchy fits one workgroup: one launch instead of ~8 per level (b of this level is already in place)
        FY_TRY(wait_coarse());
        PMat A[kMgTailMax]; double* x0[kMgTailMax]; double* x1[kMgTailMax]; double* b[kMgTailMax];
        const int n = (int)(mg.size() - l);
        for (int q = 0; q < n; ++q) {
            MgLev& M = *mg[l + (size_t)q];
            A[q] = M.A; x0[q] = M.x0.p; x1[q] = M.x1.p; b[q] = q == 0 ? const_cast<double*>(L.bptr) : M.b.p;
        }
        FY_TRY(launch_mg_tail(stream, A, x0, x1, b, n, w, coarse_sweeps, W, mg_inv.p));
        L.xcur = n > 1 ? L.x1.p : L.x0.p; L.xalt = n > 1 ? L.x0.p : L.x1.p;
        return FY_OK;
    }
    if (l + 1 == mg.size()) {
        FY_TRY(wait_coarse());
        FY_TRY(launch_mg_coarse_solve(stream, L.A, L.bptr, L.x0.p, L.x1.p, coarse_sweeps, w, mg_inv.p));
        L.xcur = L.x0.p; L.xalt = L.x1.p;
        return FY_OK;
    }
    MgLev& Cc = *mg[l + 1];
    if (!L.distributed) {
        // first iterate and first sweep in one pass (bit-identical, see the kernel): one launch fewer on the latency-bound small
        // levels, and on level 0 the first iterate never travels through memory (pressure 2.25 -> 2.21 ms)
        FY_TRY(launch_mg_smooth_two_from_zero(stream, L.A, L.bptr, L.xcur, W.w[0], W.w[1]));
    } else {
        FY_TRY(launch_mg_smooth_first(stream, L.A, L.bptr, L.xcur, W.w[0]));
        FY_TRY(smooth(l, L, W.w[1]));
    }
    for (int s = 2; s < W.n; ++s) FY_TRY(smooth(l, L, W.w[s]));
    FY_TRY(halo_level(L, L.xcur));
    const bool handover = L.distributed && !Cc.distributed;
    if (handover) {
        // hand-over to the replicated hierarchy: restrict into the local slice, all-gather the coarse right-hand side
        PMat loc = slice_of(L, Cc);
        FY_TRY(launch_mg_residual_restrict(stream, L.A, L.bptr, L.xcur, loc, rep_stage.p));
        FY_TRY(comm->allgather(stream, rep_stage.p, Cc.b.p, (size_t)loc.N));
    } else {
        FY_TRY(launch_mg_residual_restrict(stream, L.A, L.bptr, L.xcur, Cc.A, Cc.b.p));
    }
    Cc.bptr = Cc.b.p;
    FY_TRY(vcycle(l + 1));
    if (handover) {
        PMat loc = Cc.A;                      // this rank's slice of the replicated coarse solution
        loc.c0 = (int)(Cc.plane * (size_t)(L.A.nz / 2) * (size_t)comm->rank);
        FY_TRY(launch_mg_prolong_add(stream, L.A, L.xcur, loc, Cc.xcur));
        for (int s = W.n - 1; s >= 1; --s) FY_TRY(smooth(l, L, W.w[s]));
    } else if (!L.distributed && W.n >= 2 && fuse_prolong) {
        // prolongation fused into the first post-smoothing sweep (bit-identical; one launch and one pass over the level fewer)
        FY_TRY(launch_mg_smooth_prolong(stream, L.A, L.bptr, L.xcur, Cc.A, Cc.xcur, L.xalt, W.w[W.n - 1]));
        std::swap(L.xcur, L.xalt);
        for (int s = W.n - 2; s >= 1; --s) FY_TRY(smooth(l, L, W.w[s]));
    } else {
        FY_TRY(launch_mg_prolong_add(stream, L.A, L.xcur, Cc.A, Cc.xcur));
        for (int s = W.n - 1; s >= 1; --s) FY_TRY(smooth(l, L, W.w[s]));
    }
    // the last sweep of the whole cycle also leaves the partials of z.r where PCG's launch_dot would (vcycle_dot_done)
    vcycle_dot_done = l == 0 && want_vcycle_dot && !L.distributed && L.A.N == Nc && L.A.c0 == g.c0;
    FY_TRY(smooth(l, L, W.w[0], vcycle_dot_done));
    return FY_OK;
}

// coarse operators: A_{l+1} = 1/2 P^T A_l P level by level; the first replicated level is all-gathered from the slabs' slices
// index of the coarse cell that holds the pressure reference cell, local to the level-`lvl` operator `A` whose first plane is global
// coarse plane `k0` (-1: no reference cell, or not in A's planes)
int Solver::ref_cell_at(size_t lvl, const PMat& A, int k0) const {
    if (!g.need_ref) return -1;
    const int i = (g.p_ref_cell % g.nx) >> lvl, j = ((g.p_ref_cell / g.nx) % g.ny) >> lvl, k = (g.p_ref_cell / (g.nx * g.ny)) >> lvl;
    const int kl = k - k0;
    if (kl < 0 || kl >= A.nz) return -1;
    return i + A.nx * (j + A.ny * kl);
}
int Solver::build_coarse_operators() {
    // the reference cell's point term (k_mg_coarsen): level 0's value, known to every rank
    const double* ref_term = nullptr;
    if (g.need_ref && mg.size() > 1) {
        if (!mg_ref.p) FY_TRY(mg_ref.alloc_exact(1));
        FY_TRY(launch_mg_ref_term(stream, mg[0]->A, ref_cell_at(0, mg[0]->A, mg[0]->distributed ? g.kglob0 : 0), mg_ref.p));
        if (mg[0]->distributed) FY_TRY(comm->allreduce(stream, mg_ref.p, 1, false));
        ref_term = mg_ref.p;
    }
    for (size_t l = 0; l + 1 < mg.size(); ++l) {
        MgLev& F = *mg[l]; MgLev& Cc = *mg[l + 1];
        if (F.distributed && !Cc.distributed) {
            PMat loc = slice_of(F, Cc);
            const size_t cnt = (size_t)loc.N;
            loc.diag = rep_stage.p; loc.ux = rep_stage.p + cnt; loc.uy = rep_stage.p + 2 * cnt; loc.uz = rep_stage.p + 3 * cnt;
            FY_TRY(launch_mg_coarsen(stream, F.A, loc, ref_cell_at(l + 1, loc, comm->rank * loc.nz), ref_term));
            FY_TRY(comm->allgather(stream, loc.diag, Cc.A.diag, cnt));
            FY_TRY(comm->allgather(stream, loc.ux, Cc.A.ux, cnt));
            FY_TRY(comm->allgather(stream, loc.uy, Cc.A.uy, cnt));
            FY_TRY(comm->allgather(stream, loc.uz, Cc.A.uz, cnt));
        } else {
            FY_TRY(launch_mg_coarsen(stream, F.A, Cc.A, ref_cell_at(l + 1, Cc.A, Cc.distributed ? comm->rank * Cc.A.nz : 0), ref_term));
            if (Cc.distributed && comm->has_down()) FY_TRY(launch_mg_coarsen_ghost(stream, F.A, Cc.A));
        }
    }
    // the coarsest operator's banded Cholesky factor (k_mg_coarse_factor): rebuilt with the operators, used by every V-cycle until the next assembly
    if (cs.p_solver == FY_PSOLVER_PCG_MG) {
        const MgLev& Lc = *mg.back();
        if (!Lc.distributed && mg_coarse_direct_ok(Lc.A)) {
            if (!mg_inv.p) FY_TRY(mg_inv.alloc_exact((size_t)mg_coarse_factor_doubles(Lc.A)));
            FY_TRY(launch_mg_coarse_factor(stream, Lc.A, mg_inv.p));
        }
    }
    return FY_OK;
}

// OpenFOAM PCG.C with lduMatrix::solver::normFactor; preconditioner = MG V-cycle or Jacobi
int Solver::solve_pressure(bool final_iter) {
    MgLev& L = *mg[0];
    const double tol = final_iter ? cs.p_final_tol : cs.p_tol, rel = final_iter ? cs.p_final_rel_tol : cs.p_rel_tol;
    double h[2];
    // xbar = average(p) for lduMatrix::solver::normFactor: sum(p) over the owned cells, all-reduced.  The last PCG update of p left it with the
    // host (k_pcg_update_xr's second slot: the same partition and order as the sum below, the same bits); p_sum_valid falls when anything else writes p
    if (!p_sum_valid) {
        FY_TRY(launch_dot(stream, Nc, g.c0, p.p, nullptr, partials.p));
        FY_TRY(reduce_to_device(sc.p + 3));
    }
    FY_TRY(halo_cells(p, 1, 1));
    FY_TRY(launch_p_init(stream, L.A, prhs.p, p.p, p_sum_valid ? nullptr : sc.p + 3, p_sum, 1.0 / (double)Nglob, pr.p, partials.p));
    FY_TRY(reduce_read(2, false, h));
    const double norm = h[1] + 1e-20;
    double res = h[0] / norm;
    const double res0 = res;
    st.p_initial_residual = res0;
    auto converged = [&](double r) { return r < tol || (rel > 0 && r < rel * res0); };
    int it = 0;
    if (!converged(res)) {
        do {
            const double* z;
            vcycle_dot_done = false;
            if (cs.p_solver == FY_PSOLVER_PCG_MG) { L.bptr = pr.p; want_vcycle_dot = true; FY_TRY(vcycle(0)); want_vcycle_dot = false; z = L.xcur; }
            else { FY_TRY(launch_jacobi_precond(stream, L.A, pr.p, pzj.p)); z = pzj.p; }
            if (!vcycle_dot_done) FY_TRY(launch_dot(stream, Nc, g.c0, z, pr.p, partials.p));
            FY_TRY(reduce_to_device(sc.p + 0));                                                   // wArA
            FY_TRY(launch_pcg_update_p(stream, Nc, g.c0, z, pp.p, sc.p, it == 0 ? 1 : 0));
            FY_TRY(halo_cells(pp, 1, 1));
            kc[KC_P_APPLY_DOT].begin(stream);
            FY_TRY(launch_p_apply_dot(stream, L.A, pp.p, pw.p, partials.p));
            kc[KC_P_APPLY_DOT].end(stream);
            FY_TRY(reduce_to_device(sc.p + 2));                                                   // wApA
            FY_TRY(launch_pcg_update_xr(stream, Nc, g.c0, p.p, pr.p, pp.p, pw.p, sc.p, partials.p));
            FY_TRY(reduce_read(2, false, h));
            res = h[0] / norm;
            p_sum = h[1]; p_sum_valid = true;
        } while (++it < cs.p_max_iter && !converged(res));
    }
    st.p_final_residual = res;
    st.p_iters_total += it; st.p_solves += 1;
    return FY_OK;
}

}  // namespace fy
