// The pressure solver of fy::Solver: OpenFOAM's PCG (PCG.C, lduMatrix::solver::normFactor) [OF-6] for the pEqn of icoFoamYade.C:118-125 /
// pEqn.H:26-35, preconditioned by a geometric multigrid V-cycle (2 x 2 x 2 aggregation, Chebyshev-weighted Jacobi pairs, exact coarsest solve).
//
// Round 4: the loop is shaped for eight GPUs (hard part 4; 8e).
//   * PCG in its single-reduction (Chronopoulos-Gear) form: gamma = u.r and delta = u.Au come out of ONE fold + all-reduce per iteration
//     (k_p_apply_dot, k_pcg_cg_update); the residual norm the stopping rule reads is the only other one (PCG.C's loop: three).
//   * z-slabs whose levels carry >= 5 ghost planes run a communication-avoiding V-cycle (vcycle_deep): ONE neighbour exchange per distributed
//     level per cycle -- the right-hand side, E + 3 planes deep -- instead of one per sweep; every sweep then also covers the ghost planes its
//     output is still needed on.  The cycle's result arrives valid one plane into the ghosts, so the matrix-vector product that follows needs
//     no exchange either.  A PCG iteration at two slabs of 160^3: 1 exchange + 1 all-gather + 2 all-reduces (round 3: 5 + 1 + 3).
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

// the owned planes of a distributed level and `ext` ghost planes on each side that has a neighbour (a physical boundary has nothing beyond it)
PMat Solver::planes_of(const MgLev& L, int ext, int* kofs) const {
    const int lo = comm->has_down() ? -ext : 0, hi = L.A.nz + (comm->has_up() ? ext : 0);
    PMat R = L.A;
    R.c0 = L.A.c0 + lo * (int)L.plane;
    R.N = (hi - lo) * (int)L.plane;
    if (kofs) *kofs = lo;
    return R;
}

// The communication-avoiding V-cycle of a distributed level (every distributed level carries kMgDeepGhost ghost planes: Solver::mg_deep).
// Wanted: the level's correction valid on the owned planes and E ghost planes either side.  Working backwards through V(2,2): the last
// sweep writes +-E from an iterate on +-(E + 1), that one from +-(E + 2) -- where the prolongated correction of the coarser level is added,
// which therefore has to be valid ceil((E + 2) / 2) COARSE planes out -- and the two pre-smoothing sweeps (fused, the first iterate formed
// inline) write +-(E + 2) from the right-hand side on +-(E + 3).  So the right-hand side is exchanged E + 3 planes deep, once, and the
// ghost rows are computed here as the neighbour computes them: same operands, same operations, same bits.  The operator's ghost planes
// hold the neighbour's coefficients (exchange_operator_ghosts, once per assembly).
int Solver::vcycle_deep(size_t l, int E) {
    MgLev& L = *mg[l];
    MgLev& Cc = *mg[l + 1];
    const MgWeights& W = mgw;
    if (l >= 1) FY_TRY(wait_coarse());
    if (E + 3 > L.gz || E + 3 > L.A.nz) return fail(FY_ERR_INVALID, "vcycle_deep: level %zu has %d ghost planes, %d are needed", l, L.gz, E + 3);
    int kofs = 0;
    const PMat R2 = planes_of(L, E + 2, &kofs);
    if (overlap_halos && L.A.nz >= 4) {
        // the cycle's one exchange overlapped with interior stencil work: the fused pre-smoothing sweeps of the planes whose rows read owned
        // values only start at once, the right-hand side's ghost planes travel meanwhile on comm_stream (the communicator's second channel),
        // and the end planes + ghost planes are swept when they have landed -- the same rows, the same bits
        const int pl = (int)L.plane;
        FY_HIP(hipEventRecord(ev_ready, stream));                      // b is final
        FY_HIP(hipStreamWaitEvent(comm_stream, ev_ready, 0));
        FY_TRY(halo(const_cast<double*>(L.bptr), 1, L.plane, L.A.nz, L.gz, E + 3, comm_stream));
        FY_HIP(hipEventRecord(ev_halo, comm_stream));
        PMat in = L.A; in.c0 += pl; in.N -= 2 * pl;
        FY_TRY(launch_mg_smooth_two_from_zero(stream, in, L.bptr, L.xcur, W.w[0], W.w[1]));
        FY_HIP(hipStreamWaitEvent(stream, ev_halo, 0));
        PMat lo = R2; lo.N = (1 - kofs) * pl;                          // planes [kofs, 1)
        PMat hi = R2; hi.c0 = L.A.c0 + (L.A.nz - 1) * pl; hi.N = R2.N - (L.A.nz - 1 - kofs) * pl;      // planes [nz - 1, end)
        FY_TRY(launch_mg_smooth_two_from_zero(stream, lo, L.bptr, L.xcur, W.w[0], W.w[1]));
        FY_TRY(launch_mg_smooth_two_from_zero(stream, hi, L.bptr, L.xcur, W.w[0], W.w[1]));
    } else {
        FY_TRY(halo(const_cast<double*>(L.bptr), 1, L.plane, L.A.nz, L.gz, E + 3));
        if (l == 0) kc[KC_MG_SMOOTH0].begin(stream);
        FY_TRY(launch_mg_smooth_two_from_zero(stream, R2, L.bptr, L.xcur, W.w[0], W.w[1]));
        if (l == 0) kc[KC_MG_SMOOTH0].end(stream);
    }
    if (!Cc.distributed) {
        // hand-over to the replicated hierarchy: restrict into the local slice, all-gather the coarse right-hand side
        PMat loc = slice_of(L, Cc);
        FY_TRY(launch_mg_residual_restrict(stream, L.A, L.bptr, L.xcur, loc, rep_stage.p));
        FY_TRY(comm->allgather(stream, rep_stage.p, Cc.b.p, (size_t)loc.N));
        Cc.bptr = Cc.b.p;
        FY_TRY(vcycle(l + 1));
        PMat under = Cc.A;                    // c0: the coarse cell under my first owned cell; the replicated level holds every plane around it
        under.c0 = (int)(Cc.plane * (size_t)(L.A.nz / 2) * (size_t)comm->rank);
        FY_TRY(launch_mg_prolong_add_planes(stream, R2, kofs, L.xcur, under, Cc.xcur));
    } else {
        FY_TRY(launch_mg_residual_restrict(stream, L.A, L.bptr, L.xcur, Cc.A, Cc.b.p));
        Cc.bptr = Cc.b.p;
        FY_TRY(vcycle_deep(l + 1, (E + 3) / 2));
        FY_TRY(launch_mg_prolong_add_planes(stream, R2, kofs, L.xcur, Cc.A, Cc.xcur));
    }
    FY_TRY(launch_mg_smooth(stream, planes_of(L, E + 1), L.bptr, L.xcur, L.xalt, W.w[1]));
    std::swap(L.xcur, L.xalt);
    if (l == 0) kc[KC_MG_SMOOTH0].begin(stream);
    FY_TRY(launch_mg_smooth(stream, planes_of(L, E), L.bptr, L.xcur, L.xalt, W.w[0]));
    if (l == 0) kc[KC_MG_SMOOTH0].end(stream);
    std::swap(L.xcur, L.xalt);
    return FY_OK;
}

int Solver::vcycle(size_t l) {
    const double w = 0.8;
    const MgWeights& W = mgw;
    if (l >= 1) FY_TRY(wait_coarse());                        // (level 0's operator is the assembled one; everything below comes from build_coarse_operators)
    MgLev& L = *mg[l];
    if (!L.distributed && L.A.N <= kMgTailCells && mg.size() - l <= (size_t)kMgTailMax) {
        // the rest of the hierarchy fits one workgroup: one launch instead of ~8 per level (b of this level is already in place)
        FY_TRY(wait_factor());
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
        FY_TRY(wait_factor());
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
    // the last sweep of the whole cycle also leaves the partials of z.r where PCG's fold expects them (vcycle_dot_done)
    vcycle_dot_done = l == 0 && want_vcycle_dot && !L.distributed && L.A.N == Nc && L.A.c0 == g.c0;
    FY_TRY(smooth(l, L, W.w[0], vcycle_dot_done));
    return FY_OK;
}

// index of the coarse cell that holds the pressure reference cell, local to the level-`lvl` operator `A` whose first plane is global
// coarse plane `k0` (-1: no reference cell, or not in A's planes)
int Solver::ref_cell_at(size_t lvl, const PMat& A, int k0) const {
    if (!g.need_ref) return -1;
    const int i = (g.p_ref_cell % g.nx) >> lvl, j = ((g.p_ref_cell / g.nx) % g.ny) >> lvl, k = (g.p_ref_cell / (g.nx * g.ny)) >> lvl;
    const int kl = k - k0;
    if (kl < 0 || kl >= A.nz) return -1;
    return i + A.nx * (j + A.ny * kl);
}

// a distributed level's operator on its ghost planes = the neighbours' rows: all four arrays, every ghost plane, in one grouped exchange
int Solver::exchange_operator_ghosts(MgLev& L) {
    if (!L.distributed) return FY_OK;
    comm->group_begin();
    double* arr[4] = {L.A.diag, L.A.ux, L.A.uy, L.A.uz};
    for (double* a : arr) FY_TRY(halo(a, 1, L.plane, L.A.nz, L.gz, std::min(L.gz, L.A.nz)));
    return comm->group_end(stream);
}

// coarse operators: A_{l+1} = 1/2 P^T A_l P level by level; the first replicated level is all-gathered from the slabs' slices
int Solver::build_coarse_operators() {
    Comm::Tag tag(comm, "p_operators");
    // the reference cell's point term (k_mg_coarsen): level 0's value, known to every rank
    const double* ref_term = nullptr;
    if (g.need_ref && mg.size() > 1) {
        if (!mg_ref.p) FY_TRY(mg_ref.alloc_exact(1));
        FY_TRY(launch_mg_ref_term(stream, mg[0]->A, ref_cell_at(0, mg[0]->A, mg[0]->distributed ? g.kglob0 : 0), mg_ref.p));
        if (mg[0]->distributed) FY_TRY(comm->allreduce(stream, mg_ref.p, 1, false));
        ref_term = mg_ref.p;
    }
    if (mg_deep) FY_TRY(exchange_operator_ghosts(*mg[0]));
    for (size_t l = 0; l + 1 < mg.size(); ++l) {
        MgLev& F = *mg[l]; MgLev& Cc = *mg[l + 1];
        if (F.distributed && !Cc.distributed) {
            // the four arrays of this rank's slice go out in ONE all-gather ([diag | ux | uy | uz] per rank) and are put in place by a copy each
            PMat loc = slice_of(F, Cc);
            const size_t cnt = (size_t)loc.N;
            loc.diag = rep_stage.p; loc.ux = rep_stage.p + cnt; loc.uy = rep_stage.p + 2 * cnt; loc.uz = rep_stage.p + 3 * cnt;
            FY_TRY(launch_mg_coarsen(stream, F.A, loc, ref_cell_at(l + 1, loc, comm->rank * loc.nz), ref_term));
            if (rep_gather.n < 4 * cnt * (size_t)comm->size) FY_TRY(rep_gather.alloc_exact(4 * cnt * (size_t)comm->size));
            FY_TRY(comm->allgather(stream, rep_stage.p, rep_gather.p, 4 * cnt));
            double* dst[4] = {Cc.A.diag, Cc.A.ux, Cc.A.uy, Cc.A.uz};
            for (int r = 0; r < comm->size; ++r)
                for (int a = 0; a < 4; ++a)
                    FY_TRY(launch_copy_f64(stream, dst[a] + (size_t)r * cnt, rep_gather.p + ((size_t)r * 4 + (size_t)a) * cnt, cnt));
        } else {
            FY_TRY(launch_mg_coarsen(stream, F.A, Cc.A, ref_cell_at(l + 1, Cc.A, Cc.distributed ? comm->rank * Cc.A.nz : 0), ref_term));
            if (Cc.distributed) {
                if (mg_deep) FY_TRY(exchange_operator_ghosts(Cc));
                else if (comm->has_down()) FY_TRY(launch_mg_coarsen_ghost(stream, F.A, Cc.A));
            }
        }
    }
    // the coarsest operator's banded Cholesky factor (k_mg_coarse_factor): rebuilt with the operators, used by every V-cycle until the next assembly
    if (cs.p_solver == FY_PSOLVER_PCG_MG) {
        const MgLev& Lc = *mg.back();
        if (!Lc.distributed && mg_coarse_direct_ok(Lc.A)) {
            if (!mg_inv.p) FY_TRY(mg_inv.alloc_exact((size_t)mg_coarse_factor_doubles(Lc.A)));
            // on the side stream: the operators are complete here -- the V-cycle's way down may start; only its tail waits for the factor (70 us on one CU)
            if (coarse_on_side) { FY_HIP(hipEventRecord(ev_coarse, stream)); coarse_marked = true; }
            FY_TRY(launch_mg_coarse_factor(stream, Lc.A, mg_inv.p));
        }
    }
    return FY_OK;
}

// OpenFOAM PCG.C with lduMatrix::solver::normFactor, in the single-reduction form (see k_pcg_cg_update); preconditioner = MG V-cycle or Jacobi.
// sc: [0] gamma = u.r, [1] delta = u.Au, [2..5] two sets {gamma_old, alpha_old}, [6] sum(p)
int Solver::prepare_p_init(bool with_halo) {
    // xbar = average(p) for lduMatrix::solver::normFactor: sum(p) over the owned cells, all-reduced.  The last PCG update of p left it with the
    // host (k_pcg_cg_update's second slot: the same partition and order as the sum below, the same bits); p_sum_valid falls when anything else writes p
    if (!p_sum_valid) {
        FY_TRY(launch_dot(stream, Nc, g.c0, p.p, nullptr, partials.p));
        FY_TRY(reduce_to_device(sc.p + 6));
    }
    return with_halo ? halo_p() : FY_OK;
}

int Solver::solve_pressure(bool final_iter, bool init_done) {
    Comm::Tag tag(comm, "pcg");
    MgLev& L = *mg[0];
    const double tol = final_iter ? cs.p_final_tol : cs.p_tol, rel = final_iter ? cs.p_final_rel_tol : cs.p_rel_tol;
    double h[2];
    if (!init_done) {
        FY_TRY(prepare_p_init());
        FY_TRY(launch_p_init(stream, L.A, prhs.p, p.p, p_sum_valid ? nullptr : sc.p + 6, p_sum, 1.0 / (double)Nglob, pr.p, partials.p));
    }
    FY_TRY(reduce_read(2, false, h));
    const double norm = h[1] + 1e-20;
    double res = h[0] / norm;
    const double res0 = res;
    st.p_initial_residual = res0;
    auto converged = [&](double r) { return r < tol || (rel > 0 && r < rel * res0); };
    int it = 0;
    if (!converged(res)) {
        do {
            const double* u;
            bool u_ghosts = false;                             // u valid one plane into the ghosts already
            vcycle_dot_done = false;
            if (cs.p_solver == FY_PSOLVER_PCG_MG) {
                L.bptr = pr.p;
                Comm::Tag vt(comm, "vcycle");
                if (mg_deep) { FY_TRY(vcycle_deep(0, 1)); u_ghosts = true; }
                else { want_vcycle_dot = true; FY_TRY(vcycle(0)); want_vcycle_dot = false; }
                u = L.xcur;
            } else {
                FY_TRY(launch_jacobi_precond(stream, L.A, pr.p, pzj.p));
                u = pzj.p;
            }
            if (!u_ghosts) FY_TRY(halo(const_cast<double*>(u), 1, plane, g.nz, g.gz, 1));
            kc[KC_P_APPLY_DOT].begin(stream);
            FY_TRY(launch_p_apply_dot(stream, L.A, u, vcycle_dot_done ? nullptr : pr.p, pw.p, partials.p));      // w = A u; gamma, delta
            kc[KC_P_APPLY_DOT].end(stream);
            FY_TRY(launch_reduce_finalize(stream, partials.p, Nc, 2, nullptr, sc.p));
            FY_TRY(comm->allreduce(stream, sc.p, 2, false));
            FY_TRY(launch_pcg_cg_update(stream, Nc, g.c0, u, pw.p, pp.p, ps.p, p.p, pr.p, sc.p, it, partials.p));
            if (it == 0) {
                // p = u and s = w without a pass: the buffers trade places (the level's buffer that held u becomes the old pp: scratch for the next cycle)
                std::swap(ps.p, pw.p);
                if (u == L.xcur) {
                    double*& held = (L.xcur == L.x0.p) ? L.x0.p : L.x1.p;
                    std::swap(pp.p, held);
                    L.xcur = held; L.xalt = (L.xcur == L.x0.p) ? L.x1.p : L.x0.p;
                } else {
                    std::swap(pp.p, pzj.p);
                }
            }
            p_ghosts_fresh = false;
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
