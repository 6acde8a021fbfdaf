// extern "C" entry points of the solver (include/foamyade_hip.h): thin shims over fy::Solver
#include "fv_solver.hpp"

extern "C" {

// documented defaults: icoFoam cavity / DPMFoam tutorial settings (the reference ships no case)
void fy_case_defaults(fy_case_desc* c, int solver) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->solver = solver;
    c->nx = c->ny = c->nz = 32; c->dx = 0.1 / 32; c->dt = 0.005; c->nu = 0.01; c->rho_fluid = 1000.0; c->rho_particle = 2650.0;
    for (int q = 0; q < 6; ++q) { c->u_bc[q] = FY_BC_U_FIXED_VALUE; c->p_bc[q] = FY_BC_P_ZERO_GRADIENT; }
    c->n_outer_correctors = 1; c->n_correctors = 2; c->n_non_orth_correctors = 0; c->momentum_predictor = 1;
    c->p_ref_cell = 0; c->p_ref_value = 0.0;
    c->p_solver = FY_PSOLVER_PCG_MG;
    c->p_tol = 1e-6; c->p_rel_tol = 0.05; c->p_final_tol = 1e-6; c->p_final_rel_tol = 0.0; c->p_max_iter = 1000;
    c->u_tol = 1e-5; c->u_rel_tol = 0.0; c->u_max_iter = 1000;
    c->adjust_time_step = 0; c->max_co = 1.0; c->max_delta_t = 1e300;
    c->u_relax = 1.0; c->u_relax_final = 0.0; c->p_relax = 0.0; c->p_relax_final = 0.0;
    c->convection_limiter_k = 1.0;
    c->k_tol = 1e-6; c->k_rel_tol = 0.0; c->k_max_iter = 1000; c->k_relax = 0.0; c->k_convection_scheme = FY_CONVECTION_UPWIND;
    c->eps_tol = 1e-6; c->eps_rel_tol = 0.0; c->eps_max_iter = 1000; c->eps_relax = 0.0; c->eps_convection_scheme = FY_CONVECTION_UPWIND;
    c->wf_kappa = 0.41; c->wf_E = 9.8;                      // [OF-6 nutWallFunctionFvPatchScalarField defaults]
    c->ras_cmu = 0.09; c->ras_c1 = 1.44; c->ras_c2 = 1.92; c->ras_c3 = 0.0; c->ras_sigmak = 1.0; c->ras_sigmaeps = 1.3;      // [OF-6 kEpsilon.C defaults]
    c->turbulence_model = FY_TURBULENCE_LAMINAR; c->les_ck = 0.094; c->les_ce = 1.048; c->les_delta_coeff = 1.0;     // [OF-6 Smagorinsky.C, cubeRootVolDelta.C defaults]
}

static int solver_create_impl(const fy_case_desc* c, const fy_transport* tr, int device_ordinal, fy::Comm* cm, fy_solver** out) {
    if (!out) return fy::fail(FY_ERR_INVALID, "null out");
    *out = nullptr;
    fy_solver* s = new (std::nothrow) fy_solver();
    if (!s) return fy::fail(FY_ERR_INVALID, "out of host memory");
    int rc = s->s.create(c, tr, device_ordinal, cm);
    if (rc != FY_OK) { delete s; return rc; }
    *out = s;
    return FY_OK;
}

int fy_solver_create(const fy_case_desc* c, const fy_transport* tr, int device_ordinal, fy_solver** out) {
    return solver_create_impl(c, tr, device_ordinal, nullptr, out);
}

int fy_solver_create_slab(const fy_case_desc* c, const fy_transport* tr, int device_ordinal, fy_comm* comm, fy_solver** out) {
    if (!comm || !comm->c) return fy::fail(FY_ERR_INVALID, "null communicator");
    return solver_create_impl(c, tr, device_ordinal, comm->c, out);
}

int fy_comm_create_local_group(int n, fy_comm** out) {
    if (n < 1 || !out) return fy::fail(FY_ERR_INVALID, "bad arguments");
    std::vector<fy::Comm*> cs((size_t)n);
    FY_TRY(fy::local_comm_group_create(n, cs.data()));
    for (int r = 0; r < n; ++r) { out[r] = new fy_comm(); out[r]->c = cs[(size_t)r]; }
    return FY_OK;
}
int fy_comm_create_host(int rank, int size, const fy_comm_callbacks* cb, fy_comm** out) {
    if (!out) return fy::fail(FY_ERR_INVALID, "null out");
    fy::Comm* c = nullptr;
    FY_TRY(fy::host_comm_create(rank, size, cb, &c));
    *out = new fy_comm(); (*out)->c = c;
    return FY_OK;
}
int fy_comm_create_ipc(int rank, int size, const fy_comm_callbacks* cb, int device, fy_comm** out) {
    if (!out) return fy::fail(FY_ERR_INVALID, "null out");
    fy::Comm* c = nullptr;
    FY_TRY(fy::ipc_comm_create(rank, size, cb, device, &c));
    *out = new fy_comm(); (*out)->c = c;
    return FY_OK;
}
int fy_rccl_unique_id(void* out128) { return fy::rccl_unique_id(out128); }
int fy_comm_create_rccl(int rank, int size, const void* id128, int device, fy_comm** out) {
    if (!out) return fy::fail(FY_ERR_INVALID, "null out");
    fy::Comm* c = nullptr;
    FY_TRY(fy::rccl_comm_create(rank, size, id128, device, &c));
    *out = new fy_comm(); (*out)->c = c;
    return FY_OK;
}
int fy_comm_destroy(fy_comm* c) { if (c) { delete c->c; delete c; } return FY_OK; }
int fy_comm_stats(fy_comm* c, uint64_t* out4) {
    if (!c || !c->c || !out4) return fy::fail(FY_ERR_INVALID, "fy_comm_stats: null argument");
    out4[0] = c->c->n_exchange; out4[1] = c->c->n_allreduce; out4[2] = c->c->n_allgather; out4[3] = c->c->exchange_bytes;
    return FY_OK;
}
int fy_comm_stats_by_tag(fy_comm* c, char* buf, size_t cap) {
    if (!c || !c->c || !buf || cap == 0) return fy::fail(FY_ERR_INVALID, "fy_comm_stats_by_tag: null argument");
    std::string out;
    for (const auto& kv : c->c->by_tag)
        out += (kv.first.empty() ? std::string("-") : kv.first) + " " + std::to_string(kv.second[0]) + " " + std::to_string(kv.second[1]) + " " + std::to_string(kv.second[2]) + "\n";
    if (out.size() + 1 > cap) return fy::fail(FY_ERR_INVALID, "fy_comm_stats_by_tag: buffer too small (%zu needed)", out.size() + 1);
    std::memcpy(buf, out.c_str(), out.size() + 1);
    return FY_OK;
}
int fy_comm_selftest(fy_comm* c, int device_ordinal) { if (!c || !c->c) return fy::fail(FY_ERR_INVALID, "null communicator"); return fy::comm_selftest(c->c, device_ordinal); }
int fy_comm_rank(fy_comm* c) { return c && c->c ? c->c->rank : -1; }
int fy_comm_size(fy_comm* c) { return c && c->c ? c->c->size : -1; }

#define FY_S(s) if (!(s)) return fy::fail(FY_ERR_INVALID, "null fy_solver")

fy_ctx* fy_solver_coupling(fy_solver* s) { return s ? s->s.cpl : nullptr; }
int fy_solver_step(fy_solver* s) { FY_S(s); return s->s.step(); }
int fy_solver_get_stats(fy_solver* s, fy_step_stats* out) { FY_S(s); if (!out) return fy::fail(FY_ERR_INVALID, "null out"); *out = s->s.st; return FY_OK; }
int fy_solver_local_cells(fy_solver* s) { return s ? s->s.Nc : -1; }

int fy_solver_hold_sources(fy_solver* s, int hold) {
    FY_S(s);
    s->s.hold_sources = hold != 0;
    if (!hold && s->s.sources_pending) { FY_HIP(hipSetDevice(s->s.device)); FY_TRY(s->s.cpl->c.set_source_zero()); s->s.sources_pending = false; }
    return FY_OK;
}

int fy_solver_field_count(fy_solver* s, const char* name, int64_t* count) {
    FY_S(s);
    if (!count) return fy::fail(FY_ERR_INVALID, "null count");
    double* p; size_t n;
    FY_TRY(s->s.field(name, &p, &n));
    *count = (int64_t)n;
    return FY_OK;
}

int fy_solver_read_field_host(fy_solver* s, const char* name, double* out) {
    FY_S(s);
    double* p; size_t n;
    FY_TRY(s->s.field(name, &p, &n));
    FY_HIP(hipSetDevice(s->s.device));
    FY_HIP(hipMemcpyAsync(out, p, n * sizeof(double), hipMemcpyDeviceToHost, s->s.stream));
    FY_HIP(hipStreamSynchronize(s->s.stream));
    return FY_OK;
}

int fy_solver_write_field_host(fy_solver* s, const char* name, const double* in) {
    FY_S(s);
    double* p; size_t n;
    FY_TRY(s->s.field(name, &p, &n));
    FY_HIP(hipSetDevice(s->s.device));
    FY_HIP(hipMemcpyAsync(p, in, n * sizeof(double), hipMemcpyHostToDevice, s->s.stream));
    s->s.carry_valid = false;                 // whatever was written, the carried Courant sums may no longer describe phi
    s->s.p_sum_valid = false; s->s.p_ghosts_fresh = false;
    if (std::string(name) == "U") {           // createPhi (collective when there are several slabs)
        s->s.U_ghosts_fresh = false;
        FY_TRY(s->s.halo_U());
        FY_TRY(fy::launch_flux_of(s->s.stream, s->s.g, s->s.U.p, s->s.F3(s->s.phi)));
    }
    if (std::string(name) == "nut") FY_TRY(s->s.halo_cells(s->s.nut, 1, 1));
    if (std::string(name) == "k") FY_TRY(s->s.halo_cells(s->s.kturb, 1, 1));
    if (std::string(name) == "epsilon") FY_TRY(s->s.halo_cells(s->s.epsturb, 1, 1));
    FY_HIP(hipStreamSynchronize(s->s.stream));
    return FY_OK;
}

int fy_solver_destroy(fy_solver* s) { delete s; return FY_OK; }

int fy_solver_apply_p_matrix_host(fy_solver* s, const double* x, double* y) {
    FY_S(s);
    fy::Solver& S = s->s;
    FY_HIP(hipSetDevice(S.device));
    FY_HIP(hipMemcpyAsync(S.pp.p + S.g.c0, x, (size_t)S.Nc * sizeof(double), hipMemcpyHostToDevice, S.stream));
    FY_TRY(S.halo_cells(S.pp, 1, 1));
    FY_TRY(fy::launch_p_apply(S.stream, S.mg[0]->A, S.pp.p, S.pw.p));
    FY_HIP(hipMemcpyAsync(y, S.pw.p + S.g.c0, (size_t)S.Nc * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    FY_HIP(hipStreamSynchronize(S.stream));
    return FY_OK;
}

int fy_solver_solve_p_host(fy_solver* s, const double* rhs, double* x, int* iterations) {
    FY_S(s);
    if (!rhs || !x) return fy::fail(FY_ERR_INVALID, "fy_solver_solve_p_host: null argument");
    fy::Solver& S = s->s;
    FY_HIP(hipSetDevice(S.device));
    FY_HIP(hipMemcpyAsync(S.prhs.p + S.g.c0, rhs, (size_t)S.Nc * sizeof(double), hipMemcpyHostToDevice, S.stream));
    FY_HIP(hipMemcpyAsync(S.p.p + S.g.c0, x, (size_t)S.Nc * sizeof(double), hipMemcpyHostToDevice, S.stream));
    S.p_sum_valid = false; S.p_ghosts_fresh = false;
    const int before = S.st.p_iters_total;
    FY_TRY(S.solve_pressure(true));
    FY_HIP(hipMemcpyAsync(x, S.p.p + S.g.c0, (size_t)S.Nc * sizeof(double), hipMemcpyDeviceToHost, S.stream));
    FY_HIP(hipStreamSynchronize(S.stream));
    if (iterations) *iterations = S.st.p_iters_total - before;
    return FY_OK;
}

int fy_solver_time_p_apply(fy_solver* s, int reps, double* avg_ms) {
    FY_S(s);
    if (reps < 1 || !avg_ms) return fy::fail(FY_ERR_INVALID, "bad arguments");
    fy::Solver& S = s->s;
    fy::EventTimer t;
    FY_TRY(t.init());
    for (int i = 0; i < 3; ++i) FY_TRY(fy::launch_p_apply(S.stream, S.mg[0]->A, S.pp.p, S.pw.p));
    t.start(S.stream);
    for (int i = 0; i < reps; ++i) FY_TRY(fy::launch_p_apply(S.stream, S.mg[0]->A, S.pp.p, S.pw.p));
    t.stop(S.stream);
    *avg_ms = t.ms() / reps;
    t.destroy();
    return FY_OK;
}

int fy_solver_enable_kernel_timing(fy_solver* s, int on) {
    FY_S(s);
    for (auto& k : s->s.kc) { k.reset(); k.on = on != 0; }
    return FY_OK;
}

int fy_solver_enable_exchange_timing(fy_solver* s, int on) {
    FY_S(s);
    s->s.xwait_timing = on != 0;
    for (auto& k : s->s.clk_xwait) { k.reset(); k.on = on != 0; k.per_collect = 4096; }
    return FY_OK;
}

int fy_solver_get_exchange_wait(fy_solver* s, double ms[4], int64_t waits[4]) {
    FY_S(s);
    if (!ms) return fy::fail(FY_ERR_INVALID, "fy_solver_get_exchange_wait: null argument");
    for (int q = 0; q < fy::Solver::XW_COUNT; ++q) { ms[q] = s->s.clk_xwait[q].total_ms; if (waits) waits[q] = s->s.clk_xwait[q].launches; }
    return FY_OK;
}

int fy_solver_get_kernel_timing(fy_solver* s, const char* kernel, double* total_ms, int64_t* launches) {
    FY_S(s);
    const std::string k = kernel ? kernel : "";
    int idx = k == "mg_smooth_l0" ? fy::Solver::KC_MG_SMOOTH0 : k == "p_apply_dot" ? fy::Solver::KC_P_APPLY_DOT : k == "mom_pass" ? fy::Solver::KC_MOM_PASS : -1;
    if (idx < 0 || !total_ms || !launches) return fy::fail(FY_ERR_INVALID, "unknown kernel clock '%s'", k.c_str());
    *total_ms = s->s.kc[idx].total_ms; *launches = s->s.kc[idx].launches;
    return FY_OK;
}

}  // extern "C"
