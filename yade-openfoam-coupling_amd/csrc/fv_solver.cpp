// placeholder until the FV half lands: every fy_solver_* symbol of include/foamyade_hip.h is exported and fails loudly
#include "common.hpp"
extern "C" {
void fy_case_defaults(fy_case_desc* c, int solver) { if (c) { std::memset(c, 0, sizeof(*c)); c->solver = solver; } }
int fy_solver_create(const fy_case_desc*, const fy_transport*, int, fy_solver** out) { if (out) *out = nullptr; return fy::fail(FY_ERR_UNSUPPORTED, "fy_solver not built yet"); }
fy_ctx* fy_solver_coupling(fy_solver*) { return nullptr; }
int fy_solver_step(fy_solver*) { return fy::fail(FY_ERR_UNSUPPORTED, "fy_solver not built yet"); }
int fy_solver_get_stats(fy_solver*, fy_step_stats*) { return fy::fail(FY_ERR_UNSUPPORTED, "fy_solver not built yet"); }
int fy_solver_read_field_host(fy_solver*, const char*, double*) { return fy::fail(FY_ERR_UNSUPPORTED, "fy_solver not built yet"); }
int fy_solver_write_field_host(fy_solver*, const char*, const double*) { return fy::fail(FY_ERR_UNSUPPORTED, "fy_solver not built yet"); }
int fy_solver_destroy(fy_solver*) { return FY_OK; }
int fy_solver_apply_p_matrix_host(fy_solver*, const double*, double*) { return fy::fail(FY_ERR_UNSUPPORTED, "fy_solver not built yet"); }
int fy_solver_time_p_apply(fy_solver*, int, double*) { return fy::fail(FY_ERR_UNSUPPORTED, "fy_solver not built yet"); }
}
