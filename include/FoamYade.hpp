// FoamYade.hpp -- header-only C++ facade over the C-ABI (include/foamyade_hip.h) with the shape of the reference class
// Foam::FoamYade (FoamYade/FoamYade.H:57-161): same constructor argument order, same three methods, so the two solver
// main()s change only the type name and how the field storage is named.
//
// OpenFOAM fields are contiguous AoS arrays (vector = 3 doubles, tensor = 9 doubles row-major), so a caller passes
// `U.primitiveField().cdata()`-style pointers; nothing here depends on OpenFOAM headers.
#pragma once
#include <stdexcept>
#include <string>

#include "foamyade_hip.h"

namespace fyhip {

struct MeshView {                 // what the reference reads from fvMesh (FoamYade.H:121; FoamYade.C:69,86,304)
    int n_cells;
    const double* centres;        // mesh.C()   [n][3]
    const double* volumes;        // mesh.V()   [n]
    double bbox_min[3], bbox_max[3];
    int nx = 0, ny = 0, nz = 0;   // uniform hex block description (required for point-force mode); 0 = unstructured
    double dx = 0.0;
    double origin[3] = {0, 0, 0};
};

class FoamYade {
public:
    // argument order of FoamYade::FoamYade (FoamYade.H:106-117)
    FoamYade(const MeshView& mesh, const double* U, const double* gradP, const double* vGrad, const double* divT, const double* ddtU,
             const double g[3], double* uSourceDrag, double* alpha, double* uSource, double* uParticle, bool gaussianInterp,
             const fy_transport* transport = nullptr, int fields_location = FY_MEM_HOST, int device = 0) {
        fy_mesh_desc m{};
        m.n_cells = mesh.n_cells; m.centres = mesh.centres; m.volumes = mesh.volumes;
        for (int a = 0; a < 3; ++a) { m.bbox_min[a] = mesh.bbox_min[a]; m.bbox_max[a] = mesh.bbox_max[a]; m.origin[a] = mesh.origin[a]; }
        m.nx = mesh.nx; m.ny = mesh.ny; m.nz = mesh.nz; m.dx = mesh.dx;
        fy_field_ptrs f{};
        f.location = fields_location;
        f.U = U; f.gradP = gradP; f.vGrad = vGrad; f.divT = divT; f.ddtU = ddtU;
        for (int a = 0; a < 3; ++a) f.g[a] = g[a];
        f.uSourceDrag = uSourceDrag; f.alpha = alpha; f.uSource = uSource; f.uParticle = uParticle;
        check(fy_create(&m, &f, gaussianInterp ? 1 : 0, transport, device, &ctx_));
    }
    FoamYade(const FoamYade&) = delete;
    FoamYade& operator=(const FoamYade&) = delete;
    virtual ~FoamYade() { fy_destroy(ctx_); }

    void setScalarProperties(double rhoP, double rhoF, double nu) { check(fy_set_scalar_properties(ctx_, rhoP, rhoF, nu)); }   // FoamYade.C:9-11
    void setParticleAction(double dt) { syncFibre(); check(fy_set_particle_action(ctx_, dt)); }                                  // FoamYade.C:605-632
    int finalizeRun() { int v = -1; check(fy_finalize_run(ctx_, &v)); return v; }     // FoamYade.C:595-599: 10 = the caller finalizes MPI
    void setSourceZero() { check(fy_set_source_zero(ctx_)); }                                                                    // FoamYade.C:556-566
    // opt-in: the two force models the reference carries without a call site (FoamYade.C:392-413, 465-479); default off = shipped behaviour
    void setForceModels(unsigned flags) { check(fy_set_force_models(ctx_, flags)); }
    // FoamYade.H:102: a PUBLIC flag the reference's callers set by assignment (`yadeCoupling.fibreCpl = true;`); it is read at the top of
    // setParticleAction here, so that an unchanged caller gets the 15-double records it asked for (setFibreCoupling does the same at once)
    bool fibreCpl = false;
    void setFibreCoupling(bool on) { fibreCpl = on; syncFibre(); }
    // FoamYade.C:582-590: both are empty in the reference ("TODO", immediate return); kept so that callers compile unchanged
    void calcHydroTimeScale() {}
    void sendHydroTimeScale(void* /*yProc*/) {}
    double yadeDT() const { return fy_yade_dt(ctx_); }                                                                           // FoamYade.H:94
    fy_ctx* handle() { return ctx_; }

private:
    void syncFibre() {
        if (fibreCpl != fibreSent_) { check(fy_set_fibre_coupling(ctx_, fibreCpl ? 1 : 0)); fibreSent_ = fibreCpl; }
    }
    bool fibreSent_ = false;
    static void check(int rc) {
        if (rc != FY_OK) throw std::runtime_error(std::string("libfoamyade_hip: ") + fy_last_error());
    }
    fy_ctx* ctx_ = nullptr;
};

}  // namespace fyhip
