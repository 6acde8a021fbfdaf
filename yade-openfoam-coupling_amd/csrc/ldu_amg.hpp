// The pressure matrix of fy_ldu_solver in cell-major ELL form, and an agglomeration multigrid V-cycle on it as the PCG preconditioner -- what
// `solver GAMG;` / `preconditioner GAMG;` in fvSolution.solvers.p asks OpenFOAM for [OF-6 GAMGSolver, pairGAMGAgglomeration: faceAreaPair], re-designed
// for the device:
//   * rows as ELL slots: nbr[k n + c], coef[k n + c] (slot-major, so the lanes of a wave read consecutive words); padding = the cell itself with
//     coefficient 0.  One gather per neighbour instead of the three (cell -> face -> owner / neighbour -> value) of the face-addressed form
//   * the hierarchy is built ONCE on the host from the face areas (faceAreaPair's weights): three pairwise matching passes per level (aggregates of
//     about eight cells, like the 2 x 2 x 2 of the structured solver), coarse cells numbered in the order of their first fine cell
//   * the Galerkin product P^T A P with piecewise-constant P is a GATHER through index lists precomputed with the hierarchy (no atomics: every
//     coefficient is summed in a fixed order), scaled by (cells per aggregate)^(-1/3) -- the 1/2 of fv_pressure.cpp under 2 x 2 x 2 coarsening (the
//     over-correction of piecewise-constant transfer) --, the reference cell's point term carried unscaled
//   * smoother: the Chebyshev-weighted Jacobi pair of the structured V-cycle (fv_solver.hpp: kMgWa, kMgWb), post-smoothing in reverse order, so the
//     cycle is a symmetric positive definite operator; the coarsest level (<= kAmgCoarsest cells) is solved exactly (dense inverse, one workgroup)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <vector>

#include "common.hpp"

namespace fy {

struct EllMat {                 // device pointers
    int n, W;
    const int32_t* nbr;         // [W n]
    const double* coef;         // [W n]  a >= 0: the row is  diag x_c - sum_k coef x_nbr
    const double* diag;         // [n]
};

constexpr int kAmgTailCells = 1024;     // levels of at most this many cells run inside the one-workgroup tail kernel of the V-cycle
constexpr int kAmgTailMax = 4;          // ... at most this many of them
constexpr int kAmgCoarsest = 64;        // cells the coarsest level may have (its dense inverse is formed in LDS: 32 KB)

struct AmgLevel {
    int n = 0, W = 0;
    DevBuf<int32_t> nbr, agg, child_off, child, ent_off, ent_idx, din_off, din_idx;      // agg: this level's cell -> the next level's; the lists build the NEXT level
    DevBuf<double> coef, diag, invd, x0, x1, b;
    int ref_cell = -1;                                                                   // the cell that holds the reference cell (or -1)
    EllMat mat() const { return EllMat{n, W, nbr.p, coef.p, diag.p}; }
};

struct LduAmg {
    std::vector<std::unique_ptr<AmgLevel> > lev;   // lev[0] = the mesh's cells
    DevBuf<int32_t> ell_face;                      // [W0 n0]: the internal face of slot k of cell c, or -1
    int ref_cell0 = -1;
    bool hier = false;
    int passes = 3;
    DevBuf<double> coarse_inv;                     // the coarsest level's inverse, dense [n][n]
    const double* diag0_ = nullptr;               // the fine diagonal of the last setup (the solver's array)
    // host: adjacency of the mesh (per cell its internal faces in ascending order: neighbour, face, weight) -> the ELL pattern and the hierarchy
    int build(hipStream_t s, int n_cells, int n_internal, const int32_t* own, const int32_t* nei, const std::vector<int32_t>& cf_off, const std::vector<int32_t>& cf_face,
              const double* face_weight, int ref_cell, bool with_hierarchy);
    // per matrix: the fine level's coefficients from the face coefficients (diag = the solver's own array), then every coarse level
    int setup(hipStream_t s, const double* pcoef, const double* pdiag);
    // u = V-cycle(r) from a zero first guess
    int vcycle(hipStream_t s, const double* r, double* u);
    bool has_hierarchy() const { return hier; }
};

int launch_ell_fill(hipStream_t s, int n, int W, const int32_t* ell_face, const double* pcoef, double* coef);
int launch_ell_jacobi(hipStream_t s, int n, const double* diag, const double* r, double* u);                               // u = r / diag
int launch_ell_apply_dot(hipStream_t s, EllMat A, const double* u, const double* r, double* w, double* partials);         // w = A u; slot 0 = u.r, slot 1 = u.w

}  // namespace fy
