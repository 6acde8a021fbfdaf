// OpenFOAM case directories for fy_solver: what icoFoamYade / pimpleFoamYade get from OpenFOAM's runTime / mesh / field
// constructors (icoFoamYade.C:38-46, createFields.H of both solvers) and give back with runTime.write() (icoFoamYade.C:142,
// pimpleFoamYade.C:107), for the one mesh class this library computes on: a single axis-aligned blockMesh hex block (uniform cubes, or graded: simpleGrading (ex ey ez)).
//   read   system/blockMeshDict  (vertices, one `hex` block, simpleGrading (ex ey ez), `boundary` patches -> the 6 box sides)
//          system/controlDict    (startTime, endTime, deltaT, writeControl, writeInterval)
//          system/fvSolution     (PISO | PIMPLE controls, solvers.p / pFinal / U tolerances)
//          constant/transportProperties (nu, partDensity, fluidDensity | continuousPhaseName + rho.<phase>), constant/g
//          <startTime>/U | U.<phase>, <startTime>/p  (boundary types fixedValue / noSlip / zeroGradient / fixedFluxPressure;
//                                                    internalField uniform or nonuniform)
//   write  <time>/U | U.<phase>, p, and in Gaussian mode alpha.<phase>: ASCII volFields with the case's own patch entries
// Anything outside that subset is refused with FY_ERR_UNSUPPORTED and a message naming the file and keyword -- never guessed.
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "foam_dict.hpp"

struct fy_foam_case {
    std::string dir;
    int solver = FY_SOLVER_ICO;
    fy_case_desc desc{};
    std::string phase, u_name, start_name;
    double start_time = 0, end_time = 0;
    int write_interval_steps = 0;
    std::string patch_of_side[6];               // blockMesh patch name covering XMIN, XMAX, YMIN, YMAX, ZMIN, ZMAX
    std::vector<double> grading[3];             // graded block: cell sizes per axis (fy_case_desc.hx / hy / hz point here)
    std::vector<std::string> patch_order;       // patch names in blockMeshDict order (one side each here)
    std::string u_bc_text[6], p_bc_text[6];     // the boundaryField entries as read, re-emitted on write
    std::vector<double> U0, p0, nut0, k0, eps0; // internalField of the start time (nut0, k0, eps0: turbulence cases only)
    std::string nut_bc_text[6], k_bc_text[6], eps_bc_text[6];
    // controlDict's output settings [OF-6 Time::readDict]: writeFormat ascii | binary, writePrecision (ASCII digits; absent: 17, lossless --
    // OpenFOAM's own default of 6 would not restart a run where it stopped), purgeWrite N (keep the N newest time directories this run wrote)
    bool write_binary = false;
    int write_precision = 17;
    int purge_write = 0;
    mutable std::vector<std::string> written;   // time directories written so far (purgeWrite's ring)
    // A DECOMPOSED case (decomposePar, simple (1 1 N): processorR = the R-th z-slab, its cells in the global order): the field files are read
    // from and written to <case>/processorR, hold the slab's cells only, and carry the processor patches (procBoundaryRtoS) next to the case's own
    std::string fdir;                           // where the time directories are: dir, or dir/processorR
    size_t fcells = 0, foffset = 0;             // cells per field file, global number of the first one
    int proc_rank = -1, proc_count = 0;
    std::vector<std::pair<std::string, std::string> > extra_patches[5];   // per field (U, p, nut, k, epsilon): boundaryField entries of other patches, as read
    // constant/polyMesh read instead of blockMeshDict: the mesh's cell numbers may differ from the lattice's (several blocks): file_cell[L] = the mesh's
    // cell at lattice index L = i + nx (j + ny k); empty = the same numbering (one block)
    std::vector<int32_t> file_cell;
    // A GENERAL polyhedral mesh (fy_foam_case_open_general: whatever createMesh.H would hand icoFoamYade, icoFoamYade.C:42): constant/polyMesh's arrays
    // as read, every patch of the boundary file in its order, and per patch the conditions fy_ldu_solver takes.  desc then only carries the controls
    bool general = false;
    int g_cells = 0, g_internal = 0;
    std::vector<double> g_points;
    std::vector<int32_t> g_face_off, g_face_pts, g_own, g_nei, g_patch_start, g_patch_size, g_u_bc, g_p_bc;
    std::vector<std::string> g_patch_name, g_u_text, g_p_text, g_nut_text, g_k_text;
    std::vector<int32_t> g_k_bc; std::vector<double> g_k_val;
    std::vector<std::string> g_eps_text; std::vector<int32_t> g_eps_bc; std::vector<double> g_eps_val;
    std::vector<int32_t> g_patch_neighbour;      // per patch: its cyclic partner, or -1
    std::vector<std::string> g_patch_class;      // the boundary file's `type` per patch (wall | patch | symmetryPlane | symmetry)
    std::vector<double> g_u_val, g_p_val, g_nut_val;
    std::vector<int32_t> g_nut_bc;
};

namespace {

using fy::fail;
using fy::FoamDict;

std::string join(const std::string& a, const std::string& b) { return a + "/" + b; }

bool file_exists(const std::string& path) { struct stat st; return stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

int need_file(const std::string& path, FoamDict* d) {
    std::string err;
    if (!fy::foam_parse_file(path, d, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    return FY_OK;
}

// `name { ... } name { ... }` inside a token list -> named dictionaries, in order
int named_dicts(const std::vector<std::string>& tok, const std::string& what, std::vector<std::pair<std::string, FoamDict> >* out) {
    size_t i = 0;
    double cnt;
    if (i < tok.size() && fy::foam_tok_is_number(tok[i], &cnt)) ++i;
    if (i >= tok.size() || tok[i] != "(") return fail(FY_ERR_INVALID, "%s: expected a list", what.c_str());
    ++i;
    while (i < tok.size() && tok[i] != ")") {
        const std::string name = tok[i++];
        if (i >= tok.size() || tok[i] != "{") return fail(FY_ERR_INVALID, "%s: expected '{' after '%s'", what.c_str(), name.c_str());
        int depth = 0;
        std::string text;
        for (; i < tok.size(); ++i) {
            if (tok[i] == "{") { if (depth++ == 0) continue; }
            if (tok[i] == "}") { if (--depth == 0) { ++i; break; } }
            text += tok[i];
            text += ' ';
        }
        if (depth != 0) return fail(FY_ERR_INVALID, "%s: unbalanced braces in '%s'", what.c_str(), name.c_str());
        FoamDict d;
        std::string err;
        if (!fy::foam_parse(text, &d, &err)) return fail(FY_ERR_INVALID, "%s: patch '%s': %s", what.c_str(), name.c_str(), err.c_str());
        out->emplace_back(name, d);
    }
    return FY_OK;
}

bool near(double a, double b, double scale) { return std::fabs(a - b) <= 1e-9 * scale; }

// one direction of one hex block: its cell sizes from the grading [OF-6 blockMesh lineDivide.C, gradingDescriptor(s).C].  A grading is a list of
// sections (fraction of the length, fraction of the cells, expansion ratio = last cell / first cell of the section); `simpleGrading (2 ...)` is the
// one-section list ((1 1 2)).  Fractions are normalised to sum 1; a section holds label(nDivFrac * n + 0.5) cells, the last one what is left;
// inside a section the sizes form a geometric progression with factor ratio^(1 / (cells - 1)); a negative ratio r means 1 / |r|
struct GradSection { double len, cells, ratio; };
int axis_sizes(const std::string& path, int n, double L, std::vector<GradSection> sec, std::vector<double>* h) {
    double sl = 0.0, sc = 0.0;
    for (GradSection& g : sec) {
        if (!(g.len > 0) || !(g.cells > 0) || g.ratio == 0.0) return fail(FY_ERR_INVALID, "%s: grading section (%g %g %g): fractions must be positive, the ratio non-zero", path.c_str(), g.len, g.cells, g.ratio);
        if (g.ratio < 0) g.ratio = 1.0 / -g.ratio;
        sl += g.len; sc += g.cells;
    }
    h->clear();
    int start = 0;
    for (size_t q = 0; q < sec.size(); ++q) {
        int m = (int)(sec[q].cells / sc * n + 0.5);
        if (q + 1 == sec.size() || m > n - start) m = n - start;
        if (m < 1) return fail(FY_ERR_INVALID, "%s: a grading section is left without cells (%d cells over %zu sections)", path.c_str(), n, sec.size());
        const double len = sec[q].len / sl * L;
        const double r = (sec[q].ratio != 1.0 && m > 1) ? std::pow(sec[q].ratio, 1.0 / (m - 1)) : 1.0;
        double sum = 0.0, w = 1.0;
        const size_t at = h->size();
        for (int i = 0; i < m; ++i) { h->push_back(w); sum += w; w *= r; }
        for (size_t i = at; i < h->size(); ++i) (*h)[i] *= len / sum;
        start += m;
    }
    if (start != n) return fail(FY_ERR_INVALID, "%s: the grading sections hold %d of %d cells", path.c_str(), start, n);
    return FY_OK;
}

struct HexBlock { int hv[8]; int nn[3]; double lo[3], hi[3]; std::vector<double> h[3]; };

// system/blockMeshDict: one hex block, or several that tile a box as a tensor product (Bx x By x Bz blocks, every one present, neighbours
// agreeing on the cell sizes along the faces they share -- what blockMesh merges into one rectilinear lattice); simpleGrading with one
// expansion ratio or a multi-grading list per direction.  Curved edges, edgeGrading, mergePatchPairs and sides shared by two patches are refused.
int read_block_mesh(fy_foam_case* c) {
    const std::string path = join(c->dir, "system/blockMeshDict");
    FoamDict d;
    FY_TRY(need_file(path, &d));
    double scale = 1.0;
    if (!d.scalar("convertToMeters", &scale)) d.scalar("scale", &scale);
    const auto* vt = d.tokens("vertices");
    std::vector<double> v;
    if (!vt || !fy::foam_read_list(*vt, 0, 3, &v) || v.size() < 24 || v.size() % 3 != 0)
        return fail(FY_ERR_UNSUPPORTED, "%s: need at least 8 vertices (x y z)", path.c_str());
    for (double& x : v) x *= scale;
    const int nvert = (int)(v.size() / 3);
    for (const char* key : {"edges", "mergePatchPairs"}) {
        const auto* et = d.tokens(key);
        if (et) for (const std::string& t : *et) if (t != "(" && t != ")") return fail(FY_ERR_UNSUPPORTED, "%s: a non-empty '%s' list is not supported (straight edges, conforming blocks)", path.c_str(), key);
    }
    const auto* bt = d.tokens("blocks");
    double lead;
    const size_t t0 = (bt && !bt->empty() && fy::foam_tok_is_number((*bt)[0], &lead)) ? 1 : 0;      // (an optional element count before the list)
    if (!bt || bt->size() < 24 + t0 || (*bt)[t0] != "(" || (*bt)[t0 + 1] != "hex") return fail(FY_ERR_UNSUPPORTED, "%s: blocks must hold 'hex' blocks", path.c_str());
    const std::vector<std::string>& T = *bt;
    std::vector<HexBlock> blocks;
    size_t i = t0 + 1;
    auto num = [&](double* x) { return i < T.size() && fy::foam_tok_is_number(T[i], x) ? (++i, true) : false; };
    auto tok = [&](const char* w) { return i < T.size() && T[i] == w ? (++i, true) : false; };
    double scl = 0.0;
    while (i < T.size() && T[i] != ")") {
        HexBlock B;
        if (!tok("hex") || !tok("(")) return fail(FY_ERR_UNSUPPORTED, "%s: blocks must hold 'hex' blocks", path.c_str());
        for (int q = 0; q < 8; ++q) { double x; if (!num(&x)) return fail(FY_ERR_INVALID, "%s: malformed hex vertex list", path.c_str()); B.hv[q] = (int)x; }
        if (!tok(")")) return fail(FY_ERR_INVALID, "%s: malformed hex vertex list", path.c_str());
        if (i < T.size() && T[i] != "(") ++i;                                   // (an optional cellZone name)
        if (!tok("(")) return fail(FY_ERR_INVALID, "%s: malformed hex cell counts", path.c_str());
        for (int q = 0; q < 3; ++q) { double x; if (!num(&x)) return fail(FY_ERR_INVALID, "%s: malformed hex cell counts", path.c_str()); B.nn[q] = (int)x; }
        if (!tok(")")) return fail(FY_ERR_INVALID, "%s: malformed hex cell counts", path.c_str());
        if (!tok("simpleGrading")) return fail(FY_ERR_UNSUPPORTED, "%s: only simpleGrading is supported (no edgeGrading)", path.c_str());
        if (!tok("(")) return fail(FY_ERR_INVALID, "%s: malformed simpleGrading", path.c_str());
        std::vector<GradSection> grad[3];
        for (int a = 0; a < 3; ++a) {
            double r;
            if (num(&r)) { grad[a].push_back(GradSection{1.0, 1.0, r}); continue; }
            if (!tok("(")) return fail(FY_ERR_INVALID, "%s: simpleGrading takes an expansion ratio or a list ((fraction cells ratio) ...) per direction", path.c_str());
            while (tok("(")) {
                GradSection g;
                if (!num(&g.len) || !num(&g.cells) || !num(&g.ratio) || !tok(")")) return fail(FY_ERR_INVALID, "%s: malformed multi-grading section (fraction cells ratio)", path.c_str());
                grad[a].push_back(g);
            }
            if (!tok(")") || grad[a].empty()) return fail(FY_ERR_INVALID, "%s: malformed multi-grading list", path.c_str());
        }
        if (!tok(")")) return fail(FY_ERR_INVALID, "%s: simpleGrading takes three entries", path.c_str());
        for (int q = 0; q < 8; ++q) if (B.hv[q] < 0 || B.hv[q] >= nvert) return fail(FY_ERR_INVALID, "%s: hex vertex label out of range", path.c_str());
        auto P = [&](int q, int a) { return v[3 * (size_t)B.hv[q] + a]; };
        // blockMesh's hex: 0-1 = local x, 0-3 = local y, 0-4 = local z; require them to be +x, +y, +z of an axis-aligned box
        const double L[3] = {P(1, 0) - P(0, 0), P(3, 1) - P(0, 1), P(4, 2) - P(0, 2)};
        const double bs = std::fabs(L[0]) + std::fabs(L[1]) + std::fabs(L[2]);
        scl = std::max(scl, bs);
        const int bits[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
        for (int q = 0; q < 8; ++q)
            for (int a = 0; a < 3; ++a)
                if (!near(P(q, a), P(0, a) + bits[q][a] * L[a], bs)) return fail(FY_ERR_UNSUPPORTED, "%s: every block must be an axis-aligned box with the standard hex vertex order", path.c_str());
        if (!(L[0] > 0 && L[1] > 0 && L[2] > 0) || B.nn[0] < 1 || B.nn[1] < 1 || B.nn[2] < 1) return fail(FY_ERR_INVALID, "%s: degenerate block", path.c_str());
        for (int a = 0; a < 3; ++a) {
            B.lo[a] = P(0, a); B.hi[a] = P(0, a) + L[a];
            FY_TRY(axis_sizes(path, B.nn[a], L[a], grad[a], &B.h[a]));
        }
        blocks.push_back(B);
    }
    if (blocks.empty()) return fail(FY_ERR_UNSUPPORTED, "%s: blocks must hold 'hex' blocks", path.c_str());
    // ---- the blocks as a tensor product: break points per axis, one block per slot, equal sizes along shared intervals
    std::vector<double> brk[3];
    for (int a = 0; a < 3; ++a) {
        for (const HexBlock& B : blocks) { brk[a].push_back(B.lo[a]); brk[a].push_back(B.hi[a]); }
        std::sort(brk[a].begin(), brk[a].end());
        std::vector<double> u;
        for (double x : brk[a]) if (u.empty() || !near(x, u.back(), scl)) u.push_back(x);
        brk[a].swap(u);
    }
    const size_t nslot[3] = {brk[0].size() - 1, brk[1].size() - 1, brk[2].size() - 1};
    if (nslot[0] * nslot[1] * nslot[2] != blocks.size())
        return fail(FY_ERR_UNSUPPORTED, "%s: the %zu blocks do not tile a box as a %zu x %zu x %zu tensor product (L-shaped or partly refined arrangements are not supported)", path.c_str(),
                    blocks.size(), nslot[0], nslot[1], nslot[2]);
    std::vector<int> taken(blocks.size(), 0);
    std::vector<const std::vector<double>*> hs[3];
    for (int a = 0; a < 3; ++a) hs[a].assign(nslot[a], nullptr);
    for (const HexBlock& B : blocks) {
        size_t at[3];
        for (int a = 0; a < 3; ++a) {
            size_t q = 0;
            while (q < nslot[a] && !near(brk[a][q], B.lo[a], scl)) ++q;
            if (q == nslot[a] || !near(brk[a][q + 1], B.hi[a], scl)) return fail(FY_ERR_UNSUPPORTED, "%s: a block spans more than one interval of the block lattice (not a tensor-product arrangement)", path.c_str());
            at[a] = q;
            if (!hs[a][q]) hs[a][q] = &B.h[a];
            else {
                const std::vector<double>& o = *hs[a][q];
                bool same = o.size() == B.h[a].size();
                for (size_t m = 0; same && m < o.size(); ++m) same = near(o[m], B.h[a][m], o[m]);
                if (!same) return fail(FY_ERR_UNSUPPORTED, "%s: neighbouring blocks disagree on the cell sizes along a shared direction (non-conforming blocks)", path.c_str());
            }
        }
        int& t = taken[at[0] + nslot[0] * (at[1] + nslot[1] * at[2])];
        if (t++) return fail(FY_ERR_INVALID, "%s: two blocks occupy the same place", path.c_str());
    }
    double L[3], lo[3];
    int nn[3];
    std::vector<double> h[3];
    for (int a = 0; a < 3; ++a) {
        for (size_t q = 0; q < nslot[a]; ++q) h[a].insert(h[a].end(), hs[a][q]->begin(), hs[a][q]->end());
        nn[a] = (int)h[a].size(); lo[a] = brk[a].front(); L[a] = brk[a].back() - brk[a].front();
    }
    const double dx = L[0] / nn[0];
    bool cubes = true;
    for (int a = 0; a < 3; ++a) for (double x : h[a]) cubes = cubes && near(x, dx, dx);
    c->desc.nx = nn[0]; c->desc.ny = nn[1]; c->desc.nz = nn[2]; c->desc.dx = dx;
    c->desc.hx = c->desc.hy = c->desc.hz = nullptr;
    if (!cubes) {       // a graded block (or uniform cells that are not cubes): per-axis cell sizes
        for (int a = 0; a < 3; ++a) c->grading[a] = h[a];
        c->desc.hx = c->grading[0].data(); c->desc.hy = c->grading[1].data(); c->desc.hz = c->grading[2].data();
    }
    for (int a = 0; a < 3; ++a) c->desc.origin[a] = lo[a];

    const auto* bd = d.tokens("boundary");
    if (!bd) return fail(FY_ERR_UNSUPPORTED, "%s: no 'boundary' list (the old 'patches' syntax is not supported)", path.c_str());
    std::vector<std::pair<std::string, FoamDict> > patches;
    FY_TRY(named_dicts(*bd, path + ": boundary", &patches));
    double covered[6] = {0, 0, 0, 0, 0, 0};
    for (auto& pd : patches) {
        const auto* ft = pd.second.tokens("faces");
        std::vector<double> f;
        if (!ft) return fail(FY_ERR_INVALID, "%s: patch '%s' has no faces", path.c_str(), pd.first.c_str());
        // ( (a b c d) (a b c d) ... ): read as 4-component tuples
        if (!fy::foam_read_list(*ft, 0, 4, &f) || f.empty()) return fail(FY_ERR_INVALID, "%s: patch '%s': malformed faces list", path.c_str(), pd.first.c_str());
        std::string ty;
        if (pd.second.word("type", &ty) && (ty == "empty" || ty == "cyclic" || ty == "wedge"))
            return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s' of type '%s' is not supported (wall / patch / symmetryPlane sides of a 3-D box)", path.c_str(), pd.first.c_str(), ty.c_str());
        for (size_t q = 0; q + 3 < f.size(); q += 4) {
            int side = -1;
            double flo[3] = {1e300, 1e300, 1e300}, fhi[3] = {-1e300, -1e300, -1e300};
            for (int m = 0; m < 4; ++m) {
                const int vi = (int)f[q + m];
                if (vi < 0 || vi >= nvert) return fail(FY_ERR_INVALID, "%s: patch '%s': vertex label out of range", path.c_str(), pd.first.c_str());
                for (int a = 0; a < 3; ++a) { flo[a] = std::min(flo[a], v[3 * (size_t)vi + a]); fhi[a] = std::max(fhi[a], v[3 * (size_t)vi + a]); }
            }
            for (int a = 0; a < 3 && side < 0; ++a)
                for (int s = 0; s < 2 && side < 0; ++s)
                    if (near(flo[a], lo[a] + s * L[a], scl) && near(fhi[a], lo[a] + s * L[a], scl)) side = 2 * a + s;
            if (side < 0) return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s' has a face that is not a side of the block", path.c_str(), pd.first.c_str());
            if (!c->patch_of_side[side].empty() && c->patch_of_side[side] != pd.first)
                return fail(FY_ERR_UNSUPPORTED, "%s: side %d of the box is shared by the patches '%s' and '%s' (one boundary condition per side)", path.c_str(), side, c->patch_of_side[side].c_str(), pd.first.c_str());
            c->patch_of_side[side] = pd.first;
            const int a = side / 2, t1 = (a + 1) % 3, t2 = (a + 2) % 3;
            covered[side] += (fhi[t1] - flo[t1]) * (fhi[t2] - flo[t2]);
        }
        c->patch_order.push_back(pd.first);
    }
    for (int s = 0; s < 6; ++s) {
        if (c->patch_of_side[s].empty()) return fail(FY_ERR_INVALID, "%s: block side %d belongs to no patch", path.c_str(), s);
        const int a = s / 2;
        const double area = L[(a + 1) % 3] * L[(a + 2) % 3];
        if (!near(covered[s], area, area)) return fail(FY_ERR_INVALID, "%s: block side %d is covered %s by its patch's faces (%g of %g)", path.c_str(), s, covered[s] > area ? "more than once" : "only in part", covered[s], area);
    }
    return FY_OK;
}

// constant/polyMesh (points, faces, owner, neighbour, boundary: what the reference's solvers read -- createMesh.H -- and blockMesh wrote): accepted
// when it is a rectilinear lattice of hexahedra filling an axis-aligned box, i.e. what blockMesh makes of the block arrangements read_block_mesh
// takes.  The lattice comes from the distinct point coordinates; a cell's place in it from the smallest corner of its faces' points; its number in the
// mesh need not be the lattice's (blocks are numbered one after the other): file_cell keeps the map, and the field files are read and written through it.
int read_poly_mesh(fy_foam_case* c) {
    const std::string base = join(c->dir, "constant/polyMesh");
    std::vector<std::string> tk;
    std::string err;
    auto code_of = [&]() { return err.find("not supported") != std::string::npos ? FY_ERR_UNSUPPORTED : FY_ERR_INVALID; };      // (an unsupported file format falls back to blockMeshDict)
    auto list_of = [&](const char* name) -> int { return fy::foam_list_file_tokens(join(base, name), &tk, &err) ? FY_OK : fail(code_of(), "%s", err.c_str()); };
    // ---- points: N (x y z) ...
    std::vector<double> pts;
    if (!fy::foam_numeric_list_file(join(base, "points"), &pts, &err)) return fail(code_of(), "%s", err.c_str());
    if (pts.size() < 25 || (pts.size() - 1) % 3 != 0 || (double)((pts.size() - 1) / 3) != pts[0]) return fail(FY_ERR_INVALID, "%s/points: malformed point list", base.c_str());
    pts.erase(pts.begin());
    const size_t npts = pts.size() / 3;
    std::vector<double> ax[3];
    double span = 0.0;
    for (int a = 0; a < 3; ++a) {
        std::vector<double> v(npts);
        for (size_t q = 0; q < npts; ++q) v[q] = pts[3 * q + a];
        std::sort(v.begin(), v.end());
        span = std::max(span, v.back() - v.front());
        ax[a].swap(v);
    }
    for (int a = 0; a < 3; ++a) {
        std::vector<double> u;
        for (double x : ax[a]) if (u.empty() || !(std::fabs(x - u.back()) <= 1e-9 * span)) u.push_back(x);
        ax[a].swap(u);
        if (ax[a].size() < 2) return fail(FY_ERR_INVALID, "%s/points: the mesh is flat along axis %d", base.c_str(), a);
    }
    const size_t np1[3] = {ax[0].size(), ax[1].size(), ax[2].size()};
    if (np1[0] * np1[1] * np1[2] != npts)
        return fail(FY_ERR_UNSUPPORTED, "%s/points: %zu points on %zu x %zu x %zu distinct coordinates -- not a rectilinear lattice (only axis-aligned boxes of hexahedra are supported)", base.c_str(),
                    npts, np1[0], np1[1], np1[2]);
    const int nn[3] = {(int)np1[0] - 1, (int)np1[1] - 1, (int)np1[2] - 1};
    const size_t ncell = (size_t)nn[0] * nn[1] * nn[2];
    auto index_of = [&](int a, double x) { return (int)(std::lower_bound(ax[a].begin(), ax[a].end(), x - 1e-9 * span) - ax[a].begin()); };
    std::vector<int32_t> pidx(3 * npts);
    for (size_t q = 0; q < npts; ++q) for (int a = 0; a < 3; ++a) pidx[3 * q + a] = index_of(a, pts[3 * q + a]);
    { std::vector<double>().swap(pts); }
    // ---- faces: N 4(a b c d) ...
    std::vector<int32_t> fl;
    if (!fy::foam_label_list_file(join(base, "faces"), &fl, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    if (fl.empty() || (fl.size() - 1) % 5 != 0 || (size_t)fl[0] != (fl.size() - 1) / 5) return fail(FY_ERR_UNSUPPORTED, "%s/faces: only quadrilateral faces (hexahedral cells) are supported", base.c_str());
    const size_t nfaces = (fl.size() - 1) / 5;
    std::vector<int32_t> fpt(4 * nfaces);
    for (size_t f = 0; f < nfaces; ++f) {
        if (fl[1 + 5 * f] != 4) return fail(FY_ERR_UNSUPPORTED, "%s/faces: only quadrilateral faces (hexahedral cells) are supported", base.c_str());
        for (int m = 0; m < 4; ++m) {
            const int32_t l = fl[2 + 5 * f + m];
            if (l < 0 || (size_t)l >= npts) return fail(FY_ERR_INVALID, "%s/faces: point label out of range", base.c_str());
            fpt[4 * f + m] = l;
        }
    }
    { std::vector<int32_t>().swap(fl); }
    std::vector<int32_t> own, nei;
    if (!fy::foam_label_list_file(join(base, "owner"), &own, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    if (own.empty() || (size_t)own[0] != own.size() - 1 || own.size() - 1 != nfaces) return fail(FY_ERR_INVALID, "%s/owner: %zu entries for %zu faces", base.c_str(), own.size() - 1, nfaces);
    own.erase(own.begin());
    if (!fy::foam_label_list_file(join(base, "neighbour"), &nei, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    if (nei.empty() || (size_t)nei[0] != nei.size() - 1 || nei.size() - 1 > nfaces) return fail(FY_ERR_INVALID, "%s/neighbour: malformed", base.c_str());
    nei.erase(nei.begin());
    // ---- every cell's smallest corner -> its place in the lattice
    std::vector<int32_t> cmin(3 * ncell, INT32_MAX);
    auto touch = [&](int32_t cell, size_t f) -> bool {
        if (cell < 0 || (size_t)cell >= ncell) return false;
        for (int m = 0; m < 4; ++m) for (int a = 0; a < 3; ++a) {
            int32_t& v = cmin[3 * (size_t)cell + a];
            v = std::min(v, pidx[3 * (size_t)fpt[4 * f + m] + a]);
        }
        return true;
    };
    for (size_t f = 0; f < nfaces; ++f) {
        if (!touch(own[f], f) || (f < nei.size() && !touch(nei[f], f)))
            return fail(FY_ERR_UNSUPPORTED, "%s: cell labels beyond %zu = the lattice's cell count (not a box filled with hexahedra)", base.c_str(), ncell);
    }
    c->file_cell.assign(ncell, -1);
    bool identity = true;
    for (size_t q = 0; q < ncell; ++q) {
        const int i = cmin[3 * q], j = cmin[3 * q + 1], k = cmin[3 * q + 2];
        if (i < 0 || i >= nn[0] || j < 0 || j >= nn[1] || k < 0 || k >= nn[2]) return fail(FY_ERR_UNSUPPORTED, "%s: cell %zu does not sit in the lattice", base.c_str(), q);
        const size_t L = (size_t)i + (size_t)nn[0] * ((size_t)j + (size_t)nn[1] * (size_t)k);
        if (c->file_cell[L] != -1) return fail(FY_ERR_UNSUPPORTED, "%s: two cells share the lattice place (%d %d %d)", base.c_str(), i, j, k);
        c->file_cell[L] = (int32_t)q;
        identity = identity && L == q;
    }
    if (identity) c->file_cell.clear();
    // ---- boundary: N ( name { type ...; nFaces n; startFace s; } ... ): every patch's faces lie on sides of the box, every side belongs to one patch
    FY_TRY(list_of("boundary"));
    std::vector<std::pair<std::string, FoamDict> > patches;
    FY_TRY(named_dicts(tk, base + "/boundary", &patches));
    std::vector<size_t> side_faces(6, 0);
    for (auto& pd : patches) {
        int nf = 0, sf = 0;
        std::string ty;
        pd.second.word("type", &ty);
        if (!pd.second.integer("nFaces", &nf) || !pd.second.integer("startFace", &sf) || nf < 0 || sf < 0 || (size_t)sf + (size_t)nf > nfaces)
            return fail(FY_ERR_INVALID, "%s/boundary: patch '%s' needs nFaces and startFace inside the face list", base.c_str(), pd.first.c_str());
        if (nf == 0) continue;                               // (a patch without faces -- blockMesh's empty `defaultFaces` -- constrains nothing, whatever its type)
        if (ty == "empty" || ty == "cyclic" || ty == "wedge" || ty == "processor")
            return fail(FY_ERR_UNSUPPORTED, "%s/boundary: patch '%s' of type '%s' is not supported (wall / patch / symmetryPlane sides of a 3-D box)", base.c_str(), pd.first.c_str(), ty.c_str());
        for (int q = 0; q < nf; ++q) {
            const size_t f = (size_t)sf + (size_t)q;
            int side = -1;
            for (int a = 0; a < 3 && side < 0; ++a)
                for (int sd = 0; sd < 2 && side < 0; ++sd) {
                    bool all = true;
                    for (int m = 0; m < 4; ++m) all = all && pidx[3 * (size_t)fpt[4 * f + m] + a] == (sd ? nn[a] : 0);
                    if (all) side = 2 * a + sd;
                }
            if (side < 0) return fail(FY_ERR_UNSUPPORTED, "%s/boundary: patch '%s' has a face that is not on a side of the box", base.c_str(), pd.first.c_str());
            if (!c->patch_of_side[side].empty() && c->patch_of_side[side] != pd.first)
                return fail(FY_ERR_UNSUPPORTED, "%s/boundary: side %d of the box is shared by the patches '%s' and '%s' (one boundary condition per side)", base.c_str(), side, c->patch_of_side[side].c_str(), pd.first.c_str());
            c->patch_of_side[side] = pd.first;
            ++side_faces[(size_t)side];
        }
        if (nf > 0) c->patch_order.push_back(pd.first);
    }
    for (int sd = 0; sd < 6; ++sd) {
        const int a = sd / 2;
        const size_t want = (size_t)nn[(a + 1) % 3] * (size_t)nn[(a + 2) % 3];
        if (side_faces[(size_t)sd] != want) return fail(FY_ERR_INVALID, "%s/boundary: side %d of the box has %zu of its %zu faces in patches", base.c_str(), sd, side_faces[(size_t)sd], want);
    }
    // ---- the block: cell sizes per axis, cubes or graded
    std::vector<double> h[3];
    for (int a = 0; a < 3; ++a) for (int q = 0; q < nn[a]; ++q) h[a].push_back(ax[a][(size_t)q + 1] - ax[a][(size_t)q]);
    const double dx = (ax[0].back() - ax[0].front()) / nn[0];
    bool cubes = true;
    for (int a = 0; a < 3; ++a) for (double x : h[a]) cubes = cubes && near(x, dx, dx);
    c->desc.nx = nn[0]; c->desc.ny = nn[1]; c->desc.nz = nn[2]; c->desc.dx = dx;
    c->desc.hx = c->desc.hy = c->desc.hz = nullptr;
    if (!cubes) {
        for (int a = 0; a < 3; ++a) c->grading[a] = h[a];
        c->desc.hx = c->grading[0].data(); c->desc.hy = c->grading[1].data(); c->desc.hz = c->grading[2].data();
    }
    for (int a = 0; a < 3; ++a) c->desc.origin[a] = ax[a].front();
    return FY_OK;
}

std::string entry_text(const FoamDict& d) {
    std::string t;
    for (const std::string& k : d.order) {
        const auto* tk = d.tokens(k);
        if (!tk) continue;
        bool blob = false;
        for (const std::string& s : *tk) blob = blob || (!s.empty() && s[0] == '\x01');
        if (blob) continue;                                // (a binary list: not text; the types that need a value here take `uniform`)
        t += "        " + k;
        for (const std::string& s : *tk) t += " " + s;
        t += ";\n";
    }
    return t;
}

int read_internal(const FoamDict& f, const std::string& path, int ncomp, size_t ncell, std::vector<double>* out, const std::vector<int32_t>* file_cell = nullptr) {
    const auto* t = f.tokens("internalField");
    if (!t || t->empty()) return fail(FY_ERR_INVALID, "%s: no internalField", path.c_str());
    out->assign(ncell * (size_t)ncomp, 0.0);
    if ((*t)[0] == "uniform") {
        double v[3] = {0, 0, 0};
        if (ncomp == 1) { if (t->size() < 2 || !fy::foam_tok_is_number((*t)[1], &v[0])) return fail(FY_ERR_INVALID, "%s: malformed uniform internalField", path.c_str()); }
        else if (!f.vector3("internalField", v)) return fail(FY_ERR_INVALID, "%s: malformed uniform internalField", path.c_str());
        for (size_t c = 0; c < ncell; ++c) for (int q = 0; q < ncomp; ++q) (*out)[c * ncomp + q] = v[q];
        return FY_OK;
    }
    if ((*t)[0] == "nonuniform" && t->size() > 2) {
        std::vector<double> v;
        if (!fy::foam_read_list(*t, 2, ncomp, &v) || v.size() != ncell * (size_t)ncomp)
            return fail(FY_ERR_INVALID, "%s: nonuniform internalField does not hold %zu values", path.c_str(), ncell);
        if (file_cell && !file_cell->empty()) {           // the mesh's cell numbers -> lattice order
            for (size_t L = 0; L < ncell; ++L) for (int q = 0; q < ncomp; ++q) (*out)[L * ncomp + q] = v[(size_t)(*file_cell)[L] * ncomp + q];
        } else {
            out->swap(v);
        }
        return FY_OK;
    }
    return fail(FY_ERR_UNSUPPORTED, "%s: internalField must be 'uniform' or 'nonuniform List<...>'", path.c_str());
}

// boundaryField entries that belong to none of the six sides (a decomposed case's processor patches): kept as text, written back as they are
// A processor patch is written back with a `value` entry in any case: processorFvPatchField's dictionary constructor reads one [OF-6
// processorFvPatchField.C], so reconstructPar would refuse a file without it -- where the file's value was a binary list (not kept as text) or
// absent, `uniform 0` stands in (the patch's values are re-evaluated from the neighbour's cells on the first use)
void keep_extra_patches(const fy_foam_case* c, const FoamDict& bf, int ncomp, std::vector<std::pair<std::string, std::string> >* out) {
    out->clear();
    for (const std::string& name : bf.order) {
        bool side = false;
        for (int s = 0; s < 6; ++s) side = side || c->patch_of_side[s] == name;
        const FoamDict* pd = bf.subdict(name);
        if (side || !pd) continue;
        std::string text = entry_text(*pd);
        if (text.find("        value ") == std::string::npos) text += ncomp == 3 ? "        value uniform (0 0 0);\n" : "        value uniform 0;\n";
        out->emplace_back(name, text);
    }
}
// `value nonuniform List<...> 0()`: what decomposePar writes for a patch that has no face on this processor (every z side of the block on the ranks
// that do not touch it): the patch's value is then nobody's business here
bool empty_patch_value(const std::vector<std::string>* vt, int ncomp) {
    if (!vt || vt->size() < 3 || (*vt)[0] != "nonuniform") return false;
    std::vector<double> v;
    return fy::foam_read_list(*vt, 2, ncomp, &v) && v.empty();
}

int read_fields(fy_foam_case* c) {
    const size_t ncell = c->fcells;
    {
        const std::string path = join(c->fdir, c->start_name + "/" + c->u_name);
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 3, ncell, &c->U0, &c->file_cell));
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        keep_extra_patches(c, *bf, 3, &c->extra_patches[0]);
        for (int s = 0; s < 6; ++s) {
            const FoamDict* pd = bf->subdict(c->patch_of_side[s]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), c->patch_of_side[s].c_str());
            c->u_bc_text[s] = entry_text(*pd);
            for (int q = 0; q < 3; ++q) c->desc.u_value[s][q] = 0.0;
            if (ty == "fixedValue") {
                c->desc.u_bc[s] = FY_BC_U_FIXED_VALUE;
                const auto* vt = pd->tokens("value");
                if (empty_patch_value(vt, 3)) { /* no face of this patch on this processor */ }
                else if (!vt || vt->empty() || (*vt)[0] != "uniform" || !pd->vector3("value", c->desc.u_value[s]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform (x y z)'", path.c_str(), c->patch_of_side[s].c_str());
            } else if (ty == "noSlip") {
                c->desc.u_bc[s] = FY_BC_U_FIXED_VALUE;
            } else if (ty == "zeroGradient") {
                c->desc.u_bc[s] = FY_BC_U_ZERO_GRADIENT;
            } else if (ty == "symmetryPlane" || ty == "symmetry" || ty == "slip") {       // (one and the same on a planar patch)
                c->desc.u_bc[s] = FY_BC_U_SLIP;
            } else {
                return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': velocity boundary type '%s' is not supported (fixedValue, noSlip, zeroGradient, symmetryPlane, symmetry, slip)", path.c_str(),
                            c->patch_of_side[s].c_str(), ty.c_str());
            }
        }
    }
    {
        const std::string path = join(c->fdir, c->start_name + "/p");
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 1, ncell, &c->p0, &c->file_cell));
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        keep_extra_patches(c, *bf, 1, &c->extra_patches[1]);
        for (int s = 0; s < 6; ++s) {
            const FoamDict* pd = bf->subdict(c->patch_of_side[s]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), c->patch_of_side[s].c_str());
            c->p_bc_text[s] = entry_text(*pd);
            c->desc.p_value[s] = 0.0;
            if (ty == "zeroGradient" || ty == "symmetryPlane" || ty == "symmetry") c->desc.p_bc[s] = FY_BC_P_ZERO_GRADIENT;       // (a scalar on a symmetry plane: the cell value)
            else if (ty == "fixedFluxPressure") c->desc.p_bc[s] = FY_BC_P_FIXED_FLUX;
            else if (ty == "fixedValue") {
                c->desc.p_bc[s] = FY_BC_P_FIXED_VALUE;
                const auto* vt = pd->tokens("value");
                if (empty_patch_value(vt, 1)) { /* no face of this patch on this processor */ }
                else if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->desc.p_value[s]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform <p>'", path.c_str(), c->patch_of_side[s].c_str());
            } else {
                return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': pressure boundary type '%s' is not supported (zeroGradient, fixedValue, fixedFluxPressure)", path.c_str(),
                            c->patch_of_side[s].c_str(), ty.c_str());
            }
        }
    }
    if (c->desc.turbulence_model != FY_TURBULENCE_LAMINAR) {
        // nut.<phase> [OF-6 eddyViscosity: nut_ is MUST_READ, named with the velocity's group]; uniform or nonuniform, patches zeroGradient |
        // fixedValue (uniform) | calculated with a uniform value -- the latter only keeps its value, which is what fixedValue does here
        const std::string path = join(c->fdir, c->start_name + "/nut." + c->phase);
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 1, ncell, &c->nut0, &c->file_cell));
        c->desc.nut_initial = c->nut0.empty() ? 0.0 : c->nut0[0];
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        keep_extra_patches(c, *bf, 1, &c->extra_patches[2]);
        for (int s = 0; s < 6; ++s) {
            const FoamDict* pd = bf->subdict(c->patch_of_side[s]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), c->patch_of_side[s].c_str());
            c->nut_bc_text[s] = entry_text(*pd);
            c->desc.nut_value[s] = 0.0;
            if (ty == "zeroGradient" || ty == "symmetryPlane" || ty == "symmetry") c->desc.nut_bc[s] = FY_BC_NUT_ZERO_GRADIENT;
            else if (ty == "fixedValue") {
                c->desc.nut_bc[s] = FY_BC_NUT_FIXED_VALUE;
                const auto* vt = pd->tokens("value");
                if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->desc.nut_value[s]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform <nut>'", path.c_str(), c->patch_of_side[s].c_str());
            } else if (ty == "calculated" && (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON)) {
                c->desc.nut_bc[s] = FY_BC_NUT_CALCULATED;              // file value until the first correct(), the model's expression afterwards
                const auto* vt = pd->tokens("value");
                if (vt && vt->size() >= 2 && (*vt)[0] == "uniform") fy::foam_tok_is_number((*vt)[1], &c->desc.nut_value[s]);
            } else if (ty == "nutkWallFunction" && (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON)) {
                // [OF-6 nutkWallFunctionFvPatchScalarField]: Cmu / kappa / E may be given per patch; one set serves the case here
                c->desc.nut_bc[s] = FY_BC_WALL_FUNCTION;
                const auto* vt = pd->tokens("value");
                if (vt && vt->size() >= 2 && (*vt)[0] == "uniform") fy::foam_tok_is_number((*vt)[1], &c->desc.nut_value[s]);
                double cmu = c->desc.ras_cmu;
                pd->scalar("kappa", &c->desc.wf_kappa); pd->scalar("E", &c->desc.wf_E);
                if (pd->scalar("Cmu", &cmu) && cmu != c->desc.ras_cmu) return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': a wall-function Cmu other than the model's is not supported", path.c_str(), c->patch_of_side[s].c_str());
            } else {
                return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': nut boundary type '%s' is not supported (zeroGradient, fixedValue; calculated / nutkWallFunction with kEqn / kEpsilon)", path.c_str(),
                            c->patch_of_side[s].c_str(), ty.c_str());
            }
        }
    }
    if (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) {
        // k.<phase> [OF-6 kEqn / kEpsilon: k_ is MUST_READ]; patches zeroGradient | fixedValue (uniform); kqRWallFunction is a zeroGradient condition
        const std::string path = join(c->fdir, c->start_name + "/k." + c->phase);
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 1, ncell, &c->k0, &c->file_cell));
        c->desc.k_initial = c->k0.empty() ? 0.0 : c->k0[0];
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        keep_extra_patches(c, *bf, 1, &c->extra_patches[3]);
        for (int s = 0; s < 6; ++s) {
            const FoamDict* pd = bf->subdict(c->patch_of_side[s]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), c->patch_of_side[s].c_str());
            c->k_bc_text[s] = entry_text(*pd);
            c->desc.k_value[s] = 0.0;
            if ((ty == "zeroGradient" || ty == "symmetryPlane" || ty == "symmetry") || ty == "kqRWallFunction") c->desc.k_bc[s] = FY_BC_NUT_ZERO_GRADIENT;
            else if (ty == "fixedValue") {
                c->desc.k_bc[s] = FY_BC_NUT_FIXED_VALUE;
                const auto* vt = pd->tokens("value");
                if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->desc.k_value[s]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform <k>'", path.c_str(), c->patch_of_side[s].c_str());
            } else {
                return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': k boundary type '%s' is not supported (zeroGradient, kqRWallFunction, fixedValue)", path.c_str(),
                            c->patch_of_side[s].c_str(), ty.c_str());
            }
        }
    }
    if (c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) {
        // epsilon.<phase> [OF-6 kEpsilon: epsilon_ is MUST_READ]; zeroGradient | fixedValue (uniform) | epsilonWallFunction
        const std::string path = join(c->fdir, c->start_name + "/epsilon." + c->phase);
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 1, ncell, &c->eps0, &c->file_cell));
        c->desc.eps_initial = c->eps0.empty() ? 0.0 : c->eps0[0];
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        keep_extra_patches(c, *bf, 1, &c->extra_patches[4]);
        for (int s = 0; s < 6; ++s) {
            const FoamDict* pd = bf->subdict(c->patch_of_side[s]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), c->patch_of_side[s].c_str());
            c->eps_bc_text[s] = entry_text(*pd);
            c->desc.eps_value[s] = 0.0;
            if (ty == "zeroGradient" || ty == "symmetryPlane" || ty == "symmetry") c->desc.eps_bc[s] = FY_BC_NUT_ZERO_GRADIENT;
            else if (ty == "fixedValue") {
                c->desc.eps_bc[s] = FY_BC_NUT_FIXED_VALUE;
                const auto* vt = pd->tokens("value");
                if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->desc.eps_value[s]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform <epsilon>'", path.c_str(), c->patch_of_side[s].c_str());
            } else if (ty == "epsilonWallFunction") {
                c->desc.eps_bc[s] = FY_BC_WALL_FUNCTION;
            } else {
                return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': epsilon boundary type '%s' is not supported (zeroGradient, fixedValue, epsilonWallFunction)", path.c_str(),
                            c->patch_of_side[s].c_str(), ty.c_str());
            }
        }
    }
    return FY_OK;
}

// constant/polyMesh of ANY mesh, kept in OpenFOAM's own addressing for fy_ldu_solver [OF-6 polyMesh: points, faces (n(p0 .. pn-1) each), owner,
// neighbour (internal faces first), boundary].  The cell count is not stored in the files: it is the largest owner + 1 [OF-6 polyMesh::initMesh]
int read_general_mesh(fy_foam_case* c) {
    const std::string base = join(c->dir, "constant/polyMesh");
    std::string err;
    std::vector<double> pts;
    if (!fy::foam_numeric_list_file(join(base, "points"), &pts, &err)) return fail(err.find("not supported") != std::string::npos ? FY_ERR_UNSUPPORTED : FY_ERR_INVALID, "%s", err.c_str());
    if (pts.size() < 13 || (pts.size() - 1) % 3 != 0 || (double)((pts.size() - 1) / 3) != pts[0]) return fail(FY_ERR_INVALID, "%s/points: malformed point list", base.c_str());
    c->g_points.assign(pts.begin() + 1, pts.end());
    { std::vector<double>().swap(pts); }
    const size_t npts = c->g_points.size() / 3;
    std::vector<int32_t> fl;
    if (!fy::foam_label_list_file(join(base, "faces"), &fl, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    if (fl.size() < 2 || fl[0] < 4) return fail(FY_ERR_INVALID, "%s/faces: malformed face list", base.c_str());
    const size_t nfaces = (size_t)fl[0];
    c->g_face_off.assign(1, 0);
    c->g_face_off.reserve(nfaces + 1);
    c->g_face_pts.reserve(fl.size());
    size_t q = 1;
    for (size_t f = 0; f < nfaces; ++f) {
        if (q >= fl.size() || fl[q] < 3 || q + (size_t)fl[q] >= fl.size()) return fail(FY_ERR_INVALID, "%s/faces: face %zu is cut short or has fewer than three points", base.c_str(), f);
        const int n = fl[q++];
        for (int m = 0; m < n; ++m, ++q) {
            if (fl[q] < 0 || (size_t)fl[q] >= npts) return fail(FY_ERR_INVALID, "%s/faces: face %zu names point %d of %zu", base.c_str(), f, fl[q], npts);
            c->g_face_pts.push_back(fl[q]);
        }
        c->g_face_off.push_back((int32_t)c->g_face_pts.size());
    }
    if (q != fl.size()) return fail(FY_ERR_INVALID, "%s/faces: %zu labels follow the last of the %zu faces", base.c_str(), fl.size() - q, nfaces);
    { std::vector<int32_t>().swap(fl); }
    if (!fy::foam_label_list_file(join(base, "owner"), &c->g_own, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    if (c->g_own.empty() || (size_t)c->g_own[0] != c->g_own.size() - 1 || c->g_own.size() - 1 != nfaces) return fail(FY_ERR_INVALID, "%s/owner: %zu entries for %zu faces", base.c_str(), c->g_own.size() - 1, nfaces);
    c->g_own.erase(c->g_own.begin());
    if (!fy::foam_label_list_file(join(base, "neighbour"), &c->g_nei, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    if (c->g_nei.empty() || (size_t)c->g_nei[0] != c->g_nei.size() - 1 || c->g_nei.size() - 1 > nfaces) return fail(FY_ERR_INVALID, "%s/neighbour: malformed", base.c_str());
    c->g_nei.erase(c->g_nei.begin());
    c->g_internal = (int)c->g_nei.size();
    int32_t top = -1;
    // nCells = the highest cell label in owner OR neighbour, + 1 [OF-6 polyMesh::initMesh]: the highest-numbered cells own no internal face (owner <
    // neighbour) and, when they lie in the interior, no boundary face either -- they appear in `neighbour` only (tests/test_case_vs_oracle.py found it)
    for (int32_t v : c->g_own) top = std::max(top, v);
    for (int32_t v : c->g_nei) top = std::max(top, v);
    c->g_cells = top + 1;
    std::vector<std::string> tk;
    if (!fy::foam_list_file_tokens(join(base, "boundary"), &tk, &err)) return fail(FY_ERR_INVALID, "%s", err.c_str());
    std::vector<std::pair<std::string, FoamDict> > patches;
    FY_TRY(named_dicts(tk, base + "/boundary", &patches));
    std::vector<std::string> cyc_nbr;
    for (auto& pd : patches) {
        int nf = 0, sf = 0;
        std::string ty;
        pd.second.word("type", &ty);
        if (!pd.second.integer("nFaces", &nf) || !pd.second.integer("startFace", &sf) || nf < 0 || sf < c->g_internal || (size_t)sf + (size_t)nf > nfaces)
            return fail(FY_ERR_INVALID, "%s/boundary: patch '%s' needs nFaces and startFace among the boundary faces", base.c_str(), pd.first.c_str());
        // the patch classes fy_ldu_solver has conditions for: another constraint patch (empty, wedge, processor) would be silently treated as a wall
        if (nf > 0 && ty != "wall" && ty != "patch" && ty != "symmetryPlane" && ty != "symmetry" && ty != "cyclic")
            return fail(FY_ERR_UNSUPPORTED, "%s/boundary: patch '%s' of type '%s' is not supported on a general mesh (wall, patch, symmetryPlane, symmetry, cyclic)", base.c_str(), pd.first.c_str(), ty.c_str());
        std::string nbr, tf;
        if (ty == "cyclic") {
            if (!pd.second.word("neighbourPatch", &nbr)) return fail(FY_ERR_INVALID, "%s/boundary: cyclic patch '%s' needs neighbourPatch", base.c_str(), pd.first.c_str());
            if (pd.second.word("transform", &tf) && tf != "translational" && tf != "unknown" && tf != "noOrdering" && tf != "none")
                return fail(FY_ERR_UNSUPPORTED, "%s/boundary: cyclic patch '%s': transform '%s' is not supported (translational cyclics)", base.c_str(), pd.first.c_str(), tf.c_str());
        }
        cyc_nbr.push_back(nbr);
        c->g_patch_name.push_back(pd.first);
        c->g_patch_class.push_back(ty);
        c->g_patch_start.push_back(sf);
        c->g_patch_size.push_back(nf);
    }
    if (c->g_patch_name.empty()) return fail(FY_ERR_INVALID, "%s/boundary: no patches", base.c_str());
    c->g_patch_neighbour.assign(c->g_patch_name.size(), -1);
    for (size_t a = 0; a < cyc_nbr.size(); ++a) {
        if (cyc_nbr[a].empty()) continue;
        for (size_t b = 0; b < c->g_patch_name.size(); ++b) if (c->g_patch_name[b] == cyc_nbr[a] && b != a && c->g_patch_class[b] == "cyclic") c->g_patch_neighbour[a] = (int32_t)b;
        if (c->g_patch_neighbour[a] < 0) return fail(FY_ERR_INVALID, "%s/boundary: cyclic patch '%s': neighbourPatch '%s' is not another cyclic patch", base.c_str(), c->g_patch_name[a].c_str(), cyc_nbr[a].c_str());
    }
    c->patch_order = c->g_patch_name;
    return FY_OK;
}

// <startTime>/U and p of a general mesh: one entry per patch, the types fy_ldu_solver takes (fixedValue / noSlip / zeroGradient; zeroGradient / fixedValue)
int read_general_fields(fy_foam_case* c) {
    const size_t ncell = (size_t)c->g_cells, np = c->g_patch_name.size();
    c->g_u_bc.assign(np, FY_BC_U_FIXED_VALUE); c->g_p_bc.assign(np, FY_BC_P_ZERO_GRADIENT);
    c->g_u_val.assign(3 * np, 0.0); c->g_p_val.assign(np, 0.0);
    c->g_u_text.assign(np, std::string()); c->g_p_text.assign(np, std::string());
    for (int which = 0; which < 2; ++which) {
        const std::string path = join(c->fdir, c->start_name + "/" + (which ? std::string("p") : c->u_name));
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, which ? 1 : 3, ncell, which ? &c->p0 : &c->U0));
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        for (size_t pa = 0; pa < np; ++pa) {
            const char* pn = c->g_patch_name[pa].c_str();
            const FoamDict* pd = bf->subdict(c->g_patch_name[pa]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), pn);
            (which ? c->g_p_text : c->g_u_text)[pa] = entry_text(*pd);
            // [OF-6 fvPatchField::New]: a field's entry on a constraint patch must carry the patch's own type
            if ((c->g_patch_class[pa] == "symmetryPlane" || c->g_patch_class[pa] == "symmetry" || c->g_patch_class[pa] == "cyclic") && c->g_patch_size[pa] > 0 && ty != c->g_patch_class[pa])
                return fail(FY_ERR_INVALID, "%s: patch '%s' is a %s patch (constant/polyMesh/boundary): its entry must be of that type, not '%s'", path.c_str(), pn, c->g_patch_class[pa].c_str(), ty.c_str());
            if (ty == "cyclic") { (which ? c->g_p_bc[pa] : c->g_u_bc[pa]) = which ? FY_BC_P_ZERO_GRADIENT : FY_BC_U_ZERO_GRADIENT; continue; }      // (folded into internal faces: the codes are not used)
            const auto* vt = pd->tokens("value");
            if (!which) {
                if (ty == "fixedValue") {
                    if (!vt || vt->empty() || (*vt)[0] != "uniform" || !pd->vector3("value", &c->g_u_val[3 * pa]))
                        return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform (x y z)'", path.c_str(), pn);
                } else if (ty == "zeroGradient") c->g_u_bc[pa] = FY_BC_U_ZERO_GRADIENT;
                else if (ty == "symmetryPlane" || ty == "symmetry" || ty == "slip") c->g_u_bc[pa] = FY_BC_U_SLIP;      // (each face with its own normal: one and the same on a planar patch)
                else if (ty != "noSlip") return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': velocity boundary type '%s' is not supported on a general mesh (fixedValue, noSlip, zeroGradient, symmetryPlane, symmetry, slip)", path.c_str(), pn, ty.c_str());
            } else {
                if (ty == "fixedValue") {
                    c->g_p_bc[pa] = FY_BC_P_FIXED_VALUE;
                    if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->g_p_val[pa]))
                        return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform <p>'", path.c_str(), pn);
                } else if (ty == "fixedFluxPressure" && c->solver == FY_SOLVER_PIMPLE) c->g_p_bc[pa] = FY_BC_P_FIXED_FLUX;
                else if (ty != "zeroGradient" && ty != "symmetryPlane" && ty != "symmetry")      // (a scalar on a symmetry patch: the cell value)
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': pressure boundary type '%s' is not supported on a general mesh (zeroGradient, symmetryPlane, symmetry, fixedValue; fixedFluxPressure with pimpleFoamYade)", path.c_str(), pn, ty.c_str());
            }
        }
    }
    if (c->desc.turbulence_model != FY_TURBULENCE_LAMINAR) {
        // nut.<phase> [OF-6 eddyViscosity: MUST_READ]: patches zeroGradient | fixedValue (uniform) | calculated with a uniform value (= keeps its value: fixedValue here)
        const std::string path = join(c->fdir, c->start_name + "/nut." + c->phase);
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 1, ncell, &c->nut0));
        c->desc.nut_initial = c->nut0.empty() ? 0.0 : c->nut0[0];
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        c->g_nut_bc.assign(np, FY_BC_NUT_ZERO_GRADIENT); c->g_nut_val.assign(np, 0.0); c->g_nut_text.assign(np, std::string());
        for (size_t pa = 0; pa < np; ++pa) {
            const char* pn = c->g_patch_name[pa].c_str();
            const FoamDict* pd = bf->subdict(c->g_patch_name[pa]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), pn);
            c->g_nut_text[pa] = entry_text(*pd);
            const auto* vt = pd->tokens("value");
            if (ty == "calculated" && (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON)) {      // the file's value until the first correctNut(), the model's expression afterwards
                c->g_nut_bc[pa] = FY_BC_NUT_CALCULATED;
                if (vt && vt->size() >= 2 && (*vt)[0] == "uniform") fy::foam_tok_is_number((*vt)[1], &c->g_nut_val[pa]);
            } else if (ty == "fixedValue" || ty == "calculated") {
                c->g_nut_bc[pa] = FY_BC_NUT_FIXED_VALUE;
                if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->g_nut_val[pa]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': %s needs 'value uniform <nut>'", path.c_str(), pn, ty.c_str());
            } else if (ty != "zeroGradient" && ty != "symmetryPlane" && ty != "symmetry" && ty != "cyclic") return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': nut boundary type '%s' is not supported on a general mesh (zeroGradient, symmetryPlane, symmetry, cyclic, fixedValue, calculated)", path.c_str(), pn, ty.c_str());
        }
    }
    if (c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) {     // epsilon.<phase> [OF-6 kEpsilon: epsilon_ is MUST_READ]; zeroGradient | fixedValue (uniform); no epsilonWallFunction on a general mesh
        const std::string path = join(c->fdir, c->start_name + "/epsilon." + c->phase);
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 1, ncell, &c->eps0));
        c->desc.eps_initial = c->eps0.empty() ? 0.0 : c->eps0[0];
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        c->g_eps_bc.assign(np, FY_BC_NUT_ZERO_GRADIENT); c->g_eps_val.assign(np, 0.0); c->g_eps_text.assign(np, std::string());
        for (size_t pa = 0; pa < np; ++pa) {
            const char* pn = c->g_patch_name[pa].c_str();
            const FoamDict* pd = bf->subdict(c->g_patch_name[pa]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), pn);
            c->g_eps_text[pa] = entry_text(*pd);
            const auto* vt = pd->tokens("value");
            if (ty == "fixedValue") {
                c->g_eps_bc[pa] = FY_BC_NUT_FIXED_VALUE;
                if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->g_eps_val[pa]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform <epsilon>'", path.c_str(), pn);
            } else if (ty != "zeroGradient" && ty != "symmetryPlane" && ty != "symmetry" && ty != "cyclic")
                return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': epsilon boundary type '%s' is not supported on a general mesh (zeroGradient, symmetryPlane, symmetry, cyclic, fixedValue; the wall functions need the block solver)", path.c_str(), pn, ty.c_str());
        }
    }
    if (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) {         // k.<phase> [OF-6: k_ is MUST_READ]; patches zeroGradient | fixedValue (uniform); kqRWallFunction is a zeroGradient condition
        const std::string path = join(c->fdir, c->start_name + "/k." + c->phase);
        FoamDict f;
        FY_TRY(need_file(path, &f));
        FY_TRY(read_internal(f, path, 1, ncell, &c->k0));
        c->desc.k_initial = c->k0.empty() ? 0.0 : c->k0[0];
        const FoamDict* bf = f.subdict("boundaryField");
        if (!bf) return fail(FY_ERR_INVALID, "%s: no boundaryField", path.c_str());
        c->g_k_bc.assign(np, FY_BC_NUT_ZERO_GRADIENT); c->g_k_val.assign(np, 0.0); c->g_k_text.assign(np, std::string());
        for (size_t pa = 0; pa < np; ++pa) {
            const char* pn = c->g_patch_name[pa].c_str();
            const FoamDict* pd = bf->subdict(c->g_patch_name[pa]);
            std::string ty;
            if (!pd || !pd->word("type", &ty)) return fail(FY_ERR_INVALID, "%s: boundaryField has no (typed) entry for patch '%s'", path.c_str(), pn);
            c->g_k_text[pa] = entry_text(*pd);
            const auto* vt = pd->tokens("value");
            if (ty == "fixedValue") {
                c->g_k_bc[pa] = FY_BC_NUT_FIXED_VALUE;
                if (!vt || vt->size() < 2 || (*vt)[0] != "uniform" || !fy::foam_tok_is_number((*vt)[1], &c->g_k_val[pa]))
                    return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': fixedValue needs 'value uniform <k>'", path.c_str(), pn);
            } else if (ty != "zeroGradient" && ty != "kqRWallFunction" && ty != "symmetryPlane" && ty != "symmetry" && ty != "cyclic")
                return fail(FY_ERR_UNSUPPORTED, "%s: patch '%s': k boundary type '%s' is not supported on a general mesh (zeroGradient, kqRWallFunction, symmetryPlane, symmetry, cyclic, fixedValue)", path.c_str(), pn, ty.c_str());
        }
    }
    return FY_OK;
}

// what fy_ldu_solver discretises with (include/foamyade_hip.h): Gauss linear convection, Gauss linear corrected laplacian, corrected snGrad.  read_controls
// has refused everything the lattice solver cannot do; on a general mesh "orthogonal" / "uncorrected" are different schemes, and so are the upwinded ones
int check_general_schemes(const fy_foam_case* c) {
    const std::string path = join(c->dir, "system/fvSchemes");
    FoamDict d;
    FY_TRY(need_file(path, &d));
    for (const char* dn : {"laplacianSchemes", "snGradSchemes"}) {
        const FoamDict* sd = d.subdict(dn);
        for (const std::string& k : sd->order) {
            const auto* tk = sd->tokens(k);
            if (!tk) continue;
            bool corrected = false;
            for (const std::string& t : *tk) corrected = corrected || t == "corrected";
            if (!corrected) return fail(FY_ERR_UNSUPPORTED, "%s: %s.%s: on a general mesh the scheme must be 'corrected' (the full non-orthogonal correction, icoFoamYade.C:114-131)", path.c_str(), dn, k.c_str());
        }
    }
    return FY_OK;
}

int read_controls(fy_foam_case* c) {
    {
        const std::string path = join(c->dir, "system/controlDict");
        FoamDict d;
        FY_TRY(need_file(path, &d));
        if (!d.scalar("deltaT", &c->desc.dt) || !(c->desc.dt > 0)) return fail(FY_ERR_INVALID, "%s: deltaT missing or not positive", path.c_str());
        if (!d.scalar("endTime", &c->end_time)) return fail(FY_ERR_INVALID, "%s: endTime missing", path.c_str());
        c->start_time = 0.0; c->start_name = "0";
        std::string from;
        if (d.word("startFrom", &from) && from != "startTime" && from != "latestTime" && from != "firstTime")
            return fail(FY_ERR_UNSUPPORTED, "%s: startFrom %s is not supported (startTime, firstTime, latestTime)", path.c_str(), from.c_str());
        if (from == "latestTime" || from == "firstTime") {
            // the time directories of the case: the entries of the case directory whose names read as numbers [OF-6 Time::findTimes]
            DIR* dd = opendir(c->fdir.c_str());
            if (!dd) return fail(FY_ERR_INVALID, "%s: cannot list %s", path.c_str(), c->fdir.c_str());
            bool any = false;
            while (struct dirent* de = readdir(dd)) {
                double t;
                struct stat st;
                if (!fy::foam_tok_is_number(de->d_name, &t) || stat(join(c->fdir, de->d_name).c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) continue;
                if (!any || (from == "latestTime" ? t > c->start_time : t < c->start_time)) { c->start_time = t; c->start_name = de->d_name; }
                any = true;
            }
            closedir(dd);
            if (!any) return fail(FY_ERR_INVALID, "%s: startFrom %s: the case holds no time directory", path.c_str(), from.c_str());
        } else if (d.word("startTime", &c->start_name)) d.scalar("startTime", &c->start_time);
        std::string wc = "timeStep";
        d.word("writeControl", &wc);
        double wi = 0;
        d.scalar("writeInterval", &wi);
        if (wc == "timeStep") c->write_interval_steps = (int)wi;
        else if (wc == "runTime" || wc == "adjustableRunTime") c->write_interval_steps = (int)std::llround(wi / c->desc.dt);
        else return fail(FY_ERR_UNSUPPORTED, "%s: writeControl %s is not supported (timeStep, runTime, adjustableRunTime)", path.c_str(), wc.c_str());
        std::string wf = "ascii";
        if (d.word("writeFormat", &wf) && wf != "ascii" && wf != "binary") return fail(FY_ERR_INVALID, "%s: writeFormat %s (ascii or binary)", path.c_str(), wf.c_str());
        c->write_binary = wf == "binary";
        std::string comp;
        if (d.word("writeCompression", &comp) && comp != "off" && comp != "no" && comp != "false" && comp != "none" && comp != "uncompressed")
            return fail(FY_ERR_UNSUPPORTED, "%s: writeCompression %s is not supported (off)", path.c_str(), comp.c_str());
        double wp = 0, pw = 0;
        if (d.scalar("writePrecision", &wp)) { if (!(wp >= 1 && wp <= 17)) return fail(FY_ERR_INVALID, "%s: writePrecision %g", path.c_str(), wp); c->write_precision = (int)wp; }
        if (d.scalar("purgeWrite", &pw)) { if (pw < 0) return fail(FY_ERR_INVALID, "%s: purgeWrite %g", path.c_str(), pw); c->purge_write = (int)pw; }
        // readTimeControls.H [OF-6]: adjustTimeStep (default no), maxCo (default 1), maxDeltaT (default great).  Only pimpleFoamYade
        // includes setDeltaT.H (pimpleFoamYade.C:62-64); icoFoamYade's loop (icoFoamYade.C:65-70) never reads the switch, so there it is
        // ignored exactly as the reference ignores it
        bool adj = false;
        if (d.boolean("adjustTimeStep", &adj) && adj && c->solver == FY_SOLVER_PIMPLE) {
            if (wc != "timeStep") return fail(FY_ERR_UNSUPPORTED, "%s: adjustTimeStep with writeControl %s (output times that cut the time step) is not supported; use writeControl timeStep", path.c_str(), wc.c_str());
            c->desc.adjust_time_step = 1;
            c->desc.max_co = 1.0; c->desc.max_delta_t = 1e300;
            d.scalar("maxCo", &c->desc.max_co);
            d.scalar("maxDeltaT", &c->desc.max_delta_t);
        }
    }
    {
        const std::string path = join(c->dir, "constant/transportProperties");
        FoamDict d;
        FY_TRY(need_file(path, &d));
        if (!d.scalar("nu", &c->desc.nu)) return fail(FY_ERR_INVALID, "%s: nu missing (createFields.H: transportProperties.lookup(\"nu\"))", path.c_str());
        if (!d.scalar("partDensity", &c->desc.rho_particle)) return fail(FY_ERR_INVALID, "%s: partDensity missing (createFields.H)", path.c_str());
        c->phase.clear();
        c->u_name = "U";
        if (c->solver == FY_SOLVER_PIMPLE) {
            // pimpleFoamYade/createFields.H:3-15,91-99: continuousPhaseName and rho.<phase>
            if (!d.word("continuousPhaseName", &c->phase)) return fail(FY_ERR_INVALID, "%s: continuousPhaseName missing (pimpleFoamYade/createFields.H:3-15)", path.c_str());
            c->u_name = "U." + c->phase;
            if (!d.scalar("rho." + c->phase, &c->desc.rho_fluid)) return fail(FY_ERR_INVALID, "%s: rho.%s missing (pimpleFoamYade/createFields.H:91-99)", path.c_str(), c->phase.c_str());
        } else {
            if (!d.scalar("fluidDensity", &c->desc.rho_fluid)) return fail(FY_ERR_INVALID, "%s: fluidDensity missing (icoFoamYade/createFields.H:41-46)", path.c_str());
        }
    }
    {
        const std::string path = join(c->dir, "constant/g");
        FoamDict d;
        FY_TRY(need_file(path, &d));          // readGravitationalAcceleration.H (createFields.H:1 of both solvers)
        if (!d.vector3("value", c->desc.g)) return fail(FY_ERR_INVALID, "%s: 'value (gx gy gz)' missing", path.c_str());
    }
    if (c->solver == FY_SOLVER_PIMPLE) {
        // continuousPhaseTurbulence (pimpleFoamYade/createFields.H): PhaseIncompressibleTurbulenceModel::New reads
        // constant/turbulenceProperties.<phase> [OF-6: IOobject::groupName(turbulenceModel::propertiesName, U.group())]; the plain name is
        // accepted too, and a case without the file runs the laminar model (the reference would stop).  DPMTurbulenceModels.C:67-77
        // instantiates laminar Stokes, RAS kEpsilon, LES Smagorinsky and LES kEqn; Stokes and Smagorinsky are implemented.
        std::string path = join(c->dir, "constant/turbulenceProperties." + c->phase);
        FoamDict d;
        bool have = file_exists(path);
        if (!have) { path = join(c->dir, "constant/turbulenceProperties"); have = file_exists(path); }
        if (have) {
            FY_TRY(need_file(path, &d));
            std::string sim;
            if (!d.word("simulationType", &sim)) return fail(FY_ERR_INVALID, "%s: simulationType missing", path.c_str());
            if (sim == "laminar") {
                std::string lm;
                const FoamDict* ld = d.subdict("laminar");
                if (ld && ld->word("laminarModel", &lm) && lm != "Stokes") return fail(FY_ERR_UNSUPPORTED, "%s: laminarModel %s (DPMTurbulenceModels.C:67-68 instantiates Stokes only)", path.c_str(), lm.c_str());
            } else if (sim == "LES") {
                const FoamDict* ld = d.subdict("LES");
                std::string model, delta;
                if (!ld || !ld->word("LESModel", &model)) return fail(FY_ERR_INVALID, "%s: LES { LESModel ...; } missing", path.c_str());
                if (model != "Smagorinsky" && model != "kEqn") return fail(FY_ERR_UNSUPPORTED, "%s: LESModel %s is not implemented (DPMTurbulenceModels.C:73-77 instantiates Smagorinsky and kEqn; both are)", path.c_str(), model.c_str());
                bool on = true;
                if (ld->boolean("turbulence", &on) && !on) return fail(FY_ERR_UNSUPPORTED, "%s: 'turbulence off' (frozen nut) is not supported: use simulationType laminar", path.c_str());
                if (!ld->word("delta", &delta) || delta != "cubeRootVol") return fail(FY_ERR_UNSUPPORTED, "%s: LES delta must be cubeRootVol (got '%s')", path.c_str(), delta.c_str());
                c->desc.turbulence_model = model == "kEqn" ? FY_TURBULENCE_KEQN : FY_TURBULENCE_SMAGORINSKY;
                if (const FoamDict* sc = ld->subdict(model + "Coeffs")) { sc->scalar("Ck", &c->desc.les_ck); sc->scalar("Ce", &c->desc.les_ce); }
                if (const FoamDict* dc = ld->subdict("cubeRootVolCoeffs")) dc->scalar("deltaCoeff", &c->desc.les_delta_coeff);
            } else if (sim == "RAS") {
                const FoamDict* rd = d.subdict("RAS");
                std::string model;
                if (!rd || !rd->word("RASModel", &model)) return fail(FY_ERR_INVALID, "%s: RAS { RASModel ...; } missing", path.c_str());
                if (model != "kEpsilon") return fail(FY_ERR_UNSUPPORTED, "%s: RASModel %s is not implemented (DPMTurbulenceModels.C:70-71 instantiates kEpsilon only)", path.c_str(), model.c_str());
                bool on = true;
                if (rd->boolean("turbulence", &on) && !on) return fail(FY_ERR_UNSUPPORTED, "%s: 'turbulence off' (frozen nut) is not supported: use simulationType laminar", path.c_str());
                c->desc.turbulence_model = FY_TURBULENCE_KEPSILON;
                if (const FoamDict* kc = rd->subdict("kEpsilonCoeffs")) {
                    kc->scalar("Cmu", &c->desc.ras_cmu); kc->scalar("C1", &c->desc.ras_c1); kc->scalar("C2", &c->desc.ras_c2); kc->scalar("C3", &c->desc.ras_c3);
                    kc->scalar("sigmak", &c->desc.ras_sigmak); kc->scalar("sigmaEps", &c->desc.ras_sigmaeps);
                }
            } else {
                return fail(FY_ERR_UNSUPPORTED, "%s: simulationType %s is not one of laminar, RAS, LES", path.c_str(), sim.c_str());
            }
        }
    }
    {
        // the discretisation is fixed in this library (Euler ddt, Gauss linear grad / div / laplacian, linear
        // interpolation, orthogonal snGrad): a case that asks for anything else would be silently mis-solved, so it is refused
        const std::string path = join(c->dir, "system/fvSchemes");
        FoamDict d;
        FY_TRY(need_file(path, &d));
        struct Want { const char* dict; const char* must; const char* alt; const char* forbid; };
        const Want wants[] = {{"ddtSchemes", "Euler", nullptr, nullptr},
                              {"gradSchemes", "linear", nullptr, "Limited"},
                              {"divSchemes", "linear", "upwind", "UpwindV"},         // parsed by name below: Gauss linear | upwind | linearUpwind grad(U) | the limited schemes; not the V variants
                              {"laplacianSchemes", "linear", nullptr, nullptr},
                              {"interpolationSchemes", "linear", nullptr, "pwind"},
                              {"snGradSchemes", "corrected", "orthogonal", nullptr}};   // corrected == uncorrected == orthogonal on this mesh
        int n_div = 0;
        for (const Want& w : wants) {
            const FoamDict* sd = d.subdict(w.dict);
            if (!sd) return fail(FY_ERR_INVALID, "%s: %s missing", path.c_str(), w.dict);
            for (const std::string& k : sd->order) {
                const auto* tk = sd->tokens(k);
                if (!tk) continue;
                std::string joined;
                for (const std::string& t : *tk) joined += t + " ";
                const bool is_div = std::string(w.dict) == "divSchemes";
                // divSchemes: [bounded] Gauss <scheme> [args]; the scheme by name
                int sch = -1;
                double lim_k = 1.0;
                if (is_div && joined.find("none") != 0) {
                    size_t q = 0;
                    if (q < tk->size() && (*tk)[q] == "bounded") ++q;
                    if (q + 1 < tk->size() && (*tk)[q] == "Gauss") {
                        const std::string& nm = (*tk)[q + 1];
                        static const struct { const char* name; int id; } known[] = {
                            {"linear", FY_CONVECTION_LINEAR}, {"upwind", FY_CONVECTION_UPWIND}, {"linearUpwind", FY_CONVECTION_LINEAR_UPWIND},
                            {"limitedLinear", FY_CONVECTION_LIMITED_LINEAR}, {"vanLeer", FY_CONVECTION_VAN_LEER}, {"MUSCL", FY_CONVECTION_MUSCL},
                            {"Minmod", FY_CONVECTION_MINMOD}, {"SuperBee", FY_CONVECTION_SUPERBEE}, {"QUICK", FY_CONVECTION_QUICK}};
                        for (const auto& e : known) if (nm == e.name) sch = e.id;
                        if (sch == FY_CONVECTION_LIMITED_LINEAR && !(q + 2 < tk->size() && fy::foam_tok_is_number((*tk)[q + 2], &lim_k) && lim_k >= 0 && lim_k <= 1))
                            return fail(FY_ERR_INVALID, "%s: divSchemes.%s = '%s': limitedLinear takes a coefficient in [0, 1]", path.c_str(), k.c_str(), joined.c_str());
                    }
                }
                const bool ok = is_div ? (sch >= 0 || joined.find("none") == 0)
                              : ((joined.find(w.must) != std::string::npos || (w.alt && joined.find(w.alt) != std::string::npos) || joined.find("none") == 0) &&
                                 !(w.forbid && joined.find(w.forbid) != std::string::npos) && joined.find("limited") == std::string::npos &&
                                 joined.find("vanLeer") == std::string::npos && joined.find("QUICK") == std::string::npos);
                if (!ok) return fail(FY_ERR_UNSUPPORTED, "%s: %s.%s = '%s' is not supported (Euler; Gauss linear; div(phi,U) Gauss linear | upwind | linearUpwind | limitedLinear k | vanLeer | MUSCL | Minmod | SuperBee | QUICK)", path.c_str(), w.dict, k.c_str(), joined.c_str());
                if (is_div && joined.find("none") != 0) {
                    // only the convection terms select the scheme (icoFoamYade.C:82 div(phi,U), UcEqn.H:5-6 div(alphaPhic,Uc), or `default`);
                    // every other entry (the explicit stress term div(((alpha*nuEff)*dev2(T(grad(U))))) ...) must be plain Gauss linear
                    // (the fields' registered names are alphaPhi.<phase>, U.<phase>, k.<phase>: pimpleFoamYade/createFields.H:35-45,238-246)
                    const bool ico_term = k == "div(phi,U)";
                    const bool pimple_term = k == "div(alphaPhic,Uc)" || k == "div(phic,Uc)" || k == "div(alphaPhi." + c->phase + ",U." + c->phase + ")";
                    if ((ico_term && c->solver != FY_SOLVER_ICO) || (pimple_term && c->solver != FY_SOLVER_PIMPLE)) continue;      // (the other executable's term)
                    const bool convection = k == "default" || ico_term || pimple_term;
                    const bool k_convection = k == "div(alphaPhic,k)" || k == "div(alphaPhi." + c->phase + ",k." + c->phase + ")";
                    if (k == "div(alphaPhic,epsilon)" || k == "div(alphaPhi." + c->phase + ",epsilon." + c->phase + ")") {     // fvm::div(alphaRhoPhi, epsilon)
                        if (sch >= FY_CONVECTION_LINEAR_UPWIND) return fail(FY_ERR_UNSUPPORTED, "%s: divSchemes.%s = '%s': Gauss linear or Gauss upwind for epsilon", path.c_str(), k.c_str(), joined.c_str());
                        c->desc.eps_convection_scheme = sch;
                        continue;
                    }
                    if (k_convection) {                                  // fvm::div(alphaRhoPhi, k) of the kEqn / kEpsilon models
                        if (sch >= FY_CONVECTION_LINEAR_UPWIND) return fail(FY_ERR_UNSUPPORTED, "%s: divSchemes.%s = '%s': Gauss linear or Gauss upwind for k", path.c_str(), k.c_str(), joined.c_str());
                        c->desc.k_convection_scheme = sch;
                        continue;
                    }
                    if (!convection) {
                        if (sch != FY_CONVECTION_LINEAR) return fail(FY_ERR_UNSUPPORTED, "%s: divSchemes.%s = '%s': only the convection terms may be upwinded, this one must be Gauss linear", path.c_str(), k.c_str(), joined.c_str());
                        continue;
                    }
                    if (k == "default") { if (!n_div) { c->desc.convection_scheme = sch; c->desc.convection_limiter_k = lim_k; } c->desc.k_convection_scheme = c->desc.eps_convection_scheme = sch == FY_CONVECTION_LINEAR ? sch : FY_CONVECTION_UPWIND; continue; }      // a named convection entry overrides it
                    if (n_div++ && sch != c->desc.convection_scheme) return fail(FY_ERR_UNSUPPORTED, "%s: divSchemes mixes different convection schemes", path.c_str());
                    c->desc.convection_scheme = sch;
                    c->desc.convection_limiter_k = lim_k;
                }
            }
        }
    }
    {
        const std::string path = join(c->dir, "system/fvSolution");
        FoamDict d;
        FY_TRY(need_file(path, &d));
        const char* alg = c->solver == FY_SOLVER_PIMPLE ? "PIMPLE" : "PISO";
        const FoamDict* a = d.subdict(alg);
        if (!a) return fail(FY_ERR_INVALID, "%s: no %s dictionary", path.c_str(), alg);
        a->integer("nCorrectors", &c->desc.n_correctors);
        a->integer("nNonOrthogonalCorrectors", &c->desc.n_non_orth_correctors);
        a->integer("nOuterCorrectors", &c->desc.n_outer_correctors);
        bool mp;
        if (a->boolean("momentumPredictor", &mp)) c->desc.momentum_predictor = mp ? 1 : 0;
        a->integer("pRefCell", &c->desc.p_ref_cell);
        a->scalar("pRefValue", &c->desc.p_ref_value);
        const FoamDict* sv = d.subdict("solvers");
        if (!sv) return fail(FY_ERR_INVALID, "%s: no solvers dictionary", path.c_str());
        const FoamDict* ps = sv->subdict("p");
        if (!ps) return fail(FY_ERR_INVALID, "%s: solvers.p missing", path.c_str());
        ps->scalar("tolerance", &c->desc.p_tol); ps->scalar("relTol", &c->desc.p_rel_tol); ps->integer("maxIter", &c->desc.p_max_iter);
        std::string sname, pre;
        ps->word("solver", &sname); ps->word("preconditioner", &pre);
        c->desc.p_solver = (sname == "GAMG" || pre == "GAMG") ? FY_PSOLVER_PCG_MG : FY_PSOLVER_PCG_JACOBI;    // the two pressure solvers this library has
        c->desc.p_final_tol = c->desc.p_tol; c->desc.p_final_rel_tol = 0.0;
        if (const FoamDict* pf = sv->subdict("pFinal")) { pf->scalar("tolerance", &c->desc.p_final_tol); pf->scalar("relTol", &c->desc.p_final_rel_tol); }
        const FoamDict* us = sv->subdict(c->u_name);
        if (!us)
            for (const std::string& k : sv->order)
                if (k.find(c->u_name) != std::string::npos || (k.find("U") != std::string::npos && k.find("Final") == std::string::npos)) { us = sv->subdict(k); if (us) break; }
        if (us) { us->scalar("tolerance", &c->desc.u_tol); us->scalar("relTol", &c->desc.u_rel_tol); us->integer("maxIter", &c->desc.u_max_iter); }
        if (c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) {      // solvers.epsilon.<phase> (or a pattern naming epsilon)
            c->desc.eps_tol = c->desc.u_tol; c->desc.eps_rel_tol = c->desc.u_rel_tol; c->desc.eps_max_iter = c->desc.u_max_iter;
            const std::string en = "epsilon." + c->phase;
            const FoamDict* es = sv->subdict(en);
            if (!es)
                for (const std::string& key : sv->order)
                    if (key.find("epsilon") != std::string::npos) { es = sv->subdict(key); if (es) break; }
            if (!es) return fail(FY_ERR_INVALID, "%s: solvers has no entry for %s (kEpsilon solves a transport equation for it)", path.c_str(), en.c_str());
            es->scalar("tolerance", &c->desc.eps_tol); es->scalar("relTol", &c->desc.eps_rel_tol); es->integer("maxIter", &c->desc.eps_max_iter);
        }
        if (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) {          // solvers.k.<phase> (or a pattern naming k, e.g. "(U.water|k.water)")
            c->desc.k_tol = c->desc.u_tol; c->desc.k_rel_tol = c->desc.u_rel_tol; c->desc.k_max_iter = c->desc.u_max_iter;
            const std::string kn = "k." + c->phase;
            const FoamDict* ks = sv->subdict(kn);
            if (!ks)
                for (const std::string& key : sv->order)
                    if (key.find(kn) != std::string::npos || key.find("|k|") != std::string::npos || key.find("|k)") != std::string::npos) { ks = sv->subdict(key); if (ks) break; }
            if (!ks) return fail(FY_ERR_INVALID, "%s: solvers has no entry for %s (kEqn solves a transport equation for it)", path.c_str(), kn.c_str());
            ks->scalar("tolerance", &c->desc.k_tol); ks->scalar("relTol", &c->desc.k_rel_tol); ks->integer("maxIter", &c->desc.k_max_iter);
        }
        // relaxationFactors: equations { <U>; <U>Final; ".*" } for UcEqn.relax() (UcEqn.H:12), fields { p; pFinal } for p.relax() (pEqn.H:41).
        // No entry = the call does nothing [OF-6 fvMatrix::relax(), GeometricField::relax()]; icoFoamYade relaxes nothing.
        c->desc.u_relax = c->desc.u_relax_final = c->desc.p_relax = c->desc.p_relax_final = 0.0;
        if (const FoamDict* rf = d.subdict("relaxationFactors")) {
            if (c->solver == FY_SOLVER_PIMPLE) {
                auto lookup = [&](const FoamDict* sd, const std::string& name, double* out) {
                    if (!sd) return;
                    double v = 0;
                    if (sd->scalar(name, &v) || sd->scalar("\"" + name + "\"", &v)) { *out = v; return; }
                    for (const std::string& k : sd->order) {            // the catch-all patterns the tutorials use
                        const bool any = k == "\".*\"" || k == ".*" || k == "\"(.*)\"";
                        const bool fin = k == "\".*Final\"" || k == "\"(.*)Final\"";
                        const bool is_final = name.size() > 5 && name.compare(name.size() - 5, 5, "Final") == 0;
                        if ((fin && is_final) || (any && sd->scalar(k, &v))) { if (sd->scalar(k, &v)) *out = v; }
                    }
                };
                lookup(rf->subdict("equations"), c->u_name, &c->desc.u_relax);
                lookup(rf->subdict("equations"), c->u_name + "Final", &c->desc.u_relax_final);
                if (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) lookup(rf->subdict("equations"), "k." + c->phase, &c->desc.k_relax);
                if (c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) lookup(rf->subdict("equations"), "epsilon." + c->phase, &c->desc.eps_relax);
                lookup(rf->subdict("fields"), "p", &c->desc.p_relax);
                lookup(rf->subdict("fields"), "pFinal", &c->desc.p_relax_final);
                for (double v : {c->desc.u_relax, c->desc.u_relax_final, c->desc.p_relax, c->desc.p_relax_final})
                    if (v > 1.0) return fail(FY_ERR_INVALID, "%s: relaxationFactors above 1", path.c_str());
            }
        }
    }
    return FY_OK;
}

int write_field(const fy_foam_case* c, const std::string& tdir, const std::string& tname, const std::string& name, const char* cls, const char* dims, int ncomp,
                const std::vector<double>& v, const std::string bc_text[6], const char* default_bc,
                const std::vector<std::pair<std::string, std::string> >* extras = nullptr) {
    const std::string path = tdir + "/" + name;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return fail(FY_ERR_INVALID, "cannot write %s", path.c_str());
    const size_t n = v.size() / (size_t)ncomp;
    std::fprintf(f, "FoamFile\n{\n    version     2.0;\n    format      %s;\n", c->write_binary ? "binary" : "ascii");
    if (c->write_binary) std::fprintf(f, "    arch        \"LSB;label=32;scalar=64\";\n");
    std::fprintf(f, "    class       %s;\n    location    \"%s\";\n    object      %s;\n}\n\n", cls, tname.c_str(), name.c_str());
    std::fprintf(f, "dimensions      %s;\n\ninternalField   nonuniform List<%s> %zu\n(", dims, ncomp == 3 ? "vector" : "scalar", n);
    std::vector<double> vf;                                // lattice order -> the mesh's cell numbers
    if (!c->file_cell.empty() && n == c->file_cell.size()) {
        vf.resize(v.size());
        for (size_t L = 0; L < n; ++L) for (int q = 0; q < ncomp; ++q) vf[(size_t)c->file_cell[L] * ncomp + q] = v[L * ncomp + q];
    }
    const std::vector<double>& v_out = vf.empty() ? v : vf;
    if (c->write_binary) {
        // the list's values as they lie in memory, between the parentheses [OF-6 UList<T>::writeEntry, binary stream]
        if (std::fwrite(v_out.data(), sizeof(double), v_out.size(), f) != v_out.size()) { std::fclose(f); return fail(FY_ERR_INVALID, "cannot write %s", path.c_str()); }
    } else {
        const int pr = c->write_precision;
        std::fputc('\n', f);
        for (size_t q = 0; q < n; ++q) {
            if (ncomp == 3) std::fprintf(f, "(%.*g %.*g %.*g)\n", pr, v_out[3 * q], pr, v_out[3 * q + 1], pr, v_out[3 * q + 2]);
            else std::fprintf(f, "%.*g\n", pr, v_out[q]);
        }
    }
    std::fprintf(f, ")\n;\n\nboundaryField\n{\n");
    if (c->general) {
        const std::vector<std::string>& tx = bc_text == c->u_bc_text ? c->g_u_text : (bc_text == c->nut_bc_text ? c->g_nut_text : (bc_text == c->k_bc_text ? c->g_k_text : (bc_text == c->eps_bc_text ? c->g_eps_text : c->g_p_text)));
        for (size_t pa = 0; pa < c->g_patch_name.size(); ++pa) std::fprintf(f, "    %s\n    {\n%s    }\n", c->g_patch_name[pa].c_str(), (!bc_text || tx[pa].empty()) ? default_bc : tx[pa].c_str());
    }
    for (const std::string& pn : c->general ? std::vector<std::string>() : c->patch_order) {
        int side = -1;
        for (int s = 0; s < 6; ++s) if (c->patch_of_side[s] == pn) side = s;
        std::fprintf(f, "    %s\n    {\n%s    }\n", pn.c_str(), (bc_text && side >= 0 && !bc_text[side].empty()) ? bc_text[side].c_str() : default_bc);
    }
    // a decomposed case's processor patches, as the start time's file had them (fields without a start-time file: those of U, type only)
    if (extras) for (const auto& e : *extras) std::fprintf(f, "    %s\n    {\n%s    }\n", e.first.c_str(), e.second.c_str());
    std::fprintf(f, "}\n");
    std::fclose(f);
    return FY_OK;
}

}  // namespace

extern "C" {

static int open_case(const char* case_dir, int solver, int rank, int nranks, fy_foam_case** out) {
    if (!case_dir || !out) return fail(FY_ERR_INVALID, "fy_foam_case_open: null argument");
    if (solver != FY_SOLVER_ICO && solver != FY_SOLVER_PIMPLE) return fail(FY_ERR_INVALID, "fy_foam_case_open: solver must be FY_SOLVER_ICO or FY_SOLVER_PIMPLE");
    *out = nullptr;
    fy_foam_case* c = new (std::nothrow) fy_foam_case();
    if (!c) return fail(FY_ERR_INVALID, "out of host memory");
    c->dir = case_dir;
    c->fdir = c->dir;
    c->solver = solver;
    fy_case_defaults(&c->desc, solver);
    // the mesh: constant/polyMesh when the case has one (what the reference's solvers read), else system/blockMeshDict (what blockMesh would make of it)
    int rc = file_exists(join(c->dir, "constant/polyMesh/points")) ? read_poly_mesh(c) : read_block_mesh(c);
    if (rc == FY_ERR_UNSUPPORTED && file_exists(join(c->dir, "constant/polyMesh/points")) && file_exists(join(c->dir, "system/blockMeshDict"))) {
        // a polyMesh this reader does not take (binary, or not a lattice of hexahedra): the dictionary it was made from may still describe the block
        const std::string why = fy_last_error();
        *c = fy_foam_case();
        c->dir = case_dir; c->fdir = c->dir; c->solver = solver;
        fy_case_defaults(&c->desc, solver);
        rc = read_block_mesh(c);
        if (rc != FY_OK) rc = fail(rc, "%s (constant/polyMesh was refused first: %s)", std::string(fy_last_error()).c_str(), why.c_str());
    }
    if (rc == FY_OK) {
        c->fcells = (size_t)c->desc.nx * c->desc.ny * c->desc.nz;
        if (nranks > 0) {
            // processorR of a decomposed case = the R-th of nranks equal z-slabs (decomposePar, simple (1 1 N)), cells in the global order
            struct stat st;
            const std::string pd = join(c->dir, "processor" + std::to_string(rank));
            if (rank < 0 || rank >= nranks) rc = fail(FY_ERR_INVALID, "fy_foam_case_open_processor: rank %d of %d", rank, nranks);
            else if (stat(pd.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) rc = fail(FY_ERR_INVALID, "%s: no such processor directory (decomposePar first, or open the undecomposed case)", pd.c_str());
            else if (stat(join(c->dir, "processor" + std::to_string(nranks)).c_str(), &st) == 0) rc = fail(FY_ERR_INVALID, "%s is decomposed into more than %d parts", c->dir.c_str(), nranks);
            else if (c->desc.nz % nranks != 0) rc = fail(FY_ERR_UNSUPPORTED, "%s: %d planes do not split into %d equal z-slabs", c->dir.c_str(), c->desc.nz, nranks);
            else if (!c->file_cell.empty()) rc = fail(FY_ERR_UNSUPPORTED, "%s: a decomposed case whose mesh numbers its cells block by block is not supported", c->dir.c_str());
            else {
                c->fdir = pd; c->proc_rank = rank; c->proc_count = nranks;
                c->fcells /= (size_t)nranks; c->foffset = c->fcells * (size_t)rank;
            }
        }
    }
    if (rc == FY_OK) rc = read_controls(c);
    if (rc == FY_OK) rc = read_fields(c);
    if (rc != FY_OK) { delete c; return rc; }
    *out = c;
    return FY_OK;
}

int fy_foam_case_open(const char* case_dir, int solver, fy_foam_case** out) { return open_case(case_dir, solver, 0, 0, out); }
int fy_foam_case_open_processor(const char* case_dir, int solver, int rank, int nranks, fy_foam_case** out) {
    if (nranks < 1) return fail(FY_ERR_INVALID, "fy_foam_case_open_processor: nranks must be positive");
    return open_case(case_dir, solver, rank, nranks, out);
}

int fy_foam_case_desc(const fy_foam_case* c, fy_case_desc* out) {
    if (!c || !out) return fail(FY_ERR_INVALID, "fy_foam_case_desc: null argument");
    if (c->general) return fail(FY_ERR_UNSUPPORTED, "fy_foam_case_desc: the case holds a general mesh: fy_foam_case_poly_mesh + fy_foam_case_ldu_desc for fy_ldu_solver_create");
    *out = c->desc;
    return FY_OK;
}

int fy_foam_case_info_get(const fy_foam_case* c, fy_foam_case_info* out) {
    if (!c || !out) return fail(FY_ERR_INVALID, "fy_foam_case_info_get: null argument");
    std::memset(out, 0, sizeof(*out));
    out->start_time = c->start_time; out->end_time = c->end_time; out->delta_t = c->desc.dt;
    out->write_interval_steps = c->write_interval_steps;
    out->n_cells = c->general ? (int64_t)c->g_cells : (int64_t)c->desc.nx * c->desc.ny * c->desc.nz;
    out->field_cells = (int64_t)c->fcells; out->field_offset = (int64_t)c->foffset;
    std::snprintf(out->u_name, sizeof(out->u_name), "%s", c->u_name.c_str());
    std::snprintf(out->phase, sizeof(out->phase), "%s", c->phase.c_str());
    std::snprintf(out->start_name, sizeof(out->start_name), "%s", c->start_name.c_str());
    for (int s = 0; s < 6; ++s) std::snprintf(out->patch_of_side[s], sizeof(out->patch_of_side[s]), "%s", c->patch_of_side[s].c_str());
    return FY_OK;
}

int fy_foam_case_initial_fields(const fy_foam_case* c, double* U, double* p) {
    if (!c) return fail(FY_ERR_INVALID, "fy_foam_case_initial_fields: null case");
    if (U) std::memcpy(U, c->U0.data(), c->U0.size() * sizeof(double));
    if (p) std::memcpy(p, c->p0.data(), c->p0.size() * sizeof(double));
    return FY_OK;
}

int fy_foam_case_initial_nut(const fy_foam_case* c, double* nut) {
    if (!c || !nut) return fail(FY_ERR_INVALID, "fy_foam_case_initial_nut: null argument");
    if (c->nut0.empty()) return fail(FY_ERR_INVALID, "fy_foam_case_initial_nut: the case has no turbulence model");
    std::memcpy(nut, c->nut0.data(), c->nut0.size() * sizeof(double));
    return FY_OK;
}

int fy_foam_case_initial_k(const fy_foam_case* c, double* k) {
    if (!c || !k) return fail(FY_ERR_INVALID, "fy_foam_case_initial_k: null argument");
    if (c->k0.empty()) return fail(FY_ERR_INVALID, "fy_foam_case_initial_k: the case has no k equation");
    std::memcpy(k, c->k0.data(), c->k0.size() * sizeof(double));
    return FY_OK;
}

int fy_foam_case_initial_epsilon(const fy_foam_case* c, double* eps) {
    if (!c || !eps) return fail(FY_ERR_INVALID, "fy_foam_case_initial_epsilon: null argument");
    if (c->eps0.empty()) return fail(FY_ERR_INVALID, "fy_foam_case_initial_epsilon: the case has no epsilon equation");
    std::memcpy(eps, c->eps0.data(), c->eps0.size() * sizeof(double));
    return FY_OK;
}

int fy_foam_case_write_fields(const fy_foam_case* c, const char* time_name, const double* U, const double* p, const double* alpha, const double* nut,
                              const double* k, const double* epsilon) {
    if (!c || !time_name || !*time_name || !U || !p) return fail(FY_ERR_INVALID, "fy_foam_case_write_fields: null argument");
    const std::string tdir = join(c->fdir, time_name);
    if (mkdir(tdir.c_str(), 0777) != 0) {
        struct stat st;
        if (stat(tdir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return fail(FY_ERR_INVALID, "cannot create %s", tdir.c_str());
    }
    const size_t n = c->fcells;
    // (a field the start time had no file for -- alpha -- takes the processor patches of U with their type alone)
    std::vector<std::pair<std::string, std::string> > bare;
    for (const auto& e : c->extra_patches[0]) {
        const size_t a = e.second.find("type");
        const size_t b = a == std::string::npos ? a : e.second.find('\n', a);
        bare.emplace_back(e.first, (a == std::string::npos ? e.second : e.second.substr(0, b + 1)) + "        value uniform 1;\n");      // (alphac = 1 where nothing was deposited)
    }
    const char* zg = "        type            zeroGradient;\n";
    auto vec = [&](const double* src, int nc) { return std::vector<double>(src, src + n * (size_t)nc); };
    FY_TRY(write_field(c, tdir, time_name, c->u_name, "volVectorField", "[0 1 -1 0 0 0 0]", 3, vec(U, 3), c->u_bc_text, zg, &c->extra_patches[0]));
    FY_TRY(write_field(c, tdir, time_name, "p", "volScalarField", "[0 2 -2 0 0 0 0]", 1, vec(p, 1), c->p_bc_text, zg, &c->extra_patches[1]));
    // alphac is AUTO_WRITE (pimpleFoamYade/createFields.H:139-150) and is written BEFORE setSourceZero resets it (pimpleFoamYade.C:106-108):
    // run the solver with fy_solver_hold_sources(s, 1) to get that
    if (c->solver == FY_SOLVER_PIMPLE && alpha) FY_TRY(write_field(c, tdir, time_name, "alpha." + c->phase, "volScalarField", "[0 0 0 0 0 0 0]", 1, vec(alpha, 1), nullptr, zg, &bare));
    if (c->desc.turbulence_model != FY_TURBULENCE_LAMINAR && nut)          // eddyViscosity::nut_ is AUTO_WRITE
        FY_TRY(write_field(c, tdir, time_name, "nut." + c->phase, "volScalarField", "[0 2 -1 0 0 0 0]", 1, vec(nut, 1), c->nut_bc_text, zg, &c->extra_patches[2]));
    if (c->desc.turbulence_model == FY_TURBULENCE_KEPSILON && epsilon)
        FY_TRY(write_field(c, tdir, time_name, "epsilon." + c->phase, "volScalarField", "[0 2 -3 0 0 0 0]", 1, vec(epsilon, 1), c->eps_bc_text, zg, &c->extra_patches[4]));
    if ((c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) && k)
        FY_TRY(write_field(c, tdir, time_name, "k." + c->phase, "volScalarField", "[0 2 -2 0 0 0 0]", 1, vec(k, 1), c->k_bc_text, zg, &c->extra_patches[3]));
    if (c->purge_write > 0) {
        // purgeWrite [OF-6 Time::writeObject]: once more than N time directories have been written, the oldest of them goes
        bool known = false;
        for (const std::string& w : c->written) known = known || w == time_name;
        if (!known) c->written.push_back(time_name);
        while ((int)c->written.size() > c->purge_write) {
            const std::string old = join(c->fdir, c->written.front());
            c->written.erase(c->written.begin());
            if (DIR* dd = opendir(old.c_str())) {
                while (struct dirent* de = readdir(dd)) {
                    const std::string nm = de->d_name;
                    if (nm != "." && nm != "..") ::unlink(join(old, nm).c_str());
                }
                closedir(dd);
                ::rmdir(old.c_str());
            }
        }
    }
    return FY_OK;
}

int fy_foam_case_write_time(const fy_foam_case* c, fy_solver* s, const char* time_name) {
    if (!c || !s || !time_name || !*time_name) return fail(FY_ERR_INVALID, "fy_foam_case_write_time: null argument");
    const size_t n = c->fcells;
    int64_t cnt = 0;
    FY_TRY(fy_solver_field_count(s, "p", &cnt));
    if ((size_t)cnt != n) return fail(FY_ERR_UNSUPPORTED, "fy_foam_case_write_time: the solver holds %lld cells, the case's field files %zu (a slab of an undecomposed case: gather the slabs and use fy_foam_case_write_fields)", (long long)cnt, n);
    std::vector<double> U(3 * n), p(n), a, nt, kk, ee;
    FY_TRY(fy_solver_read_field_host(s, "U", U.data()));
    FY_TRY(fy_solver_read_field_host(s, "p", p.data()));
    if (c->solver == FY_SOLVER_PIMPLE) { a.resize(n); FY_TRY(fy_solver_read_field_host(s, "alpha", a.data())); }
    if (c->desc.turbulence_model != FY_TURBULENCE_LAMINAR) { nt.resize(n); FY_TRY(fy_solver_read_field_host(s, "nut", nt.data())); }
    if (c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) { ee.resize(n); FY_TRY(fy_solver_read_field_host(s, "epsilon", ee.data())); }
    if (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) { kk.resize(n); FY_TRY(fy_solver_read_field_host(s, "k", kk.data())); }
    return fy_foam_case_write_fields(c, time_name, U.data(), p.data(), a.empty() ? nullptr : a.data(), nt.empty() ? nullptr : nt.data(), kk.empty() ? nullptr : kk.data(),
                                     ee.empty() ? nullptr : ee.data());
}

int fy_foam_case_open_general(const char* case_dir, int solver, fy_foam_case** out) {
    if (!case_dir || !out) return fail(FY_ERR_INVALID, "fy_foam_case_open_general: null argument");
    if (solver != FY_SOLVER_ICO && solver != FY_SOLVER_PIMPLE) return fail(FY_ERR_INVALID, "fy_foam_case_open_general: solver must be FY_SOLVER_ICO or FY_SOLVER_PIMPLE");
    *out = nullptr;
    fy_foam_case* c = new (std::nothrow) fy_foam_case();
    if (!c) return fail(FY_ERR_INVALID, "out of host memory");
    c->dir = case_dir; c->fdir = c->dir; c->solver = solver; c->general = true;
    fy_case_defaults(&c->desc, solver);
    int rc = read_general_mesh(c);
    if (rc == FY_OK) { c->fcells = (size_t)c->g_cells; rc = read_controls(c); }
    if (rc == FY_OK) rc = check_general_schemes(c);
    if (rc == FY_OK) rc = read_general_fields(c);
    if (rc != FY_OK) { delete c; return rc; }
    *out = c;
    return FY_OK;
}

int fy_foam_case_poly_mesh(const fy_foam_case* c, fy_poly_mesh* out) {
    if (!c || !out) return fail(FY_ERR_INVALID, "fy_foam_case_poly_mesh: null argument");
    if (!c->general) return fail(FY_ERR_INVALID, "fy_foam_case_poly_mesh: the case was opened as a block (fy_foam_case_open): use fy_foam_case_desc, or open it with fy_foam_case_open_general");
    std::memset(out, 0, sizeof(*out));
    out->n_points = (int32_t)(c->g_points.size() / 3); out->points = c->g_points.data();
    out->n_faces = (int32_t)c->g_own.size(); out->n_internal_faces = c->g_internal;
    out->face_offsets = c->g_face_off.data(); out->face_points = c->g_face_pts.data();
    out->owner = c->g_own.data(); out->neighbour = c->g_nei.data();
    out->n_cells = c->g_cells;
    out->n_patches = (int32_t)c->g_patch_name.size(); out->patch_start = c->g_patch_start.data(); out->patch_size = c->g_patch_size.data();
    out->patch_neighbour = nullptr;
    for (int32_t v : c->g_patch_neighbour) if (v >= 0) out->patch_neighbour = c->g_patch_neighbour.data();
    return FY_OK;
}

int fy_foam_case_ldu_desc(const fy_foam_case* c, fy_ldu_case* out) {
    if (!c || !out) return fail(FY_ERR_INVALID, "fy_foam_case_ldu_desc: null argument");
    if (!c->general) return fail(FY_ERR_INVALID, "fy_foam_case_ldu_desc: the case was opened as a block (fy_foam_case_open)");
    fy_ldu_case_defaults(out);
    const fy_case_desc& d = c->desc;
    out->dt = d.dt; out->nu = d.nu; out->rho_fluid = d.rho_fluid; out->rho_particle = d.rho_particle;
    out->n_correctors = d.n_correctors; out->n_non_orth_correctors = d.n_non_orth_correctors; out->momentum_predictor = d.momentum_predictor;
    out->p_ref_cell = d.p_ref_cell; out->p_ref_value = d.p_ref_value;
    out->p_tol = d.p_tol; out->p_rel_tol = d.p_rel_tol; out->p_final_tol = d.p_final_tol; out->p_final_rel_tol = d.p_final_rel_tol; out->p_max_iter = d.p_max_iter;
    out->u_tol = d.u_tol; out->u_rel_tol = d.u_rel_tol; out->u_max_iter = d.u_max_iter; out->p_solver = d.p_solver;
    out->solver = c->solver; out->n_outer_correctors = d.n_outer_correctors;
    for (int a = 0; a < 3; ++a) out->g[a] = d.g[a];
    out->u_relax = d.u_relax; out->u_relax_final = d.u_relax_final; out->p_relax = d.p_relax; out->p_relax_final = d.p_relax_final;
    out->adjust_time_step = d.adjust_time_step; out->max_co = d.max_co; out->max_delta_t = d.max_delta_t;
    out->turbulence_model = d.turbulence_model; out->les_ck = d.les_ck; out->les_ce = d.les_ce; out->les_delta_coeff = d.les_delta_coeff; out->nut_initial = d.nut_initial;
    out->nut_bc = c->g_nut_bc.empty() ? nullptr : c->g_nut_bc.data(); out->nut_value = c->g_nut_val.empty() ? nullptr : c->g_nut_val.data();
    out->convection_scheme = d.convection_scheme; out->convection_limiter_k = d.convection_limiter_k;
    out->k_initial = d.k_initial; out->k_bc = c->g_k_bc.empty() ? nullptr : c->g_k_bc.data(); out->k_value = c->g_k_val.empty() ? nullptr : c->g_k_val.data();
    out->k_convection_scheme = d.k_convection_scheme; out->k_tol = d.k_tol; out->k_rel_tol = d.k_rel_tol; out->k_max_iter = d.k_max_iter; out->k_relax = d.k_relax;
    out->ras_cmu = d.ras_cmu; out->ras_c1 = d.ras_c1; out->ras_c2 = d.ras_c2; out->ras_c3 = d.ras_c3; out->ras_sigmak = d.ras_sigmak; out->ras_sigmaeps = d.ras_sigmaeps;
    out->eps_initial = d.eps_initial; out->eps_bc = c->g_eps_bc.empty() ? nullptr : c->g_eps_bc.data(); out->eps_value = c->g_eps_val.empty() ? nullptr : c->g_eps_val.data();
    out->eps_convection_scheme = d.eps_convection_scheme; out->eps_tol = d.eps_tol; out->eps_rel_tol = d.eps_rel_tol; out->eps_max_iter = d.eps_max_iter; out->eps_relax = d.eps_relax;
    out->u_bc = c->g_u_bc.data(); out->u_value = c->g_u_val.data(); out->p_bc = c->g_p_bc.data(); out->p_value = c->g_p_val.data();
    return FY_OK;
}

int fy_foam_case_patch_name(const fy_foam_case* c, int patch, char* out, int cap) {
    if (!c || !out || cap < 1) return fail(FY_ERR_INVALID, "fy_foam_case_patch_name: null argument");
    const std::vector<std::string>& names = c->general ? c->g_patch_name : c->patch_order;
    if (patch < 0 || (size_t)patch >= names.size()) return fail(FY_ERR_INVALID, "fy_foam_case_patch_name: patch %d of %zu", patch, names.size());
    std::snprintf(out, (size_t)cap, "%s", names[(size_t)patch].c_str());
    return FY_OK;
}

int fy_foam_case_write_time_ldu(const fy_foam_case* c, fy_ldu_solver* s, const char* time_name) {
    if (!c || !s || !time_name || !*time_name) return fail(FY_ERR_INVALID, "fy_foam_case_write_time_ldu: null argument");
    if (!c->general) return fail(FY_ERR_INVALID, "fy_foam_case_write_time_ldu: the case was opened as a block (fy_foam_case_open)");
    int64_t cnt = 0;
    FY_TRY(fy_ldu_solver_field_count(s, "p", &cnt));
    if ((size_t)cnt != c->fcells) return fail(FY_ERR_INVALID, "fy_foam_case_write_time_ldu: the solver holds %lld cells, the case %zu", (long long)cnt, c->fcells);
    std::vector<double> U(3 * c->fcells), p(c->fcells), a;
    FY_TRY(fy_ldu_solver_read_field_host(s, "U", U.data()));
    FY_TRY(fy_ldu_solver_read_field_host(s, "p", p.data()));
    std::vector<double> nt;
    if (c->solver == FY_SOLVER_PIMPLE) { a.resize(c->fcells); FY_TRY(fy_ldu_solver_read_field_host(s, "alpha", a.data())); }       // (with fy_ldu_solver_hold_sources: before setSourceZero)
    if (c->desc.turbulence_model != FY_TURBULENCE_LAMINAR) { nt.resize(c->fcells); FY_TRY(fy_ldu_solver_read_field_host(s, "nut", nt.data())); }
    std::vector<double> kt;
    std::vector<double> et;
    if (c->desc.turbulence_model == FY_TURBULENCE_KEQN || c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) { kt.resize(c->fcells); FY_TRY(fy_ldu_solver_read_field_host(s, "k", kt.data())); }
    if (c->desc.turbulence_model == FY_TURBULENCE_KEPSILON) { et.resize(c->fcells); FY_TRY(fy_ldu_solver_read_field_host(s, "epsilon", et.data())); }
    return fy_foam_case_write_fields(c, time_name, U.data(), p.data(), a.empty() ? nullptr : a.data(), nt.empty() ? nullptr : nt.data(), kt.empty() ? nullptr : kt.data(), et.empty() ? nullptr : et.data());
}

int fy_foam_case_close(fy_foam_case* c) {
    delete c;
    return FY_OK;
}

}  // extern "C"
