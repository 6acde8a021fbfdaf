// Host side of fy_ldu_solver: OpenFOAM's mesh geometry from constant/polyMesh's arrays (what createMesh.H builds before icoFoamYade.C:42 hands
// the mesh to everything else) [OF-6, restated: primitiveMeshFaceCentresAndAreas.C, primitiveMeshCellCentresAndVols.C, surfaceInterpolation.C]:
//   face centre / area vector   triangle decomposition about the face's point average (a triangle: its own centroid and half cross product)
//   cell centre / volume        pyramid decomposition about the average of the cell's face centres
//   linear weights              w = |Sf.(C_N - Cf)| / (|Sf.(Cf - C_P)| + |Sf.(C_N - Cf)|)
//   nonOrthDeltaCoeffs          1 / max(n.d, 0.05 |d|), d = C_N - C_P; boundary faces: 1 / (n.(Cf - C_P))
//   nonOrthCorrectionVectors    n - d nonOrthDeltaCoeffs (internal faces; none on non-coupled boundary faces)
// Cyclic patches (fy_poly_mesh.patch_neighbour; translational, the two halves' faces matched one to one in order [OF-6 cyclicPolyPatch, cyclicFvPatch]) are FOLDED
// before anything else: the pair (face i of A, face i of B) becomes ONE more internal face between the two cells behind it -- owner the lower-numbered cell, the face's
// points those of that cell's half --, numbered after the mesh's own internal faces; the neighbour cell is seen at its IMAGE, C_N + sep_f with sep_f = Cf(own half) -
// Cf(other half) [OF-6 cyclicFvPatch::delta: patchD - nbrPatchD; makeWeights: the two normal distances], which is all a coupled patch differs by in weights,
// nonOrthDeltaCoeffs and correction vectors.  From then on every operator treats it like any internal face: implicit in the matrices, no patch code.
#include <algorithm>
#include <cmath>

#include "ldu.hpp"

namespace fy {

namespace {
struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double mag(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 at(const double* p, int q) { return {p[3 * (size_t)q], p[3 * (size_t)q + 1], p[3 * (size_t)q + 2]}; }
inline void put(std::vector<double>& v, int q, V3 a) { v[3 * (size_t)q] = a.x; v[3 * (size_t)q + 1] = a.y; v[3 * (size_t)q + 2] = a.z; }
const double VSMALL = 1e-300;
// centre and area vector of a face [OF-6 primitiveMeshFaceCentresAndAreas.C]
void face_geometry(const double* pts, const int32_t* q, int n, V3* cOut, V3* sOut) {
    V3 c, S;
    if (n == 3) {
        c = (1.0 / 3.0) * (at(pts, q[0]) + at(pts, q[1]) + at(pts, q[2]));
        S = 0.5 * cross(at(pts, q[1]) - at(pts, q[0]), at(pts, q[2]) - at(pts, q[0]));
    } else {
        V3 fc{0, 0, 0};
        for (int a = 0; a < n; ++a) fc = fc + at(pts, q[a]);
        fc = (1.0 / n) * fc;
        V3 sumN{0, 0, 0}, sumAc{0, 0, 0};
        double sumA = 0.0;
        for (int a = 0; a < n; ++a) {
            const V3 p0 = at(pts, q[a]), p1 = at(pts, q[(a + 1) % n]);
            const V3 nn = cross(p1 - p0, fc - p0);
            const double aa = mag(nn);
            sumN = sumN + nn; sumA += aa; sumAc = sumAc + aa * (p0 + p1 + fc);
        }
        c = sumA < VSMALL ? fc : (1.0 / 3.0) * ((1.0 / sumA) * sumAc);
        S = 0.5 * sumN;
    }
    *cOut = c; *sOut = S;
}
}  // namespace

// the cyclic pairs of m folded into internal faces: the arrays of the folded mesh (held by the LduHostMesh), *m re-pointed at them
int LduHostMesh::fold_cyclics(fy_poly_mesh* m) {
    const int np = m->n_patches, nI = m->n_internal_faces, nF = m->n_faces;
    std::vector<char> is_cyc((size_t)np, 0);
    int nCyc = 0;
    for (int a = 0; a < np; ++a) {
        const int b = m->patch_neighbour[a];
        if (b < 0) continue;
        if (b >= np || b == a || m->patch_neighbour[b] != a) return fail(FY_ERR_INVALID, "fy_poly_mesh: patch %d names patch %d as its cyclic neighbour, which does not name it back", a, b);
        if (m->patch_size[a] != m->patch_size[b]) return fail(FY_ERR_INVALID, "fy_poly_mesh: cyclic patches %d and %d have %d and %d faces", a, b, m->patch_size[a], m->patch_size[b]);
        is_cyc[(size_t)a] = 1;
        if (a < b) nCyc += m->patch_size[a];
    }
    for (int a = 0; a < np; ++a)
        for (int q = 0; q < m->patch_size[a]; ++q) { const int f = m->patch_start[a] + q; if (f < nI || f >= nF) return fail(FY_ERR_INVALID, "fy_poly_mesh: patch %d does not hold boundary faces of its own", a); }
    n_real_internal = nI;
    f_off.clear(); f_pts.clear(); f_own.clear(); f_nei.clear(); sep.clear(); orig_face.clear();
    f_off.push_back(0);
    auto add_face = [&](int f) {
        for (int q = m->face_offsets[f]; q < m->face_offsets[f + 1]; ++q) f_pts.push_back(m->face_points[q]);
        f_off.push_back((int32_t)f_pts.size());
        orig_face.push_back(f);
    };
    for (int f = 0; f < nI; ++f) { add_face(f); f_own.push_back(m->owner[f]); f_nei.push_back(m->neighbour[f]); }
    sep.assign(3 * (size_t)(nI + nCyc), 0.0);
    for (int a = 0; a < np; ++a) {
        const int b = m->patch_neighbour[a];
        if (b < 0 || b < a) continue;
        for (int q = 0; q < m->patch_size[a]; ++q) {
            const int fA = m->patch_start[a] + q, fB = m->patch_start[b] + q, cA = m->owner[fA], cB = m->owner[fB];
            for (int f : {fA, fB}) {
                if (m->face_offsets[f + 1] - m->face_offsets[f] < 3) return fail(FY_ERR_INVALID, "fy_poly_mesh: face %d has fewer than three points", f);
                for (int e = m->face_offsets[f]; e < m->face_offsets[f + 1]; ++e) if (m->face_points[e] < 0 || m->face_points[e] >= m->n_points) return fail(FY_ERR_INVALID, "fy_poly_mesh: face %d names point %d", f, m->face_points[e]);
            }
            if (cA < 0 || cA >= m->n_cells || cB < 0 || cB >= m->n_cells) return fail(FY_ERR_INVALID, "fy_poly_mesh: owner of face %d out of range", fA);
            if (cA == cB) return fail(FY_ERR_UNSUPPORTED, "fy_poly_mesh: cell %d lies on both halves of a cyclic pair (patches %d and %d): one cell between them is its own neighbour", cA, a, b);
            V3 ca, sa, cb, sb;
            face_geometry(m->points, m->face_points + m->face_offsets[fA], m->face_offsets[fA + 1] - m->face_offsets[fA], &ca, &sa);
            face_geometry(m->points, m->face_points + m->face_offsets[fB], m->face_offsets[fB + 1] - m->face_offsets[fB], &cb, &sb);
            if (!(mag(sa + sb) <= 1e-6 * mag(sa)))
                return fail(FY_ERR_UNSUPPORTED, "fy_poly_mesh: face %d of cyclic patch %d and face %d of patch %d are not translates of each other (translational cyclics whose faces match one to one, in order)", q, a, q, b);
            const bool baseA = cA < cB;
            const size_t nf = f_own.size();
            add_face(baseA ? fA : fB);
            f_own.push_back(baseA ? cA : cB); f_nei.push_back(baseA ? cB : cA);
            const V3 sp = baseA ? ca - cb : cb - ca;                 // the neighbour's image = C_N + sep
            sep[3 * nf] = sp.x; sep[3 * nf + 1] = sp.y; sep[3 * nf + 2] = sp.z;
        }
    }
    f_pstart.assign((size_t)np, 0); f_psize.assign((size_t)np, 0);
    for (int a = 0; a < np; ++a) {
        f_pstart[(size_t)a] = (int32_t)f_own.size();
        if (is_cyc[(size_t)a]) continue;
        f_psize[(size_t)a] = m->patch_size[a];
        for (int q = 0; q < m->patch_size[a]; ++q) { const int f = m->patch_start[a] + q; add_face(f); f_own.push_back(m->owner[f]); }
    }
    if ((int)f_own.size() != nF - nCyc) return fail(FY_ERR_INVALID, "fy_poly_mesh: the patches do not cover the boundary faces once each");
    m->n_faces = (int32_t)f_own.size(); m->n_internal_faces = nI + nCyc;
    m->face_offsets = f_off.data(); m->face_points = f_pts.data(); m->owner = f_own.data(); m->neighbour = f_nei.data();
    m->patch_start = f_pstart.data(); m->patch_size = f_psize.data(); m->patch_neighbour = nullptr;
    return FY_OK;
}

int LduHostMesh::build(const fy_poly_mesh* m_in) {
    fy_poly_mesh folded, *m = nullptr;
    if (m_in) { folded = *m_in; m = &folded; }
    if (!m || m->n_points < 4 || m->n_faces < 4 || m->n_cells < 1 || m->n_internal_faces < 0 || m->n_internal_faces > m->n_faces || !m->points || !m->face_offsets ||
        !m->face_points || !m->owner || (m->n_internal_faces > 0 && !m->neighbour) || m->n_patches < 1 || !m->patch_start || !m->patch_size)
        return fail(FY_ERR_INVALID, "fy_poly_mesh: missing arrays or empty mesh");
    sep.clear(); orig_face.clear(); n_real_internal = m->n_internal_faces;
    bool any_cyclic = false;
    if (m->patch_neighbour) for (int a = 0; a < m->n_patches; ++a) any_cyclic = any_cyclic || m->patch_neighbour[a] >= 0;
    if (any_cyclic) FY_TRY(fold_cyclics(m));
    nPoints = m->n_points; nFaces = m->n_faces; nInt = m->n_internal_faces; nCells = m->n_cells; nPatches = m->n_patches;
    own.assign(m->owner, m->owner + nFaces);
    nei.assign(m->neighbour, m->neighbour + nInt);
    for (int f = 0; f < nFaces; ++f) {
        if (own[f] < 0 || own[f] >= nCells) return fail(FY_ERR_INVALID, "fy_poly_mesh: owner of face %d out of range", f);
        if (f < nInt && (nei[f] <= own[f] || nei[f] >= nCells)) return fail(FY_ERR_INVALID, "fy_poly_mesh: internal face %d needs owner < neighbour < n_cells", f);
        const int n = m->face_offsets[f + 1] - m->face_offsets[f];
        if (n < 3) return fail(FY_ERR_INVALID, "fy_poly_mesh: face %d has fewer than three points", f);
        for (int a = 0; a < n; ++a) { const int q = m->face_points[m->face_offsets[f] + a]; if (q < 0 || q >= nPoints) return fail(FY_ERR_INVALID, "fy_poly_mesh: face %d names point %d", f, q); }
    }
    patch_of.assign((size_t)(nFaces - nInt), -1);
    for (int pa = 0; pa < nPatches; ++pa)
        for (int q = 0; q < m->patch_size[pa]; ++q) {
            const int f = m->patch_start[pa] + q;
            if (f < nInt || f >= nFaces || patch_of[(size_t)(f - nInt)] >= 0) return fail(FY_ERR_INVALID, "fy_poly_mesh: patch %d does not hold boundary faces of its own", pa);
            patch_of[(size_t)(f - nInt)] = pa;
        }
    for (int32_t v : patch_of) if (v < 0) return fail(FY_ERR_INVALID, "fy_poly_mesh: a boundary face belongs to no patch");
    // ---- faces
    Cf.assign(3 * (size_t)nFaces, 0.0); Sf = Cf; magSf.assign((size_t)nFaces, 0.0);
    for (int f = 0; f < nFaces; ++f) {
        const int n = m->face_offsets[f + 1] - m->face_offsets[f];
        const int32_t* q = m->face_points + m->face_offsets[f];
        V3 c, S;
        face_geometry(m->points, q, n, &c, &S);
        put(Cf, f, c); put(Sf, f, S); magSf[(size_t)f] = mag(S);
        if (!(magSf[(size_t)f] > 0)) return fail(FY_ERR_INVALID, "fy_poly_mesh: face %d has no area", f);
    }
    // ---- cell -> faces
    cf_off.assign((size_t)nCells + 1, 0);
    for (int f = 0; f < nFaces; ++f) { ++cf_off[(size_t)own[f] + 1]; if (f < nInt) ++cf_off[(size_t)nei[f] + 1]; }
    for (int c = 0; c < nCells; ++c) { if (cf_off[(size_t)c + 1] < 4) return fail(FY_ERR_INVALID, "fy_poly_mesh: cell %d has fewer than four faces", c); cf_off[(size_t)c + 1] += cf_off[(size_t)c]; }
    cf_face.assign((size_t)cf_off[(size_t)nCells], 0);
    {
        std::vector<int32_t> fill(cf_off.begin(), cf_off.end() - 1);
        for (int f = 0; f < nFaces; ++f) { cf_face[(size_t)fill[(size_t)own[f]]++] = f; if (f < nInt) cf_face[(size_t)fill[(size_t)nei[f]]++] = f; }
    }
    Wall = 0;
    for (int c = 0; c < nCells; ++c) Wall = std::max(Wall, cf_off[(size_t)c + 1] - cf_off[(size_t)c]);
    ef.assign((size_t)Wall * nCells, -1); en.assign((size_t)Wall * nCells, -2);      // (pads: face -1, neighbour -2; a boundary face's neighbour is -1)
    for (int c = 0; c < nCells; ++c)
        for (int32_t q = cf_off[(size_t)c]; q < cf_off[(size_t)c + 1]; ++q) {
            const int f = cf_face[(size_t)q];
            const size_t e = (size_t)(q - cf_off[(size_t)c]) * nCells + c;
            ef[e] = f;
            en[e] = f < nInt ? (own[f] == c ? nei[f] : own[f]) : -1;
        }
    // ---- cells (a folded cyclic face seen from its neighbour cell lies at that cell's own half: Cf - sep)
    auto face_centre_seen_from = [&](int f, int c) {
        V3 fc = at(Cf.data(), f);
        if (!sep.empty() && f >= n_real_internal && f < nInt && nei[(size_t)f] == c) fc = fc - at(sep.data(), f);
        return fc;
    };
    std::vector<double> cEst(3 * (size_t)nCells, 0.0);
    for (int c = 0; c < nCells; ++c) {
        V3 e{0, 0, 0};
        for (int32_t q = cf_off[(size_t)c]; q < cf_off[(size_t)c + 1]; ++q) e = e + face_centre_seen_from(cf_face[(size_t)q], c);
        put(cEst, c, (1.0 / (cf_off[(size_t)c + 1] - cf_off[(size_t)c])) * e);
    }
    C.assign(3 * (size_t)nCells, 0.0); V.assign((size_t)nCells, 0.0);
    auto pyramid = [&](int c, int f, double sgn) {
        const V3 fc = face_centre_seen_from(f, c), ce = at(cEst.data(), c);
        const double pyr3 = std::max(sgn * dot(at(Sf.data(), f), fc - ce), VSMALL);
        const V3 pc = 0.75 * fc + 0.25 * ce;
        put(C, c, at(C.data(), c) + pyr3 * pc);
        V[(size_t)c] += pyr3;
    };
    for (int f = 0; f < nFaces; ++f) { pyramid(own[f], f, 1.0); if (f < nInt) pyramid(nei[f], f, -1.0); }
    for (int a = 0; a < 3; ++a) { bbox_min[a] = 1e300; bbox_max[a] = -1e300; }
    for (int q = 0; q < nPoints; ++q) for (int a = 0; a < 3; ++a) { bbox_min[a] = std::min(bbox_min[a], m->points[3 * (size_t)q + a]); bbox_max[a] = std::max(bbox_max[a], m->points[3 * (size_t)q + a]); }
    for (int c = 0; c < nCells; ++c) {
        if (!(V[(size_t)c] > 0)) return fail(FY_ERR_INVALID, "fy_poly_mesh: cell %d has no volume (are the faces' points ordered outwards of their owners?)", c);
        put(C, c, (1.0 / V[(size_t)c]) * at(C.data(), c));
        V[(size_t)c] *= (1.0 / 3.0);
    }
    // ---- fvc::reconstruct's tensor: inv(sum over the cell's faces of Sf Sf / |Sf|) (symmetric positive definite for any closed cell)
    recon.assign(9 * (size_t)nCells, 0.0);
    {
        std::vector<double> T(9 * (size_t)nCells, 0.0);
        for (int f = 0; f < nFaces; ++f) {
            const V3 S = at(Sf.data(), f);
            const double s[3] = {S.x, S.y, S.z}, r = 1.0 / magSf[(size_t)f];
            for (int side = 0; side < (f < nInt ? 2 : 1); ++side) {
                double* t = &T[9 * (size_t)(side ? nei[f] : own[f])];
                for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) t[3 * a + b] += s[a] * s[b] * r;
            }
        }
        for (int c = 0; c < nCells; ++c) {
            const double* t = &T[9 * (size_t)c];
            double* q = &recon[9 * (size_t)c];
            const double det = t[0] * (t[4] * t[8] - t[5] * t[7]) - t[1] * (t[3] * t[8] - t[5] * t[6]) + t[2] * (t[3] * t[7] - t[4] * t[6]);
            if (!(std::fabs(det) > 0)) return fail(FY_ERR_INVALID, "fy_poly_mesh: cell %d is flat (its face-area tensor cannot be inverted)", c);
            const double r = 1.0 / det;
            q[0] = (t[4] * t[8] - t[5] * t[7]) * r; q[1] = (t[2] * t[7] - t[1] * t[8]) * r; q[2] = (t[1] * t[5] - t[2] * t[4]) * r;
            q[3] = (t[5] * t[6] - t[3] * t[8]) * r; q[4] = (t[0] * t[8] - t[2] * t[6]) * r; q[5] = (t[2] * t[3] - t[0] * t[5]) * r;
            q[6] = (t[3] * t[7] - t[4] * t[6]) * r; q[7] = (t[1] * t[6] - t[0] * t[7]) * r; q[8] = (t[0] * t[4] - t[1] * t[3]) * r;
        }
    }
    // ---- interpolation / gradient coefficients
    w.assign((size_t)nInt, 0.5); dcNO.assign((size_t)nFaces, 0.0); kvec.assign(3 * (size_t)nInt, 0.0);
    for (int f = 0; f < nInt; ++f) {
        V3 cn = at(C.data(), nei[f]);
        if (!sep.empty()) cn = cn + at(sep.data(), f);
        const V3 S = at(Sf.data(), f), cf = at(Cf.data(), f), cp = at(C.data(), own[f]);
        const double sOwn = std::fabs(dot(S, cf - cp)), sNei = std::fabs(dot(S, cn - cf));
        w[(size_t)f] = sNei / (sOwn + sNei);
        const V3 d = cn - cp, n = (1.0 / magSf[(size_t)f]) * S;
        dcNO[(size_t)f] = 1.0 / std::max(dot(n, d), 0.05 * mag(d));
        put(kvec, f, n - dcNO[(size_t)f] * d);
    }
    for (int f = nInt; f < nFaces; ++f) {
        const V3 n = (1.0 / magSf[(size_t)f]) * at(Sf.data(), f);
        const double nd = dot(n, at(Cf.data(), f) - at(C.data(), own[f]));
        if (!(nd > 0)) return fail(FY_ERR_INVALID, "fy_poly_mesh: boundary face %d does not point out of its cell", f);
        dcNO[(size_t)f] = 1.0 / nd;
    }
    return FY_OK;
}

}  // namespace fy
