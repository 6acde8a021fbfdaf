"""Polyhedral meshes in OpenFOAM's addressing (points / faces / owner / neighbour / boundary: constant/polyMesh), built here because there is no blockMesh to
write them: a block of hexahedra whose vertices go through an arbitrary map (sheared, wavy: non-orthogonal), optionally with its cells renumbered at random
(no lattice left in the numbering).  Test infrastructure."""
import numpy as np

SIDES = ("xmin", "xmax", "ymin", "ymax", "zmin", "zmax")


def hex_block(nx, ny, nz, lengths=(1.0, 1.0, 1.0), vertex_map=None, patches=None, renumber_seed=None):
    """patches: [(name, [side indices 0..5])] (default: one patch per side, in SIDES order).  Faces in OpenFOAM's order: internal faces by owner, for one
    owner by ascending neighbour (upper-triangular order); then the patches' faces.  A face's points turn counter-clockwise seen from outside its owner."""
    lx, ly, lz = lengths
    pid = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)
    ii, jj, kk = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    P = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    P[pid(ii, jj, kk).ravel()] = np.stack([ii.ravel() * lx / nx, jj.ravel() * ly / ny, kk.ravel() * lz / nz], axis=1)
    if vertex_map is not None:
        P = np.asarray(vertex_map(P), dtype=np.float64)
    cid = lambda i, j, k: i + nx * (j + ny * k)
    perm = np.arange(nx * ny * nz)
    if renumber_seed is not None:
        perm = np.random.RandomState(renumber_seed).permutation(nx * ny * nz)        # new number of the lattice cell
    quad = {0: lambda i, j, k: (pid(i, j, k), pid(i, j + 1, k), pid(i, j + 1, k + 1), pid(i, j, k + 1)),        # normal +x
            1: lambda i, j, k: (pid(i, j, k), pid(i, j, k + 1), pid(i + 1, j, k + 1), pid(i + 1, j, k)),        # normal +y
            2: lambda i, j, k: (pid(i, j, k), pid(i + 1, j, k), pid(i + 1, j + 1, k), pid(i, j + 1, k))}        # normal +z
    internal = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                c = perm[cid(i, j, k)]
                for d, (di, dj, dk) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
                    if i + di < nx and j + dj < ny and k + dk < nz:
                        n = perm[cid(i + di, j + dj, k + dk)]
                        q = quad[d](i + di, j + dj, k + dk)
                        internal.append((c, n, q) if c < n else (n, c, q[::-1]))      # owner = the lower number; the normal points owner -> neighbour
    internal.sort(key=lambda t: (t[0], t[1]))
    faces = [t[2] for t in internal]; owner = [t[0] for t in internal]; neigh = [t[1] for t in internal]
    if patches is None:
        patches = [(SIDES[s], [s]) for s in range(6)]
    pstart, psize, pnames = [], [], []
    for name, sides in patches:
        pstart.append(len(faces)); pnames.append(name)
        bf = []
        for s in sides:
            d, hi = s // 2, s % 2
            rng = [range(nx), range(ny), range(nz)]
            rng[d] = [([nx, ny, nz][d] - 1) if hi else 0]
            for k in rng[2]:
                for j in rng[1]:
                    for i in rng[0]:
                        f = quad[d](i + (d == 0 and hi), j + (d == 1 and hi), k + (d == 2 and hi))
                        bf.append((perm[cid(i, j, k)], f if hi else f[::-1]))          # outward normal
        bf.sort(key=lambda t: t[0])
        for c, f in bf:
            faces.append(f); owner.append(c)
        psize.append(len(bf))
    return dict(points=P, face_offsets=np.arange(0, 4 * len(faces) + 1, 4, dtype=np.int32), face_points=np.asarray(faces, np.int32).ravel(),
                owner=np.asarray(owner, np.int32), neighbour=np.asarray(neigh, np.int32), n_cells=nx * ny * nz, patch_start=np.asarray(pstart, np.int32),
                patch_size=np.asarray(psize, np.int32), patch_names=pnames, perm=perm, shape=(nx, ny, nz))


def make_cyclic(mesh, pairs):
    """pairs: [(patch A, patch B)] (indices): the two patches become translational cyclic halves (mesh["patch_neighbour"]); B's faces are put in the order of A's
    partners (the face whose point average is A's plus the mean separation of the two patches), as OpenFOAM's cyclic patches must be"""
    m = dict(mesh)
    off, fp = m["face_offsets"], m["face_points"]
    faces = [list(fp[off[f]:off[f + 1]]) for f in range(len(off) - 1)]
    own = np.array(m["owner"]).copy()
    P = m["points"]
    ctr = lambda f: P[faces[f]].mean(axis=0)
    nbr = -np.ones(len(m["patch_start"]), np.int32)
    for a, b in pairs:
        sa, sb, n = int(m["patch_start"][a]), int(m["patch_start"][b]), int(m["patch_size"][a])
        assert n == int(m["patch_size"][b])
        ca = np.array([ctr(sa + q) for q in range(n)]); cb = np.array([ctr(sb + q) for q in range(n)])
        sep = cb.mean(axis=0) - ca.mean(axis=0)
        order = []
        for q in range(n):
            d = np.abs(cb - (ca[q] + sep)).sum(axis=1)
            j = int(np.argmin(d))
            assert d[j] < 1e-9 * (np.abs(sep).sum() + 1e-300), "the two patches are not translates of each other"
            order.append(j)
        assert sorted(order) == list(range(n))
        nf = [faces[sb + j] for j in order]; no = [own[sb + j] for j in order]
        for q in range(n):
            faces[sb + q] = nf[q]; own[sb + q] = no[q]
        nbr[a], nbr[b] = b, a
    m["face_points"] = np.asarray([v for f in faces for v in f], np.int32)
    m["face_offsets"] = np.concatenate([[0], np.cumsum([len(f) for f in faces])]).astype(np.int32)
    m["owner"] = own.astype(np.int32)
    m["patch_neighbour"] = nbr
    return m


def shear(a_xy=0.0, a_xz=0.0, a_yz=0.0):
    """x += a_xy y + a_xz z, y += a_yz z: a block of parallelepipeds (non-orthogonal, no skewness: face centres stay on the lines between cell centres)"""
    def m(P):
        Q = P.copy()
        Q[:, 0] += a_xy * P[:, 1] + a_xz * P[:, 2]
        Q[:, 1] += a_yz * P[:, 2]
        return Q
    return m


def wavy(amp, lengths=(1.0, 1.0, 1.0)):
    """interior vertices displaced by amp sin sin sin (boundary vertices stay): non-orthogonal AND skewed cells that differ from one another"""
    def m(P):
        s = np.sin(np.pi * P[:, 0] / lengths[0]) * np.sin(np.pi * P[:, 1] / lengths[1]) * np.sin(np.pi * P[:, 2] / lengths[2])
        Q = P.copy()
        Q[:, 0] += amp * s * np.cos(3 * P[:, 1]); Q[:, 1] += amp * s * np.cos(2 * P[:, 2] + 1); Q[:, 2] += amp * s * np.cos(4 * P[:, 0] + 2)
        return Q
    return m


def wavy_periodic(amp, lengths=(1.0, 1.0, 1.0)):
    """every vertex displaced by a field with the box's periods (opposite sides stay translates of each other: cyclic patches): non-orthogonal, skewed cells"""
    def m(P):
        x, y, z = (2 * np.pi * P[:, a] / lengths[a] for a in range(3))
        Q = P.copy()
        Q[:, 0] += amp * np.sin(y) * np.sin(z + 0.4); Q[:, 1] += amp * np.sin(x + 0.9) * np.sin(z); Q[:, 2] += amp * np.sin(x) * np.sin(y + 0.3)
        return Q
    return m


def to_lattice(mesh, field):
    """a cell field of a (renumbered) block back in lattice order i + nx (j + ny k)"""
    return np.asarray(field)[mesh["perm"]]


def from_cells(points, cells, patch_of_face, patch_names, shape=None, _tag=None):
    """a mesh in OpenFOAM's addressing from cells given as lists of faces (point tuples turning counter-clockwise seen from OUTSIDE the cell).  A face two
    cells share becomes an internal face (owner = the lower cell, point order = the owner's); the others go to the patch patch_of_face(face centre) names.
    Internal faces in upper-triangular order, boundary faces patch by patch sorted by owner [OF-6 polyMesh ordering]."""
    seen = {}
    for c, faces in enumerate(cells):
        for f in faces:
            key = tuple(sorted(f))
            if key in seen:
                seen[key].append((c, tuple(f)))
            else:
                seen[key] = [(c, tuple(f))]
    internal, boundary = [], [[] for _ in patch_names]
    P = np.asarray(points, np.float64)
    for key, users in seen.items():
        if len(users) == 2:
            (c0, f0), (c1, f1) = sorted(users)
            internal.append((c0, c1, f0))
        else:
            c0, f0 = users[0]
            boundary[patch_names.index(_tag[key] if _tag is not None else patch_of_face(P[list(f0)].mean(axis=0)))].append((c0, f0))
    internal.sort(key=lambda t: (t[0], t[1]))
    faces = [t[2] for t in internal]; owner = [t[0] for t in internal]; neigh = [t[1] for t in internal]
    pstart, psize = [], []
    for bf in boundary:
        pstart.append(len(faces)); psize.append(len(bf))
        for c, f in sorted(bf, key=lambda t: t[0]):
            faces.append(f); owner.append(c)
    off = np.concatenate([[0], np.cumsum([len(f) for f in faces])]).astype(np.int32)
    return dict(points=P, face_offsets=off, face_points=np.asarray([q for f in faces for q in f], np.int32), owner=np.asarray(owner, np.int32),
                neighbour=np.asarray(neigh, np.int32), n_cells=len(cells), patch_start=np.asarray(pstart, np.int32), patch_size=np.asarray(psize, np.int32),
                patch_names=list(patch_names), perm=np.arange(len(cells)), shape=shape)


def refined_block(nx, ny, nz, kr, lengths=(1.0, 1.0, 1.0), vertex_map=None):
    """the box with its layers k >= kr refined 2 x 2 in x and y (hexRef8-like, 2:1): the cells of layer kr - 1 are polyhedra with NINE faces -- their top is the four
    faces of the fine cells above, and their side faces carry the hanging mid-edge point (five-point polygons), as OpenFOAM keeps such meshes closed and conformal.
    Patches: one per side of the box (SIDES)."""
    lx, ly, lz = lengths
    NX, NY = 2 * nx, 2 * ny
    pid = lambda i, j, k: i + (NX + 1) * (j + (NY + 1) * k)                      # the fine lattice's points on every plane (the coarse layers use the even ones)
    ii, jj, kk = np.meshgrid(np.arange(NX + 1), np.arange(NY + 1), np.arange(nz + 1), indexing="ij")
    P0 = np.zeros(((NX + 1) * (NY + 1) * (nz + 1), 3))
    P0[pid(ii, jj, kk).ravel()] = np.stack([ii.ravel() * lx / NX, jj.ravel() * ly / NY, kk.ravel() * lz / nz], axis=1)
    P = P0 if vertex_map is None else np.asarray(vertex_map(P0), np.float64)
    cells = []
    for k in range(nz):
        fine = k >= kr
        st = 1 if fine else 2
        for j in range(0, NY, st):
            for i in range(0, NX, st):
                a, b = i + st, j + st
                hang = (not fine) and k == kr - 1                                 # the coarse layer under the fine ones
                bottom = [(pid(i, j, k), pid(i, b, k), pid(a, b, k), pid(a, j, k))]
                if hang:
                    m, n = i + 1, j + 1
                    top = [(pid(i, j, k + 1), pid(m, j, k + 1), pid(m, n, k + 1), pid(i, n, k + 1)), (pid(m, j, k + 1), pid(a, j, k + 1), pid(a, n, k + 1), pid(m, n, k + 1)),
                           (pid(i, n, k + 1), pid(m, n, k + 1), pid(m, b, k + 1), pid(i, b, k + 1)), (pid(m, n, k + 1), pid(a, n, k + 1), pid(a, b, k + 1), pid(m, b, k + 1))]
                    sides = [(pid(i, j, k), pid(a, j, k), pid(a, j, k + 1), pid(m, j, k + 1), pid(i, j, k + 1)),              # ymin (outward -y)
                             (pid(a, j, k), pid(a, b, k), pid(a, b, k + 1), pid(a, n, k + 1), pid(a, j, k + 1)),              # xmax
                             (pid(a, b, k), pid(i, b, k), pid(i, b, k + 1), pid(m, b, k + 1), pid(a, b, k + 1)),              # ymax
                             (pid(i, b, k), pid(i, j, k), pid(i, j, k + 1), pid(i, n, k + 1), pid(i, b, k + 1))]              # xmin
                else:
                    top = [(pid(i, j, k + 1), pid(a, j, k + 1), pid(a, b, k + 1), pid(i, b, k + 1))]
                    sides = [(pid(i, j, k), pid(a, j, k), pid(a, j, k + 1), pid(i, j, k + 1)), (pid(a, j, k), pid(a, b, k), pid(a, b, k + 1), pid(a, j, k + 1)),
                             (pid(a, b, k), pid(i, b, k), pid(i, b, k + 1), pid(a, b, k + 1)), (pid(i, b, k), pid(i, j, k), pid(i, j, k + 1), pid(i, b, k + 1))]
                cells.append(bottom + top + sides)
    eps = 1e-9
    P_un = P0
    seen = {}                                                                     # boundary faces go to the side their UNMAPPED centre lies on
    for c, faces in enumerate(cells):
        for f in faces:
            seen.setdefault(tuple(sorted(f)), []).append((c, tuple(f)))
    names = list(SIDES)
    def which(f):
        m = P_un[list(f)].mean(axis=0)
        for a, (lo, hi) in enumerate(((0.0, lx), (0.0, ly), (0.0, lz))):
            if abs(m[a] - lo) < eps: return SIDES[2 * a]
            if abs(m[a] - hi) < eps: return SIDES[2 * a + 1]
        raise AssertionError("an unmatched interior face: %r" % (m,))
    tag = {key: which(users[0][1]) for key, users in seen.items() if len(users) == 1}
    return from_cells(P, cells, None, names, shape=(nx, ny, nz), _tag=tag)


def prism_block(nx, ny, nz, lengths=(1.0, 1.0, 1.0), vertex_map=None):
    """the box cut into triangular prisms: every hexahedron of the nx x ny x nz lattice split along the diagonal of its z faces (alternating direction from
    cell to cell), so the mesh has triangular AND quadrilateral faces and cells with five faces.  Patches: one per side of the box (SIDES)."""
    lx, ly, lz = lengths
    pid = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)
    ii, jj, kk = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    P0 = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    P0[pid(ii, jj, kk).ravel()] = np.stack([ii.ravel() * lx / nx, jj.ravel() * ly / ny, kk.ravel() * lz / nz], axis=1)
    P = P0 if vertex_map is None else np.asarray(vertex_map(P0), np.float64)

    def prism(a, b, c, k):          # triangle a b c (counter-clockwise seen from +z) extruded from plane k to k + 1
        lo = [pid(i, j, k) for i, j in (a, b, c)]; hi = [pid(i, j, k + 1) for i, j in (a, b, c)]
        return [(lo[0], lo[2], lo[1]), (hi[0], hi[1], hi[2]), (lo[0], lo[1], hi[1], hi[0]), (lo[1], lo[2], hi[2], hi[1]), (lo[2], lo[0], hi[0], hi[2])]
    cells = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                a, b, c, d = (i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)
                if (i + j) % 2 == 0:
                    cells += [prism(a, b, c, k), prism(a, c, d, k)]
                else:
                    cells += [prism(a, b, d, k), prism(b, c, d, k)]
    # the side of the box a boundary face lies on comes from the UNMAPPED lattice (the map may move the sides)
    eps = 1e-9

    def patch_of_face_pts(f):
        c = P0[list(f)].mean(axis=0)
        for a, L in enumerate((lx, ly, lz)):
            if abs(c[a]) < eps: return SIDES[2 * a]
            if abs(c[a] - L) < eps: return SIDES[2 * a + 1]
        raise ValueError("face not on the box")
    # from_cells looks a patch up by the (mapped) face centre: hand it a lookup keyed on that centre
    table = {}
    for faces in cells:
        for f in faces:
            c0 = P0[list(f)].mean(axis=0)
            on_side = any(abs(c0[a]) < eps or abs(c0[a] - L) < eps for a, L in enumerate((lx, ly, lz)))
            if on_side:
                table[tuple(np.round(P[list(f)].mean(axis=0), 12))] = patch_of_face_pts(f)
    return from_cells(P, cells, lambda ctr: table[tuple(np.round(ctr, 12))], list(SIDES), shape=(nx, ny, nz))


def write_poly_mesh_files(case_dir, mesh, patch_types=None, binary=False, label64=False):
    """constant/polyMesh of `mesh`: ASCII (the layout OpenFOAM writes), or binary = the stream format of `writeFormat binary` (points as raw doubles, owner / neighbour as raw
    labels, faces as a faceCompactList: offsets + labels; the boundary file stays text).  Test infrastructure: there is no blockMesh / snappyHexMesh here"""
    import os
    pm = os.path.join(str(case_dir), "constant", "polyMesh")
    os.makedirs(pm, exist_ok=True)
    lab = np.int64 if label64 else np.int32
    arch = '    arch        "LSB;label=%d;scalar=64";\n' % (64 if label64 else 32) if binary else ""
    head = lambda cls, obj: ("FoamFile\n{\n    version     2.0;\n    format      %s;\n%s    class       %s;\n    location    \"constant/polyMesh\";\n    object      %s;\n}\n\n"
                             % ("binary" if binary else "ascii", arch, cls, obj)).encode()
    P, off, fp = mesh["points"], mesh["face_offsets"], mesh["face_points"]
    blob = lambda a: b"%d\n(" % len(a) + np.ascontiguousarray(a).tobytes() + b")\n"
    with open(os.path.join(pm, "points"), "wb") as f:
        if binary:
            f.write(head("vectorField", "points") + b"%d\n(" % len(P) + np.ascontiguousarray(P, "<f8").tobytes() + b")\n")
        else:
            f.write(head("vectorField", "points") + ("%d\n(\n" % len(P) + "".join("(%r %r %r)\n" % (float(x), float(y), float(z)) for x, y, z in P) + ")\n").encode())
    with open(os.path.join(pm, "faces"), "wb") as f:
        if binary:
            f.write(head("faceCompactList", "faces") + blob(np.asarray(off, lab)) + b"\n" + blob(np.asarray(fp, lab)))
        else:
            f.write(head("faceList", "faces") + ("%d\n(\n" % (len(off) - 1) +
                    "".join("%d(%s)\n" % (off[q + 1] - off[q], " ".join(str(int(v)) for v in fp[off[q]:off[q + 1]])) for q in range(len(off) - 1)) + ")\n").encode())
    for name in ("owner", "neighbour"):
        with open(os.path.join(pm, name), "wb") as f:
            if binary:
                f.write(head("labelList", name) + blob(np.asarray(mesh[name], lab)))
            else:
                f.write(head("labelList", name) + ("%d\n(\n" % len(mesh[name]) + "".join("%d\n" % int(v) for v in mesh[name]) + ")\n").encode())
    with open(os.path.join(pm, "boundary"), "w") as f:
        f.write("FoamFile\n{\n    version     2.0;\n    format      ascii;\n    class       polyBoundaryMesh;\n    location    \"constant/polyMesh\";\n    object      boundary;\n}\n\n%d\n(\n" % len(mesh["patch_names"]))
        for q, name in enumerate(mesh["patch_names"]):
            ty = (patch_types or {}).get(name, "wall")
            f.write("    %s\n    {\n        type            %s;\n        nFaces          %d;\n        startFace       %d;\n    }\n" % (name, ty, int(mesh["patch_size"][q]), int(mesh["patch_start"][q])))
        f.write(")\n")


def hex_block_fast(nx, ny, nz, lengths=(1.0, 1.0, 1.0), vertex_map=None):
    """hex_block(nx, ny, nz, lengths, vertex_map) with one patch per side and the lattice numbering, built with array operations (millions of cells)"""
    lx, ly, lz = lengths
    I, J, K = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    pid = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)
    P = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    P[pid(I, J, K).ravel()] = np.stack([I.ravel() * lx / nx, J.ravel() * ly / ny, K.ravel() * lz / nz], axis=1)
    if vertex_map is not None:
        P = np.asarray(vertex_map(P), np.float64)
    ci, cj, ck = [a.ravel(order="F") for a in np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")]      # i fastest
    cid = ci + nx * (cj + ny * ck)

    def quad(d, i, j, k):
        if d == 0: return np.stack([pid(i, j, k), pid(i, j + 1, k), pid(i, j + 1, k + 1), pid(i, j, k + 1)], axis=1)
        if d == 1: return np.stack([pid(i, j, k), pid(i, j, k + 1), pid(i + 1, j, k + 1), pid(i + 1, j, k)], axis=1)
        return np.stack([pid(i, j, k), pid(i + 1, j, k), pid(i + 1, j + 1, k), pid(i, j + 1, k)], axis=1)
    own_l, nei_l, f_l, key_l = [], [], [], []
    for d, (di, dj, dk) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
        ok = (ci + di < nx) & (cj + dj < ny) & (ck + dk < nz)
        o = cid[ok]
        own_l.append(o); nei_l.append(o + (1, nx, nx * ny)[d]); f_l.append(quad(d, ci[ok] + di, cj[ok] + dj, ck[ok] + dk)); key_l.append(3 * o.astype(np.int64) + d)
    order = np.argsort(np.concatenate(key_l), kind="stable")
    faces = [np.concatenate(f_l)[order]]; owner = [np.concatenate(own_l)[order]]; neigh = np.concatenate(nei_l)[order]
    pstart, psize = [], []
    nfaces = len(neigh)
    for s in range(6):
        d, hi = s // 2, s % 2
        on = (ci, cj, ck)[d] == (((nx, ny, nz)[d] - 1) if hi else 0)
        i, j, k = ci[on], cj[on], ck[on]
        q = quad(d, i + (d == 0 and hi), j + (d == 1 and hi), k + (d == 2 and hi))
        faces.append(q if hi else q[:, ::-1]); owner.append(cid[on])
        pstart.append(nfaces); psize.append(int(on.sum())); nfaces += int(on.sum())
    F = np.concatenate(faces)
    return dict(points=P, face_offsets=np.arange(0, 4 * len(F) + 1, 4, dtype=np.int32), face_points=F.astype(np.int32).ravel(), owner=np.concatenate(owner).astype(np.int32),
                neighbour=neigh.astype(np.int32), n_cells=nx * ny * nz, patch_start=np.asarray(pstart, np.int32), patch_size=np.asarray(psize, np.int32),
                patch_names=list(SIDES), perm=np.arange(nx * ny * nz), shape=(nx, ny, nz))


def tet_block(nx, ny, nz, lengths=(1.0, 1.0, 1.0), vertex_map=None):
    """the box cut into tetrahedra: every hexahedron of the lattice split into the six Kuhn tetrahedra about its main diagonal (the face diagonals then agree between
    neighbouring hexahedra: a conforming mesh of triangular faces only, four-faced cells, strongly non-orthogonal).  Patches: one per side of the box (SIDES)."""
    import itertools
    lx, ly, lz = lengths
    pid = lambda i, j, k: i + (nx + 1) * (j + (ny + 1) * k)
    ii, jj, kk = np.meshgrid(np.arange(nx + 1), np.arange(ny + 1), np.arange(nz + 1), indexing="ij")
    P0 = np.zeros(((nx + 1) * (ny + 1) * (nz + 1), 3))
    P0[pid(ii, jj, kk).ravel()] = np.stack([ii.ravel() * lx / nx, jj.ravel() * ly / ny, kk.ravel() * lz / nz], axis=1)
    P = P0 if vertex_map is None else np.asarray(vertex_map(P0), np.float64)
    cells = []
    for k in range(nz):
        for j in range(ny):
            for i in range(nx):
                for perm in itertools.permutations(range(3)):
                    v = [np.array([i, j, k])]
                    for a in perm:
                        e = np.zeros(3, int); e[a] = 1
                        v.append(v[-1] + e)
                    q = [pid(*w) for w in v]
                    x = P0[q]
                    vol = np.dot(np.cross(x[1] - x[0], x[2] - x[0]), x[3] - x[0])
                    if vol < 0:
                        q[1], q[2] = q[2], q[1]
                    a, b, c, d = q                      # positively oriented: the faces below turn counter-clockwise seen from outside
                    cells.append([(a, c, b), (a, b, d), (b, c, d), (a, d, c)])
    eps = 1e-9
    table = {}
    for faces in cells:
        for f in faces:
            c0 = P0[list(f)].mean(axis=0)
            for a, L in enumerate((lx, ly, lz)):
                if abs(c0[a]) < eps: table[tuple(np.round(P[list(f)].mean(axis=0), 12))] = SIDES[2 * a]
                elif abs(c0[a] - L) < eps: table[tuple(np.round(P[list(f)].mean(axis=0), 12))] = SIDES[2 * a + 1]
    return from_cells(P, cells, lambda ctr: table[tuple(np.round(ctr, 12))], list(SIDES), shape=(nx, ny, nz))
