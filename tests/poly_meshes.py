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


def to_lattice(mesh, field):
    """a cell field of a (renumbered) block back in lattice order i + nx (j + ny k)"""
    return np.asarray(field)[mesh["perm"]]
