"""Synthetic inputs for the particle-path parity cases (shared by the fixture generator and the tests).

Everything here is regenerable anywhere without the reference:
  * meshes are uniform hex blocks in blockMesh order, cell = i + nx*(j + ny*k), centre = o + (i+0.5)*dx
    (same expression as oracle/shim/fvCFD.H's driver, oracle/particle_oracle.cpp and the product);
  * fluid fields are low-order polynomials of the cell centre evaluated one IEEE operation at a time
    (numpy elementwise ops never fuse), so the doubles are bit-reproducible on any IEEE-754 host;
  * particle records come from numpy's frozen legacy MT19937 stream (RandomState), stride-10 AoS
    [x y z vx vy vz wx wy wz radius] as FoamYade.C:190-219 unpacks them.
The golden fixtures additionally store a sha256 of every generated input so drift is detected, not assumed.
"""
import hashlib
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Case:
    name: str
    nx: int
    ny: int
    nz: int
    L: float                      # extent in x; dx = L / nx
    origin: tuple = (0.0, 0.0, 0.0)
    gaussian: int = 1
    n_yade: int = 1               # 1 = serial Yade (FoamYade.C:31), >1 = master + (n_yade-1) workers
    nsteps: int = 1
    np_: int = 1000
    seed: int = 1
    rhoP: float = 2650.0
    rhoF: float = 1000.0
    nu: float = 1e-6
    dt: float = 1e-4
    g: tuple = (0.0, 0.0, -9.81)
    radius_dx: float = 0.2        # particle radius in units of dx
    vel_scale: float = 0.1
    cluster: int = 0              # extra particles packed into a 2dx cube (alpha floor / Ergun branch)
    fast: int = 0                 # extra particles with |v| ~ 2 m/s (Re > 1000 branch)
    outside: int = 0              # extra particles placed just outside the block (Q8) and far outside
    grading: tuple = (1.0, 1.0, 1.0)   # blockMesh simpleGrading (last / first cell size per axis); anything but ones = a non-uniform mesh (explicit k-d nodes)
    fibre: int = 0                # FoamYade::fibreCpl (FoamYade.H:102): 15 doubles per particle on the wire (FoamYade.C:131-136)
    extra: dict = field(default_factory=dict)

    @property
    def dx(self):
        return self.L / self.nx

    @property
    def ncells(self):
        return self.nx * self.ny * self.nz


CASES = [
    Case("g8_serial", 8, 8, 8, 0.1, np_=300, seed=11, cluster=60, fast=10, outside=10),
    Case("g16_serial_2step", 16, 16, 16, 0.1, np_=1000, seed=12, nsteps=2, cluster=150, fast=20, outside=20),
    Case("g32_serial", 32, 32, 32, 0.1, np_=1800, seed=13, cluster=150, fast=30, outside=20, nu=1e-6),
    Case("g12x10x6_offset", 12, 10, 6, 0.24, origin=(-0.05, 0.3, 1.0), np_=600, seed=14, cluster=40, fast=10, outside=10,
         nu=1e-5, rhoF=1.2, rhoP=2500.0),
    Case("g16_parallel3", 16, 16, 16, 0.1, np_=900, seed=15, n_yade=4, nsteps=2, cluster=120, fast=10, outside=10),
    Case("p32_serial_c1", 32, 32, 32, 0.1, gaussian=0, np_=1000, seed=12345, nu=0.01, dt=0.005, radius_dx=0.1,
         outside=16),
    Case("p16_parallel2", 16, 16, 16, 0.1, gaussian=0, np_=500, seed=17, n_yade=3, nsteps=2, nu=0.01, outside=10),
    # second batch: shapes the first seven do not have
    Case("g24x18x10_parallel5", 24, 18, 10, 0.36, origin=(-0.2, -0.1, 0.05), np_=1100, seed=21, n_yade=6, nsteps=2, cluster=100, fast=15,
         outside=15, nu=2e-6),                                         # non-cubic block, five parallel-Yade workers, two steps
    Case("g48_serial", 48, 48, 48, 0.12, np_=1400, seed=22, cluster=120, fast=20, outside=20),          # 110 592 cells: a 17-level tree
    Case("g9x7x5_odd", 9, 7, 5, 0.09, origin=(0.5, -0.5, 0.0), np_=300, seed=23, cluster=40, fast=5, outside=10),   # odd extents: median ties everywhere
    Case("p20x12x8_parallel4", 20, 12, 8, 0.2, origin=(0.0, 0.1, -0.3), gaussian=0, np_=700, seed=24, n_yade=5, nsteps=2, nu=0.005, outside=12),
]
# fibre coupling (public flag, never set by the shipped solvers): own list, own tests (tests/test_fibre_coupling.py)
FIBRE_CASES = [
    Case("g8_fibre_serial", 8, 8, 8, 0.1, np_=200, seed=31, cluster=30, fast=6, outside=6, fibre=1),
    Case("p16_fibre_parallel2", 16, 16, 16, 0.1, gaussian=0, np_=300, seed=32, n_yade=3, nu=0.01, outside=8, fibre=1),
]
# a non-uniform mesh (graded block): the particle half on explicit k-d nodes, Gaussian mode (the point-force findCell stand-in of the
# reference driver is a uniform-block one).  Own list, own tests (tests/test_graded_mesh.py)
GRADED_CASES = [
    Case("g16x12x10_graded", 16, 12, 10, 0.16, origin=(0.1, -0.05, 0.0), np_=700, seed=41, cluster=60, fast=10, outside=10, grading=(3.0, 1.0, 0.4)),
    Case("g10_graded_parallel2", 10, 10, 10, 0.1, np_=400, seed=42, n_yade=3, cluster=40, fast=6, outside=8, grading=(0.5, 2.0, 2.5)),
]
CASES_BY_NAME = {c.name: c for c in CASES + FIBRE_CASES + GRADED_CASES}


def is_graded(c: Case):
    return tuple(c.grading) != (1.0, 1.0, 1.0)


def axis_nodes(n, length, o, ratio):
    """node coordinates of one block edge, blockMesh simpleGrading: geometric cell sizes with last / first = ratio"""
    if ratio == 1.0:
        return o + np.arange(n + 1, dtype=np.float64) * (length / n)
    r = ratio ** (1.0 / (n - 1))
    i = np.arange(n + 1, dtype=np.float64)
    return o + length * ((r ** i - 1.0) / (r ** n - 1.0))


def mesh_nodes(c: Case):
    return [axis_nodes(n, n * c.dx, c.origin[a], c.grading[a]) for a, n in enumerate((c.nx, c.ny, c.nz))]


def cell_volumes(c: Case):
    if not is_graded(c):
        return np.full(c.ncells, c.dx * c.dx * c.dx)
    X, Y, Z = mesh_nodes(c)
    V = np.diff(Z)[:, None, None] * (np.diff(Y)[None, :, None] * np.diff(X)[None, None, :])
    return np.ascontiguousarray(V.reshape(-1))


def mesh_points(c: Case):
    """(Np,3) mesh.points() in blockMesh order (what sendMeshBbox scans, FoamYade.C:82-94)"""
    X, Y, Z = mesh_nodes(c)
    P = np.empty((c.nz + 1, c.ny + 1, c.nx + 1, 3))
    P[..., 0] = X[None, None, :]; P[..., 1] = Y[None, :, None]; P[..., 2] = Z[:, None, None]
    return np.ascontiguousarray(P.reshape(-1, 3))


def cell_centres(c: Case):
    """(Nc,3) float64, bit-identical to the C++ sides: o + (i + 0.5) * dx (graded blocks: midpoints of the node coordinates)."""
    if is_graded(c):
        X, Y, Z = mesh_nodes(c)
        C = np.empty((c.nz, c.ny, c.nx, 3), dtype=np.float64)
        C[..., 0] = (0.5 * (X[:-1] + X[1:]))[None, None, :]
        C[..., 1] = (0.5 * (Y[:-1] + Y[1:]))[None, :, None]
        C[..., 2] = (0.5 * (Z[:-1] + Z[1:]))[:, None, None]
        return np.ascontiguousarray(C.reshape(-1, 3))
    dx = c.dx
    i = np.arange(c.nx, dtype=np.float64)
    j = np.arange(c.ny, dtype=np.float64)
    k = np.arange(c.nz, dtype=np.float64)
    x = c.origin[0] + (i + 0.5) * dx
    y = c.origin[1] + (j + 0.5) * dx
    z = c.origin[2] + (k + 0.5) * dx
    C = np.empty((c.nz, c.ny, c.nx, 3), dtype=np.float64)
    C[..., 0] = x[None, None, :]
    C[..., 1] = y[None, :, None]
    C[..., 2] = z[:, None, None]
    return C.reshape(-1, 3)


def fluid_fields(c: Case):
    """U, gradP, divT, ddtU (Nc,3) and vGrad (Nc,9): polynomials, one IEEE op at a time."""
    C = cell_centres(c)
    x = C[:, 0] - c.origin[0]
    y = C[:, 1] - c.origin[1]
    z = C[:, 2] - c.origin[2]
    U = np.empty((c.ncells, 3))
    U[:, 0] = (0.3 + 1.5 * y) - (0.7 * x) * z
    U[:, 1] = -0.2 + (0.9 * z) * x
    U[:, 2] = 0.1 * x - (0.4 * y) * z
    gradP = np.empty((c.ncells, 3))
    gradP[:, 0] = -50.0 + 300.0 * x
    gradP[:, 1] = (2000.0 * y) * z
    gradP[:, 2] = -9810.0 + 100.0 * z
    divT = np.empty((c.ncells, 3))
    divT[:, 0] = 10.0 * y
    divT[:, 1] = -5.0 * x
    divT[:, 2] = (3.0 * z) * x + 0.25
    ddtU = np.empty((c.ncells, 3))
    ddtU[:, 0] = 1.0 + x
    ddtU[:, 1] = 2.0 * y
    ddtU[:, 2] = -1.0 * z
    vGrad = np.empty((c.ncells, 9))
    for q in range(9):
        vGrad[:, q] = (0.5 * (q + 1)) * x - (0.25 * (9 - q)) * y + (0.125 * (q - 4)) * z
    return dict(U=U, gradP=gradP, divT=divT, ddtU=ddtU, vGrad=vGrad)


def particle_records(c: Case, step: int = 0):
    """(Np_total,10) float64 AoS records for `step` (different stream per step)."""
    rs = np.random.RandomState(c.seed + 1000 * step)
    dx = c.dx
    ext = np.array([c.nx, c.ny, c.nz], dtype=np.float64) * dx
    o = np.array(c.origin, dtype=np.float64)
    parts = []

    def rec(pos, vel_scale, r):
        n = pos.shape[0]
        out = np.empty((n, 10))
        out[:, 0:3] = pos
        out[:, 3:6] = (rs.random_sample((n, 3)) * 2.0 - 1.0) * vel_scale
        out[:, 6:9] = (rs.random_sample((n, 3)) * 2.0 - 1.0) * 0.1
        out[:, 9] = r
        return out

    pos = o + (0.02 + 0.96 * rs.random_sample((c.np_, 3))) * ext
    parts.append(rec(pos, c.vel_scale, c.radius_dx * dx))
    if c.cluster:
        corner = o + 0.37 * ext
        pos = corner + rs.random_sample((c.cluster, 3)) * (2.0 * dx)
        parts.append(rec(pos, c.vel_scale, c.radius_dx * dx * 1.5))
    if c.fast:
        pos = o + (0.1 + 0.8 * rs.random_sample((c.fast, 3))) * ext
        parts.append(rec(pos, 2.5, c.radius_dx * dx * 2.0))
    if c.outside:
        n = c.outside
        pos = o + rs.random_sample((n, 3)) * ext
        # push half of them 0.3..4 dx beyond a face (found in Gaussian mode only, Q8), the rest 6..9 dx beyond
        axis = rs.randint(0, 3, size=n)
        side = rs.randint(0, 2, size=n)
        dist = np.where(np.arange(n) % 2 == 0, 0.3 + 3.7 * rs.random_sample(n), 6.0 + 3.0 * rs.random_sample(n)) * dx
        for q in range(n):
            a = axis[q]
            pos[q, a] = (o[a] - dist[q]) if side[q] == 0 else (o[a] + ext[a] + dist[q])
        parts.append(rec(pos, c.vel_scale, c.radius_dx * dx))
    R = np.ascontiguousarray(np.concatenate(parts, axis=0))
    # exact-on-face / exact-on-centre probes (deterministic positions)
    if R.shape[0] >= 4:
        R[0, 0:3] = o + np.array([0.5, 0.5, 0.5]) * dx            # centre of cell 0
        R[1, 0:3] = o + np.array([1.0, 1.0, 1.0]) * dx            # a mesh vertex
        R[2, 0:3] = o                                             # block corner
        R[3, 0:3] = o + ext                                       # opposite corner (on bbox max)
    return fibre_wide(c, R, step) if c.fibre else R


def batch_ranges(c: Case, n):
    if c.n_yade == 1:
        return [(0, n)]
    W = c.n_yade - 1
    return [split_range(n, W, w) for w in range(W)]


def fibre_wide(c: Case, R, step):
    """(n,15) records for a fibreCpl run whose fields, AS THE REFERENCE READS THEM, are those of R (n,10).

    With fibreCpl the reference takes the position from buf[np*15 + 0..2] (FoamYade.C:194-198) but velocity, spin and radius from
    buf[np*10 + 3..9] of the same per-Yade-proc buffer (FoamYade.C:211-221).  Positions sit at residues {0,1,2,5,6,7} mod 10, the
    radius at residue 9, so a buffer exists in which every radius the reference reads is a real radius: fill the stride-10 view first,
    then the stride-15 positions (an odd particle's position overwrites some other particle's vz / wx / wy: still velocity-sized
    numbers).  What lies beyond 10 m doubles of a buffer and is not a position is fibre payload nobody reads."""
    rs = np.random.RandomState(c.seed + 1000 * step + 77)
    n = R.shape[0]
    out = np.empty((n, 15))
    for lo, hi in batch_ranges(c, n):
        m = hi - lo
        B = (rs.random_sample(15 * m) * 2.0 - 1.0) * 0.1
        for i in range(m):
            B[10 * i + 3:10 * i + 10] = R[lo + i, 3:10]
        for i in range(m):
            B[15 * i:15 * i + 3] = R[lo + i, 0:3]
        out[lo:hi] = B.reshape(m, 15)
    return out


def split_range(n, w_count, w):
    """contiguous split used by the fake parallel Yade workers (matches oracle/ref_driver.cpp)."""
    return (n * w) // w_count, (n * (w + 1)) // w_count


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
