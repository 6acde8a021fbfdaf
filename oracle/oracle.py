"""ctypes front-end of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product package must
never import this module (tests/test_abi.py greps for that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
MAXK = 16

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """compile liboracle.so.  The reference driver (oracle/_ref/ref_driver: the reference's FoamYade.C / meshTree.C compiled against the stand-in OpenFOAM
    header oracle/shim/fvCFD.H) is NOT built here any more (round 6): this tier's rules do not admit a reference build against stand-in headers, so it is
    neither a pin nor a baseline.  The recipe stays (`make -C oracle ref`, FOAMYADE_BUILD_REF=1 here) as the provenance of the fixtures under tests/golden,
    which tests/golden/gen_golden.py produced with it in round 1 and which are regression data, not a pin (parity unpinned)."""
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in os.listdir(HERE) if f.endswith(".cpp") and f != "ref_driver.cpp"):
        subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)
    if os.environ.get("FOAMYADE_BUILD_REF") == "1" and os.path.isdir("/root/reference/FoamYade") and os.path.exists("/opt/conda/lib/libmpi.so"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


class StepArgs(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("Nc", C.c_int),
        ("dx", C.c_double), ("bbmin", C.c_double * 3), ("bbmax", C.c_double * 3),
        ("C", _dp), ("V", _dp), ("pre", _ip),
        ("U", _dp), ("gradP", _dp), ("vGrad", _dp), ("divT", _dp),
        ("uSourceDrag", _dp), ("alpha", _dp), ("uSource", _dp), ("uParticle", _dp),
        ("gaussian", C.c_int), ("rhoP", C.c_double), ("rhoF", C.c_double), ("nu", C.c_double),
        ("nbatch", C.c_int), ("off", _ip), ("records", _dp),
        ("k", _ip), ("ids", _ip), ("w", _dp), ("chain_len", _ip), ("force", _dp), ("found", _ip),
        ("threads", C.c_int),
    ]


_lib = None


def use_native_build():
    """bench.py's cpu_baseline leg: compile the restatement for THIS host (-O3 -march=native; contraction off as
    always) and use that library from now on.  Call before anything else of this module; returns False (and keeps liboracle.so) if the
    compile fails."""
    global LIB_PATH, _lib
    native = os.path.join(HERE, "liboracle_native.so")
    try:
        if os.path.exists(native):
            os.remove(native)                  # it may have been built on another host: never trust a travelled -march=native binary
        subprocess.run(["make", "-C", HERE, "liboracle_native.so"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (subprocess.CalledProcessError, OSError):
        return False
    LIB_PATH, _lib = native, None
    return True


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_build_tree.argtypes = [C.c_int, _dp, _ip]
        L.orc_build_tree.restype = C.c_int
        L.orc_range_search.argtypes = [C.c_int, _dp, _ip, C.c_int, _dp, C.c_int, C.c_double, _ip, _ip, _ip]
        L.orc_range_search.restype = C.c_long
        L.orc_nearest_cell.argtypes = [C.c_int, _dp, _ip, C.c_int, _dp, C.c_int, _ip]
        L.orc_particle_action.argtypes = [C.POINTER(StepArgs)]
        L.orc_set_source_zero.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_extra_force_models.argtypes = [C.POINTER(StepArgs), _dp, C.c_double, C.c_int]
        L.orc_extra_force_models.restype = None
        _lib = L
    return _lib


def _d(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _i(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def build_tree(centres):
    centres = np.ascontiguousarray(centres, dtype=np.float64)
    pre = np.empty(centres.shape[0], dtype=np.int32)
    n = lib().orc_build_tree(centres.shape[0], _d(centres), _i(pre))
    assert n == centres.shape[0]
    return pre


def range_search(centres, pre, pos, rng):
    """pos (Np,3) or records (Np,10).  returns k, ids(Np,16), chain_len, visits"""
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    n = pos.shape[0]
    k = np.zeros(n, dtype=np.int32)
    ids = np.full((n, MAXK), -1, dtype=np.int32)
    chain = np.zeros(n, dtype=np.int32)
    visits = lib().orc_range_search(centres.shape[0], _d(centres), _i(pre), n, _d(pos), pos.shape[1], float(rng),
                                    _i(k), _i(ids), _i(chain))
    return k, ids, chain, visits


def nearest_cell(centres, pre, pos):
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    cell = np.empty(pos.shape[0], dtype=np.int32)
    lib().orc_nearest_cell(centres.shape[0], _d(centres), _i(pre), pos.shape[0], _d(pos), pos.shape[1], _i(cell))
    return cell


class Mesh:
    """uniform hex block, blockMesh order (same convention as tests/golden_cases.py)"""

    def __init__(self, nx, ny, nz, dx, origin=(0.0, 0.0, 0.0), centres=None, pre=None, volumes=None, bbmin=None, bbmax=None):
        """centres / volumes / bbmin / bbmax: a non-uniform mesh (mesh.C(), mesh.V(), bounding box of mesh.points()); nx, ny, nz, dx then
        only serve the uniform-block findCell stand-in of the point-force mode, which such a mesh cannot use"""
        self.nx, self.ny, self.nz, self.dx = nx, ny, nz, float(dx)
        self.origin = tuple(float(o) for o in origin)
        self.Nc = nx * ny * nz
        if centres is None:
            i = np.arange(nx, dtype=np.float64); j = np.arange(ny, dtype=np.float64); k = np.arange(nz, dtype=np.float64)
            Cc = np.empty((nz, ny, nx, 3))
            Cc[..., 0] = (self.origin[0] + (i + 0.5) * dx)[None, None, :]
            Cc[..., 1] = (self.origin[1] + (j + 0.5) * dx)[None, :, None]
            Cc[..., 2] = (self.origin[2] + (k + 0.5) * dx)[:, None, None]
            centres = Cc.reshape(-1, 3)
        self.C = np.ascontiguousarray(centres, dtype=np.float64)
        self.V = np.full(self.Nc, dx * dx * dx, dtype=np.float64) if volumes is None else np.ascontiguousarray(volumes, dtype=np.float64)
        self.bbmin = np.array(self.origin, dtype=np.float64) if bbmin is None else np.array(bbmin, dtype=np.float64)
        self.bbmax = np.array([self.origin[0] + nx * dx, self.origin[1] + ny * dx, self.origin[2] + nz * dx]) if bbmax is None else np.array(bbmax, dtype=np.float64)
        self.pre = build_tree(self.C) if pre is None else np.ascontiguousarray(pre, dtype=np.int32)


def graded_block_geometry(grading, origin=(0.0, 0.0, 0.0)):
    """cell centres, volumes, face coordinates of a rectilinear block whose cell sizes along x, y, z are `grading` = (hx, hy, hz), blockMesh order"""
    hx, hy, hz = (np.asarray(a, dtype=np.float64) for a in grading)
    xf, yf, zf = (o + np.concatenate([[0.0], np.cumsum(h)]) for o, h in zip(origin, (hx, hy, hz)))
    xc, yc, zc = 0.5 * (xf[1:] + xf[:-1]), 0.5 * (yf[1:] + yf[:-1]), 0.5 * (zf[1:] + zf[:-1])
    nx, ny, nz = hx.size, hy.size, hz.size
    Cc = np.empty((nz, ny, nx, 3))
    Cc[..., 0] = xc[None, None, :]; Cc[..., 1] = yc[None, :, None]; Cc[..., 2] = zc[:, None, None]
    V = (hz[:, None, None] * hy[None, :, None]) * hx[None, None, :]
    return Cc.reshape(-1, 3), V.reshape(-1), (xf, yf, zf)


def graded_mesh(grading, origin=(0.0, 0.0, 0.0)):
    Cc, V, (xf, yf, zf) = graded_block_geometry(grading, origin)
    m = Mesh(len(grading[0]), len(grading[1]), len(grading[2]), float(np.cbrt(V[0])), origin, centres=Cc, volumes=V,
             bbmin=(xf[0], yf[0], zf[0]), bbmax=(xf[-1], yf[-1], zf[-1]))
    m.faces = (xf, yf, zf)
    return m


FORCE_ADDED_MASS, FORCE_GAUSSIAN_TORQUE = 1, 2


def fibre_narrow(records15, batch_off):
    """The (n,10) records the reference actually uses when fibreCpl is set (FoamYade.H:102): every Yade proc's buffer holds 15
    doubles per particle (FoamYade.C:131-136,161-165); the position is buf[np*15 + 0..2] (FoamYade.C:194-198) while velocity, spin
    and radius stay buf[np*10 + 3..9] (FoamYade.C:211-221) -- indices into the same buffer, restated literally."""
    wide = np.ascontiguousarray(records15, dtype=np.float64).reshape(-1, 15)
    off = np.asarray(batch_off, dtype=np.int64)
    out = np.empty((wide.shape[0], 10))
    for b in range(len(off) - 1):
        lo, hi = int(off[b]), int(off[b + 1])
        buf = wide[lo:hi].reshape(-1)
        for np_ in range(hi - lo):
            out[lo + np_, 0:3] = buf[np_ * 15:np_ * 15 + 3]
            out[lo + np_, 3:10] = buf[np_ * 10 + 3:np_ * 10 + 10]
    return out


def particle_action(mesh: Mesh, fields: dict, mutable: dict, records, batch_off, gaussian, rhoP, rhoF, nu, threads=1,
                    force_models=0, dt=None, fibre=False):
    """FoamYade::setParticleAction without MPI.  `mutable` arrays (alpha, uParticle, uSourceDrag, uSource) are
    updated in place.  returns dict(k, ids, w, chain_len, force, found).
    force_models != 0 additionally applies the reference's call-site-less models (FoamYade.C:392-413, 465-479) on top
    (needs fields["ddtU"] and dt for the added mass)."""
    if fibre:
        records = fibre_narrow(records, batch_off)
    records = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 10)
    n = records.shape[0]
    off = np.ascontiguousarray(batch_off, dtype=np.int32)
    out = dict(k=np.zeros(n, np.int32), ids=np.full((n, MAXK), -1, np.int32), w=np.zeros((n, MAXK)),
               chain_len=np.zeros(n, np.int32), force=np.zeros((n, 6)), found=np.zeros(n, np.int32))
    a = StepArgs()
    a.nx, a.ny, a.nz, a.Nc, a.dx = mesh.nx, mesh.ny, mesh.nz, mesh.Nc, mesh.dx
    for q in range(3):
        a.bbmin[q] = mesh.bbmin[q]
        a.bbmax[q] = mesh.bbmax[q]
    a.C, a.V, a.pre = _d(mesh.C), _d(mesh.V), _i(mesh.pre)
    keep = []
    for nm in ("U", "gradP", "vGrad", "divT"):
        arr = np.ascontiguousarray(fields[nm], dtype=np.float64)
        keep.append(arr)
        setattr(a, nm, _d(arr))
    for nm in ("uSourceDrag", "alpha", "uSource", "uParticle"):
        setattr(a, nm, _d(mutable[nm]))
    a.gaussian, a.rhoP, a.rhoF, a.nu = int(gaussian), float(rhoP), float(rhoF), float(nu)
    a.nbatch, a.off, a.records = len(off) - 1, _i(off), _d(records)
    a.k, a.ids, a.w, a.chain_len = _i(out["k"]), _i(out["ids"]), _d(out["w"]), _i(out["chain_len"])
    a.force, a.found = _d(out["force"]), _i(out["found"])
    a.threads = int(threads)
    lib().orc_particle_action(C.byref(a))
    if force_models:
        ddt = np.ascontiguousarray(fields["ddtU"], dtype=np.float64)
        lib().orc_extra_force_models(C.byref(a), _d(ddt), C.c_double(float(dt)), int(force_models))
    return out


def fresh_mutable(Nc):
    """state after FoamYade::initFields (FoamYade.C:56-68)"""
    return dict(uSourceDrag=np.zeros(Nc), alpha=np.ones(Nc), uSource=np.zeros((Nc, 3)), uParticle=np.zeros((Nc, 3)))


def set_source_zero(mutable, gaussian):
    Nc = mutable["alpha"].shape[0]
    lib().orc_set_source_zero(Nc, int(gaussian), _d(mutable["uSourceDrag"]), _d(mutable["alpha"]),
                              _d(mutable["uSource"]), _d(mutable["uParticle"]))


# =====================================================================================================================
# FV half (parity unpinned -- see oracle/fv_oracle.cpp header)
# =====================================================================================================================
class FvCase(C.Structure):
    _fields_ = [("solver", C.c_int), ("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("dx", C.c_double),
                ("origin", C.c_double * 3), ("dt", C.c_double), ("nu", C.c_double), ("rho_fluid", C.c_double),
                ("rho_particle", C.c_double), ("g", C.c_double * 3),
                ("u_bc", C.c_int * 6), ("u_value", (C.c_double * 3) * 6), ("p_bc", C.c_int * 6), ("p_value", C.c_double * 6),
                ("n_outer", C.c_int), ("n_corr", C.c_int), ("n_non_orth", C.c_int), ("momentum_predictor", C.c_int),
                ("p_ref_cell", C.c_int), ("p_ref_value", C.c_double), ("p_solver", C.c_int),
                ("p_tol", C.c_double), ("p_rel_tol", C.c_double), ("p_final_tol", C.c_double), ("p_final_rel_tol", C.c_double),
                ("p_max_iter", C.c_int), ("u_tol", C.c_double), ("u_rel_tol", C.c_double), ("u_max_iter", C.c_int), ("convection_scheme", C.c_int),
                ("adjust_time_step", C.c_int), ("max_co", C.c_double), ("max_delta_t", C.c_double),
                ("u_relax", C.c_double), ("u_relax_final", C.c_double), ("p_relax", C.c_double), ("p_relax_final", C.c_double),
                ("turbulence_model", C.c_int), ("les_ck", C.c_double), ("les_ce", C.c_double), ("les_delta_coeff", C.c_double),
                ("nut_bc", C.c_int * 6), ("nut_value", C.c_double * 6), ("nut_initial", C.c_double),
                ("k_bc", C.c_int * 6), ("k_value", C.c_double * 6), ("k_initial", C.c_double), ("k_convection_scheme", C.c_int),
                ("k_tol", C.c_double), ("k_rel_tol", C.c_double), ("k_max_iter", C.c_int), ("k_relax", C.c_double),
                ("ras_cmu", C.c_double), ("ras_c1", C.c_double), ("ras_c2", C.c_double), ("ras_c3", C.c_double), ("ras_sigmak", C.c_double),
                ("ras_sigmaeps", C.c_double), ("eps_bc", C.c_int * 6), ("eps_value", C.c_double * 6), ("eps_initial", C.c_double),
                ("eps_convection_scheme", C.c_int), ("eps_tol", C.c_double), ("eps_rel_tol", C.c_double), ("eps_max_iter", C.c_int),
                ("eps_relax", C.c_double), ("wf_kappa", C.c_double), ("wf_E", C.c_double),
                ("hx", _dp), ("hy", _dp), ("hz", _dp), ("convection_limiter_k", C.c_double)]


class FvStats(C.Structure):
    _fields_ = [("courant_mean", C.c_double), ("courant_max", C.c_double), ("cont_sum_local", C.c_double),
                ("cont_global", C.c_double), ("cont_cumulative", C.c_double), ("p_iters_total", C.c_int),
                ("p_solves", C.c_int), ("u_iters_total", C.c_int), ("p_initial_residual", C.c_double),
                ("p_final_residual", C.c_double), ("delta_t", C.c_double)]


# div(phi,U): Gauss linear | upwind | linearUpwind | the NVD / TVD limited schemes
LINEAR, UPWIND, LINEAR_UPWIND, LIMITED_LINEAR, VAN_LEER, MUSCL, MINMOD, SUPERBEE, QUICK = range(9)
U_FIXED, U_ZEROGRAD, U_SLIP = 0, 1, 2         # U_SLIP: symmetryPlane / slip (normal component 0, tangential zeroGradient)
P_ZEROGRAD, P_FIXED, P_FIXEDFLUX = 0, 1, 2
XMIN, XMAX, YMIN, YMAX, ZMIN, ZMAX = range(6)


def fv_case(solver, nx, ny, nz, dx, dt, nu, rho_f=1000.0, rho_p=2650.0, g=(0, 0, 0), u_bc=None, u_val=None, p_bc=None,
            p_val=None, n_outer=1, n_corr=2, p_solver=1, origin=(0, 0, 0), momentum_predictor=1, p_tol=1e-6, p_rel_tol=0.05,
            p_final_tol=1e-6, p_final_rel_tol=0.0, u_tol=1e-5, u_rel_tol=0.0, p_max_iter=1000, u_max_iter=1000, convection_scheme=0, p_ref_cell=0,
            p_ref_value=0.0, n_non_orth=0, adjust_time_step=0, max_co=1.0, max_delta_t=1e300, u_relax=1.0, u_relax_final=0.0, p_relax=0.0,
            p_relax_final=0.0, turbulence_model=0, les_ck=0.094, les_ce=1.048, les_delta_coeff=1.0, nut_bc=None, nut_value=None, nut_initial=0.0,
            k_bc=None, k_value=None, k_initial=0.0, k_convection_scheme=1, k_tol=1e-6, k_rel_tol=0.0, k_max_iter=1000, k_relax=0.0,
            ras_cmu=0.09, ras_c1=1.44, ras_c2=1.92, ras_c3=0.0, ras_sigmak=1.0, ras_sigmaeps=1.3, eps_bc=None, eps_value=None, eps_initial=0.0,
            eps_convection_scheme=1, eps_tol=1e-6, eps_rel_tol=0.0, eps_max_iter=1000, eps_relax=0.0, wf_kappa=0.41, wf_E=9.8, grading=None, limiter_k=1.0):
    """documented defaults = the icoFoam cavity / DPMFoam tutorial settings of SURVEY.md Appendix C"""
    c = FvCase()
    c.solver, c.nx, c.ny, c.nz, c.dx, c.dt, c.nu = solver, nx, ny, nz, dx, dt, nu
    c.rho_fluid, c.rho_particle = rho_f, rho_p
    for q in range(3):
        c.origin[q] = origin[q]
        c.g[q] = g[q]
    u_bc = u_bc or [U_FIXED] * 6
    u_val = u_val or [(0, 0, 0)] * 6
    p_bc = p_bc or [P_ZEROGRAD] * 6
    p_val = p_val or [0.0] * 6
    for q in range(6):
        c.u_bc[q] = u_bc[q]
        c.p_bc[q] = p_bc[q]
        c.p_value[q] = p_val[q]
        for a in range(3):
            c.u_value[q][a] = u_val[q][a]
    c.n_outer, c.n_corr, c.n_non_orth, c.momentum_predictor = n_outer, n_corr, n_non_orth, momentum_predictor
    c.p_ref_cell, c.p_ref_value, c.p_solver = p_ref_cell, p_ref_value, p_solver
    c.p_tol, c.p_rel_tol, c.p_final_tol, c.p_final_rel_tol, c.p_max_iter = p_tol, p_rel_tol, p_final_tol, p_final_rel_tol, p_max_iter
    c.u_tol, c.u_rel_tol, c.u_max_iter = u_tol, u_rel_tol, u_max_iter
    c.convection_scheme = int(convection_scheme)
    c.convection_limiter_k = float(limiter_k)
    c.adjust_time_step, c.max_co, c.max_delta_t = int(adjust_time_step), max_co, max_delta_t
    c.u_relax, c.u_relax_final, c.p_relax, c.p_relax_final = u_relax, u_relax_final, p_relax, p_relax_final
    c.turbulence_model, c.les_ck, c.les_ce, c.les_delta_coeff, c.nut_initial = int(turbulence_model), les_ck, les_ce, les_delta_coeff, nut_initial
    for q in range(6):
        c.nut_bc[q] = (nut_bc or [0] * 6)[q]
        c.nut_value[q] = (nut_value or [0.0] * 6)[q]
        c.k_bc[q] = (k_bc or [0] * 6)[q]
        c.k_value[q] = (k_value or [0.0] * 6)[q]
    c.k_initial, c.k_convection_scheme, c.k_tol, c.k_rel_tol, c.k_max_iter, c.k_relax = k_initial, int(k_convection_scheme), k_tol, k_rel_tol, int(k_max_iter), k_relax
    c.ras_cmu, c.ras_c1, c.ras_c2, c.ras_c3, c.ras_sigmak, c.ras_sigmaeps = ras_cmu, ras_c1, ras_c2, ras_c3, ras_sigmak, ras_sigmaeps
    for q in range(6):
        c.eps_bc[q] = (eps_bc or [0] * 6)[q]
        c.eps_value[q] = (eps_value or [0.0] * 6)[q]
    c.eps_initial, c.eps_convection_scheme, c.eps_tol, c.eps_rel_tol, c.eps_max_iter, c.eps_relax = eps_initial, int(eps_convection_scheme), eps_tol, eps_rel_tol, int(eps_max_iter), eps_relax
    c.wf_kappa, c.wf_E = wf_kappa, wf_E
    if grading is not None:                      # (hx, hy, hz): cell sizes along the axes of a graded block (kept alive on the case object)
        c._grading = [np.ascontiguousarray(a, dtype=np.float64) for a in grading]
        assert [a.size for a in c._grading] == [nx, ny, nz]
        c.hx, c.hy, c.hz = (_d(a) for a in c._grading)
    return c


_fv_ready = False


def _fv_lib():
    global _fv_ready
    L = lib()
    if not _fv_ready:
        L.orc_fv_create.argtypes = [C.POINTER(FvCase)]
        L.orc_fv_create.restype = C.c_void_p
        L.orc_fv_destroy.argtypes = [C.c_void_p]
        L.orc_fv_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_fv_turbulence_correct.argtypes = [C.c_void_p]
        L.orc_fv_field_size.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_fv_get.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.orc_fv_set.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.orc_fv_ptr.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_fv_ptr.restype = _dp
        L.orc_fv_step_begin.argtypes = [C.c_void_p]
        L.orc_fv_step_end.argtypes = [C.c_void_p]
        L.orc_fv_get_stats.argtypes = [C.c_void_p, C.POINTER(FvStats)]
        L.orc_fv_apply_p.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_fv_solve_p.argtypes = [C.c_void_p, _dp, _dp]
        L.orc_fv_solve_p.restype = C.c_int
        _fv_ready = True
    return L


class FvSolver:
    """icoFoamYade / pimpleFoamYade time loop on the CPU (oracle).  step(records) = one pass of the loop body."""

    def __init__(self, case: FvCase, threads=1):
        self.case = case
        self.L = _fv_lib()
        self.h = self.L.orc_fv_create(C.byref(case))
        if not self.h:
            raise ValueError("oracle: this case is outside what the restatement carries (graded block with a turbulence model or linearUpwind)")
        self.L.orc_fv_set_threads(self.h, threads)
        self.Nc = case.nx * case.ny * case.nz
        self.gaussian = case.solver == 1
        self.mesh = None
        self.threads = threads

    def turbulence_correct(self):
        """continuousPhaseTurbulence->correct() on the current U (Smagorinsky cases)"""
        self.L.orc_fv_turbulence_correct(self.h)

    def view(self, name):
        """numpy view of the oracle's own storage (no copy)"""
        n = self.L.orc_fv_field_size(self.h, name.encode())
        assert n >= 0, name
        if n == 0:
            return np.zeros(0)
        return np.ctypeslib.as_array(self.L.orc_fv_ptr(self.h, name.encode()), shape=(n,))

    def get(self, name):
        return self.view(name).copy()

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        assert self.L.orc_fv_set(self.h, name.encode(), _d(arr)) == 0

    def stats(self):
        s = FvStats()
        self.L.orc_fv_get_stats(self.h, C.byref(s))
        return {n: getattr(s, n) for n, _ in FvStats._fields_}

    def step(self, records=None, capture=None):
        """records: (n,10) particle records or None (no particles).  Returns the per-particle force array or None.  capture: a dict that receives copies of
        alpha / uSourceDrag / uSource as setParticleAction left them (they are reset at the end of the step)"""
        self.L.orc_fv_step_begin(self.h)
        out = None
        if records is not None:
            c = self.case
            if self.mesh is None:
                g = getattr(c, "_grading", None)
                if g is None:
                    self.mesh = Mesh(c.nx, c.ny, c.nz, c.dx, tuple(c.origin))
                else:
                    self.mesh = graded_mesh(g, tuple(c.origin))
            fields = dict(U=self.view("U").reshape(-1, 3), gradP=self.view("gradP").reshape(-1, 3),
                          vGrad=self.view("vGrad").reshape(-1, 9), divT=self.view("divT").reshape(-1, 3),
                          ddtU=self.view("ddtU").reshape(-1, 3))
            mut = dict(uSourceDrag=self.view("uSourceDrag"), alpha=self.view("alpha"),
                       uSource=self.view("uSource").reshape(-1, 3), uParticle=self.view("uParticle").reshape(-1, 3))
            n = records.shape[0]
            out = particle_action(self.mesh, fields, mut, records, np.array([0, n], np.int32), self.gaussian,
                                  c.rho_particle, c.rho_fluid, c.nu, threads=self.threads,
                                  force_models=getattr(self, "force_models", 0), dt=self.stats()["delta_t"])
        if capture is not None:
            for k in ("alpha", "uSourceDrag", "uSource"):
                capture[k] = self.get(k)
        self.L.orc_fv_step_end(self.h)
        # yadeCoupling.setSourceZero() (icoFoamYade.C:147, pimpleFoamYade.C:109)
        self.L.orc_set_source_zero(self.Nc, int(self.gaussian), _d(self.view("uSourceDrag")), _d(self.view("alpha")),
                                   _d(self.view("uSource")), _d(self.view("uParticle")))
        return out

    def apply_p(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self.L.orc_fv_apply_p(self.h, _d(x), _d(y))
        return y

    def solve_p(self, rhs, x0=None):
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.zeros_like(rhs) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        it = self.L.orc_fv_solve_p(self.h, _d(rhs), _d(x))
        return x, it

    def close(self):
        if self.h:
            self.L.orc_fv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- icoFoamYade on a general polyhedral mesh in OpenFOAM's addressing (oracle/ldu_oracle.cpp) -------------------------------------------------
class LduCase(C.Structure):
    _fields_ = [("solver", C.c_int), ("dt", C.c_double), ("nu", C.c_double), ("rho_fluid", C.c_double), ("rho_particle", C.c_double),
                ("n_correctors", C.c_int), ("n_non_orth_correctors", C.c_int), ("momentum_predictor", C.c_int), ("p_ref_cell", C.c_int),
                ("p_ref_value", C.c_double), ("p_tol", C.c_double), ("p_rel_tol", C.c_double), ("p_final_tol", C.c_double),
                ("p_final_rel_tol", C.c_double), ("p_max_iter", C.c_int), ("u_tol", C.c_double), ("u_rel_tol", C.c_double), ("u_max_iter", C.c_int),
                ("u_bc", _ip), ("u_value", _dp), ("p_bc", _ip), ("p_value", _dp), ("g", C.c_double * 3), ("n_outer", C.c_int), ("u_relax", C.c_double),
                ("u_relax_final", C.c_double), ("p_relax", C.c_double), ("p_relax_final", C.c_double), ("adjust_time_step", C.c_int), ("max_co", C.c_double),
                ("max_delta_t", C.c_double), ("turbulence_model", C.c_int), ("les_ck", C.c_double), ("les_ce", C.c_double), ("les_delta_coeff", C.c_double),
                ("nut_initial", C.c_double), ("nut_bc", _ip), ("nut_value", _dp), ("convection_scheme", C.c_int), ("convection_limiter_k", C.c_double),
                ("k_initial", C.c_double), ("k_bc", _ip), ("k_value", _dp), ("k_convection_scheme", C.c_int), ("k_tol", C.c_double), ("k_rel_tol", C.c_double),
                ("k_max_iter", C.c_int), ("k_relax", C.c_double), ("ras_cmu", C.c_double), ("ras_c1", C.c_double), ("ras_c2", C.c_double), ("ras_c3", C.c_double),
                ("ras_sigmak", C.c_double), ("ras_sigmaeps", C.c_double), ("eps_initial", C.c_double), ("eps_bc", _ip), ("eps_value", _dp), ("eps_convection_scheme", C.c_int),
                ("eps_tol", C.c_double), ("eps_rel_tol", C.c_double), ("eps_max_iter", C.c_int), ("eps_relax", C.c_double)]


class LduStats(C.Structure):
    _fields_ = [("courant_mean", C.c_double), ("courant_max", C.c_double), ("cont_sum_local", C.c_double), ("cont_global", C.c_double),
                ("cont_cumulative", C.c_double), ("p_iters_total", C.c_int), ("p_solves", C.c_int), ("u_iters_total", C.c_int),
                ("p_initial_residual", C.c_double), ("p_final_residual", C.c_double), ("delta_t", C.c_double)]


_ldu_ready = False


def _ldu_lib():
    global _ldu_ready
    L = lib()
    if not _ldu_ready:
        L.orc_ldu_create.argtypes = [C.c_int, _dp, C.c_int, C.c_int, _ip, _ip, _ip, _ip, C.c_int, C.c_int, _ip, _ip, _ip, C.POINTER(LduCase)]
        L.orc_ldu_create.restype = C.c_void_p
        L.orc_ldu_destroy.argtypes = [C.c_void_p]
        L.orc_ldu_geometry.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.orc_ldu_ptr.argtypes = [C.c_void_p, C.c_char_p, _ip]
        L.orc_ldu_ptr.restype = C.POINTER(C.c_double)
        for f in ("orc_ldu_step_begin", "orc_ldu_step_end", "orc_ldu_refresh_phi"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_ldu_get_stats.argtypes = [C.c_void_p, C.POINTER(LduStats)]
        L.orc_ldu_adjust_phi_failed.argtypes = [C.c_void_p]
        L.orc_ldu_sngrad.argtypes = [C.c_void_p, _dp, _dp, C.c_int, _dp]
        _ldu_ready = True
    return L


class LduSolver:
    """mesh: dict with points (n,3), face_offsets, face_points, owner, neighbour (internal faces first), n_cells, patch_start, patch_size
    (tests/poly_meshes.py builds them); u_bc / p_bc / values per patch"""

    def __init__(self, mesh, dt, nu, u_bc, u_val, p_bc, p_val=None, n_correctors=2, n_non_orth=0, momentum_predictor=1, p_ref_cell=0, p_ref_value=0.0,
                 p_tol=1e-6, p_rel_tol=0.05, p_final_tol=1e-6, p_final_rel_tol=0.0, p_max_iter=5000, u_tol=1e-5, u_rel_tol=0.0, u_max_iter=1000,
                 rho_f=1000.0, rho_p=2650.0, solver=0, g=(0.0, 0.0, 0.0), n_outer=1, u_relax=0.0, u_relax_final=0.0, p_relax=0.0, p_relax_final=0.0,
                 adjust_time_step=0, max_co=1.0, max_delta_t=1e300, turbulence_model=0, les_ck=0.094, les_ce=1.048, les_delta_coeff=1.0, nut_initial=0.0,
                 nut_bc=None, nut_val=None, convection_scheme=0, convection_limiter_k=1.0, k_initial=0.0, k_bc=None, k_val=None, k_convection_scheme=0, k_tol=1e-6,
                 k_rel_tol=0.0, k_max_iter=1000, k_relax=0.0, ras_cmu=0.09, ras_c1=1.44, ras_c2=1.92, ras_c3=0.0, ras_sigmak=1.0, ras_sigmaeps=1.3, eps_initial=0.0,
                 eps_bc=None, eps_val=None, eps_convection_scheme=0, eps_tol=1e-6, eps_rel_tol=0.0, eps_max_iter=1000, eps_relax=0.0):
        """solver = 1: pimpleFoamYade -- step(source, alpha, drag) then takes the void fraction and the implicit drag coefficient the coupling would leave"""
        self.L = _ldu_lib()
        self.mesh = mesh
        npatch = len(mesh["patch_start"])
        self._keep = dict(points=np.ascontiguousarray(mesh["points"], np.float64), foff=np.ascontiguousarray(mesh["face_offsets"], np.int32),
                          fpts=np.ascontiguousarray(mesh["face_points"], np.int32), own=np.ascontiguousarray(mesh["owner"], np.int32),
                          nei=np.ascontiguousarray(mesh["neighbour"], np.int32), ps=np.ascontiguousarray(mesh["patch_start"], np.int32),
                          pz=np.ascontiguousarray(mesh["patch_size"], np.int32), ub=np.ascontiguousarray(u_bc, np.int32),
                          uv=np.ascontiguousarray(u_val, np.float64).reshape(npatch, 3), pb=np.ascontiguousarray(p_bc, np.int32),
                          pv=np.ascontiguousarray(p_val if p_val is not None else np.zeros(npatch), np.float64),
                          nb=np.ascontiguousarray(nut_bc if nut_bc is not None else np.zeros(npatch), np.int32),
                          nv=np.ascontiguousarray(nut_val if nut_val is not None else np.zeros(npatch), np.float64),
                          pn=(np.ascontiguousarray(mesh["patch_neighbour"], np.int32) if mesh.get("patch_neighbour") is not None else None),
                          kb=np.ascontiguousarray(k_bc if k_bc is not None else np.zeros(npatch), np.int32),
                          kv=np.ascontiguousarray(k_val if k_val is not None else np.zeros(npatch), np.float64),
                          eb=np.ascontiguousarray(eps_bc if eps_bc is not None else np.zeros(npatch), np.int32),
                          ev=np.ascontiguousarray(eps_val if eps_val is not None else np.zeros(npatch), np.float64))
        k = self._keep
        self.case = LduCase(solver, dt, nu, rho_f, rho_p, n_correctors, n_non_orth, momentum_predictor, p_ref_cell, p_ref_value, p_tol, p_rel_tol, p_final_tol,
                            p_final_rel_tol, p_max_iter, u_tol, u_rel_tol, u_max_iter, _i(k["ub"]), _d(k["uv"]), _i(k["pb"]), _d(k["pv"]), (C.c_double * 3)(*g), n_outer,
                            u_relax, u_relax_final, p_relax, p_relax_final, adjust_time_step, max_co, max_delta_t, turbulence_model, les_ck, les_ce,
                            les_delta_coeff, nut_initial, _i(k["nb"]), _d(k["nv"]), convection_scheme, convection_limiter_k, k_initial, _i(k["kb"]), _d(k["kv"]),
                            k_convection_scheme, k_tol, k_rel_tol, k_max_iter, k_relax, ras_cmu, ras_c1, ras_c2, ras_c3, ras_sigmak, ras_sigmaeps, eps_initial,
                            _i(k["eb"]), _d(k["ev"]), eps_convection_scheme, eps_tol, eps_rel_tol, eps_max_iter, eps_relax)
        self.pimple = solver == 1
        self.nc, self.nf, self.ni = int(mesh["n_cells"]), len(k["own"]), len(k["nei"])
        self.h = self.L.orc_ldu_create(k["points"].shape[0], _d(k["points"]), self.nf, self.ni, _i(k["foff"]), _i(k["fpts"]), _i(k["own"]), _i(k["nei"]),
                                       self.nc, npatch, _i(k["ps"]), _i(k["pz"]), _i(k["pn"]) if k["pn"] is not None else None, C.byref(self.case))
        if not self.h:
            raise ValueError("oracle: malformed polyhedral mesh (a boundary face outside every patch, or cyclic halves that do not pair)")
        if k["pn"] is not None:                       # cyclic pairs folded into internal faces: the solver's own face counts
            cnt = np.zeros(3)
            self.L.orc_ldu_geometry(self.h, b"counts", _d(cnt))
            self.nf, self.ni = int(cnt[0]), int(cnt[1])

    def geometry(self, name):
        size = {"C": 3 * self.nc, "V": self.nc, "Cf": 3 * self.nf, "Sf": 3 * self.nf, "magSf": self.nf, "w": self.ni, "dcNO": self.nf, "kvec": 3 * self.ni,
                "sep": 3 * self.ni, "orig_face": self.nf}[name]
        out = np.empty(size)
        assert self.L.orc_ldu_geometry(self.h, name.encode(), _d(out)) == size
        return out.reshape(-1, 3) if name in ("C", "Cf", "Sf", "kvec", "sep") else out

    def sngrad(self, cell_values, boundary_values, corrected=True):
        """corrected surface-normal gradient on the internal faces (correctedSnGrad [OF-6])"""
        out = np.empty(self.ni)
        self.L.orc_ldu_sngrad(self.h, _d(np.ascontiguousarray(cell_values, np.float64)), _d(np.ascontiguousarray(boundary_values, np.float64)), 0 if corrected else 1, _d(out))
        return out

    def view(self, name):
        n = C.c_int(0)
        ptr = self.L.orc_ldu_ptr(self.h, name.encode(), C.byref(n))
        if not ptr:
            raise KeyError(name)
        return np.ctypeslib.as_array(ptr, shape=(n.value,))

    def get(self, name):
        return self.view(name).copy()

    def set(self, name, arr):
        self.view(name)[:] = np.ascontiguousarray(arr, np.float64).ravel()
        if name == "U":
            self.L.orc_ldu_refresh_phi(self.h)

    def step(self, source=None, alpha=None, drag=None):
        """source: (nc,3) explicit momentum source uSource for this step (what the coupling would leave), or None; pimpleFoamYade: alpha (the void fraction) and
        drag (uSourceDrag, the implicit coefficient) too.  They are placed where setParticleAction sits in the loop: after the pre-coupling fields"""
        self.L.orc_ldu_step_begin(self.h)
        self.view("uSource")[:] = 0.0 if source is None else np.ascontiguousarray(source, np.float64).ravel()
        if self.pimple:
            self.view("alpha")[:] = 1.0 if alpha is None else np.ascontiguousarray(alpha, np.float64).ravel()
            self.view("uSourceDrag")[:] = 0.0 if drag is None else np.ascontiguousarray(drag, np.float64).ravel()
        self.L.orc_ldu_step_end(self.h)
        if self.pimple:                                   # yadeCoupling.setSourceZero() (pimpleFoamYade.C:109)
            self.view("alpha")[:] = 1.0; self.view("uSourceDrag")[:] = 0.0; self.view("uSource")[:] = 0.0

    def stats(self):
        s = LduStats()
        self.L.orc_ldu_get_stats(self.h, C.byref(s))
        return {n: getattr(s, n) for n, _ in LduStats._fields_}

    def close(self):
        if self.h:
            self.L.orc_ldu_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
