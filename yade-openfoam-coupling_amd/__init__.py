"""Python mirror of the reference's interface for the hot path, over the C-ABI of libfoamyade_hip.so.

The product is the shared library (HIP kernels + C++ host code, `csrc/`, ABI in `include/foamyade_hip.h`); this module
is only the ctypes stub a Python caller (tests, bench.py) needs.  Names follow the reference:

    FoamYade(mesh, U, gradP, vGrad, divT, ddtU, g, uSourceDrag, alpha, uSource, uParticle, gaussianInterp)
        .setScalarProperties(rhoP, rhoF, nu)   FoamYade.C:9-11
        .setParticleAction(dt)                 FoamYade.C:605-632
        .setSourceZero()                       FoamYade.C:556-566

There is no CPU fallback: if the library is missing or no HIP device is visible, construction raises.
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("FOAMYADE_HIP_LIB") or os.path.join(_HERE, "lib", "libfoamyade_hip.so")   # override: A/B builds only
CSRC = os.path.join(_HERE, "csrc")
MAXK = 16

FY_OK = 0
FY_ERR_NO_DEVICE = 2
FORCE_ADDED_MASS, FORCE_GAUSSIAN_TORQUE = 1, 2        # fy_set_force_models flags
FY_MEM_HOST, FY_MEM_DEVICE = 0, 1
FY_T_INT, FY_T_DOUBLE = 0, 1
FY_OP_MAX, FY_OP_SUM = 0, 1
FY_SOLVER_ICO, FY_SOLVER_PIMPLE = 0, 1
FY_BC_U_FIXED_VALUE, FY_BC_U_ZERO_GRADIENT, FY_BC_U_SLIP = 0, 1, 2
FY_BC_P_ZERO_GRADIENT, FY_BC_P_FIXED_VALUE, FY_BC_P_FIXED_FLUX = 0, 1, 2
FY_PSOLVER_PCG_JACOBI, FY_PSOLVER_PCG_MG = 0, 1
FY_CONVECTION_LINEAR, FY_CONVECTION_UPWIND, FY_CONVECTION_LINEAR_UPWIND = 0, 1, 2
FY_CONVECTION_LIMITED_LINEAR, FY_CONVECTION_VAN_LEER, FY_CONVECTION_MUSCL, FY_CONVECTION_MINMOD, FY_CONVECTION_SUPERBEE, FY_CONVECTION_QUICK = 3, 4, 5, 6, 7, 8

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class FoamYadeError(RuntimeError):
    pass


def build(verbose=False):
    """compile libfoamyade_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.run(["make", "-C", CSRC, "all"], check=True, stdout=out)
    return LIB_PATH


class MeshDesc(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("centres", _dp), ("volumes", _dp), ("bbox_min", C.c_double * 3),
                ("bbox_max", C.c_double * 3), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("dx", C.c_double),
                ("origin", C.c_double * 3), ("xf", _dp), ("yf", _dp), ("zf", _dp)]


class FieldPtrs(C.Structure):
    _fields_ = [("location", C.c_int32), ("U", _dp), ("gradP", _dp), ("vGrad", _dp), ("divT", _dp), ("ddtU", _dp),
                ("g", C.c_double * 3), ("uSourceDrag", _dp), ("alpha", _dp), ("uSource", _dp), ("uParticle", _dp)]


_SEND = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
_RECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
_BCAST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int)
_ALLRED = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("world_rank", C.c_int32), ("world_size", C.c_int32), ("local_rank", C.c_int32),
                ("local_size", C.c_int32), ("send", _SEND), ("recv", _RECV), ("bcast_world", _BCAST),
                ("bcast_local", _BCAST), ("allreduce_world", _ALLRED),
                # the optional zero-copy wire of round 4 (include/foamyade_hip.h): absent (NULL) in the Python transports
                ("describe_block", C.c_void_p), ("recv_view", C.c_void_p), ("recv_view_layout", C.c_void_p), ("recv_view_next", C.c_void_p),
                ("send_reserve", C.c_void_p), ("send_commit", C.c_void_p), ("view_region", C.c_void_p)]


class PolyMesh(C.Structure):
    _fields_ = [("n_points", C.c_int32), ("points", _dp), ("n_faces", C.c_int32), ("n_internal_faces", C.c_int32), ("face_offsets", _ip), ("face_points", _ip),
                ("owner", _ip), ("neighbour", _ip), ("n_cells", C.c_int32), ("n_patches", C.c_int32), ("patch_start", _ip), ("patch_size", _ip), ("patch_neighbour", _ip)]


class LduCase(C.Structure):
    _fields_ = [("dt", C.c_double), ("nu", C.c_double), ("rho_fluid", C.c_double), ("rho_particle", C.c_double), ("n_correctors", C.c_int32),
                ("n_non_orth_correctors", C.c_int32), ("momentum_predictor", C.c_int32), ("p_ref_cell", C.c_int32), ("p_ref_value", C.c_double),
                ("p_tol", C.c_double), ("p_rel_tol", C.c_double), ("p_final_tol", C.c_double), ("p_final_rel_tol", C.c_double), ("p_max_iter", C.c_int32),
                ("u_tol", C.c_double), ("u_rel_tol", C.c_double), ("u_max_iter", C.c_int32), ("p_solver", C.c_int32), ("u_bc", _ip), ("u_value", _dp), ("p_bc", _ip),
                ("p_value", _dp), ("solver", C.c_int32), ("n_outer_correctors", C.c_int32), ("g", C.c_double * 3), ("u_relax", C.c_double), ("u_relax_final", C.c_double),
                ("p_relax", C.c_double), ("p_relax_final", C.c_double), ("adjust_time_step", C.c_int32), ("max_co", C.c_double), ("max_delta_t", C.c_double),
                ("turbulence_model", C.c_int32), ("les_ck", C.c_double), ("les_ce", C.c_double), ("les_delta_coeff", C.c_double), ("nut_initial", C.c_double), ("nut_bc", _ip),
                ("nut_value", _dp), ("convection_scheme", C.c_int32), ("convection_limiter_k", C.c_double), ("k_initial", C.c_double), ("k_bc", _ip), ("k_value", _dp),
                ("k_convection_scheme", C.c_int32), ("k_tol", C.c_double), ("k_rel_tol", C.c_double), ("k_max_iter", C.c_int32), ("k_relax", C.c_double),
                ("ras_cmu", C.c_double), ("ras_c1", C.c_double), ("ras_c2", C.c_double), ("ras_c3", C.c_double), ("ras_sigmak", C.c_double), ("ras_sigmaeps", C.c_double),
                ("eps_initial", C.c_double), ("eps_bc", _ip), ("eps_value", _dp), ("eps_convection_scheme", C.c_int32), ("eps_tol", C.c_double), ("eps_rel_tol", C.c_double),
                ("eps_max_iter", C.c_int32), ("eps_relax", C.c_double)]


class ParticleTimings(C.Structure):
    _fields_ = [("h2d", C.c_double), ("bin", C.c_double), ("locate_deposit", C.c_double), ("finalize", C.c_double),
                ("force", C.c_double), ("d2h", C.c_double), ("total", C.c_double), ("n_particles", C.c_int64),
                ("n_pairs", C.c_int64), ("copy_in", C.c_double), ("copy_out", C.c_double), ("wire_recv", C.c_double),
                ("wire_send", C.c_double), ("bytes_in", C.c_int64), ("bytes_out", C.c_int64), ("fold", C.c_double)]


class CaseDesc(C.Structure):
    _fields_ = [("solver", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("dx", C.c_double),
                ("origin", C.c_double * 3), ("dt", C.c_double), ("nu", C.c_double), ("rho_fluid", C.c_double),
                ("rho_particle", C.c_double), ("g", C.c_double * 3),
                ("u_bc", C.c_int32 * 6), ("u_value", (C.c_double * 3) * 6),
                ("p_bc", C.c_int32 * 6), ("p_value", C.c_double * 6),
                ("n_outer_correctors", C.c_int32), ("n_correctors", C.c_int32), ("n_non_orth_correctors", C.c_int32),
                ("momentum_predictor", C.c_int32), ("p_ref_cell", C.c_int32), ("p_ref_value", C.c_double),
                ("p_solver", C.c_int32), ("p_tol", C.c_double), ("p_rel_tol", C.c_double), ("p_final_tol", C.c_double),
                ("p_final_rel_tol", C.c_double), ("p_max_iter", C.c_int32),
                ("u_tol", C.c_double), ("u_rel_tol", C.c_double), ("u_max_iter", C.c_int32), ("convection_scheme", C.c_int32),
                ("adjust_time_step", C.c_int32), ("max_co", C.c_double), ("max_delta_t", C.c_double),
                ("u_relax", C.c_double), ("u_relax_final", C.c_double), ("p_relax", C.c_double), ("p_relax_final", C.c_double),
                ("turbulence_model", C.c_int32), ("les_ck", C.c_double), ("les_ce", C.c_double), ("les_delta_coeff", C.c_double),
                ("nut_bc", C.c_int32 * 6), ("nut_value", C.c_double * 6), ("nut_initial", C.c_double),
                ("k_bc", C.c_int32 * 6), ("k_value", C.c_double * 6), ("k_initial", C.c_double), ("k_convection_scheme", C.c_int32),
                ("k_tol", C.c_double), ("k_rel_tol", C.c_double), ("k_max_iter", C.c_int32), ("k_relax", C.c_double),
                ("ras_cmu", C.c_double), ("ras_c1", C.c_double), ("ras_c2", C.c_double), ("ras_c3", C.c_double), ("ras_sigmak", C.c_double),
                ("ras_sigmaeps", C.c_double), ("eps_bc", C.c_int32 * 6), ("eps_value", C.c_double * 6), ("eps_initial", C.c_double),
                ("eps_convection_scheme", C.c_int32), ("eps_tol", C.c_double), ("eps_rel_tol", C.c_double), ("eps_max_iter", C.c_int32),
                ("eps_relax", C.c_double), ("wf_kappa", C.c_double), ("wf_E", C.c_double),
                ("hx", C.POINTER(C.c_double)), ("hy", C.POINTER(C.c_double)), ("hz", C.POINTER(C.c_double)), ("convection_limiter_k", C.c_double)]


BC_WALL_FUNCTION, BC_NUT_CALCULATED = 2, 3
TURBULENCE_LAMINAR, TURBULENCE_SMAGORINSKY, TURBULENCE_KEQN, TURBULENCE_KEPSILON = 0, 1, 2, 3
NUT_ZERO_GRADIENT, NUT_FIXED_VALUE = 0, 1


class FoamCaseInfo(C.Structure):
    _fields_ = [("start_time", C.c_double), ("end_time", C.c_double), ("delta_t", C.c_double), ("write_interval_steps", C.c_int32),
                ("n_cells", C.c_int64), ("u_name", C.c_char * 64), ("phase", C.c_char * 32), ("start_name", C.c_char * 32),
                ("patch_of_side", (C.c_char * 64) * 6), ("field_cells", C.c_int64), ("field_offset", C.c_int64)]


class StepStats(C.Structure):
    _fields_ = [("courant_mean", C.c_double), ("courant_max", C.c_double), ("cont_err_sum_local", C.c_double),
                ("cont_err_global", C.c_double), ("cont_err_cumulative", C.c_double),
                ("p_iters_total", C.c_int32), ("p_solves", C.c_int32), ("u_iters_total", C.c_int32),
                ("p_initial_residual", C.c_double), ("p_final_residual", C.c_double),
                ("ms_particle", C.c_double), ("ms_momentum", C.c_double), ("ms_pressure", C.c_double),
                ("ms_other", C.c_double), ("ms_total", C.c_double), ("delta_t", C.c_double)]


_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch's ROCm wheel bundles its own libamdhip64 (soname libamdhip64.so.7, file name libamdhip64.so)
    and libtorch_hip asks for it by FILE name, so a process that loads this library first (resolved to /opt/rocm's copy) and initialises
    torch.cuda later ends up with two HIP/ROCr runtimes side by side -- which works for a while and then fails in torch
    ("No HIP GPUs are available" after a few dozen contexts).  Loading the wheel's copy first makes both resolve to the same object.
    torch itself is not imported.  This is NOT enough for a process that works this library hard and imports torch afterwards (measured: any
    of the slab / particle test files followed by a test that does `import torch` ends in "double free or corruption" at interpreter exit, while
    the same tests with torch imported first are clean): a process that is going to use torch should import it BEFORE loading this library --
    tests/conftest.py and __graft_entry__.load_product() do."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except (OSError, ImportError, ValueError):
        pass


def lib():
    """load the product library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FoamYadeError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.fy_last_error.restype = C.c_char_p
    L.fy_create.argtypes = [C.POINTER(MeshDesc), C.POINTER(FieldPtrs), C.c_int, C.POINTER(Transport), C.c_int, C.POINTER(vp)]
    L.fy_set_scalar_properties.argtypes = [vp, C.c_double, C.c_double, C.c_double]
    L.fy_set_particle_action.argtypes = [vp, C.c_double]
    L.fy_set_force_models.argtypes = [vp, C.c_uint]
    L.fy_set_fibre_coupling.argtypes = [vp, C.c_int]
    L.fy_set_source_zero.argtypes = [vp]
    L.fy_finalize_run.argtypes = [vp, C.POINTER(C.c_int)]
    L.fy_destroy.argtypes = [vp]
    L.fy_set_num_batches.argtypes = [vp, C.c_int]
    L.fy_set_particles_host.argtypes = [vp, C.c_int, _dp, C.c_int64]
    L.fy_set_particles_device.argtypes = [vp, C.c_int, vp, C.c_int64]
    L.fy_get_forces_host.argtypes = [vp, C.c_int, _dp]
    L.fy_get_found_host.argtypes = [vp, C.c_int, _ip]
    L.fy_migrate_particles.argtypes = [vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.fy_get_particles_host.argtypes = [vp, C.c_int, _dp, C.POINTER(C.c_int64)]
    L.fy_forces_device.argtypes = [vp, C.c_int]
    L.fy_forces_device.restype = vp
    L.fy_get_stencils_host.argtypes = [vp, C.c_int, _ip, _ip, _dp, _ip]
    L.fy_nearest_cells_host.argtypes = [vp, _dp, C.c_int64, _ip]
    L.fy_get_tree_preorder.argtypes = [vp, _ip]
    L.fy_read_field_host.argtypes = [vp, C.c_char_p, _dp]
    L.fy_write_field_host.argtypes = [vp, C.c_char_p, _dp]
    L.fy_yade_dt.argtypes = [vp]
    L.fy_yade_dt.restype = C.c_double
    L.fy_interp_range.argtypes = [vp]
    L.fy_interp_range.restype = C.c_double
    L.fy_locate_walk_count.argtypes = [vp]
    L.fy_locate_walk_count.restype = C.c_longlong
    L.fy_get_particle_timings.argtypes = [vp, C.POINTER(ParticleTimings)]
    L.fy_enable_timing.argtypes = [vp, C.c_int]
    L.fy_case_defaults.argtypes = [C.POINTER(CaseDesc), C.c_int]
    L.fy_case_defaults.restype = None
    L.fy_solver_create.argtypes = [C.POINTER(CaseDesc), C.POINTER(Transport), C.c_int, C.POINTER(vp)]
    L.fy_solver_coupling.argtypes = [vp]
    L.fy_solver_coupling.restype = vp
    L.fy_solver_step.argtypes = [vp]
    L.fy_solver_get_stats.argtypes = [vp, C.POINTER(StepStats)]
    L.fy_solver_read_field_host.argtypes = [vp, C.c_char_p, _dp]
    L.fy_solver_field_count.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int64)]
    L.fy_solver_hold_sources.argtypes = [vp, C.c_int]
    L.fy_foam_case_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.fy_foam_case_desc.argtypes = [vp, C.POINTER(CaseDesc)]
    L.fy_foam_case_info_get.argtypes = [vp, C.POINTER(FoamCaseInfo)]
    L.fy_foam_case_initial_fields.argtypes = [vp, _dp, _dp]
    L.fy_foam_case_initial_nut.argtypes = [vp, _dp]
    L.fy_foam_case_initial_k.argtypes = [vp, _dp]
    L.fy_foam_case_initial_epsilon.argtypes = [vp, _dp]
    L.fy_foam_case_write_time.argtypes = [vp, vp, C.c_char_p]
    L.fy_foam_case_close.argtypes = [vp]
    L.fy_solver_write_field_host.argtypes = [vp, C.c_char_p, _dp]
    L.fy_solver_destroy.argtypes = [vp]
    L.fy_solver_apply_p_matrix_host.argtypes = [vp, _dp, _dp]
    L.fy_solver_time_p_apply.argtypes = [vp, C.c_int, _dp]
    L.fy_solver_solve_p_host.argtypes = [vp, _dp, _dp, C.POINTER(C.c_int)]
    L.fy_rccl_unique_id.argtypes = [C.c_void_p]
    L.fy_comm_create_rccl.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(vp)]
    L.fy_comm_create_host.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(vp)]
    L.fy_comm_create_local_group.argtypes = [C.c_int, C.POINTER(vp)]
    L.fy_comm_destroy.argtypes = [vp]
    L.fy_comm_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.fy_comm_stats_by_tag.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.fy_comm_selftest.argtypes = [vp, C.c_int]
    L.fy_comm_rank.argtypes = [vp]
    L.fy_comm_size.argtypes = [vp]
    L.fy_solver_create_slab.argtypes = [C.POINTER(CaseDesc), C.POINTER(Transport), C.c_int, vp, C.POINTER(vp)]
    L.fy_solver_local_cells.argtypes = [vp]
    L.fy_solver_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.fy_solver_get_kernel_timing.argtypes = [vp, C.c_char_p, _dp, C.POINTER(C.c_int64)]
    L.fy_solver_enable_exchange_timing.argtypes = [vp, C.c_int]
    L.fy_solver_get_exchange_wait.argtypes = [vp, _dp, C.POINTER(C.c_int64)]
    _lib = L
    return L


_hip_rt = None


def _hip():
    """the HIP runtime through ctypes (the stub needs a few bytes of device scratch for particle tags; no torch involved)"""
    global _hip_rt
    if _hip_rt is None:
        _hip_rt = C.CDLL("libamdhip64.so.7")       # by SONAME: resolves to the runtime this process already has (see _share_torch_hip_runtime)
        _hip_rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip_rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip_rt.hipFree.argtypes = [C.c_void_p]
        _hip_rt.hipSetDevice.argtypes = [C.c_int]
    return _hip_rt


def _check(rc):
    if rc != FY_OK:
        raise FoamYadeError(f"libfoamyade_hip error {rc}: {lib().fy_last_error().decode()}")


def _d(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous, "need C-contiguous float64"
    return a.ctypes.data_as(_dp)


def _i(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def _is_device(a):
    return hasattr(a, "data_ptr")        # a torch tensor: used purely as an HBM allocation


def _fptr(a):
    """numpy array -> host pointer; CUDA/HIP torch tensor (float64, contiguous) -> device pointer"""
    if a is None:
        return None
    if _is_device(a):
        assert a.is_cuda and a.is_contiguous() and a.element_size() == 8
        return C.cast(C.c_void_p(a.data_ptr()), _dp)
    return _d(a)


class BlockMesh:
    """what FoamYade reads from fvMesh, for a uniform hex block in blockMesh order (cell = i + nx*(j + ny*k))."""

    def __init__(self, nx, ny, nz, dx, origin=(0.0, 0.0, 0.0)):
        self.nx, self.ny, self.nz, self.dx = int(nx), int(ny), int(nz), float(dx)
        self.origin = tuple(float(o) for o in origin)
        self.n_cells = self.nx * self.ny * self.nz
        i = np.arange(nx, dtype=np.float64); j = np.arange(ny, dtype=np.float64); k = np.arange(nz, dtype=np.float64)
        Cc = np.empty((nz, ny, nx, 3))
        Cc[..., 0] = (self.origin[0] + (i + 0.5) * self.dx)[None, None, :]
        Cc[..., 1] = (self.origin[1] + (j + 0.5) * self.dx)[None, :, None]
        Cc[..., 2] = (self.origin[2] + (k + 0.5) * self.dx)[:, None, None]
        self.C = np.ascontiguousarray(Cc.reshape(-1, 3))                       # mesh.C()
        self.V = np.full(self.n_cells, self.dx * self.dx * self.dx)            # mesh.V()
        self.bbox_min = np.array(self.origin)
        self.bbox_max = np.array([self.origin[0] + nx * self.dx, self.origin[1] + ny * self.dx, self.origin[2] + nz * self.dx])

    def desc(self):
        m = MeshDesc()
        m.n_cells = self.n_cells
        m.centres, m.volumes = _d(self.C), _d(self.V)
        for q in range(3):
            m.bbox_min[q], m.bbox_max[q], m.origin[q] = self.bbox_min[q], self.bbox_max[q], self.origin[q]
        m.nx, m.ny, m.nz, m.dx = self.nx, self.ny, self.nz, self.dx
        return m


class GeneralMesh:
    """what FoamYade reads from fvMesh for ANY mesh: mesh.C(), mesh.V() and the bounding box of mesh.points() (fy_mesh_desc with nx = 0).
    Gaussian mode only -- the point-force mode needs polyMesh::findCell, for which the library has a uniform-block stand-in (BlockMesh)."""

    def __init__(self, centres, volumes, bbox_min, bbox_max):
        self.C = np.ascontiguousarray(centres, dtype=np.float64).reshape(-1, 3)
        self.V = np.ascontiguousarray(volumes, dtype=np.float64).reshape(-1)
        assert self.C.shape[0] == self.V.shape[0]
        self.n_cells = self.V.shape[0]
        self.bbox_min = np.array(bbox_min, dtype=np.float64); self.bbox_max = np.array(bbox_max, dtype=np.float64)

    def desc(self):
        m = MeshDesc()
        m.n_cells = self.n_cells
        m.centres, m.volumes = _d(self.C), _d(self.V)
        for q in range(3):
            m.bbox_min[q], m.bbox_max[q], m.origin[q] = self.bbox_min[q], self.bbox_max[q], 0.0
        m.nx = m.ny = m.nz = 0
        m.dx = 0.0
        return m


class GradedBlockMesh(GeneralMesh):
    """a graded (rectilinear) block in blockMesh cell order: cell sizes hx, hy, hz along the axes (fy_mesh_desc with nx, ny, nz AND the face planes
    xf, yf, zf): explicit k-d tree over the centres, findCell by a search along each axis -- so the point-force mode works on it too"""

    def __init__(self, hx, hy, hz, origin=(0.0, 0.0, 0.0)):
        h = [np.ascontiguousarray(a, dtype=np.float64) for a in (hx, hy, hz)]
        self.faces = [np.ascontiguousarray(o + np.concatenate([[0.0], np.cumsum(a)])) for o, a in zip(origin, h)]
        xc, yc, zc = (0.5 * (f[1:] + f[:-1]) for f in self.faces)
        self.nx, self.ny, self.nz = (a.size for a in h)
        Cc = np.empty((self.nz, self.ny, self.nx, 3))
        Cc[..., 0] = xc[None, None, :]; Cc[..., 1] = yc[None, :, None]; Cc[..., 2] = zc[:, None, None]
        V = (h[2][:, None, None] * h[1][None, :, None]) * h[0][None, None, :]
        super().__init__(Cc.reshape(-1, 3), V.reshape(-1), [f[0] for f in self.faces], [f[-1] for f in self.faces])
        self.origin = tuple(float(o) for o in origin)

    def desc(self):
        m = super().desc()
        m.nx, m.ny, m.nz = self.nx, self.ny, self.nz
        m.dx = float(np.cbrt(self.V[0]))
        for q in range(3):
            m.origin[q] = self.origin[q]
        m.xf, m.yf, m.zf = (_d(f) for f in self.faces)
        return m


class FoamYade:
    """Foam::FoamYade (FoamYade.H:57-161) over the C-ABI.  Field arguments are numpy arrays (host; staged per step) that
    stay owned by the caller, exactly like the reference's field references."""

    def __init__(self, mesh, U, gradP, vGrad, divT, ddtU, g, uSourceDrag, alpha, uSource, uParticle, gaussianInterp,
                 transport=None, device=0):
        L = lib()
        self._keep = [mesh, U, gradP, vGrad, divT, ddtU, uSourceDrag, alpha, uSource, uParticle, transport]
        f = FieldPtrs()
        f.location = FY_MEM_DEVICE if _is_device(U) else FY_MEM_HOST
        f.U, f.gradP, f.vGrad, f.divT = _fptr(U), _fptr(gradP), _fptr(vGrad), _fptr(divT)
        f.ddtU = _fptr(ddtU)
        for q in range(3):
            f.g[q] = float(g[q])
        f.uSourceDrag, f.alpha, f.uSource, f.uParticle = _fptr(uSourceDrag), _fptr(alpha), _fptr(uSource), _fptr(uParticle)
        self._h = C.c_void_p()
        self.mesh = mesh
        self.gaussian = bool(gaussianInterp)
        md = mesh.desc()
        _check(L.fy_create(C.byref(md), C.byref(f), int(self.gaussian), C.byref(transport) if transport is not None else None,
                           int(device), C.byref(self._h)))
        self._batch_n = []
        # FoamYade.H:102: a public flag the reference's callers set by assignment; it reaches the library before it is next needed (_sync_fibre)
        self.fibreCpl = False
        self._fibre_sent = False

    def _sync_fibre(self):
        if bool(self.fibreCpl) != self._fibre_sent:
            _check(lib().fy_set_fibre_coupling(self._h, 1 if self.fibreCpl else 0))
            self._fibre_sent = bool(self.fibreCpl)

    # ---- the reference's public methods
    def setScalarProperties(self, rhoP, rhoF, nu):
        _check(lib().fy_set_scalar_properties(self._h, float(rhoP), float(rhoF), float(nu)))

    def setParticleAction(self, dt):
        self._sync_fibre()
        _check(lib().fy_set_particle_action(self._h, float(dt)))

    def setForceModels(self, flags):
        """opt-in models the reference has no call site for: FORCE_ADDED_MASS (FoamYade.C:392-413) | FORCE_GAUSSIAN_TORQUE
        (FoamYade.C:465-479, commented out at :618)"""
        _check(lib().fy_set_force_models(self._h, int(flags)))

    def setFibreCoupling(self, on):
        """FoamYade::fibreCpl (FoamYade.H:102): records become 15 doubles per particle (FoamYade.C:131-136,161-165,189-198)"""
        self.fibreCpl = bool(on)
        self._sync_fibre()

    def calcHydroTimeScale(self):
        """FoamYade.C:582-585: empty in the reference (TODO + return)"""

    def sendHydroTimeScale(self, yProc=None):
        """FoamYade.C:587-590: empty in the reference"""

    def setSourceZero(self):
        _check(lib().fy_set_source_zero(self._h))

    def finalizeRun(self):
        """FoamYade::finalizeRun (FoamYade.C:595-599): the value Yade's rank 0 broadcasts (10: finalize MPI)"""
        v = C.c_int(-1)
        _check(lib().fy_finalize_run(self._h, C.byref(v)))
        return v.value

    # ---- direct mode (no Yade peer)
    def setParticles(self, batches):
        """batches: list of (n,10) float64 record arrays ((n,15) with fibre coupling), one per Yade proc, processed in order."""
        L = lib()
        self._sync_fibre()
        _check(L.fy_set_num_batches(self._h, len(batches)))
        self._batch_n = []
        for b, rec in enumerate(batches):
            rec = np.ascontiguousarray(rec, dtype=np.float64).reshape(-1, 15 if self.fibreCpl else 10)
            _check(L.fy_set_particles_host(self._h, b, _d(rec), rec.shape[0]))
            self._batch_n.append(rec.shape[0])

    def setParticlesDevice(self, batches):
        """batches: list of CUDA/HIP float64 tensors (n,10) already resident in HBM (borrowed, not copied)."""
        L = lib()
        self._sync_fibre()
        _check(L.fy_set_num_batches(self._h, len(batches)))
        self._batch_n = []
        self._keep_rec = list(batches)
        for b, rec in enumerate(batches):
            assert rec.is_cuda and rec.is_contiguous() and rec.element_size() == 8
            n = rec.numel() // (15 if self.fibreCpl else 10)
            _check(L.fy_set_particles_device(self._h, b, C.c_void_p(rec.data_ptr()), n))
            self._batch_n.append(n)

    def forces(self, batch=0):
        n = self._batch_n[batch]
        out = np.zeros((n, 6))
        _check(lib().fy_get_forces_host(self._h, batch, _d(out)))
        return out

    def found(self, batch=0):
        n = self._batch_n[batch]
        out = np.zeros(n, dtype=np.int32)
        _check(lib().fy_get_found_host(self._h, batch, _i(out)))
        return out

    def stencils(self, batch=0):
        n = self._batch_n[batch]
        k = np.zeros(n, np.int32); ids = np.full((n, MAXK), -1, np.int32); w = np.zeros((n, MAXK)); chain = np.zeros(n, np.int32)
        _check(lib().fy_get_stencils_host(self._h, batch, _i(k), _i(ids), _d(w), _i(chain)))
        return k, ids, w, chain

    def nearest_cells(self, pos):
        """meshTree::nearestCell (meshTree.C:66-135) of every row of pos (n,3)"""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(pos.shape[0], dtype=np.int32)
        _check(lib().fy_nearest_cells_host(self._h, _d(pos), pos.shape[0], _i(out)))
        return out

    def tree_preorder(self):
        out = np.zeros(self.mesh.n_cells, dtype=np.int32)
        _check(lib().fy_get_tree_preorder(self._h, _i(out)))
        return out

    def enable_timing(self, on=True):
        _check(lib().fy_enable_timing(self._h, int(on)))

    def timings(self):
        t = ParticleTimings()
        _check(lib().fy_get_particle_timings(self._h, C.byref(t)))
        return {n: getattr(t, n) for n, _ in ParticleTimings._fields_}

    @property
    def yadeDT(self):
        return lib().fy_yade_dt(self._h)

    @property
    def interpRange(self):
        return lib().fy_interp_range(self._h)

    @property
    def locate_walk_count(self):
        """particles of the last step's (last batch's) Gaussian locate that took the tree walk instead of the candidate lists; -1: lists not in use"""
        return lib().fy_locate_walk_count(self._h)

    @property
    def locate_stack_depth(self):
        """entries per lane of the explicit walk's LDS stack in the last step (0: the full depth; -1: no explicit tree)"""
        L = lib()
        L.fy_locate_stack_depth.argtypes = [C.c_void_p]
        return L.fy_locate_stack_depth(self._h)

    def close(self):
        if self._h:
            lib().fy_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def case_defaults(solver):
    """fy_case_defaults: the documented icoFoam-cavity / DPMFoam-tutorial settings (the reference ships no case)."""
    c = CaseDesc()
    lib().fy_case_defaults(C.byref(c), int(solver))
    return c


def make_case(solver, nx, ny, nz, dx, dt, nu, rho_f=1000.0, rho_p=2650.0, g=(0, 0, 0), u_bc=None, u_val=None, p_bc=None,
              p_val=None, origin=(0, 0, 0), grading=None, **kw):
    """grading = (hx, hy, hz): cell sizes along the axes of a graded block (fy_case_desc.hx / hy / hz); dx is then ignored"""
    c = case_defaults(solver)
    if grading is not None:
        c._grading = [np.ascontiguousarray(a, dtype=np.float64) for a in grading]      # kept alive on the case object
        assert [a.size for a in c._grading] == [nx, ny, nz]
        c.hx, c.hy, c.hz = (a.ctypes.data_as(C.POINTER(C.c_double)) for a in c._grading)
    c.nx, c.ny, c.nz, c.dx, c.dt, c.nu, c.rho_fluid, c.rho_particle = nx, ny, nz, dx, dt, nu, rho_f, rho_p
    for q in range(3):
        c.g[q] = g[q]
        c.origin[q] = origin[q]
    for q in range(6):
        if u_bc is not None:
            c.u_bc[q] = u_bc[q]
        if p_bc is not None:
            c.p_bc[q] = p_bc[q]
        if p_val is not None:
            c.p_value[q] = p_val[q]
        if u_val is not None:
            for a in range(3):
                c.u_value[q][a] = u_val[q][a]
    for k, v in kw.items():
        assert hasattr(c, k), k
        if isinstance(v, (list, tuple)):
            arr = getattr(c, k)
            for q, x in enumerate(v):
                arr[q] = x
        else:
            setattr(c, k, v)
    return c


class Solver:
    """the icoFoamYade / pimpleFoamYade executables' time loop (icoFoamYade.C:65-149, pimpleFoamYade.C:60-114): step() is one
    pass of the loop body, including yadeCoupling.setParticleAction and setSourceZero."""

    def __init__(self, case: CaseDesc, transport=None, device=0, comm=None):
        """comm: a communicator handle (c_void_p) for z-slab mode -- the case then describes the GLOBAL block and this object is
        one rank's slab (collective construction); None = single domain"""
        self.case = case
        self._h = C.c_void_p()
        self._keep = transport
        tr = C.byref(transport) if transport is not None else None
        if comm is None:
            _check(lib().fy_solver_create(C.byref(case), tr, int(device), C.byref(self._h)))
            self.n_slabs, self.rank = 1, 0
        else:
            _check(lib().fy_solver_create_slab(C.byref(case), tr, int(device), comm, C.byref(self._h)))
            self.n_slabs, self.rank = lib().fy_comm_size(comm), lib().fy_comm_rank(comm)
        self.nz_local = case.nz // self.n_slabs
        self.n_cells = case.nx * case.ny * self.nz_local        # owned cells
        self._cpl = C.c_void_p(lib().fy_solver_coupling(self._h))
        self.device = int(device)
        self._batch_n = []

    def _size(self, name):
        cnt = C.c_int64(0)
        _check(lib().fy_solver_field_count(self._h, name.encode(), C.byref(cnt)))
        return cnt.value

    def get(self, name):
        out = np.zeros(self._size(name))
        _check(lib().fy_solver_read_field_host(self._h, name.encode(), _d(out)))
        return out

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        assert arr.size == self._size(name)
        _check(lib().fy_solver_write_field_host(self._h, name.encode(), _d(arr)))

    def particles(self):
        """the records of batch 0 as the library holds them now (after a migration: this slab's new population)"""
        n = C.c_int64(0)
        _check(lib().fy_get_particles_host(self._cpl, 0, None, C.byref(n)))
        out = np.zeros((n.value, 10))
        if n.value:
            _check(lib().fy_get_particles_host(self._cpl, 0, _d(out), C.byref(n)))
        self._batch_n[0:1] = [n.value]
        return out

    def migrate(self, tags=None, capacity=None):
        """fy_migrate_particles (slab mode, collective): returns (n_local, tags of the new local population or None)"""
        n = C.c_int64(0)
        if tags is None:
            _check(lib().fy_migrate_particles(self._cpl, None, 0, C.byref(n)))
            self._batch_n[0:1] = [n.value]
            return n.value, None
        tags = np.ascontiguousarray(tags, dtype=np.int64)
        cap = int(capacity if capacity is not None else 2 * tags.size + 1024)
        hip = _hip()
        buf = C.c_void_p()
        if hip.hipSetDevice(self.device) != 0 or hip.hipMalloc(C.byref(buf), C.c_size_t(8 * cap)) != 0:
            raise FoamYadeError("hipMalloc for the tag buffer failed")
        try:
            hip.hipMemcpy(buf, tags.ctypes.data_as(C.c_void_p), C.c_size_t(8 * tags.size), 1)          # hipMemcpyHostToDevice
            _check(lib().fy_migrate_particles(self._cpl, buf, cap, C.byref(n)))
            out = np.zeros(n.value, dtype=np.int64)
            if n.value:
                hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), buf, C.c_size_t(8 * n.value), 2)        # hipMemcpyDeviceToHost
        finally:
            hip.hipFree(buf)
        self._batch_n[0:1] = [n.value]
        return n.value, out

    def hold_sources(self, on=True):
        """defer the step's closing setSourceZero to the start of the next step (runTime.write() sees this step's alpha / uSource)"""
        _check(lib().fy_solver_hold_sources(self._h, int(on)))

    def set_force_models(self, flags):
        """fy_set_force_models on the embedded coupling object (Gaussian torque / added mass, off by default)"""
        _check(lib().fy_set_force_models(self._cpl, int(flags)))

    def set_particles(self, records):
        """direct mode: the particle records the next step() will couple with ((n,10) host array, or None for none)"""
        L = lib()
        _check(L.fy_set_num_batches(self._cpl, 1))
        if records is None:
            rec = np.zeros((0, 10))
        else:
            rec = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 10)
        _check(L.fy_set_particles_host(self._cpl, 0, _d(rec) if rec.size else None, rec.shape[0]))
        self._batch_n = [rec.shape[0]]

    def set_particles_device(self, rec):
        L = lib()
        _check(L.fy_set_num_batches(self._cpl, 1))
        n = rec.numel() // 10
        self._keep_rec = rec
        _check(L.fy_set_particles_device(self._cpl, 0, C.c_void_p(rec.data_ptr()), n))
        self._batch_n = [n]

    def forces(self):
        out = np.zeros((self._batch_n[0], 6))
        _check(lib().fy_get_forces_host(self._cpl, 0, _d(out)))
        return out

    def found(self):
        out = np.zeros(self._batch_n[0], dtype=np.int32)
        _check(lib().fy_get_found_host(self._cpl, 0, _i(out)))
        return out

    def stencils(self):
        """(k, ids, w, chain) of batch 0 as the last step's setParticleAction left them (Gaussian mode)"""
        n = self._batch_n[0]
        k = np.zeros(n, np.int32); ids = np.full((n, MAXK), -1, np.int32); w = np.zeros((n, MAXK)); chain = np.zeros(n, np.int32)
        _check(lib().fy_get_stencils_host(self._cpl, 0, _i(k), _i(ids), _d(w), _i(chain)))
        return k, ids, w, chain

    def step(self):
        _check(lib().fy_solver_step(self._h))

    def stats(self):
        s = StepStats()
        _check(lib().fy_solver_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in StepStats._fields_}

    def apply_p(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        _check(lib().fy_solver_apply_p_matrix_host(self._h, _d(x), _d(y)))
        return y

    def solve_p(self, rhs, x0=None):
        """A x = rhs with the case's pressure solver and the matrix of the last step; returns (x, iterations)"""
        rhs = np.ascontiguousarray(rhs, dtype=np.float64)
        x = np.zeros_like(rhs) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
        it = C.c_int(0)
        _check(lib().fy_solver_solve_p_host(self._h, _d(rhs), _d(x), C.byref(it)))
        return x, it.value

    def time_p_apply(self, reps=50):
        ms = C.c_double()
        _check(lib().fy_solver_time_p_apply(self._h, int(reps), C.byref(ms)))
        return ms.value

    def enable_kernel_timing(self, on=True):
        _check(lib().fy_solver_enable_kernel_timing(self._h, int(on)))

    def kernel_timing(self, name):
        """(total_ms, launches) of one instrumented kernel since enable_kernel_timing(True)"""
        ms = C.c_double(); n = C.c_int64()
        _check(lib().fy_solver_get_kernel_timing(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def enable_exchange_timing(self, on=True):
        _check(lib().fy_solver_enable_exchange_timing(self._h, int(on)))

    def exchange_wait(self):
        """{phase: (ms the stream waited for slab exchanges, number of waits)} since enable_exchange_timing(True)"""
        ms = (C.c_double * 4)(); n = (C.c_int64 * 4)()
        _check(lib().fy_solver_get_exchange_wait(self._h, ms, n))
        return {nm: (ms[q], n[q]) for q, nm in enumerate(("step_start", "particle", "momentum", "corrector"))}

    def coupling_timings(self):
        t = ParticleTimings()
        _check(lib().fy_get_particle_timings(self._cpl, C.byref(t)))
        return {n: getattr(t, n) for n, _ in ParticleTimings._fields_}

    @property
    def locate_walk_count(self):
        """see FoamYade.locate_walk_count"""
        return lib().fy_locate_walk_count(self._cpl)

    def enable_particle_timing(self, on=True):
        _check(lib().fy_enable_timing(self._cpl, int(on)))

    def close(self):
        if self._h:
            lib().fy_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_unique_id():
    buf = (C.c_char * 128)()
    _check(lib().fy_rccl_unique_id(buf))
    return bytes(buf)


def rccl_comm(rank, size, id128, device):
    h = C.c_void_p()
    buf = (C.c_char * 128).from_buffer_copy(id128)
    _check(lib().fy_comm_create_rccl(int(rank), int(size), buf, int(device), C.byref(h)))
    return h


_CB_SENDRECV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
_CB_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
_CB_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class CommCallbacks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("sendrecv", _CB_SENDRECV), ("allreduce", _CB_ALLREDUCE), ("allgather", _CB_ALLGATHER)]


class GlooHostComm:
    """fy_comm_create_host over torch.distributed (gloo, CPU tensors): one process per slab with the planes staged through host memory --
    the deployment shape of the RCCL back-end on a machine where RCCL cannot run (all ranks on one GPU).  .handle goes to Solver(comm=)."""

    def __init__(self, dist, group=None):
        """group: the (gloo) process group the callbacks use; None = the default group.  Ranks are the job's global ranks either way."""
        import torch
        self.dist, self.torch, self.group = dist, torch, group
        self.rank, self.size = dist.get_rank(), dist.get_world_size()

        def view(ptr, n):
            return torch.frombuffer((C.c_double * n).from_address(ptr), dtype=torch.float64)

        def sendrecv(user, su, n_su, rd, n_rd, sd, n_sd, ru, n_ru):
            try:
                ops = []
                if n_su: ops.append(dist.isend(view(su, n_su), self.rank + 1, group=group, tag=1))
                if n_sd: ops.append(dist.isend(view(sd, n_sd), self.rank - 1, group=group, tag=2))
                if n_rd: ops.append(dist.irecv(view(rd, n_rd), self.rank - 1, group=group, tag=1))
                if n_ru: ops.append(dist.irecv(view(ru, n_ru), self.rank + 1, group=group, tag=2))
                for o in ops:
                    o.wait()
                return 0
            except Exception as e:                       # noqa: BLE001  (an exception must not cross the C frame)
                print(f"[GlooHostComm rank {self.rank}] sendrecv: {e}", file=sys.stderr, flush=True)
                return 1

        def allreduce(user, buf, n, is_max):
            try:
                dist.all_reduce(view(buf, n), op=dist.ReduceOp.MAX if is_max else dist.ReduceOp.SUM, group=group)
                return 0
            except Exception as e:                       # noqa: BLE001
                print(f"[GlooHostComm rank {self.rank}] allreduce: {e}", file=sys.stderr, flush=True)
                return 1

        def allgather(user, send, recv, n):
            try:
                out = view(recv, n * self.size)
                dist.all_gather_into_tensor(out, view(send, n).clone(), group=group)
                return 0
            except Exception as e:                       # noqa: BLE001
                print(f"[GlooHostComm rank {self.rank}] allgather: {e}", file=sys.stderr, flush=True)
                return 1

        self._keep = (_CB_SENDRECV(sendrecv), _CB_ALLREDUCE(allreduce), _CB_ALLGATHER(allgather))
        self._cb = CommCallbacks(None, *self._keep)
        self.handle = C.c_void_p()
        self._create()

    def _create(self):
        _check(lib().fy_comm_create_host(self.rank, self.size, C.byref(self._cb), C.byref(self.handle)))

    def stats(self):
        out = (C.c_uint64 * 4)()
        _check(lib().fy_comm_stats(self.handle, out))
        return dict(exchanges=out[0], allreduces=out[1], allgathers=out[2], bytes=out[3])

    def close(self):
        if self.handle:
            lib().fy_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class GlooIpcComm(GlooHostComm):
    """fy_comm_create_ipc: one process per slab, the planes and scalars written straight into the peers' device windows (hipIpc) by this rank's kernels;
    torch.distributed (gloo) carries the bootstrap only -- the all-gather of the window handles and the closing barrier.  Runs with the ranks on the GPUs of
    one node or all on ONE GPU.  close() is collective (every rank must call it, before the process group goes away)."""

    def __init__(self, dist, device=0, group=None):
        self.device = int(device)
        super().__init__(dist, group)

    def _create(self):
        L = lib()
        L.fy_comm_create_ipc.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        _check(L.fy_comm_create_ipc(self.rank, self.size, C.byref(self._cb), self.device, C.byref(self.handle)))


def comm_selftest(comm, device=0):
    """collective known-answer run of the operations the slab solver uses on this communicator (fy_comm_selftest); raises on a mismatch"""
    _check(lib().fy_comm_selftest(comm, int(device)))


def local_comm_group(n):
    """n communicators talking to each other inside this process (fy_comm_create_local_group), one per host thread"""
    arr = (C.c_void_p * n)()
    _check(lib().fy_comm_create_local_group(int(n), arr))
    return [C.c_void_p(arr[r]) for r in range(n)]


class FoamCase:
    """an OpenFOAM case directory as icoFoamYade / pimpleFoamYade would open it (fy_foam_case_*): .case is the CaseDesc for Solver(),
    .initial_fields() the start-time U and p, .write(solver, time_name) = runTime.write()"""

    def __init__(self, case_dir, solver, processor=None):
        """processor = (rank, nranks): the field files of <case>/processor<rank> of a decomposed case (fy_foam_case_open_processor)"""
        h = C.c_void_p()
        if processor is None:
            _check(lib().fy_foam_case_open(str(case_dir).encode(), int(solver), C.byref(h)))
        else:
            _check(lib().fy_foam_case_open_processor(str(case_dir).encode(), int(solver), int(processor[0]), int(processor[1]), C.byref(h)))
        self._h = h
        self.case = CaseDesc()
        _check(lib().fy_foam_case_desc(self._h, C.byref(self.case)))
        info = FoamCaseInfo()
        _check(lib().fy_foam_case_info_get(self._h, C.byref(info)))
        self.start_time, self.end_time, self.delta_t = info.start_time, info.end_time, info.delta_t
        self.write_interval_steps, self.n_cells = info.write_interval_steps, info.n_cells
        self.field_cells, self.field_offset = info.field_cells, info.field_offset
        self.u_name, self.phase, self.start_name = info.u_name.decode(), info.phase.decode(), info.start_name.decode()
        self.patch_of_side = [bytes(info.patch_of_side[s]).split(b"\0", 1)[0].decode() for s in range(6)]

    def initial_fields(self):
        U = np.zeros((self.field_cells, 3)); p = np.zeros(self.field_cells)
        _check(lib().fy_foam_case_initial_fields(self._h, _d(U), _d(p)))
        return U, p

    def initial_nut(self):
        nut = np.zeros(self.field_cells)
        _check(lib().fy_foam_case_initial_nut(self._h, _d(nut)))
        return nut

    def initial_k(self):
        k = np.zeros(self.field_cells)
        _check(lib().fy_foam_case_initial_k(self._h, _d(k)))
        return k

    def initial_epsilon(self):
        e = np.zeros(self.field_cells)
        _check(lib().fy_foam_case_initial_epsilon(self._h, _d(e)))
        return e

    def write(self, solver, time_name):
        _check(lib().fy_foam_case_write_time(self._h, solver._h, str(time_name).encode()))

    def write_fields(self, time_name, U, p, alpha=None, nut=None, k=None, epsilon=None):
        """runTime.write() from host arrays holding this case's (or processor directory's) cells: fy_foam_case_write_fields"""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (U, p, alpha, nut, k, epsilon)]
        ptr = [None if a is None else a.ctypes.data_as(_dp) for a in arrs]
        L = lib()
        L.fy_foam_case_write_fields.argtypes = [C.c_void_p, C.c_char_p] + [_dp] * 6
        _check(L.fy_foam_case_write_fields(self._h, str(time_name).encode(), *ptr))

    def close(self):
        if self._h:
            lib().fy_foam_case_close(self._h)
            self._h = None


class GeneralFoamCase(FoamCase):
    """an icoFoamYade case directory whose constant/polyMesh is any mesh of wall / patch boundaries (fy_foam_case_open_general): .pm / .ldu_case are the
    structs for fy_ldu_solver_create (LduSolver.from_foam_case), .mesh the arrays as numpy copies, .patch_names the boundary file's order"""

    def __init__(self, case_dir, solver=FY_SOLVER_ICO):
        L = lib()
        L.fy_foam_case_open_general.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.fy_foam_case_poly_mesh.argtypes = [C.c_void_p, C.POINTER(PolyMesh)]
        L.fy_foam_case_ldu_desc.argtypes = [C.c_void_p, C.POINTER(LduCase)]
        L.fy_foam_case_patch_name.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
        L.fy_foam_case_write_time_ldu.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        h = C.c_void_p()
        _check(L.fy_foam_case_open_general(str(case_dir).encode(), int(solver), C.byref(h)))
        self._h = h
        self.pm, self.ldu_case = PolyMesh(), LduCase()
        _check(L.fy_foam_case_poly_mesh(self._h, C.byref(self.pm)))
        _check(L.fy_foam_case_ldu_desc(self._h, C.byref(self.ldu_case)))
        info = FoamCaseInfo()
        _check(L.fy_foam_case_info_get(self._h, C.byref(info)))
        self.start_time, self.end_time, self.delta_t = info.start_time, info.end_time, info.delta_t
        self.write_interval_steps, self.n_cells = info.write_interval_steps, info.n_cells
        self.field_cells, self.field_offset = info.field_cells, info.field_offset
        self.u_name, self.phase, self.start_name = info.u_name.decode(), info.phase.decode(), info.start_name.decode()
        m = self.pm
        arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt) if n else np.zeros(0, dt)
        nfp = int(m.face_offsets[m.n_faces])
        self.mesh = dict(points=arr(m.points, 3 * m.n_points, np.float64).reshape(-1, 3), face_offsets=arr(m.face_offsets, m.n_faces + 1, np.int32),
                         face_points=arr(m.face_points, nfp, np.int32), owner=arr(m.owner, m.n_faces, np.int32), neighbour=arr(m.neighbour, m.n_internal_faces, np.int32),
                         n_cells=int(m.n_cells), patch_start=arr(m.patch_start, m.n_patches, np.int32), patch_size=arr(m.patch_size, m.n_patches, np.int32),
                         patch_neighbour=(arr(m.patch_neighbour, m.n_patches, np.int32) if m.patch_neighbour else None))
        self.patch_names = []
        for pa in range(m.n_patches):
            buf = C.create_string_buffer(128)
            _check(L.fy_foam_case_patch_name(self._h, pa, buf, 128))
            self.patch_names.append(buf.value.decode())
        self.u_bc = [int(self.ldu_case.u_bc[q]) for q in range(m.n_patches)]
        self.p_bc = [int(self.ldu_case.p_bc[q]) for q in range(m.n_patches)]
        self.u_value = np.array([[self.ldu_case.u_value[3 * q + a] for a in range(3)] for q in range(m.n_patches)])
        self.p_value = np.array([self.ldu_case.p_value[q] for q in range(m.n_patches)])

    def write(self, solver, time_name):
        _check(lib().fy_foam_case_write_time_ldu(self._h, solver._h, str(time_name).encode()))


class VirtualSlabs:
    """N z-slabs of one block as N Solver objects inside this process (LocalComm back-end, one thread per slab): the test double
    of the one-process-per-GPU RCCL deployment -- identical solver code, only the communicator differs."""

    def __init__(self, case: CaseDesc, n_slabs, device=0, transports=None):
        """transports: one fy_transport per slab (a Yade peer that talks to every solver rank), or None"""
        import threading
        self._threading = threading
        self.case, self.n = case, int(n_slabs)
        arr = (C.c_void_p * self.n)()
        _check(lib().fy_comm_create_local_group(self.n, arr))
        self.comms = [C.c_void_p(arr[r]) for r in range(self.n)]
        self.solvers = [None] * self.n
        self._each(lambda r: self.solvers.__setitem__(r, Solver(case, transport=transports[r] if transports else None, device=device, comm=self.comms[r])))
        self.nz_local = case.nz // self.n

    def _each(self, fn):
        """run fn(rank) on one thread per slab (the calls are collective and block on each other)"""
        errs = []

        def run(r):
            try:
                fn(r)
            except Exception as e:          # noqa: BLE001
                errs.append((r, e))
        ts = [self._threading.Thread(target=run, args=(r,)) for r in range(self.n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0][1]

    def step(self):
        self._each(lambda r: self.solvers[r].step())

    def get(self, name):
        """global field assembled from the slabs' owned parts (cell fields and phi_z; x/y face fields concatenate likewise)"""
        parts = [s.get(name) for s in self.solvers]
        if name == "phi_z":      # interface planes are held by both neighbours: keep the lower slab's copy
            pl = self.case.nx * self.case.ny
            return np.concatenate([p[:-pl] for p in parts[:-1]] + [parts[-1]])
        return np.concatenate(parts)

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        sizes = [s._size(name) for s in self.solvers]
        assert name != "phi_z" and sum(sizes) == arr.size
        offs = np.concatenate([[0], np.cumsum(sizes)])
        self._each(lambda r: self.solvers[r].set(name, arr[offs[r]:offs[r + 1]]))

    def set_particles(self, records):
        """split the records by the slab their z lies in (every rank gets exactly the particles inside its own slab)"""
        c = self.case
        rec = np.zeros((0, 10)) if records is None else np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 10)
        z = (rec[:, 2] - c.origin[2]) / c.dx
        owner = np.clip(np.floor(z / self.nz_local).astype(int), 0, self.n - 1)
        self._owner_idx = [np.nonzero(owner == r)[0] for r in range(self.n)]
        for r, s in enumerate(self.solvers):
            s.set_particles(rec[self._owner_idx[r]])
        self._n_part = rec.shape[0]

    def set_particles_all(self, records):
        """hand EVERY slab the full record set (what the reference's serial-Yade broadcast does, FoamYade.C:176-183): each rank
        locates only the particles whose containing cell lies in its own planes (SlabOwn), all others are 'not found' there"""
        rec = np.zeros((0, 10)) if records is None else np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 10)
        for s in self.solvers:
            s.set_particles(rec)
        self._owner_idx = None
        self._n_part = rec.shape[0]

    def migrate(self, tags_per_slab):
        """collective fy_migrate_particles; tags_per_slab: list of int64 arrays; returns the new per-slab tag arrays"""
        out = [None] * self.n
        self._each(lambda r: out.__setitem__(r, self.solvers[r].migrate(tags_per_slab[r])[1]))
        return out

    def forces(self):
        if self._owner_idx is None:          # full set everywhere: non-owners hold zeros, the sum is the serial protocol's all-reduce
            return sum(s.forces() for s in self.solvers)
        out = np.zeros((self._n_part, 6))
        for r, s in enumerate(self.solvers):
            if len(self._owner_idx[r]):
                out[self._owner_idx[r]] = s.forces()
        return out

    def stats(self):
        return [s.stats() for s in self.solvers]

    def comm_stats(self, rank=0):
        """(neighbour exchanges, all-reduces, all-gathers, bytes sent) issued so far by one slab's communicator"""
        out = (C.c_uint64 * 4)()
        _check(lib().fy_comm_stats(self.comms[rank], out))
        return tuple(int(v) for v in out)

    def comm_stats_by_tag(self, rank=0):
        """{phase: (exchanges, all-reduces, all-gathers)} issued so far by one slab's communicator"""
        buf = C.create_string_buffer(1 << 14)
        _check(lib().fy_comm_stats_by_tag(self.comms[rank], buf, len(buf)))
        out = {}
        for line in buf.value.decode().splitlines():
            nm, a, b, c = line.split()
            out[nm] = (int(a), int(b), int(c))
        return out

    def close(self):
        for s in self.solvers:
            if s is not None:
                s.close()
        for cm in self.comms:
            lib().fy_comm_destroy(cm)
        self.solvers, self.comms = [], []


class LduSolver:
    """icoFoamYade's loop body on a general polyhedral mesh in OpenFOAM's addressing (fy_ldu_solver, include/foamyade_hip.h).  mesh: dict with points (n,3),
    face_offsets, face_points, owner, neighbour, n_cells, patch_start, patch_size (tests/poly_meshes.py builds them); u_bc / p_bc / values per patch; the
    controls are fy_ldu_case's (defaults: the icoFoam cavity tutorial's)"""

    @staticmethod
    def _bind():
        L = lib()
        L.fy_ldu_case_defaults.argtypes = [C.POINTER(LduCase)]; L.fy_ldu_case_defaults.restype = None
        L.fy_ldu_solver_create.argtypes = [C.POINTER(PolyMesh), C.POINTER(LduCase), C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        for f in ("fy_ldu_solver_step", "fy_ldu_solver_destroy"):
            getattr(L, f).argtypes = [C.c_void_p]
        L.fy_ldu_solver_get_stats.argtypes = [C.c_void_p, C.POINTER(StepStats)]
        L.fy_ldu_solver_coupling.argtypes = [C.c_void_p]; L.fy_ldu_solver_coupling.restype = C.c_void_p
        L.fy_ldu_solver_field_count.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
        L.fy_ldu_solver_read_field_host.argtypes = [C.c_void_p, C.c_char_p, _dp]
        L.fy_ldu_solver_write_field_host.argtypes = [C.c_void_p, C.c_char_p, _dp]

    def __init__(self, mesh, dt, nu, u_bc, u_val, p_bc, p_val=None, device=0, transport=None, nut_bc=None, nut_val=None, k_bc=None, k_val=None, eps_bc=None, eps_val=None, **controls):
        L = lib()
        self._bind()
        npatch = len(mesh["patch_start"])
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        k = self._keep = dict(points=np.ascontiguousarray(mesh["points"], np.float64), foff=i32(mesh["face_offsets"]), fpts=i32(mesh["face_points"]), own=i32(mesh["owner"]),
                              nei=i32(mesh["neighbour"]), ps=i32(mesh["patch_start"]), pz=i32(mesh["patch_size"]), ub=i32(u_bc),
                              uv=np.ascontiguousarray(u_val, np.float64).reshape(npatch, 3), pb=i32(p_bc),
                              pv=np.ascontiguousarray(p_val if p_val is not None else np.zeros(npatch), np.float64))
        self.pm = PolyMesh(k["points"].shape[0], _d(k["points"]), len(k["own"]), len(k["nei"]), _i(k["foff"]), _i(k["fpts"]), _i(k["own"]), _i(k["nei"]), int(mesh["n_cells"]),
                           npatch, _i(k["ps"]), _i(k["pz"]), None)
        if mesh.get("patch_neighbour") is not None:              # cyclic pairs (fy_poly_mesh.patch_neighbour)
            k["pn"] = i32(mesh["patch_neighbour"])
            self.pm.patch_neighbour = _i(k["pn"])
        self.case = LduCase()
        L.fy_ldu_case_defaults(C.byref(self.case))
        self.case.dt, self.case.nu = dt, nu
        names = dict(n_non_orth="n_non_orth_correctors", rho_f="rho_fluid", rho_p="rho_particle")
        for key, v in controls.items():
            if isinstance(v, (list, tuple)):
                arr = getattr(self.case, names.get(key, key))
                for q, x in enumerate(v):
                    arr[q] = x
            else:
                setattr(self.case, names.get(key, key), v)
        self.case.u_bc, self.case.u_value, self.case.p_bc, self.case.p_value = _i(k["ub"]), _d(k["uv"]), _i(k["pb"]), _d(k["pv"])
        if nut_bc is not None:
            k["nb"], k["nv"] = i32(nut_bc), np.ascontiguousarray(nut_val if nut_val is not None else np.zeros(npatch), np.float64)
            self.case.nut_bc, self.case.nut_value = _i(k["nb"]), _d(k["nv"])
        if k_bc is not None:
            k["kb"], k["kv"] = i32(k_bc), np.ascontiguousarray(k_val if k_val is not None else np.zeros(npatch), np.float64)
            self.case.k_bc, self.case.k_value = _i(k["kb"]), _d(k["kv"])
        if eps_bc is not None:
            k["eb"], k["ev"] = i32(eps_bc), np.ascontiguousarray(eps_val if eps_val is not None else np.zeros(npatch), np.float64)
            self.case.eps_bc, self.case.eps_value = _i(k["eb"]), _d(k["ev"])
        self._create(device, transport)

    def _create(self, device, transport):
        L = lib()
        self._h = C.c_void_p()
        self._tr = transport
        _check(L.fy_ldu_solver_create(C.byref(self.pm), C.byref(self.case), C.byref(transport) if transport is not None else None, int(device), C.byref(self._h)))
        self._cpl = C.c_void_p(L.fy_ldu_solver_coupling(self._h))
        self.n_cells = int(self.pm.n_cells)
        self._batch_n = [0]

    @classmethod
    def from_foam_case(cls, fc, device=0, transport=None):
        """the solver of a case directory opened with GeneralFoamCase (mesh, controls and patch conditions as read), started from its start-time fields"""
        self = cls.__new__(cls)
        cls._bind()
        self._keep = fc                    # (the structs point into the case object)
        self.pm, self.case = fc.pm, fc.ldu_case
        self._create(device, transport)
        U, p = fc.initial_fields()
        self.set("U", U); self.set("p", p)
        if fc.ldu_case.turbulence_model != 0:
            self.set("nut", fc.initial_nut())
        if fc.ldu_case.turbulence_model in (TURBULENCE_KEQN, TURBULENCE_KEPSILON):
            self.set("k", fc.initial_k())
        if fc.ldu_case.turbulence_model == TURBULENCE_KEPSILON:
            self.set("epsilon", fc.initial_epsilon())
        return self

    def _size(self, name):
        cnt = C.c_int64(0)
        _check(lib().fy_ldu_solver_field_count(self._h, name.encode(), C.byref(cnt)))
        return cnt.value

    def get(self, name):
        out = np.zeros(self._size(name))
        _check(lib().fy_ldu_solver_read_field_host(self._h, name.encode(), _d(out)))
        return out.reshape(-1, 3) if name in ("C", "Cf", "Sf", "kvec", "sep") else out

    geometry = get

    def set(self, name, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        assert arr.size == self._size(name)
        _check(lib().fy_ldu_solver_write_field_host(self._h, name.encode(), _d(arr)))

    def hold_sources(self, on=True):
        L = lib()
        L.fy_ldu_solver_hold_sources.argtypes = [C.c_void_p, C.c_int]
        _check(L.fy_ldu_solver_hold_sources(self._h, 1 if on else 0))

    def mg_levels(self):
        """[(cells, slots)] of the multigrid hierarchy, finest first ([] with the diagonal preconditioner)"""
        L = lib()
        L.fy_ldu_solver_mg_levels.argtypes = [C.c_void_p, C.c_int, _ip, _ip, C.POINTER(C.c_int)]
        cells = np.zeros(32, np.int32); slots = np.zeros(32, np.int32); n = C.c_int(0)
        _check(L.fy_ldu_solver_mg_levels(self._h, 32, _i(cells), _i(slots), C.byref(n)))
        return [(int(cells[q]), int(slots[q])) for q in range(n.value)]

    def apply(self, op, x):
        """"p_matrix": A x; "p_precondition": M^-1 x -- the pressure equation's operators as the last step left them"""
        L = lib()
        L.fy_ldu_solver_apply.argtypes = [C.c_void_p, C.c_char_p, _dp, _dp]
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(self.n_cells)
        _check(L.fy_ldu_solver_apply(self._h, op.encode(), _d(x), _d(out)))
        return out

    def set_particles(self, records):
        L = lib()
        _check(L.fy_set_num_batches(self._cpl, 1))
        rec = np.zeros((0, 10)) if records is None else np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 10)
        _check(L.fy_set_particles_host(self._cpl, 0, _d(rec) if rec.size else None, rec.shape[0]))
        self._batch_n = [rec.shape[0]]

    def forces(self):
        out = np.zeros((self._batch_n[0], 6))
        _check(lib().fy_get_forces_host(self._cpl, 0, _d(out)))
        return out

    def found(self):
        out = np.zeros(self._batch_n[0], dtype=np.int32)
        _check(lib().fy_get_found_host(self._cpl, 0, _i(out)))
        return out

    def step(self, source=None):
        """source: an external momentum source [nc][3] added to what the coupling leaves (kept until set again)"""
        if source is not None:
            self.set("uSource", source)
        _check(lib().fy_ldu_solver_step(self._h))

    def stats(self):
        s = StepStats()
        _check(lib().fy_ldu_solver_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in StepStats._fields_}

    def close(self):
        if self._h:
            lib().fy_ldu_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:            # noqa: BLE001
            pass
