/* foamyade_hip.h -- C-ABI of libfoamyade_hip.so: the MI355X-native drop-in for the hot path of
 * dpkn31/Yade-OpenFOAM-coupling (class Foam::FoamYade + the icoFoamYade / pimpleFoamYade loop bodies).
 *
 * Plain C: pointers and sizes only, no C++/torch types, int status codes, never throws.
 * Every entry point names the reference interface it replaces (paths relative to /root/reference).
 *
 * Two objects:
 *   fy_ctx     <->  Foam::FoamYade            FoamYade/FoamYade.H:57-161   (coupling engine, particle half)
 *   fy_solver  <->  the solver executables    icoFoamYade/icoFoamYade.C:38-154, pimpleFoamYade/pimpleFoamYade.C:40-119
 *                                             (+UcEqn.H, pEqn.H, CourantNo.H, continuityErrs.H, createFields.H)
 *
 * Memory layout at the boundary is OpenFOAM's: scalar = double, label = int, vector = 3 contiguous doubles
 * (xyzxyz...), tensor = 9 contiguous doubles row-major xx xy xz yx yy yz zx zy zz (FoamYade.C:450,472-474).
 * Particle records are the wire layout: 10 doubles [x y z vx vy vz wx wy wz radius] (FoamYade.C:190-219);
 * force records 6 doubles [Fx Fy Fz Tx Ty Tz] (FoamYade.C:492-498).
 */
#ifndef FOAMYADE_HIP_H
#define FOAMYADE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FY_ABI_VERSION 13

/* ---- status codes ------------------------------------------------------------------------------------ */
enum {
    FY_OK = 0,
    FY_ERR_INVALID = 1,     /* bad argument / call order */
    FY_ERR_NO_DEVICE = 2,   /* no gfx950 device visible: the product NEVER falls back to a CPU path */
    FY_ERR_HIP = 3,         /* a HIP runtime call failed; see fy_last_error() */
    FY_ERR_TRANSPORT = 4,   /* a transport callback returned non-zero */
    FY_ERR_UNSUPPORTED = 5,
    FY_ERR_NOT_CONVERGED = 6
};
const char* fy_last_error(void);      /* thread-local text of the last failure */
int fy_abi_version(void);
int fy_device_count(void);            /* number of visible HIP devices (0 on a CPU-only host) */

/* ---- where a caller-owned array lives ------------------------------------------------------------------ */
enum { FY_MEM_HOST = 0, FY_MEM_DEVICE = 1 };

/* ---- mesh: what FoamYade reads from fvMesh (mesh.C(), mesh.V(), mesh.points(); FoamYade.H:121, FoamYade.C:69,86,304) */
typedef struct fy_mesh_desc {
    int32_t n_cells;
    const double* centres;      /* [n_cells][3]  mesh.C(), HOST memory (read once at create) */
    const double* volumes;      /* [n_cells]     mesh.V(), HOST memory */
    double bbox_min[3];         /* bounding box of mesh.points() (FoamYade.C:82-94) */
    double bbox_max[3];
    /* Uniform hex block in blockMesh order (cell = i + nx*(j + ny*k)).  Required for point-force mode, where it
     * stands in for polyMesh::findCell (FoamYade.C:251), and for fy_solver.  Set nx = 0 for "unstructured". */
    int32_t nx, ny, nz;
    double dx;
    double origin[3];
    /* A GRADED (rectilinear) block in the same cell order: coordinates of the nx + 1, ny + 1, nz + 1 face planes along the axes (HOST memory,
     * copied at create); all NULL = the uniform block above.  With them findCell is a search along each axis, the k-d tree carries explicit
     * node coordinates (the centres as given) and dx is unused. */
    const double *xf, *yf, *zf;
} fy_mesh_desc;

/* ---- the twelve constructor arguments of Foam::FoamYade (FoamYade.H:106-117), minus mesh and the bool ---- */
typedef struct fy_field_ptrs {
    int32_t location;           /* FY_MEM_HOST: staged over PCIe each step; FY_MEM_DEVICE: used in place */
    const double* U;            /* [n][3] */
    const double* gradP;        /* [n][3] */
    const double* vGrad;        /* [n][9] */
    const double* divT;         /* [n][3] */
    const double* ddtU;         /* [n][3]  (only consumer is addedMassForce, FoamYade.C:392-413: see fy_set_force_models) */
    double g[3];                /* uniformDimensionedVectorField g */
    double* uSourceDrag;        /* [n]     read-write */
    double* alpha;              /* [n]     read-write */
    double* uSource;            /* [n][3]  read-write */
    double* uParticle;          /* [n][3]  read-write */
} fy_field_ptrs;

/* ---- transport: the MPI calls FoamYade.C makes, as callbacks, so that this library does not link an MPI -- */
/* datatype / op selectors for the callbacks */
enum { FY_T_INT = 0, FY_T_DOUBLE = 1 };
enum { FY_OP_MAX = 0, FY_OP_SUM = 1 };
/* All callbacks return 0 on success.  "world" = MPI_COMM_WORLD (Yade ranks first, README.md:29, FoamYade.C:28-43),
 * "local" = PstreamGlobals::MPI_COMM_FOAM (FoamYade.C:21-22). */
struct fy_wire_pieces;
typedef struct fy_transport {
    void* user;
    int32_t world_rank, world_size;     /* FoamYade.C:24-25 */
    int32_t local_rank, local_size;     /* FoamYade.C:21-22 */
    int (*send)(void* user, const void* buf, int count, int dtype, int dest, int tag);              /* MPI_Send / Isend+Wait */
    int (*recv)(void* user, void* buf, int count, int dtype, int src, int tag);                     /* MPI_Recv */
    int (*bcast_world)(void* user, void* buf, int count, int dtype, int root);                      /* MPI_Bcast(WORLD) */
    int (*bcast_local)(void* user, void* buf, int count, int dtype, int root);                      /* MPI_Bcast(MPI_COMM_FOAM) */
    int (*allreduce_world)(void* user, const void* in, void* out, int count, int dtype, int op);    /* MPI_Allreduce(WORLD) */
    /* ---- optional zero-copy wire (round 4; every pointer may be NULL: a zero-initialised struct is the round-3 transport) ----
     * A transport that owns staging memory -- the shared-memory arena that wire-helper ranks fill in parallel (include/foamyade_mpi.h:
     * one receiving core copies ~9 GB/s out of the MPI library whatever it posts, so the 800 MB of a 10 M-particle step need SEVERAL
     * receiving processes; measured: tools/native/mpi_recv_rate.cpp) -- hands the library VIEWS of the messages instead of copying them
     * into buffers of the library: the PCIe copies then start from / end in that memory.  Parallel-Yade protocol only. */
    /* called once from fy_create, before the bounding box goes out (FoamYade.C:77-111): the uniform block this rank computes on */
    int (*describe_block)(void* user, const double origin[3], double dx, const int32_t n[3]);
    /* recv(..., src, FY_TAG 1002) without the copy: *buf = the message where the transport keeps it (valid until the step's last send_commit).
     * The message may be the concatenation of several PIECES, each received by another helper for its own cell layers [k0, k1) across one axis: a record is
     * located only within its piece's layers (a particle near a cut arrives in both pieces and is found in exactly one). */
    int (*recv_view)(void* user, const void** buf, int count, int dtype, int src, int tag, struct fy_wire_pieces* pieces);
    /* the same in two halves (both or neither): recv_view_layout returns at once -- where the message WILL lie and how it is cut -- and
     * recv_view_next blocks until one more piece of this step's messages has landed and says which (src = the worker, piece = index into that
     * message's fy_wire_pieces; a message without cuts is its own piece 0).  The library then starts each piece's PCIe copy when it lands, in
     * whatever order the workers deliver, instead of each message's when it is complete. */
    int (*recv_view_layout)(void* user, const void** buf, int count, int dtype, int src, int tag, struct fy_wire_pieces* pieces);
    int (*recv_view_next)(void* user, int* src, int* piece);
    /* where the library should write a message it is about to send (count elements), then the hand-over: replaces send(...) */
    int (*send_reserve)(void* user, void** buf, int count, int dtype, int dest, int tag);
    int (*send_commit)(void* user, const void* buf, int count, int dtype, int dest, int tag);
    /* the memory the views point into, for page-locking (hipHostRegister) while `generation` stays the same; bytes = 0: nothing to lock */
    int (*view_region)(void* user, void** base, size_t* bytes, uint64_t* generation);
} fy_transport;
#define FY_WIRE_MAX_PIECES 8
typedef struct fy_wire_pieces {
    int32_t n;                                  /* 0: one message, no cuts */
    int32_t axis;                               /* the block is cut across this axis (0 x, 1 y, 2 z) */
    int32_t start[FY_WIRE_MAX_PIECES];          /* first record of piece q within the message */
    int32_t k0[FY_WIRE_MAX_PIECES], k1[FY_WIRE_MAX_PIECES];      /* cell layers [k0, k1) along `axis` piece q's records may be located in */
} fy_wire_pieces;

typedef struct fy_ctx fy_ctx;

/* Foam::FoamYade::FoamYade(...) FoamYade.H:106-122 -> getRankSize() FoamYade.C:18-53:
 * rank discovery, k-d tree build over mesh.C() (meshTree.C:9-37), bbox send to every Yade rank in parallel-Yade
 * mode (FoamYade.C:77-111), initFields (FoamYade.C:56-73).
 * transport == NULL selects "direct" mode: no wire traffic; particles are handed over with
 * fy_set_particles_* and forces read back with fy_get_forces_* (used by fy_solver, the tests and bench.py). */
int fy_create(const fy_mesh_desc* mesh, const fy_field_ptrs* fields, int gaussian_interp,
              const fy_transport* transport, int device_ordinal, fy_ctx** out);
/* FoamYade::setScalarProperties FoamYade.C:9-11 */
int fy_set_scalar_properties(fy_ctx*, double rhoP, double rhoF, double nu);
/* The two Gaussian-mode force models FoamYade carries WITHOUT a live call site; both are off by default, which is the shipped
 * behaviour.  Enabling one is what re-enabling it in the reference would do:
 *   FY_FORCE_GAUSSIAN_TORQUE  calcHydroTorque's Gaussian branch, FoamYade.C:465-479 (its call is commented out at FoamYade.C:618);
 *                             reads fields.vGrad and the angular velocity of the records; fills force[3..5]
 *   FY_FORCE_ADDED_MASS       addedMassForce, FoamYade.C:392-413 (never called): reads fields.ddtU and the dt passed to
 *                             fy_set_particle_action; adds to force[0..2] and back-scatters into uSource
 * Returns FY_ERR_INVALID in point-force mode or when the field a model needs was not supplied to fy_create. */
#define FY_FORCE_ADDED_MASS 1u
#define FY_FORCE_GAUSSIAN_TORQUE 2u
int fy_set_force_models(fy_ctx*, unsigned flags);
/* FoamYade::fibreCpl (public flag, FoamYade.H:102; off by default and never set by the two solvers).  When on, Yade sends 15 doubles per
 * particle instead of 10 (FoamYade.C:131-136 parallel, :161-165 serial) and the position is read with that stride (:194-198) while
 * velocity, spin and radius are still read from buf[np*10+3..9] of the same buffer (:211-221) -- mirrored literally.  After this call
 * fy_set_particles_host/_device take [n][15] records and the transport receives 15 n doubles.  Not available with z-slabs. */
int fy_set_fibre_coupling(fy_ctx*, int on);
/* FoamYade::setParticleAction FoamYade.C:605-632 (blocking).  On return alpha, uParticle, uSourceDrag, uSource hold
 * this step's values and (with a transport) found flags / forces / dt have been exchanged with Yade. */
int fy_set_particle_action(fy_ctx*, double dt);
/* FoamYade::setSourceZero FoamYade.C:556-566 */
int fy_set_source_zero(fy_ctx*);
/* FoamYade::finalizeRun (FoamYade.C:595-599; declared FoamYade.H:159, never called by the two solvers): the value Yade's rank 0 broadcasts over
   the world communicator; 10 means "finalize MPI now" -- which the caller does, the library does not own the MPI session */
int fy_finalize_run(fy_ctx*, int* value_out);
/* FoamYade::~FoamYade FoamYade.H:160 */
int fy_destroy(fy_ctx*);

/* ---- direct mode (no Yade peer): one call per "Yade proc" batch, batches are processed in index order exactly
 *      like the loop over inCommProcs (FoamYade.C:612-628) ------------------------------------------------- */
int fy_set_num_batches(fy_ctx*, int nbatch);
int fy_set_particles_host(fy_ctx*, int batch, const double* records, int64_t n);     /* copies (H2D); `records` may be reused or freed on return */
int fy_set_particles_device(fy_ctx*, int batch, const double* d_records, int64_t n); /* borrows the device pointer */
int fy_get_forces_host(fy_ctx*, int batch, double* out_forces /* [n][6] */);
/* z-slab mode (fy_solver_create_slab): hand the particles of batch 0 whose containing cell now lies in a neighbour's planes to that
 * neighbour over the slab communicator (ncclSend / ncclRecv in one group under RCCL -- the path the halos take), compact the rest.  A
 * particle moves at most one slab per call.  d_tags (device, optional): one int64 per record, e.g. the DEM's particle id; it travels with
 * the record and is rewritten in the new local order (tag_capacity entries available).  Afterwards the records are library-owned
 * (fy_get_particles_host reads them back).  Collective.  Single domain: a no-op. */
int fy_migrate_particles(fy_ctx*, int64_t* d_tags, int64_t tag_capacity, int64_t* n_local_out);
int fy_get_particles_host(fy_ctx*, int batch, double* records_out /* [n][10] or NULL */, int64_t* n_out);
int fy_get_found_host(fy_ctx*, int batch, int32_t* out_found /* [n], 1 / -1 as foundBuff FoamYade.C:141,222 */);
const double* fy_forces_device(fy_ctx*, int batch);
/* per-particle stencil of the last step, for parity tests: k[n], ids[n][16] (-1 padded, ascending d2, ids[0] =
 * inCell FoamYade.C:208), weights[n][16] (FoamYade.C:293-316), chain_len[n] (>12 => reference behaviour is undefined,
 * meshTree.H:66-68; we append and never evict) */
int fy_get_stencils_host(fy_ctx*, int batch, int32_t* k, int32_t* ids, double* weights, int32_t* chain_len);
/* k-d tree in preorder (node, left subtree, right subtree): cell ids, for parity with meshTree.C:19-37 */
int fy_get_tree_preorder(fy_ctx*, int32_t* out_ids /* [n_cells] */);
/* meshTree::nearestCell (meshTree.C:66-135) for n points, pos[n][3] on the host: out[q] = id of the nearest cell centre (among equidistant
   ones the first the reference's depth-first descent meets).  FoamYade itself never calls it (its Gaussian locate is the range search, its
   point locate mesh.findCell); exposed because it is the tree's public API and the findCell stand-in on general meshes. */
int fy_nearest_cells_host(fy_ctx*, const double* pos, int64_t n, int32_t* out);
/* read / write any of the ctx's cell fields by name ("alpha","uParticle","uSourceDrag","uSource","U","gradP","vGrad","divT") */
int fy_read_field_host(fy_ctx*, const char* name, double* out);
int fy_write_field_host(fy_ctx*, const char* name, const double* in);
double fy_yade_dt(fy_ctx*);           /* yadeDT received in exchangeDT (FoamYade.C:537-553) */
double fy_interp_range(fy_ctx*);      /* interpRange = 4*cbrt(V[0]) (FoamYade.C:69) */
/* Gaussian locate on a uniform block: how many particles of the last fy_set_particle_action were NOT placed through the per-(cell,
 * octant) candidate lists and took the plain tree walk instead (within 8e-6 dx of a cell face, outside the block, list overflow);
 * -1 when the lists are not in use (explicit tree, FOAMYADE_NO_LOCATE_LISTS, no memory).  Same results either way. */
long long fy_locate_walk_count(fy_ctx*);
/* Gaussian locate on an explicit tree (graded block, general mesh): entries per lane of the LDS stack the last step's walk ran with -- chosen from the kernel's own histogram of
 * the depths the walks need; a walk that needs more takes a second launch with the full depth --, 0 while the walk still runs with the full depth (first steps, small clouds),
 * -1 when the tree is not explicit.  Same results either way. */
int fy_locate_stack_depth(fy_ctx*);

/* per-phase device timings of the last fy_set_particle_action, milliseconds (HIP events on the ctx stream) */
typedef struct fy_particle_timings {
    double h2d, bin, locate_deposit, finalize, force, d2h, total;
    int64_t n_particles, n_pairs;     /* n_pairs = sum of k */
    /* drop-in path (a transport is attached; all 0 otherwise).  h2d / d2h above then hold the whole receive / send phases as the
       compute stream saw them; the four below split them: time on the PCIe copy stream and host time inside the transport calls */
    double copy_in, copy_out;         /* first H2D start .. last H2D end, first D2H start .. last D2H end (copy stream, ms) */
    double wire_recv, wire_send;      /* host wall time in the transport's record / result calls (ms) */
    int64_t bytes_in, bytes_out;      /* bytes that crossed PCIe in each direction (records; forces + found flags) */
    /* Gaussian mode: locate_deposit and force above are the two big kernels ALONE (k_locate_deposit; k_force_gaussian), bracketed by HIP
       events on their stream; finalize = cell-record pack + tile reduce + k_finalize_cells (+ slab halos); fold = tile reduce +
       k_fold_sources (+ slab halos) */
    double fold;
} fy_particle_timings;
int fy_get_particle_timings(fy_ctx*, fy_particle_timings* out);
int fy_enable_timing(fy_ctx*, int on);

/* ======================================================================================================== */
/* fy_solver: the time-loop bodies of icoFoamYade (PISO, point force) and pimpleFoamYade (PIMPLE, 4-way).  */
/* ======================================================================================================== */
enum { FY_SOLVER_ICO = 0, FY_SOLVER_PIMPLE = 1 };
/* boundary patches of the block, in this order */
enum { FY_XMIN = 0, FY_XMAX = 1, FY_YMIN = 2, FY_YMAX = 3, FY_ZMIN = 4, FY_ZMAX = 5 };
enum { FY_BC_U_FIXED_VALUE = 0, FY_BC_U_ZERO_GRADIENT = 1,
       FY_BC_U_SLIP = 2 /* symmetryPlane / symmetry / slip on the block's (planar) side: normal component 0, tangential components zeroGradient */ };
enum { FY_BC_P_ZERO_GRADIENT = 0, FY_BC_P_FIXED_VALUE = 1, FY_BC_P_FIXED_FLUX = 2 };
enum { FY_PSOLVER_PCG_JACOBI = 0, FY_PSOLVER_PCG_MG = 1 };

#define FY_CONVECTION_LINEAR 0
#define FY_CONVECTION_UPWIND 1
#define FY_CONVECTION_LINEAR_UPWIND 2
/* NVD / TVD limited schemes [OF-6 LimitedScheme, NVDTVD]: face value = w U_owner + (1 - w) U_neighbour, w = limiter(r) w_linear + (1 - limiter(r)) pos0(flux),
   r = 2 (d . grad(magSqr(U))_upwind) / (magSqr(U)_N - magSqr(U)_P) - 1; one limiter for the three components, weights implicit */
#define FY_CONVECTION_LIMITED_LINEAR 3   /* Gauss limitedLinear k: max(min(2 r / k, 1), 0), k = convection_limiter_k in [0, 1] */
#define FY_CONVECTION_VAN_LEER 4         /* (r + |r|) / (1 + |r|) */
#define FY_CONVECTION_MUSCL 5            /* max(min(2 r, r / 2 + 1 / 2, 2), 0) */
#define FY_CONVECTION_MINMOD 6           /* max(min(r, 1), 0) */
#define FY_CONVECTION_SUPERBEE 7         /* max(min(2 r, 1), min(r, 2), 0) */
#define FY_CONVECTION_QUICK 8            /* max(min(2 r, (3 + r) / 4, 2), 0) */
/* continuousPhaseTurbulence (pimpleFoamYade/createFields.H, DPMTurbulenceModels.C:67-77; icoFoamYade has no turbulence model) */
#define FY_TURBULENCE_LAMINAR 0        /* simulationType laminar / laminarModel Stokes (DPMTurbulenceModels.C:67-68) */
#define FY_TURBULENCE_SMAGORINSKY 1    /* simulationType LES, LESModel Smagorinsky (DPMTurbulenceModels.C:73-74), delta cubeRootVol */
#define FY_TURBULENCE_KEQN 2           /* simulationType LES, LESModel kEqn (DPMTurbulenceModels.C:76-77), delta cubeRootVol */
#define FY_TURBULENCE_KEPSILON 3       /* simulationType RAS, RASModel kEpsilon (DPMTurbulenceModels.C:70-71); no wall functions */
#define FY_BC_NUT_ZERO_GRADIENT 0
#define FY_BC_NUT_FIXED_VALUE 1
#define FY_BC_NUT_CALCULATED 3          /* nut_bc: `calculated` patch = the model's expression on the boundary values of k (and epsilon); kEqn / kEpsilon */
#define FY_BC_WALL_FUNCTION 2           /* nut_bc: nutkWallFunction (needs a model with k); eps_bc: epsilonWallFunction; k takes zeroGradient (kqRWallFunction) */
typedef struct fy_case_desc {
    int32_t solver;                 /* FY_SOLVER_ICO | FY_SOLVER_PIMPLE */
    int32_t nx, ny, nz;
    double dx;
    double origin[3];
    double dt;                      /* controlDict deltaT */
    double nu;                      /* transportProperties nu (icoFoamYade/createFields.H:29-33) */
    double rho_fluid, rho_particle; /* fluidDensity|rho.<phase>, partDensity */
    double g[3];                    /* constant/g */
    int32_t u_bc[6];  double u_value[6][3];
    int32_t p_bc[6];  double p_value[6];
    /* fvSolution PISO / PIMPLE dictionaries (icoFoamYade/createFields.H:166-169, pimpleFoamYade/createFields.H:83-86) */
    int32_t n_outer_correctors;     /* PIMPLE nOuterCorrectors (1 for PISO) */
    int32_t n_correctors;           /* nCorrectors */
    int32_t n_non_orth_correctors;  /* nNonOrthogonalCorrectors */
    int32_t momentum_predictor;
    int32_t p_ref_cell; double p_ref_value;
    /* linear solver controls: p, pFinal, U */
    int32_t p_solver;               /* FY_PSOLVER_* */
    double p_tol, p_rel_tol, p_final_tol, p_final_rel_tol; int32_t p_max_iter;
    double u_tol, u_rel_tol; int32_t u_max_iter;
    int32_t convection_scheme;             /* divSchemes for div(phi,U): FY_CONVECTION_LINEAR (Gauss linear, default) | _UPWIND (Gauss upwind) | _LINEAR_UPWIND (Gauss linearUpwind, unlimited) | the limited schemes _LIMITED_LINEAR .. _QUICK */
    /* controlDict adjustTimeStep / maxCo / maxDeltaT: readTimeControls.H + CourantNo.H + setDeltaT.H at the top of the time loop
       (pimpleFoamYade.C:62-64); dt above is then the initial deltaT and fy_step_stats.delta_t reports what each step used */
    int32_t adjust_time_step; double max_co, max_delta_t;
    /* fvSolution relaxationFactors: equations { Uc; UcFinal } for UcEqn.relax() (UcEqn.H:12), fields { p; pFinal } for p.relax()
       (pEqn.H:41).  <= 0: no entry (the call does nothing); the *_final values apply on the last outer corrector, falling back to the
       plain ones when absent.  fy_case_defaults: u_relax = 1 (the DPMFoam tutorials' `equations { ".*" 1; }`), no field relaxation */
    double u_relax, u_relax_final, p_relax, p_relax_final;
    /* constant/turbulenceProperties (pimpleFoamYade only): FY_TURBULENCE_*.  Smagorinsky [OF-6 Smagorinsky.C]: k from
       a = Ce/delta, b = 2/3 tr(D), c = 2 Ck delta (dev(D) && D), k = ((-b + sqrt(b^2 + 4ac)) / 2a)^2, nut = Ck delta sqrt(k), evaluated by
       continuousPhaseTurbulence->correct() after the last corrector of the final outer iteration (pimpleFoamYade.C:101-104);
       delta = les_delta_coeff * cbrt(V) (LESdelta cubeRootVol); nuEff = nu + nut enters divDevRhoReff (UcEqn.H:7) as
       -fvm::laplacian(alpha nuEff, U) - fvc::div(alpha nuEff dev2(T(grad U))).  nut_initial / nut_bc / nut_value: the 0/nut file */
    int32_t turbulence_model;
    double les_ck, les_ce, les_delta_coeff;      /* fy_case_defaults: 0.094, 1.048, 1 */
    int32_t nut_bc[6]; double nut_value[6];      /* FY_BC_NUT_* per side */
    double nut_initial;                          /* uniform internalField of 0/nut (fy_solver_write_field_host("nut") for a non-uniform one) */
    /* kEqn [OF-6 LES/kEqn/kEqn.C]: fvm::ddt(alpha,k) + fvm::div(alphaPhic,k) - fvm::laplacian(alpha (nut + nu), k) == alpha G
       - fvm::SuSp(2/3 alpha div(phic), k) - fvm::Sp(Ce alpha sqrt(k)/delta, k), G = nut (gradU && dev(twoSymm(gradU))); relax; solve; bound(k, kMin);
       nut = Ck sqrt(k) delta.  The 0/k file: k_initial (uniform), k_bc / k_value per side (FY_BC_NUT_* values: zeroGradient | fixedValue);
       divSchemes div(alphaPhic,k): k_convection_scheme FY_CONVECTION_LINEAR | _UPWIND; solvers.k: k_tol / k_rel_tol / k_max_iter;
       relaxationFactors equations k: k_relax (<= 0: none) */
    int32_t k_bc[6]; double k_value[6]; double k_initial;
    int32_t k_convection_scheme;
    double k_tol, k_rel_tol; int32_t k_max_iter;
    double k_relax;
    /* kEpsilon [OF-6 RAS/kEpsilon/kEpsilon.C]: first the dissipation equation, fvm::ddt(alpha,eps) + fvm::div(alphaPhic,eps)
       - fvm::laplacian(alpha (nut/sigmaEps + nu), eps) == C1 alpha G eps/k - fvm::SuSp((2/3 C1 - C3) alpha div(phic), eps) - fvm::Sp(C2 alpha eps/k, eps),
       bound(eps, epsilonMin); then fvm::ddt(alpha,k) + fvm::div(alphaPhic,k) - fvm::laplacian(alpha (nut/sigmak + nu), k) == alpha G
       - fvm::SuSp(2/3 alpha div(phic), k) - fvm::Sp(alpha eps/k, k) with the NEW eps, bound(k, kMin); nut = Cmu k^2/eps.
       k uses the k_* fields above; epsilon its own.  Boundary types zeroGradient | fixedValue | FY_BC_WALL_FUNCTION (below) */
    double ras_cmu, ras_c1, ras_c2, ras_c3, ras_sigmak, ras_sigmaeps;   /* fy_case_defaults: 0.09, 1.44, 1.92, 0, 1, 1.3 */
    int32_t eps_bc[6]; double eps_value[6]; double eps_initial;
    int32_t eps_convection_scheme;
    double eps_tol, eps_rel_tol; int32_t eps_max_iter;
    double eps_relax;
    /* wall functions [OF-6 nutkWallFunction / epsilonWallFunction]: nut_w = nu (y+ kappa / ln(E y+) - 1) above yPlusLam, else 0, with
       y+ = Cmu^1/4 y sqrt(k)/nu; in the wall cells eps = Cmu^3/4 k^3/2/(kappa y) is imposed on the epsilon equation and the production G is
       replaced by (1/W) sum (nut_w + nu) |snGrad U| Cmu^1/4 sqrt(k)/(kappa y).  Cmu is ras_cmu */
    double wf_kappa, wf_E;                       /* fy_case_defaults: 0.41, 9.8 */
    /* A GRADED single block (blockMesh simpleGrading; icoFoamYade/createFields.H:15-162 and pimpleFoamYade/createFields.H:32-261 take any
       fvMesh): cell sizes along x, y, z -- nx, ny, nz doubles each, copied at fy_solver_create; all three NULL (fy_case_defaults) = uniform
       cubes of edge dx.  The block then starts at `origin`, dx is ignored.  Everything the uniform block carries except z-slabs (a graded
       block runs on one domain) */
    const double *hx, *hy, *hz;
    double convection_limiter_k;            /* FY_CONVECTION_LIMITED_LINEAR: the k of `Gauss limitedLinear k` (fy_case_defaults: 1) */
} fy_case_desc;

typedef struct fy_solver fy_solver;

typedef struct fy_step_stats {
    double courant_mean, courant_max;                   /* CourantNo.H:32-49 */
    double cont_err_sum_local, cont_err_global, cont_err_cumulative;   /* continuityErrs.H:32-46 */
    int32_t p_iters_total, p_solves, u_iters_total;
    double p_initial_residual, p_final_residual;
    double ms_particle, ms_momentum, ms_pressure, ms_other, ms_total;
    double delta_t;                                     /* the time step this pass used (setDeltaT.H when adjust_time_step) */
} fy_step_stats;

void fy_case_defaults(fy_case_desc* c, int solver);
int fy_solver_create(const fy_case_desc* c, const fy_transport* transport, int device_ordinal, fy_solver** out);
fy_ctx* fy_solver_coupling(fy_solver*);                 /* the yadeCoupling object (icoFoamYade.C:54, pimpleFoamYade.C:54) */
int fy_solver_step(fy_solver*);                         /* one pass of the while(runTime.loop()) body */
int fy_solver_get_stats(fy_solver*, fy_step_stats* out);
/* names: "U" [n][3], "p" [n], "phi_x" [(nx+1)*ny*nz], "phi_y", "phi_z", "nut" [n] (with a turbulence model), "k" [n] (kEqn, kEpsilon), "epsilon" [n] (kEpsilon), plus every fy_ctx field name */
int fy_solver_read_field_host(fy_solver*, const char* name, double* out);
int fy_solver_write_field_host(fy_solver*, const char* name, const double* in);
/* number of doubles fy_solver_read/write_field_host move for `name` on this rank (owned cells / local faces) */
int fy_solver_field_count(fy_solver*, const char* name, int64_t* count);
int fy_solver_destroy(fy_solver*);

/* By default fy_solver_step ends with yadeCoupling.setSourceZero() (icoFoamYade.C:147, pimpleFoamYade.C:109).  The reference calls
 * runTime.write() just BEFORE that (icoFoamYade.C:142, pimpleFoamYade.C:107), i.e. what it writes is the step's alpha / uSource.  With
 * hold = 1 the reset is deferred to the start of the next fy_solver_step (the same sequence), so that a caller can read or write
 * those fields in between. */
int fy_solver_hold_sources(fy_solver*, int hold);

/* ---- OpenFOAM case directories (what the reference's executables get from runTime / mesh / the field constructors, createFields.H
 * of both solvers, and give back with runTime.write()).  Supported subset: ONE axis-aligned blockMesh hex block of uniform cubes whose
 * six sides are covered by `boundary` patches; velocity patches fixedValue (uniform) / noSlip / zeroGradient; pressure patches
 * zeroGradient / fixedValue (uniform) / fixedFluxPressure; internalField uniform or nonuniform; fixed deltaT.  Everything else is
 * refused with FY_ERR_UNSUPPORTED and a message naming file and keyword.
 *   system/blockMeshDict, system/controlDict, system/fvSolution (PISO | PIMPLE, solvers.p / pFinal / U),
 *   system/fvSchemes (must ask for Euler / Gauss linear / linear / corrected|orthogonal, div(phi,U) Gauss linear | upwind | linearUpwind grad(U):
 *   what the solver implements),
 *   constant/transportProperties (nu, partDensity, fluidDensity | continuousPhaseName + rho.<phase>), constant/g,
 *   <startTime>/U | U.<phase>, <startTime>/p */
typedef struct fy_foam_case fy_foam_case;
typedef struct fy_foam_case_info {
    double start_time, end_time, delta_t;
    int32_t write_interval_steps;          /* writeInterval in steps (writeControl timeStep, or runTime / deltaT) */
    int64_t n_cells;
    char u_name[64];                       /* "U" (icoFoamYade) or "U.<continuousPhaseName>" (pimpleFoamYade/createFields.H:35-45) */
    char phase[32];
    char start_name[32];                   /* name of the start time directory as written in controlDict */
    char patch_of_side[6][64];             /* blockMesh patch on the XMIN, XMAX, YMIN, YMAX, ZMIN, ZMAX side */
    int64_t field_cells, field_offset;     /* cells the field files (and fy_foam_case_initial_*, fy_foam_case_write_fields) hold, global number of the first:
                                              n_cells and 0, or one processor directory's slab (fy_foam_case_open_processor) */
} fy_foam_case_info;
int fy_foam_case_open(const char* case_dir, int solver /* FY_SOLVER_ICO | FY_SOLVER_PIMPLE */, fy_foam_case** out);
/* a DECOMPOSED case (decomposePar, simple (1 1 nranks); the reference's -parallel run, README.md:29): mesh, controls and schemes from the case, the
   field files of <case>/processor<rank> -- its z-slab's cells in the global order, the processor patches kept as read and written back --; time
   directories go to the processor directory (reconstructPar's input).  fy_case_desc still describes the WHOLE block (fy_solver_create_slab) */
int fy_foam_case_open_processor(const char* case_dir, int solver, int rank, int nranks, fy_foam_case** out);
int fy_foam_case_desc(const fy_foam_case*, fy_case_desc* out);                   /* ready for fy_solver_create */
int fy_foam_case_info_get(const fy_foam_case*, fy_foam_case_info* out);
int fy_foam_case_initial_fields(const fy_foam_case*, double* U /* [n][3] or NULL */, double* p /* [n] or NULL */);
/* start-time nut.<phase> of a case with a turbulence model (hand it to fy_solver_write_field_host(s, "nut", ...) when it is not uniform) */
int fy_foam_case_initial_nut(const fy_foam_case*, double* nut /* [n] */);
int fy_foam_case_initial_k(const fy_foam_case*, double* k /* [n] */);          /* start-time k.<phase> of a kEqn / kEpsilon case */
int fy_foam_case_initial_epsilon(const fy_foam_case*, double* eps /* [n] */);  /* start-time epsilon.<phase> of a kEpsilon case */
/* runTime.write(): <case>/<time_name>/{U | U.<phase>, p [, alpha.<phase>]} as ASCII volFields with the case's own patch entries */
int fy_foam_case_write_time(const fy_foam_case*, fy_solver*, const char* time_name);
/* the same from host arrays over the WHOLE block (a slab run gathers its ranks' owned cells first: foamYadeHip_mpi -parallel); alpha / nut / k / epsilon
   may be NULL where the case has no such field */
int fy_foam_case_write_fields(const fy_foam_case*, const char* time_name, const double* U, const double* p, const double* alpha, const double* nut,
                              const double* k, const double* epsilon);
int fy_foam_case_close(fy_foam_case*);

/* ---- kernel-level entry points used by the roofline bench and the operator parity tests ------------------ */
/* y = A x for the symmetric 7-point pressure matrix (diag, ux, uy, uz) currently held by the solver; x,y host arrays */
int fy_solver_apply_p_matrix_host(fy_solver*, const double* x, double* y);
/* solve  A x = rhs  with the pressure solver of the case (PCG + multigrid / Jacobi, pFinal tolerances) and the pressure matrix the last
   step assembled; x holds the start vector on entry.  For known-answer tests of the solver itself (manufactured Poisson problems). */
int fy_solver_solve_p_host(fy_solver*, const double* rhs, double* x, int* iterations);
/* time `reps` launches of the pEqn Laplacian apply (the roofline kernel) with HIP events on the solver stream; returns avg ms */
int fy_solver_time_p_apply(fy_solver*, int reps, double* avg_ms);

/* ======================================================================================================== */
/* Multi-GPU: z-slab decomposition of the block, one slab per rank.                          */
/* ======================================================================================================== */
/* A communicator carries the neighbour exchanges (FV halos 1 plane, particle halos 5 planes, reverse sums), the tiny all-reduces
 * (Krylov scalars, Courant number, continuity errors) and the all-gather of the coarse multigrid level.
 *   fy_comm_create_rccl        one process per GPU over RCCL/xGMI; `id128` = the 128 bytes from fy_rccl_unique_id() on rank 0,
 *                              distributed by the launcher (bench.py uses torch.distributed for that)
 *   fy_comm_create_local_group n "virtual slabs" inside ONE process (one host thread per slab, device-to-device copies): the same
 *                              solver code, used to test the decomposition on a single-GPU box
 * The reference's equivalent is OpenFOAM's own domain decomposition (`-parallel`, README.md:29) [OF-6, not in the reference]. */
typedef struct fy_comm fy_comm;
int fy_rccl_unique_id(void* out128);
int fy_comm_create_rccl(int rank, int size, const void* id128, int device_ordinal, fy_comm** out);
int fy_comm_create_local_group(int n, fy_comm** out /* [n] */);
/* host-staged communicator: the library moves the planes device -> pinned host -> callback -> device and leaves the transport between the
   processes to the caller (MPI, gloo, pipes ...).  One process per slab like RCCL, so the ORDER in which separate processes reach the
   collectives is real -- what the in-process group cannot show -- on machines where RCCL cannot run (one GPU shared by the ranks).
   Callbacks return 0 on success; all buffers are host memory; a missing neighbour's buffers are NULL with count 0. */
typedef struct fy_comm_callbacks {
    void* user;
    /* send n_up doubles to rank+1 and n_down to rank-1; receive m_down from rank-1 and m_up from rank+1 (any of them may be 0) */
    int (*sendrecv)(void* user, const double* send_up, size_t n_up, double* recv_from_down, size_t m_down, const double* send_down, size_t n_down,
                    double* recv_from_up, size_t m_up);
    int (*allreduce)(void* user, double* buf, int n, int is_max);                       /* in place, identical result on every rank */
    int (*allgather)(void* user, const double* send, double* recv, size_t count_per_rank);
} fy_comm_callbacks;
int fy_comm_create_host(int rank, int size, const fy_comm_callbacks* cb, fy_comm** out);
/* One process per slab with DIRECT PEER STORES (SURVEY.md 8e "prefer direct peer stores over the fully connected xGMI mesh"; stands where the reference's
 * per-particle collectives stand, FoamYade.C:228, 511-515, and where OpenFOAM's processor patches exchange halos): every rank exports one device window
 * (hipIpcGetMemHandle) and maps the others'; a neighbour exchange is a copy kernel whose stores land in the neighbour's window + a flag, an all-reduce of
 * <= 32 doubles ONE kernel that stores this rank's values into every peer's window and folds in rank order.  The callbacks carry the bootstrap only
 * (allgather of the 64-byte handles, allreduce as the closing barrier); sendrecv may be NULL.  Works for the GPUs of one xGMI node and for N processes that
 * share ONE GPU (where RCCL refuses to run): select with FOAMYADE_COMM=ipc in bench.py.  At most 8 ranks.
 * FOAMYADE_IPC_SLOT_MB (default 8): bytes per neighbour slot (larger groups travel in chunks); FOAMYADE_IPC_TIMEOUT_MS (default 20000): bound of every
 * device-side wait -- a peer that never arrives becomes FY_ERR_TRANSPORT at the next call instead of a hung GPU. */
int fy_comm_create_ipc(int rank, int size, const fy_comm_callbacks* cb, int device_ordinal, fy_comm** out);
/* diagnostic: calls made through this communicator so far: {neighbour exchanges, all-reduces, all-gathers, bytes sent to neighbours} */
int fy_comm_stats(fy_comm*, uint64_t* out4);
/* the same call counts by the solver phase that issued them, as text: one line "<phase> <exchanges> <all-reduces> <all-gathers>" per phase
 * (step_start, particle, momentum, momentum_solve, corrector, p_operators, pcg, vcycle, turbulence ...); the per-step collective budget
 * of the slab solver is asserted on these (tests/test_slabs.py) */
int fy_comm_stats_by_tag(fy_comm*, char* buf, size_t cap);
/* collective known-answer run of every operation the slab solver uses on this communicator (a grouped two-field neighbour exchange, sum and
 * max all-reduce, all-gather); bench.py runs it in throw-away processes before it commits a multi-GPU run to the communicator */
int fy_comm_selftest(fy_comm*, int device_ordinal);
int fy_comm_destroy(fy_comm*);
int fy_comm_rank(fy_comm*);
int fy_comm_size(fy_comm*);
/* `c` describes the GLOBAL block; rank r owns z-planes [r*nz/size, (r+1)*nz/size) (nz/size must be even and >= the particle halo).
 * Collective: every rank of the communicator must call it, and later fy_solver_step, together.  Field accessors then address the
 * rank's OWNED cells only.  Each rank is given the particles that lie inside its own slab. */
int fy_solver_create_slab(const fy_case_desc* c, const fy_transport* transport, int device_ordinal, fy_comm* comm, fy_solver** out);
int fy_solver_local_cells(fy_solver*);

/* per-kernel HIP-event clocks on the solver stream, accumulated over steps since the last enable(1):
 * kernel = "mg_smooth_l0" (pEqn Laplacian apply fused with the damped-Jacobi update, fine level), "p_apply_dot" (pEqn Laplacian
 * apply + p.Ap inside PCG), "mom_pass" (fused momentum Jacobi pass) */
int fy_solver_enable_kernel_timing(fy_solver*, int on);
int fy_solver_get_kernel_timing(fy_solver*, const char* kernel, double* total_ms, int64_t* launches);
/* z-slabs: how long the solver's stream sat WAITING for slab exchanges, by phase {step start, particle, momentum, corrector}, accumulated since the
 * call that switched the clock on.  With the exchanges overlapped (the default; FOAMYADE_HALO_OVERLAP=0 switches to exchange-then-consume) a wait
 * runs from the end of the interior planes' sweep to the arrival of the ghost planes; in the serial schedule it is the exchange itself.  The clock
 * costs an event pair per exchange (5 - 10 us of idle stream each): off by default.  (The particle phase's exchanges run inside the coupling object
 * and are not sampled here.) */
int fy_solver_enable_exchange_timing(fy_solver*, int on);
int fy_solver_get_exchange_wait(fy_solver*, double ms[4], int64_t waits[4]);


/* ------------------------------------------------------------------------------------------------------------------------------------
 * icoFoamYade and pimpleFoamYade on a GENERAL polyhedral mesh (round 4).  The reference's solvers run on whatever createMesh.H
 * hands them (icoFoamYade/icoFoamYade.C:42, pimpleFoamYade/pimpleFoamYade.C:47) and carry non-orthogonal corrector loops (icoFoamYade.C:114-131,
 * pEqn.H:24-47); fy_solver above is the structured block.  fy_ldu_solver takes the mesh in OpenFOAM's own addressing -- constant/polyMesh: points,
 * faces, owner, neighbour, boundary -- builds OpenFOAM's geometry from it (face triangle / cell pyramid decomposition, linear weights,
 * nonOrthDeltaCoeffs, nonOrthCorrectionVectors, fvc::reconstruct's tensors [OF-6]) and runs the loop bodies with owner / neighbour (LDU) addressing:
 * Euler ddt, Gauss linear | upwind | linearUpwind | limited (NVD / TVD) div, Gauss linear grad, Gauss linear CORRECTED laplacian (the explicit non-orthogonal part is what the
 * correctNonOrthogonal loop iterates on), PCG in its single-reduction form with the diagonal or an agglomeration-multigrid preconditioner (p_solver),
 * Jacobi sweeps for U.  Patches: fixedValue / zeroGradient / symmetry (slip) for U (noSlip = fixedValue 0), translational cyclic pairs; zeroGradient / fixedValue for p, fixedFluxPressure with
 * pimpleFoamYade.  pimpleFoamYade (fy_ldu_case.solver): Gaussian 4-way coupling, the void-fraction-weighted UcEqn / pEqn, gravity, PIMPLE outer correctors,
 * relaxation, adjustable time step, laminar Stokes stress, LES Smagorinsky / kEqn or RAS kEpsilon (no wall functions).  The coupling object (fy_ldu_solver_coupling) works on the mesh's own
 * cell centres and volumes: explicit k-d tree, and for the point-force locate (mesh.findCell, FoamYade.C:251) the nearest centre followed by a walk
 * across the faces the point lies outside of. */
typedef struct fy_poly_mesh {
    int32_t n_points;
    const double* points;            /* [n_points][3] */
    int32_t n_faces, n_internal_faces;
    const int32_t* face_offsets;     /* [n_faces + 1] into face_points */
    const int32_t* face_points;      /* a face's points turn counter-clockwise seen from outside its owner */
    const int32_t* owner;            /* [n_faces] */
    const int32_t* neighbour;        /* [n_internal_faces]: internal faces first, owner < neighbour */
    int32_t n_cells;
    int32_t n_patches;
    const int32_t* patch_start;      /* [n_patches] first face of the patch (>= n_internal_faces) */
    const int32_t* patch_size;
    const int32_t* patch_neighbour;  /* NULL, or per patch: the index of its cyclic partner (constant/polyMesh/boundary: type cyclic; neighbourPatch), -1 for an ordinary
                                        patch.  Translational cyclics whose faces match one to one, in order [OF-6 cyclicPolyPatch]: each pair becomes one more internal
                                        face of the solver, numbered after the mesh's own (fy_ldu_solver_read_field_host "orig_face" maps the solver's faces to the
                                        caller's); the patches themselves are left without faces, their entries in the per-patch arrays unused */
} fy_poly_mesh;
typedef struct fy_ldu_case {
    double dt, nu, rho_fluid, rho_particle;
    int32_t n_correctors, n_non_orth_correctors, momentum_predictor, p_ref_cell;
    double p_ref_value;
    double p_tol, p_rel_tol, p_final_tol, p_final_rel_tol; int32_t p_max_iter;
    double u_tol, u_rel_tol; int32_t u_max_iter;
    int32_t p_solver;                /* FY_PSOLVER_PCG_JACOBI: PCG.C with the diagonal preconditioner | FY_PSOLVER_PCG_MG: preconditioned by an agglomeration
                                        multigrid V-cycle (fvSolution: solver / preconditioner GAMG) built from the face areas like faceAreaPair */
    const int32_t* u_bc;             /* per patch: FY_BC_U_FIXED_VALUE | FY_BC_U_ZERO_GRADIENT | FY_BC_U_SLIP (symmetryPlane / symmetry / slip: each face with its own normal) */
    const double* u_value;           /* [n_patches][3] */
    const int32_t* p_bc;             /* per patch: FY_BC_P_ZERO_GRADIENT | FY_BC_P_FIXED_VALUE */
    const double* p_value;           /* [n_patches] */
    /* pimpleFoamYade on the general mesh (solver = FY_SOLVER_PIMPLE; pimpleFoamYade.C:60-114, UcEqn.H, pEqn.H): Gaussian 4-way coupling, the void-fraction-
     * weighted equations, laminar Stokes stress; p patches may then be FY_BC_P_FIXED_FLUX (fixedFluxPressure) too */
    int32_t solver;                  /* FY_SOLVER_ICO (the default) | FY_SOLVER_PIMPLE */
    int32_t n_outer_correctors;
    double g[3];
    double u_relax, u_relax_final, p_relax, p_relax_final;      /* relaxationFactors; <= 0: no entry (relax() does nothing) */
    int32_t adjust_time_step;        /* pimpleFoamYade only (pimpleFoamYade.C:62-64: readTimeControls.H, CourantNo.H, setDeltaT.H) */
    double max_co, max_delta_t;
    /* continuousPhaseTurbulence (pimpleFoamYade only): FY_TURBULENCE_LAMINAR | FY_TURBULENCE_SMAGORINSKY | FY_TURBULENCE_KEQN (LES, delta cubeRootVol) | FY_TURBULENCE_KEPSILON as in fy_case_desc */
    int32_t turbulence_model;
    double les_ck, les_ce, les_delta_coeff, nut_initial;
    const int32_t* nut_bc;           /* per patch: FY_BC_NUT_ZERO_GRADIENT | FY_BC_NUT_FIXED_VALUE (NULL: zeroGradient everywhere) */
    const double* nut_value;         /* [n_patches] */
    int32_t convection_scheme;       /* FY_CONVECTION_LINEAR (default) .. FY_CONVECTION_QUICK for div(phi,U) / div(alphaPhic,Uc), as fy_case_desc.convection_scheme */
    double convection_limiter_k;     /* limitedLinear's coefficient in [0, 1] */
    /* turbulence_model FY_TURBULENCE_KEQN (LES kEqn, DPMTurbulenceModels.C:76-77): the 0/k file (k_initial uniform; per patch FY_BC_NUT_ZERO_GRADIENT | _FIXED_VALUE),
     * div(alphaPhic,k) (FY_CONVECTION_LINEAR | _UPWIND), solvers.k, relaxationFactors equations k (<= 0: none); nut patches may then be FY_BC_NUT_CALCULATED
     * (the file's value until the first correctNut(), Ck sqrt(k_b) delta afterwards) */
    double k_initial;
    const int32_t* k_bc;             /* NULL: zeroGradient everywhere */
    const double* k_value;
    int32_t k_convection_scheme;
    double k_tol, k_rel_tol; int32_t k_max_iter;
    double k_relax;
    /* turbulence_model FY_TURBULENCE_KEPSILON (RAS kEpsilon, DPMTurbulenceModels.C:70-71), WITHOUT wall functions on a general mesh (nutkWallFunction /
     * epsilonWallFunction need nearWallDist: the block solver carries them): the coefficients, the 0/epsilon file and its controls as in fy_case_desc; k as above;
     * a FY_BC_NUT_CALCULATED nut patch then carries Cmu k_b^2 / epsilon_b */
    double ras_cmu, ras_c1, ras_c2, ras_c3, ras_sigmak, ras_sigmaeps;      /* fy_ldu_case_defaults: 0.09, 1.44, 1.92, 0, 1, 1.3 */
    double eps_initial;
    const int32_t* eps_bc;           /* NULL: zeroGradient everywhere */
    const double* eps_value;
    int32_t eps_convection_scheme;
    double eps_tol, eps_rel_tol; int32_t eps_max_iter;
    double eps_relax;
} fy_ldu_case;
typedef struct fy_ldu_solver fy_ldu_solver;
void fy_ldu_case_defaults(fy_ldu_case*);        /* the icoFoam cavity tutorial's controls (as fy_case_defaults); the patch arrays stay NULL */
int fy_ldu_solver_create(const fy_poly_mesh*, const fy_ldu_case*, const fy_transport* transport /* or NULL */, int device_ordinal, fy_ldu_solver** out);
int fy_ldu_solver_step(fy_ldu_solver*);                                     /* one pass of icoFoamYade.C:65-149 */
int fy_ldu_solver_get_stats(fy_ldu_solver*, fy_step_stats* out);
int fy_ldu_solver_hold_sources(fy_ldu_solver*, int on);                    /* as fy_solver_hold_sources: setSourceZero deferred to the next step's start (runTime.write() sees alpha / uSource) */
fy_ctx* fy_ldu_solver_coupling(fy_ldu_solver*);                              /* FoamYade on this mesh (point force); fy_set_particles_* as usual */
/* fields by name, host copies: "U" [nc][3], "p", "phi" [n_faces], "uSource" [nc][3] (added to what the coupling leaves: an external momentum source),
 * "rAU", "HbyA", "phiHbyA", "p_diag", "p_coef" [n_faces], "p_rhs", "vGrad" [nc][9]; geometry: "C" "V" "Cf" "Sf" "magSf" "w" "dcNO" "kvec" */
int fy_ldu_solver_field_count(fy_ldu_solver*, const char* name, int64_t* count);
int fy_ldu_solver_read_field_host(fy_ldu_solver*, const char* name, double* out);
int fy_ldu_solver_write_field_host(fy_ldu_solver*, const char* name, const double* in);
/* the pressure equation's operators as the last step left them, on host vectors of n_cells: "p_matrix" out = A in, "p_precondition" out = M^-1 in
 * (the V-cycle, or the diagonal) -- for tests of the solver's algebra (symmetry, definiteness, the cycle's contraction) */
int fy_ldu_solver_apply(fy_ldu_solver*, const char* op, const double* in, double* out);
/* the multigrid hierarchy of FY_PSOLVER_PCG_MG: cells and matrix slots (neighbours per row, padded) per level, finest first; n_levels = 0 without one */
int fy_ldu_solver_mg_levels(fy_ldu_solver*, int cap, int32_t* cells, int32_t* slots, int* n_levels);
int fy_ldu_solver_destroy(fy_ldu_solver*);

/* An OpenFOAM case directory whose constant/polyMesh is ANY mesh of wall / patch boundaries (ASCII), for icoFoamYade or (laminar, fixed time step)
 * pimpleFoamYade: what createMesh.H + createFields.H read (icoFoamYade.C:42-44, pimpleFoamYade.C:41-43).  The object is the fy_foam_case above with the mesh kept in OpenFOAM's addressing:
 * fy_foam_case_info_get, fy_foam_case_initial_fields, fy_foam_case_write_fields and fy_foam_case_close work on it; fy_foam_case_desc refuses it
 * (there is no block to describe).  fvSchemes must ask for what fy_ldu_solver does: Gauss linear, `corrected` laplacian / snGrad. */
int fy_foam_case_open_general(const char* case_dir, int solver /* FY_SOLVER_ICO | FY_SOLVER_PIMPLE */, fy_foam_case** out);
int fy_foam_case_poly_mesh(const fy_foam_case*, fy_poly_mesh* out);              /* pointers into the case object: valid until fy_foam_case_close */
int fy_foam_case_ldu_desc(const fy_foam_case*, fy_ldu_case* out);                 /* ready for fy_ldu_solver_create (patch arrays point into the case object) */
int fy_foam_case_patch_name(const fy_foam_case*, int patch, char* out, int cap); /* the boundary file's order = fy_poly_mesh's patch numbers */
int fy_foam_case_write_time_ldu(const fy_foam_case*, fy_ldu_solver*, const char* time_name);     /* runTime.write() (icoFoamYade.C:142): <time>/U, p */

#ifdef __cplusplus
}
#endif
#endif /* FOAMYADE_HIP_H */
