// foamYadeHip: the time loop of the reference's executables (icoFoamYade/icoFoamYade.C:36-153, pimpleFoamYade/pimpleFoamYade.C:38-118)
// over an OpenFOAM case directory, on the C-ABI of libfoamyade_hip.so:
//
//     foamYadeHip -solver ico|pimple [-case DIR] [-device N]
//     mpiexec -n Y yade ... : -n N foamYadeHip_mpi -solver ... -case DIR -parallel [-nYade Y] [-hostComm | -ipcComm]
//     mpiexec -n Y yade ... : -n K foamYadeHip_mpi -solver ... -case DIR -wireHelpers [-nYade Y]
//
// read the case (fy_foam_case_open), create the solver, then  while (runTime.loop()) { step; runTime.write(); setSourceZero }.
// Built with -DFY_WITH_MPI (make mpi -> foamYadeHip_mpi) it is launched like the reference, MPMD next to Yade ("mpiexec -n 1 yade ... :
// -n 1 foamYadeHip_mpi ...", README.md:29 of the reference) and talks to Yade's FoamCoupling engine through fy_mpi_transport_create;
// without MPI it runs the fluid alone (no particles), which is what the reference does when Yade sends none.
// -parallel: N solver ranks, the reference's `-parallel` run (README.md:29) without decomposePar -- every rank reads the undecomposed case and
// takes its z-slab (fy_solver_create_slab), one GPU per rank over RCCL (or, with -hostComm / fewer GPUs than ranks, planes staged through the host
// and moved by MPI); each rank receives the particles of its slab from Yade as the reference's ranks do (FoamYade.C:77-155); time directories
// are gathered to the first solver rank and written undecomposed.
// -wireHelpers: K solver-side ranks in front of ONE GPU (include/foamyade_mpi.h): the first computes, the others receive the particles from a parallel
// Yade and send the answers back, in parallel -- one receiving process cannot take a 10 M-particle step off the wire faster than ~9 GB/s.
// This file is host glue only: no arithmetic of the path lives here.
#include <sys/stat.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/foamyade_hip.h"
#ifdef FY_WITH_MPI
#include <mpi.h>

#include "../../include/foamyade_mpi.h"
#endif

static bool g_mpi_up = false;
static int die(const char* what) {
    std::fprintf(stderr, "foamYadeHip: %s: %s\n", what, fy_last_error());
#ifdef FY_WITH_MPI
    // the other ranks of the launch (wire helpers serving, Yade waiting in a receive) would hang on a rank that just left: take the job down
    if (g_mpi_up) { std::fflush(stderr); MPI_Abort(MPI_COMM_WORLD, 1); }
#endif
    return 1;
}

// icoFoamYade on a general mesh: the same loop around fy_ldu_solver (one domain)
static int run_general(fy_foam_case* fc, const fy_transport* trp, int device) {
    fy_poly_mesh pm;
    fy_ldu_case lc;
    fy_foam_case_info info;
    if (fy_foam_case_poly_mesh(fc, &pm) != FY_OK || fy_foam_case_ldu_desc(fc, &lc) != FY_OK || fy_foam_case_info_get(fc, &info) != FY_OK) return die("reading the case");
    std::printf("             %d points, %d faces (%d internal), %d cells, %d patches\n", pm.n_points, pm.n_faces, pm.n_internal_faces, pm.n_cells, pm.n_patches);
    fy_ldu_solver* s = nullptr;
    if (fy_ldu_solver_create(&pm, &lc, trp, device, &s) != FY_OK) return die("fy_ldu_solver_create");
    fy_ldu_solver_hold_sources(s, 1);                   // runTime.write() comes before setSourceZero (icoFoamYade.C:142-147, pimpleFoamYade.C:107-109)
    {
        std::vector<double> U(3 * (size_t)info.n_cells), p((size_t)info.n_cells);
        fy_foam_case_initial_fields(fc, U.data(), p.data());
        if (fy_ldu_solver_write_field_host(s, "p", p.data()) != FY_OK || fy_ldu_solver_write_field_host(s, "U", U.data()) != FY_OK) return die("initial fields");
        if (lc.turbulence_model != FY_TURBULENCE_LAMINAR)                 // nut.<phase> of the start time (eddyViscosity: MUST_READ)
            if (fy_foam_case_initial_nut(fc, p.data()) != FY_OK || fy_ldu_solver_write_field_host(s, "nut", p.data()) != FY_OK) return die("initial nut");
        if (lc.turbulence_model == FY_TURBULENCE_KEQN || lc.turbulence_model == FY_TURBULENCE_KEPSILON)
            if (fy_foam_case_initial_k(fc, p.data()) != FY_OK || fy_ldu_solver_write_field_host(s, "k", p.data()) != FY_OK) return die("initial k");
        if (lc.turbulence_model == FY_TURBULENCE_KEPSILON)
            if (fy_foam_case_initial_epsilon(fc, p.data()) != FY_OK || fy_ldu_solver_write_field_host(s, "epsilon", p.data()) != FY_OK) return die("initial epsilon");
    }
    std::printf("\nStarting time loop\n\n");
    const long n_steps = std::lround((info.end_time - info.start_time) / info.delta_t);
    double t = info.start_time, dt_now = info.delta_t;        // adjustTimeStep: Time::run(), value() < endTime - 0.5 deltaT [OF-6 Time.C]
    for (long k = 1; lc.adjust_time_step ? t < info.end_time - 0.5 * dt_now : k <= n_steps; ++k) {
        if (fy_ldu_solver_step(s) != FY_OK) return die("fy_ldu_solver_step");
        fy_step_stats st;
        fy_ldu_solver_get_stats(s, &st);
        t = lc.adjust_time_step ? t + st.delta_t : info.start_time + (double)k * info.delta_t;
        dt_now = st.delta_t;
        char tname[64];
        std::snprintf(tname, sizeof(tname), "%.12g", t);
        std::printf("Time = %s\n\nCourant Number mean: %g max: %g\n", tname, st.courant_mean, st.courant_max);
        std::printf("pressure: %d solves, %d iterations, initial residual %g, final residual %g\n", st.p_solves, st.p_iters_total, st.p_initial_residual, st.p_final_residual);
        std::printf("time step continuity errors : sum local = %g, global = %g, cumulative = %g\n\n", st.cont_err_sum_local, st.cont_err_global, st.cont_err_cumulative);
        if (info.write_interval_steps > 0 && k % info.write_interval_steps == 0)
            if (fy_foam_case_write_time_ldu(fc, s, tname) != FY_OK) return die("writing the time directory");
    }
    std::printf("End\n");
    fy_ldu_solver_destroy(s);
    fy_foam_case_close(fc);
    return 0;
}

int main(int argc, char** argv) {
    std::string dir = ".", solver_name;
    int device = -1, n_yade_arg = -1;
    bool parallel = false, host_comm = false, ipc_comm = false, wire_helpers = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-case" && i + 1 < argc) dir = argv[++i];
        else if (a == "-solver" && i + 1 < argc) solver_name = argv[++i];
        else if (a == "-device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "-parallel") parallel = true;
        else if (a == "-nYade" && i + 1 < argc) n_yade_arg = std::atoi(argv[++i]);
        else if (a == "-hostComm") host_comm = true;
        else if (a == "-ipcComm") ipc_comm = true;
        else if (a == "-wireHelpers") wire_helpers = true;
        else { std::fprintf(stderr, "usage: foamYadeHip -solver ico|pimple [-case DIR] [-device N] [-parallel [-nYade Y] [-hostComm | -ipcComm]] | [-wireHelpers [-nYade Y]]\n"); return 2; }
    }
    if (solver_name != "ico" && solver_name != "pimple") { std::fprintf(stderr, "foamYadeHip: -solver ico|pimple is required\n"); return 2; }
    const int solver = solver_name == "ico" ? FY_SOLVER_ICO : FY_SOLVER_PIMPLE;

    fy_transport tr{};
    const fy_transport* trp = nullptr;
    fy_comm* comm = nullptr;
    int srank = 0, ssize = 1;                       // this rank among the solver ranks
#ifdef FY_WITH_MPI
    MPI_Init(&argc, &argv);
    g_mpi_up = true;
    int world = 1, wrank = 0;
    MPI_Comm_size(MPI_COMM_WORLD, &world);
    MPI_Comm_rank(MPI_COMM_WORLD, &wrank);
    // Yade ranks come first in MPI_COMM_WORLD (README.md:29, FoamYade.C:28).  Without -parallel ONE fluid rank, the rest is Yade; with it the
    // solver ranks are the world's last world - nYade ranks (-nYade defaults to 1: a serial Yade)
    // -nYade absent: the transport derives it as the reference does (commSzDff = world size - solver communicator size, FoamYade.C:28): every rank
    // running this executable is a solver rank, the Yade ranks are the ones in front of them (README.md:29)
    int n_yade = (parallel || wire_helpers) ? (n_yade_arg >= 0 ? n_yade_arg : -1) : world - 1;
    if (n_yade < 0) {                                  // (the launch's FIRST executable has nobody in front of it: the fluid alone, in parallel)
        int* appnum = nullptr; int flag = 0;
        MPI_Comm_get_attr(MPI_COMM_WORLD, MPI_APPNUM, &appnum, &flag);
        if (flag && appnum && *appnum == 0) n_yade = 0;
    }
    if (n_yade >= world || (n_yade >= 0 && wrank < n_yade)) { std::fprintf(stderr, "foamYadeHip_mpi: the solver ranks must come last in the MPMD launch (world %d, Yade ranks %d)\n", world, n_yade); MPI_Abort(MPI_COMM_WORLD, 2); }
    MPI_Comm solver_comm = MPI_COMM_WORLD;
    if (wire_helpers) {
        int is_helper = 0;
        if (fy_mpi_transport_create_wire_helpers(n_yade, &tr, &is_helper) != FY_OK) return die("fy_mpi_transport_create_wire_helpers");
        if (is_helper) {                                  // receive and answer for my slab of the block until the computing rank is done
            const int rc = fy_mpi_wire_helper_serve(&tr);
            g_mpi_up = false;
            MPI_Finalize();
            return rc == FY_OK ? 0 : 1;
        }
        trp = &tr;
        solver_comm = MPI_COMM_SELF;
    } else if (n_yade != 0) {
        if (fy_mpi_transport_create(n_yade, &tr) != FY_OK) return die("fy_mpi_transport_create");
        trp = tr.send ? &tr : nullptr;                    // (derived count 0: the fluid alone, in parallel -- no coupling transport)
        fy_mpi_local_comm(&tr, &solver_comm);
    }
    MPI_Comm_rank(solver_comm, &srank);
    MPI_Comm_size(solver_comm, &ssize);
    if (ssize > 1) {
        const int ndev = fy_device_count();
        if (ndev < 1) return die("no HIP device");
        // a GPU per rank is a question of THIS node's ranks and GPUs (two nodes of 8 GPUs run 16 ranks over RCCL)
        MPI_Comm node;
        MPI_Comm_split_type(solver_comm, MPI_COMM_TYPE_SHARED, srank, MPI_INFO_NULL, &node);
        int nrank = 0, nsize = 1;
        MPI_Comm_rank(node, &nrank); MPI_Comm_size(node, &nsize);
        MPI_Comm_free(&node);
        if (device < 0) device = nrank % ndev;
        int mine_ok = (!host_comm && !ipc_comm && ndev >= nsize) ? 1 : 0, use_rccl = 0;
        MPI_Allreduce(&mine_ok, &use_rccl, 1, MPI_INT, MPI_MIN, solver_comm);      // one GPU per rank everywhere, or the ranks share and MPI moves the planes
        // -ipcComm: the solver ranks of one node store into each other's device windows (fy_comm_create_ipc), whether or not they share GPUs
        if (ipc_comm) { int one_node = nsize == ssize ? 1 : 0, all = 0; MPI_Allreduce(&one_node, &all, 1, MPI_INT, MPI_MIN, solver_comm); if (!all) return die("-ipcComm: the solver ranks must share one node"); use_rccl = 2; }
        if (fy_mpi_comm_create(&solver_comm, use_rccl, device, &comm) != FY_OK) return die("fy_mpi_comm_create");
        if (srank == 0) std::printf("Decomposition: %d z-slabs, %s\n", ssize, use_rccl == 2 ? "peer stores into hipIpc-mapped device windows" : use_rccl ? "RCCL" : "planes staged through the host, moved by MPI");
    }
#else
    if (parallel) { std::fprintf(stderr, "foamYadeHip: -parallel needs the MPI build (foamYadeHip_mpi)\n"); return 2; }
#endif
    if (device < 0) device = 0;
    const bool master = srank == 0;

    // -parallel on a DECOMPOSED case (<case>/processor0 exists: decomposePar, simple (1 1 N)): every rank reads and writes the field files of its own
    // processor directory, as the reference's ranks do; otherwise every rank reads the undecomposed case and the first solver rank writes it
    fy_foam_case* fc = nullptr;
    bool decomposed = false;
    if (ssize > 1) { struct stat sb; decomposed = stat((dir + "/processor0").c_str(), &sb) == 0 && S_ISDIR(sb.st_mode); }
    int open_rc = decomposed ? fy_foam_case_open_processor(dir.c_str(), solver, srank, ssize, &fc) : fy_foam_case_open(dir.c_str(), solver, &fc);
    if (open_rc == FY_ERR_UNSUPPORTED && ssize == 1) {
        // not the block fy_solver computes on: the solver on the mesh as it is (owner / neighbour addressing, non-orthogonal correctors)
        const std::string why = fy_last_error();
        if (fy_foam_case_open_general(dir.c_str(), solver, &fc) == FY_OK) {
            std::printf("Create mesh: general polyhedral mesh (the block reader said: %s)\n", why.c_str());
            const int rc = run_general(fc, trp, device);
#ifdef FY_WITH_MPI
            if (tr.user) fy_mpi_transport_destroy(&tr);
            g_mpi_up = false;
            MPI_Finalize();
#endif
            return rc;
        }
    }
    if (open_rc != FY_OK) return die("reading the case");
    if (master && ssize > 1) std::printf("Case: %s\n", decomposed ? "decomposed (processor directories)" : "undecomposed (gathered on write)");
    fy_case_desc cd;
    fy_foam_case_info info;
    fy_foam_case_desc(fc, &cd);
    fy_foam_case_info_get(fc, &info);
    if (master)
        std::printf("Create mesh: %d x %d x %d cells of %g m, %s on the six sides x- x+ y- y+ z- z+: %s %s %s %s %s %s\n", cd.nx, cd.ny, cd.nz, cd.dx,
                    "patches", info.patch_of_side[0], info.patch_of_side[1], info.patch_of_side[2], info.patch_of_side[3], info.patch_of_side[4], info.patch_of_side[5]);
    fy_solver* s = nullptr;
    if ((comm ? fy_solver_create_slab(&cd, trp, device, comm, &s) : fy_solver_create(&cd, trp, device, &s)) != FY_OK) return die("fy_solver_create");
    // a slab owns the z-planes [srank nz / ssize, (srank + 1) nz / ssize): a contiguous run of the block's cells, `first` cells in
    // (the field files hold the whole block, or -- processor directories -- exactly this rank's cells)
    const size_t n_own = (size_t)fy_solver_local_cells(s), first = info.field_cells == (int64_t)n_own ? 0 : (size_t)srank * n_own;
    if (decomposed && info.field_cells != (int64_t)n_own) return die("the processor directory does not hold this rank's slab");
    {
        std::vector<double> U(3 * (size_t)info.field_cells), p((size_t)info.field_cells);
        fy_foam_case_initial_fields(fc, U.data(), p.data());
        if (fy_solver_write_field_host(s, "p", p.data() + first) != FY_OK || fy_solver_write_field_host(s, "U", U.data() + 3 * first) != FY_OK) return die("initial fields");
        if (cd.turbulence_model != FY_TURBULENCE_LAMINAR) {              // nut.<phase> of the start time (eddyViscosity: MUST_READ)
            std::vector<double> nut((size_t)info.field_cells);
            if (fy_foam_case_initial_nut(fc, nut.data()) != FY_OK || fy_solver_write_field_host(s, "nut", nut.data() + first) != FY_OK) return die("initial nut");
            if ((cd.turbulence_model == FY_TURBULENCE_KEQN || cd.turbulence_model == FY_TURBULENCE_KEPSILON) &&
                (fy_foam_case_initial_k(fc, nut.data()) != FY_OK || fy_solver_write_field_host(s, "k", nut.data() + first) != FY_OK)) return die("initial k");
            if (cd.turbulence_model == FY_TURBULENCE_KEPSILON &&
                (fy_foam_case_initial_epsilon(fc, nut.data()) != FY_OK || fy_solver_write_field_host(s, "epsilon", nut.data() + first) != FY_OK)) return die("initial epsilon");
        }
    }
    // runTime.write(): one rank writes its solver's fields; slabs are gathered to the first solver rank, which writes the whole block
    auto write_time = [&](const char* tname) -> int {
        if (!comm || decomposed) return fy_foam_case_write_time(fc, s, tname);       // one domain, or every rank into its own processor directory
#ifdef FY_WITH_MPI
        const size_t n = (size_t)info.n_cells;
        const bool turb = cd.turbulence_model != FY_TURBULENCE_LAMINAR, has_k = cd.turbulence_model == FY_TURBULENCE_KEQN || cd.turbulence_model == FY_TURBULENCE_KEPSILON;
        const bool has_eps = cd.turbulence_model == FY_TURBULENCE_KEPSILON, has_alpha = solver == FY_SOLVER_PIMPLE;
        struct Fld { const char* name; int nc; bool on; std::vector<double> all; } f[6] = {{"U", 3, true, {}}, {"p", 1, true, {}}, {"alpha", 1, has_alpha, {}},
                                                                                          {"nut", 1, turb, {}}, {"k", 1, has_k, {}}, {"epsilon", 1, has_eps, {}}};
        std::vector<double> mine;
        for (Fld& q : f) {
            if (!q.on) continue;
            mine.resize(n_own * (size_t)q.nc);
            if (fy_solver_read_field_host(s, q.name, mine.data()) != FY_OK) return FY_ERR_INVALID;
            if (master) q.all.resize(n * (size_t)q.nc);
            if (MPI_Gather(mine.data(), (int)mine.size(), MPI_DOUBLE, master ? q.all.data() : nullptr, (int)mine.size(), MPI_DOUBLE, 0, solver_comm) != MPI_SUCCESS) return FY_ERR_TRANSPORT;
        }
        if (!master) return FY_OK;
        auto ptr = [&](int i) { return f[i].on ? f[i].all.data() : nullptr; };
        return fy_foam_case_write_fields(fc, tname, ptr(0), ptr(1), ptr(2), ptr(3), ptr(4), ptr(5));
#else
        return FY_ERR_UNSUPPORTED;
#endif
    };
    fy_solver_hold_sources(s, 1);                       // runTime.write() comes before setSourceZero (icoFoamYade.C:142-147)

    if (master) std::printf("\nStarting time loop\n\n");
    // runTime.loop(): fixed deltaT -> endTime / deltaT steps named start + k deltaT; adjustTimeStep (pimpleFoamYade.C:62-64) -> the time
    // advances by what setDeltaT.H chose for each step, until endTime is reached to within half a step
    const long n_steps = std::lround((info.end_time - info.start_time) / info.delta_t);
    double t = info.start_time;
    double dt_now = info.delta_t;                      // Time::run(): value() < endTime - 0.5 deltaT [OF-6 Time.C], with the deltaT setDeltaT.H left
    for (long k = 1; cd.adjust_time_step ? t < info.end_time - 0.5 * dt_now : k <= n_steps; ++k) {
        if (fy_solver_step(s) != FY_OK) return die("fy_solver_step");
        fy_step_stats st;
        fy_solver_get_stats(s, &st);
        t = cd.adjust_time_step ? t + st.delta_t : info.start_time + (double)k * info.delta_t;
        dt_now = st.delta_t;
        char tname[64];
        std::snprintf(tname, sizeof(tname), "%.12g", t);
        if (master) {
            std::printf("Time = %s\n\nCourant Number mean: %g max: %g\n", tname, st.courant_mean, st.courant_max);
            std::printf("pressure: %d solves, %d iterations, initial residual %g, final residual %g\n", st.p_solves, st.p_iters_total, st.p_initial_residual, st.p_final_residual);
            std::printf("time step continuity errors : sum local = %g, global = %g, cumulative = %g\n\n", st.cont_err_sum_local, st.cont_err_global, st.cont_err_cumulative);
        }
        if (info.write_interval_steps > 0 && k % info.write_interval_steps == 0)
            if (write_time(tname) != FY_OK) return die("writing the time directory");
    }
    if (master) std::printf("End\n");
    fy_solver_destroy(s);
    if (comm) fy_comm_destroy(comm);
    fy_foam_case_close(fc);
#ifdef FY_WITH_MPI
    if (tr.user) fy_mpi_transport_destroy(&tr);
    g_mpi_up = false;
    MPI_Finalize();
#endif
    return 0;
}
