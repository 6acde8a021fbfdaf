// Bench infrastructure (bench.py's drop-in leg): the coupled C3-type case driven over REAL MPI in parallel-Yade mode, the way the
// reference is deployed (README.md:29 of the reference: Yade ranks first in MPI_COMM_WORLD, solver ranks after them).
//
//   mpiexec -n 1 wire_bench ARGS : -n W wire_bench ARGS : -n 1 wire_bench ARGS        ARGS = n  particles  steps  dt  [c5]
//
//   world rank 0        Yade master: bounding box (FoamYade.C:99-103), then per step the dt handshake (FoamYade.C:537-549)
//   world ranks 1..W    Yade workers: bounding box, then per step the counts (tag 1003), their slice of the cloud (1002), and back the
//                       search results (1004) and the hydrodynamic forces (1005) -- the peer of FoamYade.C:114-155, 239-243, 504-507
//   world rank W + 1    the solver rank: fy_solver (pimpleFoamYade loop body, HIP) with the MPI transport of libfoamyade_mpi
//
// Every message is a real MPI_Send / MPI_Recv between processes; nothing is shared.  The solver rank times `steps` coupled steps after
// one warm-up step and prints ONE JSON line: wall time per step, the phases the library's own clocks saw (PCIe copies, host time inside
// the MPI calls), and a checksum of what the workers received, which they send to it at the end.
#include <mpi.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "foamyade_hip.h"
#include "foamyade_mpi.h"

namespace {
enum { TAG_SZ_BUFF = 1003, TAG_YADE_DATA = 1002, TAG_SEARCH_RES = 1004, TAG_FORCE = 1005, TAG_GRID_BBOX = 1001, TAG_FLUID_DT = 1050, TAG_YADE_DT = 1060,
       TAG_BENCH_SUM = 1999 };
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    int rank = 0, world = 1;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &world);
    if (argc < 5 || world < 3) { if (rank == 0) std::fprintf(stderr, "usage: see the head of wire_bench.cpp\n"); MPI_Abort(MPI_COMM_WORLD, 2); }
    const int n = std::atoi(argv[1]);
    const long long n_part = std::atoll(argv[2]);
    const int steps = std::atoi(argv[3]);
    const double dt = std::atof(argv[4]);
    const bool c5 = argc > 5 && std::string(argv[5]) == "c5";
    const int W = world - 2, solver_rank = world - 1;
    const double dx = 1.0 / n;

    if (rank != solver_rank) {
        MPI_Comm dummy;
        MPI_Comm_split(MPI_COMM_WORLD, 2, rank, &dummy);               // the solver side splits MPI_COMM_WORLD (FoamYade.C:21-22)
        double bbox[6];
        MPI_Status st;
        MPI_Recv(bbox, 6, MPI_DOUBLE, solver_rank, TAG_GRID_BBOX, MPI_COMM_WORLD, &st);     // FoamYade.C:99-103
        if (rank == 0) {                                               // ---- Yade master
            for (int s = 0; s <= steps; ++s) {
                double fdt = -1.0;
                MPI_Recv(&fdt, 1, MPI_DOUBLE, solver_rank, TAG_FLUID_DT, MPI_COMM_WORLD, &st);
                double ydt = 1e-6;
                MPI_Send(&ydt, 1, MPI_DOUBLE, solver_rank, TAG_YADE_DT, MPI_COMM_WORLD);
            }
        } else {                                                       // ---- Yade worker `rank`: its share of the cloud, at rest, r = 0.2 dx
            const long long lo = (rank - 1) * n_part / W, hi = rank * n_part / W;
            const int cnt = (int)(hi - lo);
            std::vector<double> rec(10 * (size_t)cnt, 0.0), F(6 * (size_t)cnt);
            std::vector<int> found((size_t)cnt);
            std::mt19937_64 gen(1000 + rank);
            std::uniform_real_distribution<double> u(0.0, 1.0);
            for (int i = 0; i < cnt; ++i) {
                double* r = &rec[10 * (size_t)i];
                r[0] = u(gen); r[1] = u(gen); r[2] = u(gen) * (c5 ? 1.0 / 3.0 : 0.6);
                r[9] = 0.2 * dx;
            }
            double sum = 0.0;
            long long n_found = 0;
            for (int s = 0; s <= steps; ++s) {
                int counts[1] = {cnt};                                  // one solver rank: every particle intersects its bounding box
                MPI_Send(counts, 1, MPI_INT, solver_rank, TAG_SZ_BUFF, MPI_COMM_WORLD);
                MPI_Send(rec.data(), 10 * cnt, MPI_DOUBLE, solver_rank, TAG_YADE_DATA, MPI_COMM_WORLD);
                MPI_Recv(found.data(), cnt, MPI_INT, solver_rank, TAG_SEARCH_RES, MPI_COMM_WORLD, &st);
                MPI_Recv(F.data(), 6 * cnt, MPI_DOUBLE, solver_rank, TAG_FORCE, MPI_COMM_WORLD, &st);
                if (s == steps) {
                    for (int i = 0; i < cnt; ++i) { n_found += found[(size_t)i] == 1; sum += F[6 * (size_t)i + 2]; }
                }
            }
            double out[2] = {sum, (double)n_found};
            MPI_Send(out, 2, MPI_DOUBLE, solver_rank, TAG_BENCH_SUM, MPI_COMM_WORLD);
        }
        MPI_Finalize();
        return 0;
    }

    // ---- the solver rank
    fy_transport T{};
    if (fy_mpi_transport_create(W + 1, &T) != FY_OK) { std::fprintf(stderr, "wire_bench: fy_mpi_transport_create failed\n"); MPI_Abort(MPI_COMM_WORLD, 3); }
    fy_case_desc cd;
    fy_case_defaults(&cd, FY_SOLVER_PIMPLE);
    cd.nx = cd.ny = cd.nz = n; cd.dx = dx; cd.dt = dt; cd.nu = 1e-6; cd.rho_fluid = 1000.0; cd.rho_particle = 2650.0;
    cd.g[0] = 0; cd.g[1] = 0; cd.g[2] = -9.81;
    for (int q = 0; q < 6; ++q) { cd.u_bc[q] = FY_BC_U_FIXED_VALUE; cd.p_bc[q] = FY_BC_P_FIXED_FLUX; for (int a = 0; a < 3; ++a) cd.u_value[q][a] = 0.0; }
    if (c5) { cd.u_value[FY_ZMIN][2] = 0.05; cd.u_bc[FY_ZMAX] = FY_BC_U_ZERO_GRADIENT; cd.p_bc[FY_ZMAX] = FY_BC_P_FIXED_VALUE; cd.p_value[FY_ZMAX] = 0.0; }
    cd.n_outer_correctors = 1; cd.n_correctors = 2; cd.p_solver = FY_PSOLVER_PCG_MG;
    fy_solver* s = nullptr;
    if (fy_solver_create(&cd, &T, 0, &s) != FY_OK) { std::fprintf(stderr, "wire_bench: fy_solver_create: %s\n", fy_last_error()); MPI_Abort(MPI_COMM_WORLD, 4); }
    fy_ctx* cpl = fy_solver_coupling(s);
    fy_enable_timing(cpl, 1);
    double acc_step = 0, acc_in = 0, acc_out = 0, acc_recv = 0, acc_send = 0, acc_part = 0;
    long long bytes_in = 0, bytes_out = 0;
    for (int k = 0; k <= steps; ++k) {                                  // step 0 warms up (first touch of the pinned staging, tile capacities)
        const double t0 = now_ms();
        if (fy_solver_step(s) != FY_OK) { std::fprintf(stderr, "wire_bench: fy_solver_step: %s\n", fy_last_error()); MPI_Abort(MPI_COMM_WORLD, 5); }
        const double el = now_ms() - t0;                                // (the step returns after its final stream synchronisation)
        fy_particle_timings pt;
        fy_step_stats st;
        fy_get_particle_timings(cpl, &pt);
        fy_solver_get_stats(s, &st);
        if (k == 0) continue;
        acc_step += el; acc_in += pt.copy_in; acc_out += pt.copy_out; acc_recv += pt.wire_recv; acc_send += pt.wire_send; acc_part += st.ms_particle;
        bytes_in += pt.bytes_in; bytes_out += pt.bytes_out;
    }
    double sum = 0.0, n_found = 0.0;
    for (int w = 1; w <= W; ++w) {
        double in[2];
        MPI_Status st;
        MPI_Recv(in, 2, MPI_DOUBLE, w, TAG_BENCH_SUM, MPI_COMM_WORLD, &st);
        sum += in[0]; n_found += in[1];
    }
    const double K = (double)steps;
    std::printf("{\"ms_per_step\": %.3f, \"h2d\": %.3f, \"d2h\": %.3f, \"wire_recv\": %.3f, \"wire_send\": %.3f, \"particle_phase_incl_transfers\": %.3f, "
                "\"bytes_in\": %lld, \"bytes_out\": %lld, \"found_at_the_workers\": %.0f, \"sum_fz_at_the_workers\": %.9e, \"workers\": %d, \"steps\": %d}\n",
                acc_step / K, acc_in / K, acc_out / K, acc_recv / K, acc_send / K, acc_part / K, (long long)(bytes_in / steps), (long long)(bytes_out / steps), n_found, sum, W, steps);
    std::fflush(stdout);
    fy_solver_destroy(s);
    fy_mpi_transport_destroy(&T);
    MPI_Finalize();
    return 0;
}
