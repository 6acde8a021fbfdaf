// Bench infrastructure (bench.py's drop-in leg): the coupled C3-type case driven over REAL MPI in parallel-Yade mode, the way the
// reference is deployed (README.md:29 of the reference: Yade ranks first in MPI_COMM_WORLD, solver ranks after them).
//
//   mpiexec -n 1 wire_bench ARGS : -n W wire_bench ARGS : -n K wire_bench ARGS        ARGS = n  particles  steps  dt  [c5|-]  [K]
//
//   K = 1 (default): one solver rank receives and answers everything itself (round 3).  K > 1: the solver side is a computing rank and K - 1
//   WIRE HELPERS (include/foamyade_mpi.h): to the workers a K-rank solver, each rank with its own bounding box -- they send every rank the
//   particles whose bounding sphere touches its box and take the answers from the ranks they sent to, as Yade's FoamCoupling does.
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
    const int K = argc > 6 ? std::atoi(argv[6]) : 1;                   // solver-side ranks (the last K of the world)
    if (K < 1 || world - 1 - K < 1) { if (rank == 0) std::fprintf(stderr, "wire_bench: need a master, >= 1 worker and K solver-side ranks\n"); MPI_Abort(MPI_COMM_WORLD, 2); }
    const int W = world - 1 - K, solver_rank = world - K;
    const double dx = 1.0 / n;

    if (rank < solver_rank) {
        MPI_Comm dummy;
        MPI_Comm_split(MPI_COMM_WORLD, 2, rank, &dummy);               // the solver side splits MPI_COMM_WORLD (FoamYade.C:21-22)
        std::vector<double> bbox(6 * (size_t)K);
        MPI_Status st;
        for (int f = 0; f < K; ++f) MPI_Recv(&bbox[6 * (size_t)f], 6, MPI_DOUBLE, solver_rank + f, TAG_GRID_BBOX, MPI_COMM_WORLD, &st);     // FoamYade.C:99-103: one box per solver rank
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
            std::vector<double> rec(10 * (size_t)cnt, 0.0);
            std::mt19937_64 gen(1000 + rank);
            std::uniform_real_distribution<double> u(0.0, 1.0);
            for (int i = 0; i < cnt; ++i) {
                double* r = &rec[10 * (size_t)i];
                r[0] = u(gen); r[1] = u(gen); r[2] = u(gen) * (c5 ? 1.0 / 3.0 : 0.6);
                r[9] = 0.2 * dx;
            }
            // per solver rank: the particles whose bounding sphere touches its box (what Yade's FoamCoupling sends it), packed once -- the cloud is at rest
            std::vector<std::vector<int> > idx((size_t)K);
            std::vector<std::vector<double> > out((size_t)K), F((size_t)K);
            std::vector<std::vector<int> > found((size_t)K);
            std::vector<int> counts((size_t)K);
            for (int f = 0; f < K; ++f) {
                const double* b = &bbox[6 * (size_t)f];
                for (int i = 0; i < cnt; ++i) {
                    const double* r = &rec[10 * (size_t)i];
                    bool hit = true;
                    for (int a = 0; a < 3; ++a) hit = hit && r[a] + r[9] >= b[a] && r[a] - r[9] <= b[3 + a];
                    if (hit) idx[(size_t)f].push_back(i);
                }
                counts[(size_t)f] = (int)idx[(size_t)f].size();
                out[(size_t)f].resize(10 * idx[(size_t)f].size()); F[(size_t)f].resize(6 * idx[(size_t)f].size()); found[(size_t)f].resize(idx[(size_t)f].size());
                for (size_t q = 0; q < idx[(size_t)f].size(); ++q) std::memcpy(&out[(size_t)f][10 * q], &rec[10 * (size_t)idx[(size_t)f][q]], 10 * sizeof(double));
            }
            double sum = 0.0;
            long long n_found = 0, n_twice = 0;
            std::vector<MPI_Request> rq((size_t)K);
            for (int s = 0; s <= steps; ++s) {
                for (int f = 0; f < K; ++f) MPI_Send(counts.data(), K, MPI_INT, solver_rank + f, TAG_SZ_BUFF, MPI_COMM_WORLD);       // every rank gets every count (FoamYade.C:122-125)
                int nr = 0;
                for (int f = 0; f < K; ++f)
                    if (counts[(size_t)f] > 0) MPI_Isend(out[(size_t)f].data(), 10 * counts[(size_t)f], MPI_DOUBLE, solver_rank + f, TAG_YADE_DATA, MPI_COMM_WORLD, &rq[(size_t)nr++]);
                MPI_Waitall(nr, rq.data(), MPI_STATUSES_IGNORE);
                for (int f = 0; f < K; ++f) {
                    if (counts[(size_t)f] <= 0) continue;
                    MPI_Recv(found[(size_t)f].data(), counts[(size_t)f], MPI_INT, solver_rank + f, TAG_SEARCH_RES, MPI_COMM_WORLD, &st);
                    MPI_Recv(F[(size_t)f].data(), 6 * counts[(size_t)f], MPI_DOUBLE, solver_rank + f, TAG_FORCE, MPI_COMM_WORLD, &st);
                }
                if (s == steps) {
                    std::vector<unsigned char> hits((size_t)cnt, 0);
                    for (int f = 0; f < K; ++f)
                        for (size_t q = 0; q < idx[(size_t)f].size(); ++q) {
                            if (found[(size_t)f][q] == 1) { ++hits[(size_t)idx[(size_t)f][q]]; }
                            sum += F[(size_t)f][6 * q + 2];
                        }
                    for (int i = 0; i < cnt; ++i) { n_found += hits[(size_t)i] >= 1; n_twice += hits[(size_t)i] > 1; }
                }
            }
            double res[3] = {sum, (double)n_found, (double)n_twice};
            MPI_Send(res, 3, MPI_DOUBLE, solver_rank, TAG_BENCH_SUM, MPI_COMM_WORLD);
        }
        MPI_Finalize();
        return 0;
    }

    // ---- the solver rank
    fy_transport T{};
    if (K > 1) {
        int is_helper = 0;
        if (fy_mpi_transport_create_wire_helpers(W + 1, &T, &is_helper) != FY_OK) { std::fprintf(stderr, "wire_bench: fy_mpi_transport_create_wire_helpers failed\n"); MPI_Abort(MPI_COMM_WORLD, 3); }
        if (is_helper) {
            const int rc = fy_mpi_wire_helper_serve(&T);
            if (rc != FY_OK) { std::fprintf(stderr, "wire_bench: helper %d failed (%d)\n", rank, rc); MPI_Abort(MPI_COMM_WORLD, 6); }
            MPI_Finalize();
            return 0;
        }
    } else if (fy_mpi_transport_create(W + 1, &T) != FY_OK) { std::fprintf(stderr, "wire_bench: fy_mpi_transport_create failed\n"); MPI_Abort(MPI_COMM_WORLD, 3); }
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
        if (std::getenv("WIRE_BENCH_VERBOSE")) std::fprintf(stderr, "step %d: %.2f ms (wire_recv %.2f, wire_send %.2f, h2d %.2f, d2h %.2f, particle phase %.2f)\n", k, el, pt.wire_recv, pt.wire_send, pt.copy_in, pt.copy_out, st.ms_particle);
        if (k == 0) continue;
        acc_step += el; acc_in += pt.copy_in; acc_out += pt.copy_out; acc_recv += pt.wire_recv; acc_send += pt.wire_send; acc_part += st.ms_particle;
        bytes_in += pt.bytes_in; bytes_out += pt.bytes_out;
    }
    double sum = 0.0, n_found = 0.0, n_twice = 0.0;
    for (int w = 1; w <= W; ++w) {
        double in[3];
        MPI_Status st;
        MPI_Recv(in, 3, MPI_DOUBLE, w, TAG_BENCH_SUM, MPI_COMM_WORLD, &st);
        sum += in[0]; n_found += in[1]; n_twice += in[2];
    }
    const double KS = (double)steps;
    std::printf("{\"ms_per_step\": %.3f, \"h2d\": %.3f, \"d2h\": %.3f, \"wire_recv\": %.3f, \"wire_send\": %.3f, \"particle_phase_incl_transfers\": %.3f, "
                "\"bytes_in\": %lld, \"bytes_out\": %lld, \"found_at_the_workers\": %.0f, \"found_by_two_ranks\": %.0f, \"sum_fz_at_the_workers\": %.9e, \"workers\": %d, "
                "\"solver_side_ranks\": %d, \"steps\": %d}\n",
                acc_step / KS, acc_in / KS, acc_out / KS, acc_recv / KS, acc_send / KS, acc_part / KS, (long long)(bytes_in / steps), (long long)(bytes_out / steps), n_found, n_twice, sum, W, K, steps);
    std::fflush(stdout);
    fy_solver_destroy(s);
    fy_mpi_transport_destroy(&T);
    MPI_Finalize();
    return 0;
}
