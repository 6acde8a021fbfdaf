// Test infrastructure: a serial Yade as seen from the solver over REAL MPI (the Yade side of FoamYade.C:176,181,228,510-531,537-549),
// repeated for NSTEPS coupling steps with the same particle records.  World rank 0; the solver (foamYadeHip_mpi) is world rank 1.
//   mpiexec -n 1 fake_yade RECORDS.bin GAUSSIAN NSTEPS OUT_FORCE.bin : -n 1 foamYadeHip_mpi -solver ... -case ...
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    if (argc < 5) MPI_Abort(MPI_COMM_WORLD, 2);
    const int gaussian = std::atoi(argv[2]), nsteps = std::atoi(argv[3]);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) MPI_Abort(MPI_COMM_WORLD, 3);
    std::fseek(f, 0, SEEK_END); const long nb = std::ftell(f); std::fseek(f, 0, SEEK_SET);
    std::vector<double> rec(nb / sizeof(double));
    if (nb && std::fread(rec.data(), 1, nb, f) != (size_t)nb) MPI_Abort(MPI_COMM_WORLD, 3);
    std::fclose(f);
    MPI_Comm dummy;
    MPI_Comm_split(MPI_COMM_WORLD, 2, 0, &dummy);                       // the solver side splits MPI_COMM_WORLD (FoamYade.C:21-22)
    int N = (int)(rec.size() / 10);
    std::vector<double> F(6 * (size_t)N, 0.0);
    for (int s = 0; s < nsteps; ++s) {
        MPI_Bcast(&N, 1, MPI_INT, 0, MPI_COMM_WORLD);
        MPI_Bcast(rec.data(), 10 * N, MPI_DOUBLE, 0, MPI_COMM_WORLD);
        std::vector<int> owner(N);
        for (int i = 0; i < N; ++i) { int d = -5; MPI_Allreduce(&d, &owner[i], 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD); }
        std::fill(F.begin(), F.end(), 0.0);
        if (gaussian) for (int j = 0; j < 6 * N; ++j) { double z = 0; MPI_Allreduce(&z, &F[j], 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD); }
        else for (int i = 0; i < N; ++i) if (owner[i] > 0) { MPI_Status st; MPI_Recv(&F[6 * (size_t)i], 6, MPI_DOUBLE, owner[i], 1005, MPI_COMM_WORLD, &st); }
        double fdt = -1; MPI_Status st;
        MPI_Recv(&fdt, 1, MPI_DOUBLE, 1, 1050, MPI_COMM_WORLD, &st);
        double ydt = 1.25e-5;
        MPI_Bcast(&ydt, 1, MPI_DOUBLE, 0, MPI_COMM_WORLD);
    }
    FILE* o = std::fopen(argv[4], "wb");
    std::fwrite(F.data(), sizeof(double), F.size(), o);
    std::fclose(o);
    MPI_Finalize();
    return 0;
}
