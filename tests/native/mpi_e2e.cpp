// End-to-end check of the drop-in boundary over REAL MPI (test infrastructure): world rank 0 plays a serial Yade
// (the call sequence of FoamYade.C:176,181,228,510-531,537-549 as seen from the Yade side), world rank 1 is the solver rank
// and uses fyhip::FoamYade (C++ facade -> C-ABI -> HIP) with the MPI transport.  Inputs/outputs are raw files; the Python
// test compares them with the golden vectors produced by the reference.
//   mpiexec -n 1 mpi_e2e DIR : -n 1 mpi_e2e DIR        DIR/meta.txt as written by tests/test_mpi_e2e.py
#include <mpi.h>

#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "FoamYade.hpp"
#include "foamyade_mpi.h"

template <class T> static std::vector<T> rd(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); MPI_Abort(MPI_COMM_WORLD, 2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T)); if (n && fread(v.data(), 1, n, f) != (size_t)n) MPI_Abort(MPI_COMM_WORLD, 3); fclose(f); return v;
}
template <class T> static void wr(const std::string& p, const std::vector<T>& v) { FILE* f = fopen(p.c_str(), "wb"); fwrite(v.data(), sizeof(T), v.size(), f); fclose(f); }

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    int rank; MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    const std::string dir = argv[1];
    int nx, ny, nz, gaussian; double dx, ox, oy, oz, rhoP, rhoF, nu, dt, gx, gy, gz;
    { std::ifstream m(dir + "/meta.txt"); m >> nx >> ny >> nz >> dx >> ox >> oy >> oz >> gaussian >> rhoP >> rhoF >> nu >> dt >> gx >> gy >> gz; }
    if (rank == 0) {                                    // ---- fake serial Yade
        MPI_Comm dummy; MPI_Comm_split(MPI_COMM_WORLD, 2, rank, &dummy);
        std::vector<double> rec = rd<double>(dir + "/records.bin");
        int N = (int)(rec.size() / 10);
        MPI_Bcast(&N, 1, MPI_INT, 0, MPI_COMM_WORLD);
        MPI_Bcast(rec.data(), 10 * N, MPI_DOUBLE, 0, MPI_COMM_WORLD);
        std::vector<int> owner(N);
        for (int i = 0; i < N; ++i) { int d = -5; MPI_Allreduce(&d, &owner[i], 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD); }
        std::vector<double> F(6 * (size_t)N, 0.0);
        if (gaussian) for (int j = 0; j < 6 * N; ++j) { double z = 0; MPI_Allreduce(&z, &F[j], 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD); }
        else for (int i = 0; i < N; ++i) if (owner[i] > 0) { MPI_Status st; MPI_Recv(&F[6 * (size_t)i], 6, MPI_DOUBLE, owner[i], 1005, MPI_COMM_WORLD, &st); }
        double fdt = -1; MPI_Status st; MPI_Recv(&fdt, 1, MPI_DOUBLE, 1, 1050, MPI_COMM_WORLD, &st);
        double ydt = 1.25e-5; MPI_Bcast(&ydt, 1, MPI_DOUBLE, 0, MPI_COMM_WORLD);
        wr(dir + "/yade_force.bin", F); wr(dir + "/yade_owner.bin", owner); wr(dir + "/yade_fluiddt.bin", std::vector<double>{fdt});
    } else {                                            // ---- solver rank
        fy_transport T{};
        if (fy_mpi_transport_create(1, &T) != FY_OK) MPI_Abort(MPI_COMM_WORLD, 4);
        const int Nc = nx * ny * nz;
        std::vector<double> C(3 * (size_t)Nc), V(Nc, dx * dx * dx);
        for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
            size_t c = i + (size_t)nx * (j + (size_t)ny * k);
            C[3 * c] = ox + (i + 0.5) * dx; C[3 * c + 1] = oy + (j + 0.5) * dx; C[3 * c + 2] = oz + (k + 0.5) * dx;
        }
        std::vector<double> U = rd<double>(dir + "/U.bin"), gradP = rd<double>(dir + "/gradP.bin"), vGrad = rd<double>(dir + "/vGrad.bin"),
                            divT = rd<double>(dir + "/divT.bin"), ddtU = rd<double>(dir + "/ddtU.bin");
        std::vector<double> uSourceDrag(Nc, 5.0), alpha(Nc, 0.0), uSource(3 * (size_t)Nc, 3.0), uParticle(3 * (size_t)Nc, 4.0);
        fyhip::MeshView mv{};
        mv.n_cells = Nc; mv.centres = C.data(); mv.volumes = V.data();
        mv.bbox_min[0] = ox; mv.bbox_min[1] = oy; mv.bbox_min[2] = oz; mv.bbox_max[0] = ox + nx * dx; mv.bbox_max[1] = oy + ny * dx; mv.bbox_max[2] = oz + nz * dx;
        mv.nx = nx; mv.ny = ny; mv.nz = nz; mv.dx = dx; mv.origin[0] = ox; mv.origin[1] = oy; mv.origin[2] = oz;
        const double g[3] = {gx, gy, gz};
        try {
            fyhip::FoamYade yadeCoupling(mv, U.data(), gradP.data(), vGrad.data(), divT.data(), ddtU.data(), g, uSourceDrag.data(), alpha.data(),
                                         uSource.data(), uParticle.data(), gaussian != 0, &T);
            yadeCoupling.setScalarProperties(rhoP, rhoF, nu);
            yadeCoupling.setParticleAction(dt);
            wr(dir + "/foam_alpha.bin", alpha); wr(dir + "/foam_uSource.bin", uSource);
            wr(dir + "/foam_yadedt.bin", std::vector<double>{yadeCoupling.yadeDT()});
            yadeCoupling.setSourceZero();
        } catch (const std::exception& e) { fprintf(stderr, "%s\n", e.what()); MPI_Abort(MPI_COMM_WORLD, 5); }
        fy_mpi_transport_destroy(&T);
    }
    MPI_Finalize();
    return 0;
}
