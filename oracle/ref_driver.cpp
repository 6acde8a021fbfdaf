// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked into the product library.
//
// Drives the UNMODIFIED reference particle path (/root/reference/FoamYade/FoamYade.C and
// /root/reference/FoamYade/meshtree/meshTree.C, compiled by oracle/Makefile against
// oracle/shim/fvCFD.H) under MPICH MPMD with fake Yade rank(s), and dumps what the
// reference computed as raw little-endian arrays.  tests/golden/gen_golden.py turns
// those dumps into the committed golden fixtures.  This file is our own code; it only
// *calls* the reference's public members (FoamYade.H:69-160 are all public).
//
// Launch (serial Yade, FoamYade.C:31):   mpiexec -n 1 ref_driver DIR : -n 1 ref_driver DIR
// Launch (parallel Yade, W workers):     mpiexec -n (1+W) ref_driver DIR : -n 1 ref_driver DIR
// World rank layout follows README.md:29 / FoamYade.C:28-43: Yade ranks first
// (0 = Yade master, 1..W = workers), the single Foam rank last.
//
// DIR/meta.txt:  nx ny nz dx ox oy oz gaussian nYade nsteps rhoP rhoF nu dt gx gy gz [fibre]
//   fibre = 1 sets the public flag FoamYade::fibreCpl (FoamYade.H:102); the records files then hold Np*15 doubles (FoamYade.C:131-136)
// DIR/records_s<step>.bin : Np*10 doubles  [x y z vx vy vz wx wy wz radius]  (FoamYade.C:190-219)
// DIR/{U,gradP,divT,ddtU}.bin : Nc*3 doubles ; DIR/vGrad.bin : Nc*9 doubles
#include "FoamYade.H"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <fstream>
#include <sstream>
#include <stdexcept>

MPI_Comm Foam::PstreamGlobals::MPI_COMM_FOAM;

namespace {

struct Meta {
    int nx, ny, nz; double dx, ox, oy, oz; int gaussian, nYade, nsteps;
    double rhoP, rhoF, nu, dt, gx, gy, gz;
    int fibre = 0;
    int stride() const { return fibre ? 15 : 10; }
};

Meta read_meta(const std::string& dir) {
    std::ifstream f(dir + "/meta.txt");
    if (!f) throw std::runtime_error("cannot open meta.txt");
    Meta m;
    f >> m.nx >> m.ny >> m.nz >> m.dx >> m.ox >> m.oy >> m.oz >> m.gaussian >> m.nYade >> m.nsteps
      >> m.rhoP >> m.rhoF >> m.nu >> m.dt >> m.gx >> m.gy >> m.gz;
    if (!(f >> m.fibre)) m.fibre = 0;
    return m;
}

template <class T>
std::vector<T> read_bin(const std::string& path) {
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("cannot open " + path);
    std::fseek(fp, 0, SEEK_END);
    long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T));
    if (n && std::fread(v.data(), 1, n, fp) != (size_t)n) throw std::runtime_error("short read " + path);
    std::fclose(fp);
    return v;
}

template <class T>
void write_bin(const std::string& path, const T* p, size_t n) {
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("cannot write " + path);
    if (n) std::fwrite(p, sizeof(T), n, fp);
    std::fclose(fp);
}

std::string sfx(const std::string& base, int step) {
    std::ostringstream o; o << base << "_s" << step << ".bin"; return o.str();
}

// contiguous split of Np particles over W workers: worker w (0-based) owns [lo,hi)
void split(int Np, int W, int w, int& lo, int& hi) { lo = (int)((long)Np * w / W); hi = (int)((long)Np * (w + 1) / W); }

void preorder(const Foam::kdNode* n, std::vector<int>& out) {
    if (!n) return;
    out.push_back(n->p.id);
    preorder(n->left, out);
    preorder(n->right, out);
}

const int MAXK = 16;  // dump width; reference bound is 12 (+ UB growth, meshTree.H:64-78)

// ------------------------------------------------------------------------------------------------
// Fake Yade, serial mode (one Yade process = world rank 0).  Mirrors the Foam-side call sequence
// FoamYade.C:176,181,228,510-531,537-549.
void fake_yade_serial(const std::string& dir, const Meta& m, int foamRank) {
    for (int s = 0; s < m.nsteps; ++s) {
        std::vector<double> rec = read_bin<double>(dir + "/" + sfx("records", s));
        const int L = m.stride();
        int N = (int)(rec.size() / L);
        MPI_Bcast(&N, 1, MPI_INT, 0, MPI_COMM_WORLD);
        MPI_Bcast(rec.data(), L * N, MPI_DOUBLE, 0, MPI_COMM_WORLD);
        std::vector<int> owner(N, -7);
        for (int i = 0; i < N; ++i) { int d = -5; MPI_Allreduce(&d, &owner[i], 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD); }
        std::vector<double> F(6 * (size_t)N, 0.0);
        if (m.gaussian) {
            for (int j = 0; j < 6 * N; ++j) { double z = 0.0; MPI_Allreduce(&z, &F[j], 1, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD); }
        } else {
            for (int i = 0; i < N; ++i) {
                // owner 0 == "nobody located it" (found starts at 0, FoamYade.C:202): the reference sends nothing.
                if (owner[i] > 0) { MPI_Status st; MPI_Recv(&F[6 * (size_t)i], 6, MPI_DOUBLE, owner[i], 1005, MPI_COMM_WORLD, &st); }
            }
        }
        double fluidDt = -1.0; MPI_Status st;
        MPI_Recv(&fluidDt, 1, MPI_DOUBLE, foamRank, 1050, MPI_COMM_WORLD, &st);
        double yadeDt = 1.25e-5 * (s + 1);
        MPI_Bcast(&yadeDt, 1, MPI_DOUBLE, 0, MPI_COMM_WORLD);
        write_bin(dir + "/" + sfx("wire_owner", s), owner.data(), owner.size());
        write_bin(dir + "/" + sfx("wire_force", s), F.data(), F.size());
        write_bin(dir + "/" + sfx("wire_fluiddt", s), &fluidDt, 1);
    }
}

// Fake Yade, parallel mode.  rank 0 = master (bbox + dt handshake only), ranks 1..W = workers.
void fake_yade_parallel(const std::string& dir, const Meta& m, int rank, int foamRank) {
    const int W = m.nYade - 1;
    double bbox[6]; MPI_Status st;
    MPI_Recv(bbox, 6, MPI_DOUBLE, foamRank, 1001, MPI_COMM_WORLD, &st);  // FoamYade.C:99-108
    if (rank == 0) write_bin(dir + "/wire_bbox.bin", bbox, 6);
    for (int s = 0; s < m.nsteps; ++s) {
        if (rank == 0) {
            double fluidDt = -1.0;
            MPI_Recv(&fluidDt, 1, MPI_DOUBLE, foamRank, 1050, MPI_COMM_WORLD, &st);   // FoamYade.C:538-540
            double yadeDt = 1.25e-5 * (s + 1);
            MPI_Send(&yadeDt, 1, MPI_DOUBLE, foamRank, 1060, MPI_COMM_WORLD);          // FoamYade.C:542-545
            write_bin(dir + "/" + sfx("wire_fluiddt", s), &fluidDt, 1);
        } else {
            std::vector<double> rec = read_bin<double>(dir + "/" + sfx("records", s));
            const int L = m.stride();
            int N = (int)(rec.size() / L), lo, hi;
            split(N, W, rank - 1, lo, hi);
            int cnt = hi - lo;                                   // one Foam rank => localCommSize == 1
            MPI_Send(&cnt, 1, MPI_INT, foamRank, 1003, MPI_COMM_WORLD);                 // FoamYade.C:122-125
            if (cnt > 0) {
                MPI_Send(&rec[L * (size_t)lo], L * cnt, MPI_DOUBLE, foamRank, 1002, MPI_COMM_WORLD);  // :149-153
                std::vector<int> found(cnt, 0);
                MPI_Recv(found.data(), cnt, MPI_INT, foamRank, 1004, MPI_COMM_WORLD, &st);               // :239-243
                std::vector<double> F(6 * (size_t)cnt, -9.0);
                MPI_Recv(F.data(), 6 * cnt, MPI_DOUBLE, foamRank, 1005, MPI_COMM_WORLD, &st);            // :504-507
                std::ostringstream a, b; a << "wire_found_w" << rank; b << "wire_force_w" << rank;
                write_bin(dir + "/" + sfx(a.str(), s), found.data(), found.size());
                write_bin(dir + "/" + sfx(b.str(), s), F.data(), F.size());
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
void foam_rank(const std::string& dir, const Meta& m) {
    const int nx = m.nx, ny = m.ny, nz = m.nz, Nc = nx * ny * nz;
    Foam::fvMesh mesh;
    mesh.nx = nx; mesh.ny = ny; mesh.nz = nz; mesh.dx = m.dx;
    mesh.Cc.f.resize(Nc); mesh.Vv.f.resize(Nc);
    const double vol = m.dx * m.dx * m.dx;
    for (int k = 0; k < nz; ++k) for (int j = 0; j < ny; ++j) for (int i = 0; i < nx; ++i) {
        int c = i + nx * (j + ny * k);
        mesh.Cc.f[c] = Foam::vector(m.ox + (i + 0.5) * m.dx, m.oy + (j + 0.5) * m.dx, m.oz + (k + 0.5) * m.dx);
        mesh.Vv.f[c] = vol;
    }
    for (int k = 0; k <= nz; ++k) for (int j = 0; j <= ny; ++j) for (int i = 0; i <= nx; ++i)
        mesh.pts.f.push_back(Foam::vector(m.ox + i * m.dx, m.oy + j * m.dx, m.oz + k * m.dx));
    mesh.bbmin = Foam::vector(m.ox, m.oy, m.oz);
    mesh.bbmax = Foam::vector(m.ox + nx * m.dx, m.oy + ny * m.dx, m.oz + nz * m.dx);

    {   // a non-uniform mesh: mesh.C(), mesh.V(), mesh.points() as handed over (Gaussian mode only: findCell above is a uniform-block stand-in)
        std::ifstream probe(dir + "/mesh_centres.bin", std::ios::binary);
        if (probe.good()) {
            std::vector<double> cc = read_bin<double>(dir + "/mesh_centres.bin"), vv = read_bin<double>(dir + "/mesh_volumes.bin"),
                                pp = read_bin<double>(dir + "/mesh_points.bin");
            if ((int)vv.size() != Nc || (int)cc.size() != 3 * Nc || !m.gaussian) throw std::runtime_error("bad non-uniform mesh files");
            for (int c = 0; c < Nc; ++c) { mesh.Cc.f[c] = Foam::vector(cc[3 * (size_t)c], cc[3 * (size_t)c + 1], cc[3 * (size_t)c + 2]); mesh.Vv.f[c] = vv[c]; }
            mesh.pts.f.clear();
            for (size_t q = 0; q + 2 < pp.size(); q += 3) mesh.pts.f.push_back(Foam::vector(pp[q], pp[q + 1], pp[q + 2]));
        }
    }

    Foam::volVectorField U, gradP, divT, ddtU, uSource, uParticle;
    Foam::volTensorField vGrad;
    Foam::volScalarField uSourceDrag, alpha;
    auto loadV = [&](Foam::volVectorField& F, const char* name) {
        std::vector<double> a = read_bin<double>(dir + "/" + name);
        F.f.resize(Nc);
        for (int c = 0; c < Nc; ++c) F.f[c] = Foam::vector(a[3 * (size_t)c], a[3 * (size_t)c + 1], a[3 * (size_t)c + 2]);
    };
    loadV(U, "U.bin"); loadV(gradP, "gradP.bin"); loadV(divT, "divT.bin"); loadV(ddtU, "ddtU.bin");
    {
        std::vector<double> a = read_bin<double>(dir + "/vGrad.bin");
        vGrad.f.resize(Nc);
        for (int c = 0; c < Nc; ++c) for (int q = 0; q < 9; ++q) vGrad.f[c].t[q] = a[9 * (size_t)c + q];
    }
    // deliberately NOT the post-initFields state, to pin FoamYade::initFields (FoamYade.C:56-73)
    uSource.f.assign(Nc, Foam::vector(3, 3, 3)); uParticle.f.assign(Nc, Foam::vector(4, 4, 4));
    uSourceDrag.f.assign(Nc, 5.0); alpha.f.assign(Nc, 0.0);
    Foam::uniformDimensionedVectorField g; g.v = Foam::vector(m.gx, m.gy, m.gz);

    // `bool serialYade` is only ever assigned `true` (FoamYade.C:31) and has no initialiser (FoamYade.H:91), so in
    // parallel-Yade mode the reference reads an indeterminate value.  Construct into zero-filled storage so that it
    // reads `false` there, which is the evident intent; the reference source itself stays untouched.
    static std::aligned_storage<sizeof(Foam::FoamYade), alignof(Foam::FoamYade)>::type fyStorage;
    std::memset(&fyStorage, 0, sizeof(fyStorage));
    Foam::FoamYade& fy = *new (&fyStorage) Foam::FoamYade(mesh, U, gradP, vGrad, divT, ddtU, g, uSourceDrag, alpha, uSource,
                                                         uParticle, m.gaussian != 0);
    fy.setScalarProperties(m.rhoP, m.rhoF, m.nu);
    fy.fibreCpl = m.fibre != 0;                                // public member, FoamYade.H:102

    {   // tree, preorder (meshTree.C:19-37)
        std::vector<int> pre; pre.reserve(Nc);
        preorder(fy.mshTree.root, pre);
        write_bin(dir + "/tree_preorder.bin", pre.data(), pre.size());
        double scal[2] = {fy.interpRange, fy.sigmaInterp};
        write_bin(dir + "/interp_scalars.bin", scal, 2);
        // state right after the constructor's initFields
        write_bin(dir + "/init_alpha.bin", alpha.f.data(), (size_t)Nc);
        write_bin(dir + "/init_uSource.bin", &uSource.f[0].v[0], 3 * (size_t)Nc);
    }

    const int W = m.nYade - 1;
    for (int s = 0; s < m.nsteps; ++s) {
        fy.setParticleAction(m.dt);

        // per-particle dump in GLOBAL particle numbering
        std::vector<double> rec = read_bin<double>(dir + "/" + sfx("records", s));
        const int L = m.stride();
        const int Np = (int)(rec.size() / L);
        std::vector<int> k(Np, 0), ids((size_t)Np * MAXK, -1), incell(Np, -1);
        std::vector<double> w((size_t)Np * MAXK, 0.0), FT((size_t)Np * 6, 0.0);
        for (const auto& yp : fy.inCommProcs) {
            int lo = 0, hi = Np;
            if (m.nYade > 1) split(Np, W, yp->yRank - 1, lo, hi);
            for (const auto& prt : yp->foundParticles) {
                const int gi = lo + prt->indx;
                const int kk = (int)prt->cellIds.size();
                if (kk > MAXK) { std::fprintf(stderr, "stencil %d exceeds dump width\n", kk); MPI_Abort(MPI_COMM_WORLD, 3); }
                k[gi] = kk; incell[gi] = prt->inCell;
                for (int q = 0; q < kk; ++q) ids[(size_t)gi * MAXK + q] = prt->cellIds[q];
                for (size_t q = 0; q < prt->interpCellWeight.size(); ++q) w[(size_t)gi * MAXK + q] = prt->interpCellWeight[q].second;
                FT[6 * (size_t)gi + 0] = prt->hydroForce.x(); FT[6 * (size_t)gi + 1] = prt->hydroForce.y(); FT[6 * (size_t)gi + 2] = prt->hydroForce.z();
                FT[6 * (size_t)gi + 3] = prt->hydroTorque.x(); FT[6 * (size_t)gi + 4] = prt->hydroTorque.y(); FT[6 * (size_t)gi + 5] = prt->hydroTorque.z();
            }
        }
        {   // meshTree::nearestCell (meshTree.C:66-135; no call site in FoamYade) on every record's position, for the NN locate of row A6
            std::vector<int> nn(Np, -1);
            for (int q = 0; q < Np; ++q) nn[q] = fy.mshTree.nearestCell(Foam::vector(rec[L * (size_t)q], rec[L * (size_t)q + 1], rec[L * (size_t)q + 2]));
            write_bin(dir + "/" + sfx("part_nn", s), nn.data(), nn.size());
        }
        write_bin(dir + "/" + sfx("part_k", s), k.data(), k.size());
        write_bin(dir + "/" + sfx("part_incell", s), incell.data(), incell.size());
        write_bin(dir + "/" + sfx("part_ids", s), ids.data(), ids.size());
        write_bin(dir + "/" + sfx("part_w", s), w.data(), w.size());
        write_bin(dir + "/" + sfx("part_force", s), FT.data(), FT.size());
        write_bin(dir + "/" + sfx("alpha", s), alpha.f.data(), (size_t)Nc);
        write_bin(dir + "/" + sfx("uSourceDrag", s), uSourceDrag.f.data(), (size_t)Nc);
        write_bin(dir + "/" + sfx("uParticle", s), &uParticle.f[0].v[0], 3 * (size_t)Nc);
        write_bin(dir + "/" + sfx("uSource", s), &uSource.f[0].v[0], 3 * (size_t)Nc);
        double ydt = fy.yadeDT;
        write_bin(dir + "/" + sfx("foam_yadedt", s), &ydt, 1);

        if (m.gaussian) {
            // The two force models the reference ships WITHOUT a live call site: Gaussian calcHydroTorque (its call is commented
            // out at FoamYade.C:618) and addedMassForce (FoamYade.C:392-413, never called).  Called here on the particles the
            // step just processed, so that the optional fy_set_force_models() path has reference outputs to be checked against.
            for (const auto& yp : fy.inCommProcs) fy.calcHydroTorque(yp.get());
            for (const auto& yp : fy.inCommProcs) for (const auto& prt : yp->foundParticles) fy.addedMassForce(prt.get());
            std::vector<double> FX((size_t)Np * 6, 0.0);
            for (const auto& yp : fy.inCommProcs) {
                int lo = 0, hi = Np;
                if (m.nYade > 1) split(Np, W, yp->yRank - 1, lo, hi);
                for (const auto& prt : yp->foundParticles) {
                    const size_t gi = (size_t)(lo + prt->indx);
                    FX[6 * gi + 0] = prt->hydroForce.x(); FX[6 * gi + 1] = prt->hydroForce.y(); FX[6 * gi + 2] = prt->hydroForce.z();
                    FX[6 * gi + 3] = prt->hydroTorque.x(); FX[6 * gi + 4] = prt->hydroTorque.y(); FX[6 * gi + 5] = prt->hydroTorque.z();
                }
            }
            write_bin(dir + "/" + sfx("part_forcex", s), FX.data(), FX.size());
            write_bin(dir + "/" + sfx("uSourcex", s), &uSource.f[0].v[0], 3 * (size_t)Nc);
        }

        fy.setSourceZero();
        if (s == m.nsteps - 1) {
            write_bin(dir + "/zero_alpha.bin", alpha.f.data(), (size_t)Nc);
            write_bin(dir + "/zero_uSource.bin", &uSource.f[0].v[0], 3 * (size_t)Nc);
            write_bin(dir + "/zero_uSourceDrag.bin", uSourceDrag.f.data(), (size_t)Nc);
            write_bin(dir + "/zero_uParticle.bin", &uParticle.f[0].v[0], 3 * (size_t)Nc);
        }
    }
}

}  // namespace

int main(int argc, char** argv) {
    MPI_Init(&argc, &argv);
    int rank, size;
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    MPI_Comm_size(MPI_COMM_WORLD, &size);
    if (argc < 2) { if (!rank) std::fprintf(stderr, "usage: ref_driver DIR\n"); MPI_Abort(MPI_COMM_WORLD, 2); }
    const std::string dir = argv[1];
    int rc = 0;
    try {
        Meta m = read_meta(dir);
        if (size != m.nYade + 1) { if (!rank) std::fprintf(stderr, "need %d ranks\n", m.nYade + 1); MPI_Abort(MPI_COMM_WORLD, 2); }
        const bool isFoam = (rank == size - 1);
        // the reference relies on a patched OpenFOAM Pstream for this split (FoamYade.C:4,21-22)
        MPI_Comm_split(MPI_COMM_WORLD, isFoam ? 1 : 2, rank, &Foam::PstreamGlobals::MPI_COMM_FOAM);
        if (isFoam) foam_rank(dir, m);
        else if (m.nYade == 1) fake_yade_serial(dir, m, size - 1);
        else fake_yade_parallel(dir, m, rank, size - 1);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
        MPI_Abort(MPI_COMM_WORLD, 1);
        rc = 1;
    }
    MPI_Finalize();
    return rc;
}
