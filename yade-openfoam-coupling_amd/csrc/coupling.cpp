// fy_ctx: MI355X-native replacement of Foam::FoamYade (FoamYade/FoamYade.{H,C}).  Host orchestration + wire protocol;
// all arithmetic runs in the HIP kernels of particle_kernels.hip.  There is no CPU compute path: without a HIP
// device fy_create fails with FY_ERR_NO_DEVICE.
#include "coupling.hpp"
#include "ldu.hpp"

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <thread>

#include <sys/stat.h>
#include <unistd.h>

#include <atomic>

namespace fy {

static std::atomic<int> g_no_pairs{-1};
bool pairs_disabled() {
    int v = g_no_pairs.load(std::memory_order_relaxed);
    if (v < 0) { const char* e = getenv("FOAMYADE_NO_PAIRS"); v = (e != nullptr && *e != 0 && strcmp(e, "0") != 0) ? 1 : 0; g_no_pairs.store(v, std::memory_order_relaxed); }
    return v != 0;
}

Options options() {
    auto on = [](const char* nm) { const char* e = getenv(nm); return e != nullptr && *e != 0 && strcmp(e, "0") != 0; };
    Options q{};
    q.explicit_tree = on("FOAMYADE_EXPLICIT_TREE");
    q.no_locate_lists = on("FOAMYADE_NO_LOCATE_LISTS");
    if (const char* d = getenv("FOAMYADE_TREE_CACHE_DIR")) q.tree_cache_dir = d;
    q.rebin_interval = 32;
    if (const char* e = getenv("FOAMYADE_REBIN_INTERVAL")) q.rebin_interval = std::max(1, atoi(e));
    q.no_halo_overlap = on("FOAMYADE_NO_HALO_OVERLAP");
    q.no_aux_comm = on("FOAMYADE_NO_AUX_COMM");
    q.halo_overlap = true;
    if (const char* e = getenv("FOAMYADE_HALO_OVERLAP")) q.halo_overlap = !(*e == 0 || strcmp(e, "0") == 0);
    q.no_deep_vcycle = on("FOAMYADE_NO_DEEP_VCYCLE");
    q.no_fused_corrector = on("FOAMYADE_NO_FUSED_CORRECTOR");
    q.faces_from_arrays = on("FOAMYADE_FACES_FROM_ARRAYS");
    q.no_pairs = on("FOAMYADE_NO_PAIRS");
    g_no_pairs.store(q.no_pairs ? 1 : 0, std::memory_order_relaxed);
    q.strip_blocks = -1;
    if (const char* e = getenv("FOAMYADE_STRIP_BLOCKS")) q.strip_blocks = atoi(e);
    return q;
}

std::string& last_error() {
    static thread_local std::string e;
    return e;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

// wire tags, FoamYade.H:60-66
static const int TAG_SZ_BUFF = 1003, TAG_GRID_BBOX = 1001, TAG_YADE_DATA = 1002, TAG_FORCE = 1005, TAG_SEARCH_RES = 1004,
                 TAG_FLUID_DT = 1050, TAG_YADE_DT = 1060;

#define FY_TR(expr)                                                                             \
    do {                                                                                        \
        if ((expr) != 0) return fail(FY_ERR_TRANSPORT, "transport call failed: %s", #expr);     \
    } while (0)

int Coupling::create(const fy_mesh_desc* m, const fy_field_ptrs* f, int gaussian_interp, const fy_transport* tr, int device_ordinal) {
    if (!m || !f || m->n_cells <= 0 || !m->centres || !m->volumes) return fail(FY_ERR_INVALID, "fy_create: bad mesh/fields");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(FY_ERR_NO_DEVICE, "no HIP device visible: libfoamyade_hip has no CPU path");
    if (device_ordinal < 0 || device_ordinal >= ndev) return fail(FY_ERR_INVALID, "device ordinal %d out of range (%d devices)", device_ordinal, ndev);
    device = device_ordinal;
    FY_HIP(hipSetDevice(device));
    if (ext_stream) { stream = ext_stream; owns_stream = false; }
    else { FY_HIP(hipStreamCreate(&stream)); owns_stream = true; }
    mesh = *m;
    mesh.centres = nullptr; mesh.volumes = nullptr; mesh.xf = mesh.yf = mesh.zf = nullptr;   // not retained
    n_cells = m->n_cells;
    gaussian = gaussian_interp != 0;
    structured = m->nx > 0;
    rectilinear = structured && m->xf && m->yf && m->zf;          // graded block: lattice INDEXING, but no lattice of centres
    if (!gaussian && !structured && !ldu_geo) return fail(FY_ERR_UNSUPPORTED, "point-force mode needs the structured block description, or a general mesh's face addressing (fy_ldu_solver): the findCell stand-ins");
    if (structured && (int64_t)m->nx * m->ny * m->nz != m->n_cells) return fail(FY_ERR_INVALID, "nx*ny*nz != n_cells");
    if (tr) { transport = *tr; has_transport = true; }

    // ---- getRankSize, FoamYade.C:18-53
    if (has_transport) {
        comm_sz_diff = std::abs(transport.world_size - transport.local_size);   // FoamYade.C:28
        serial_yade = (comm_sz_diff == 1);                                      // FoamYade.C:31
    } else {
        comm_sz_diff = 0; serial_yade = true;
    }
    // z-slabs with a parallel Yade: the halo / reverse-halo collectives are issued once per batch, and the number of Yade workers whose
    // particles intersect a rank's bounding box (FoamYade.C:114-155) differs from rank to rank -- so a slab keeps ONE batch per Yade worker,
    // in worker order, also for a worker that sent it nothing (recv_yade_intrs): every rank then walks the same W batches and the
    // collectives pair up, while the per-worker order of buildCellPartList / setCellVolFraction / calcHydroForce (FoamYade.C:612-628) is
    // the single domain's

    // ---- mshTree.build_tree(), FoamYade.C:33 (always, also in point-force mode: quirk Q6 kept for get_tree parity)
    {
        // implicit-coordinate nodes: legal only if EVERY centre equals origin + (i + 0.5) * dx bit for bit (checked here, so a
        // real OpenFOAM mesh whose centres come from pyramid decomposition simply keeps the explicit path)
        bool exact = structured && !rectilinear && !options().explicit_tree && m->nx <= 1024 && m->ny <= 1024 && m->nz <= 4096 &&
                     n_cells < (1 << 25);
        if (exact)
            for (int k = 0; k < m->nz && exact; ++k) for (int j = 0; j < m->ny && exact; ++j) for (int i = 0; i < m->nx; ++i) {
                const size_t c = (size_t)i + (size_t)m->nx * (j + (size_t)m->ny * k);
                const double x = m->origin[0] + ((double)i + 0.5) * m->dx, y = m->origin[1] + ((double)j + 0.5) * m->dx, z = m->origin[2] + ((double)k + 0.5) * m->dx;
                if (x != m->centres[3 * c] || y != m->centres[3 * c + 1] || z != m->centres[3 * c + 2]) { exact = false; break; }
            }
        // The tree is a pure function of the centres.  Several ranks of one node (z-slab mode: every rank holds the tree of the
        // GLOBAL block) may share it through a cache directory instead of each running the 25-level nth_element recursion:
        // FOAMYADE_TREE_CACHE_DIR=/dev/shm ; the first rank to create the lock file builds and publishes, the others wait.
        std::vector<int32_t> pre;      // preorder cell ids
        std::string cache;
        const std::string cache_dir = options().tree_cache_dir;
        if (exact && !cache_dir.empty()) {
            char nm[512];
            snprintf(nm, sizeof(nm), "%s/fy_tree_%dx%dx%d_%016llx.bin", cache_dir.c_str(), m->nx, m->ny, m->nz,
                     (unsigned long long)(std::hash<double>()(m->dx) ^ (std::hash<double>()(m->origin[0]) << 1) ^ (std::hash<double>()(m->origin[1]) << 2) ^ (std::hash<double>()(m->origin[2]) << 3)));
            cache = nm;
        }
        auto load_cache = [&]() -> bool {
            FILE* f = fopen(cache.c_str(), "rb");
            if (!f) return false;
            pre.resize((size_t)n_cells);
            bool ok = fread(pre.data(), sizeof(int32_t), (size_t)n_cells, f) == (size_t)n_cells && fgetc(f) == EOF;
            fclose(f);
            // a foreign or truncated file must not turn into out-of-range packed (i,j,k): every id is a cell of THIS block
            for (size_t q = 0; ok && q < pre.size(); ++q) ok = pre[q] >= 0 && pre[q] < n_cells;
            if (!ok) pre.clear();
            return ok;
        };
        bool have = false, hold_lock = false;
        const std::string lock = cache + ".lock";
        if (!cache.empty()) {
            have = load_cache();
            if (!have) {
                // Exclusive create: the winner builds and publishes, the others wait for the file.  A lock left behind by a crashed run must
                // not stall later runs for ever: a lock older than kStaleLock seconds (no k-d build of a block that fits the implicit tree
                // takes that long) is stale -- it is removed and the create retried, so that somebody publishes again; and nobody waits
                // longer than that either, after which this rank builds its own copy (and publishes it, tmp + rename: harmless if doubled).
                const double kStaleLock = 60.0;
                auto lock_age = [&]() -> double {
                    struct stat sb;
                    if (stat(lock.c_str(), &sb) != 0) return -1.0;           // gone
                    return difftime(time(nullptr), sb.st_mtime);
                };
                for (int attempt = 0; attempt < 2 && !have && !hold_lock; ++attempt) {
                    FILE* lf = fopen(lock.c_str(), "wx");
                    if (lf) { fclose(lf); hold_lock = true; break; }
                    const int max_spins = (int)(kStaleLock * 20);             // 50 ms each
                    for (int spin = 0; spin < max_spins && !have; ++spin) {
                        const double age = lock_age();
                        if (age < 0 || age > kStaleLock) break;               // the builder finished (or died long ago)
                        std::this_thread::sleep_for(std::chrono::milliseconds(50));
                        have = load_cache();
                    }
                    if (!have) have = load_cache();
                    if (!have && lock_age() > kStaleLock) remove(lock.c_str());   // stale: clear it and try to become the builder
                }
            }
        }
        const bool publish = !cache.empty() && !have;     // whoever had to build publishes (the lock only decides who builds FIRST)
        std::vector<KdNode> nodes;
        if (!have) {
            unsigned hw = std::thread::hardware_concurrency();
            build_kdtree_preorder(m->centres, n_cells, nodes, (int)std::min(hw ? hw : 1u, 8u));
            pre.resize((size_t)n_cells);
            for (size_t q = 0; q < nodes.size(); ++q) pre[q] = nodes[q].id;
            if (publish) {                                        // publish: complete file first, then the atomic rename
                char sfx[64];
                snprintf(sfx, sizeof(sfx), ".tmp.%ld.%zx", (long)getpid(), std::hash<std::thread::id>()(std::this_thread::get_id()));
                const std::string tmp = cache + sfx;
                bool ok = false;
                if (FILE* f = fopen(tmp.c_str(), "wb")) {
                    ok = fwrite(pre.data(), sizeof(int32_t), pre.size(), f) == pre.size();
                    ok = (fclose(f) == 0) && ok;
                }
                if (!ok || rename(tmp.c_str(), cache.c_str()) != 0) remove(tmp.c_str());
            }
        }
        if (hold_lock) remove(lock.c_str());                      // whether or not the build was published
        tree_levels = kdtree_levels(n_cells);
        if (exact) {
            std::vector<uint32_t> packed((size_t)n_cells);
            for (size_t q = 0; q < packed.size(); ++q) {
                const int id = pre[q];
                const int i = id % m->nx, j = (id / m->nx) % m->ny, k = id / (m->nx * m->ny);
                packed[q] = (uint32_t)i | ((uint32_t)j << 10) | ((uint32_t)k << 20);
            }
            FY_TRY(d_tree_packed.alloc_exact(packed.size()));
            FY_HIP(hipMemcpyAsync(d_tree_packed.p, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            implicit.ox = m->origin[0]; implicit.oy = m->origin[1]; implicit.oz = m->origin[2]; implicit.dx = m->dx;
            implicit.nx = m->nx; implicit.ny = m->ny; implicit.nz = m->nz;
            use_implicit = true;
            FY_HIP(hipStreamSynchronize(stream));
        } else {
            if (nodes.empty()) {                                 // (cache hit on a non-implicit mesh cannot happen: caching needs `exact`)
                build_kdtree_preorder(m->centres, n_cells, nodes, 1);
            }
            FY_TRY(d_tree.alloc_exact(nodes.size()));
            FY_HIP(hipMemcpyAsync(d_tree.p, nodes.data(), nodes.size() * sizeof(KdNode), hipMemcpyHostToDevice, stream));
            FY_HIP(hipStreamSynchronize(stream));
        }
        h_tree_pre.swap(pre);
    }
    v0 = m->volumes[0];
    vol_uniform = true;
    for (int32_t c = 1; c < n_cells && vol_uniform; ++c) vol_uniform = (m->volumes[c] == v0);
    n_field = slab.active ? (int64_t)slab.n_store : (int64_t)n_cells;
    FY_TRY(d_vol.alloc_exact((size_t)n_field));
    if (slab.active) {
        if (!structured || rectilinear) return fail(FY_ERR_UNSUPPORTED, "slab mode needs the uniform structured block description");
        FY_TRY(launch_fill_f64(stream, d_vol.p, (size_t)n_field, v0));        // uniform block
        FY_TRY(halo_tmp.alloc_exact(2 * (size_t)slab.gz * slab.plane * 4));   // both directions x (1 + 3) components of a grouped reverse sum
    } else {
        FY_HIP(hipMemcpyAsync(d_vol.p, m->volumes, (size_t)n_cells * sizeof(double), hipMemcpyHostToDevice, stream));
    }

    // ---- fields
    fields = *f;
    fields_on_host = (f->location == FY_MEM_HOST);
    if (slab.active && fields_on_host) return fail(FY_ERR_UNSUPPORTED, "slab mode works on device-resident fields only");
    if (fields_on_host) {
        FY_TRY(own_U.alloc_exact(3 * (size_t)n_cells)); FY_TRY(own_gradP.alloc_exact(3 * (size_t)n_cells));
        FY_TRY(own_vGrad.alloc_exact(9 * (size_t)n_cells)); FY_TRY(own_divT.alloc_exact(3 * (size_t)n_cells));
        FY_TRY(own_uSourceDrag.alloc_exact(n_cells)); FY_TRY(own_alpha.alloc_exact(n_cells));
        FY_TRY(own_uSource.alloc_exact(3 * (size_t)n_cells)); FY_TRY(own_uParticle.alloc_exact(3 * (size_t)n_cells));
        dU = own_U.p; dGradP = own_gradP.p; dVGrad = own_vGrad.p; dDivT = own_divT.p;
        dUSourceDrag = own_uSourceDrag.p; dAlpha = own_alpha.p; dUSource = own_uSource.p; dUParticle = own_uParticle.p;
        // the caller's current content of the two fields point mode never touches must survive the round trip
        FY_TRY(stage_mutable_in());
    } else {
        dU = f->U; dGradP = f->gradP; dVGrad = f->vGrad; dDivT = f->divT; dDdtU = f->ddtU;
        dUSourceDrag = f->uSourceDrag; dAlpha = f->alpha; dUSource = f->uSource; dUParticle = f->uParticle;
    }
    if (!dU || !dUSource || !dAlpha) return fail(FY_ERR_INVALID, "fy_create: U, uSource and alpha are required");
    if (gaussian && (!dGradP || !dDivT || !dUSourceDrag || !dUParticle)) return fail(FY_ERR_INVALID, "gaussian mode needs gradP, divT, uSourceDrag, uParticle");
    if (!gaussian && !dVGrad) return fail(FY_ERR_INVALID, "point-force mode needs vGrad");

    if (gaussian) {
        const size_t nf = (size_t)n_field;
        FY_TRY(d_pvol_acc.alloc_exact(nf)); FY_TRY(d_up_acc.alloc_exact(3 * nf)); FY_TRY(d_touched.alloc_exact(nf));
        FY_HIP(hipMemsetAsync(d_pvol_acc.p, 0, nf * sizeof(double), stream));
        FY_HIP(hipMemsetAsync(d_up_acc.p, 0, 3 * nf * sizeof(double), stream));
        FY_HIP(hipMemsetAsync(d_touched.p, 0, nf, stream));
        FY_TRY(d_cellrec.alloc_exact(8 * nf)); FY_TRY(d_drag_acc.alloc_exact(nf));
        FY_HIP(hipMemsetAsync(d_cellrec.p, 0, 8 * nf * sizeof(double), stream));
        FY_HIP(hipMemsetAsync(d_drag_acc.p, 0, nf * sizeof(double), stream));
        FY_HIP(hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking));
        FY_HIP(hipEventCreateWithFlags(&side.fork, hipEventDisableTiming));
        FY_HIP(hipEventCreateWithFlags(&side.join, hipEventDisableTiming));
    }

    // ---- binning grid (locality only)
    {
        const double h0 = (structured && !rectilinear) ? m->dx : std::cbrt(v0);
        const double h = 2.0 * h0;                                     // bin edge: 2 cells; locality only (4 / 2 / 1 / 0.5 measured flat)
        bins.ox = m->bbox_min[0]; bins.oy = m->bbox_min[1]; bins.oz = m->bbox_min[2];
        bins.inv_h = 1.0 / h;
        auto nb = [&](int a) { double e = (m->bbox_max[a] - m->bbox_min[a]) / h; int v = (int)std::ceil(e - 1e-9); return std::max(v, 1); };
        bins.nbx = nb(0); bins.nby = nb(1); bins.nbz = nb(2);
        bins.bx4 = (bins.nbx + 3) / 4; bins.by4 = (bins.nby + 3) / 4;
        const int bz4 = (bins.nbz + 3) / 4;
        const uint64_t nk = (uint64_t)bins.bx4 * bins.by4 * bz4 * 64ull;
        if (nk > (1ull << 30)) return fail(FY_ERR_UNSUPPORTED, "bin grid too large");
        bins.nkeys = (uint32_t)nk;
        FY_TRY(d_hist.alloc_exact(bins.nkeys));
        FY_TRY(d_tile_sums.alloc_exact((bins.nkeys + 2047u) / 2048u + 1));
    }
    if (rectilinear) {                                            // face planes for findCell (point-force mode)
        const double* src[3] = {m->xf, m->yf, m->zf};
        const int cnt[3] = {m->nx + 1, m->ny + 1, m->nz + 1};
        for (int a = 0; a < 3; ++a) {
            FY_TRY(d_faces[a].alloc_exact((size_t)cnt[a]));
            FY_HIP(hipMemcpyAsync(d_faces[a].p, src[a], (size_t)cnt[a] * sizeof(double), hipMemcpyHostToDevice, stream));
        }
        FY_HIP(hipStreamSynchronize(stream));
    }
    for (auto& t : timers) FY_TRY(t.init());
    FY_TRY(marks.init());
    rebin_interval = options().rebin_interval;
    if (has_transport || fields_on_host) {
        // The two copy streams get priorities of their own: the runtime multiplexes a process's streams onto a handful of hardware queues (4 by default),
        // and two streams that land on one queue wait for each other's packets -- round 5's trace of the drop-in leg: the last records' H2D copies sat behind
        // the answers' D2H, the batches that needed them 6 - 10 ms late.  Streams of different priority live on different queues.
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);                     // (lowest, highest: numerically prio_hi <= prio_lo)
        if (hipStreamCreateWithPriority(&copy_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) { (void)hipGetLastError(); FY_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)); }
        // results travel the other way on a stream of their own
        if (hipStreamCreateWithPriority(&copy_out_stream, hipStreamNonBlocking, prio_lo) != hipSuccess) { (void)hipGetLastError(); FY_HIP(hipStreamCreateWithFlags(&copy_out_stream, hipStreamNonBlocking)); }
    }
    // FY_MEM_HOST field arrays are NOT page-locked in place (round 2 offered hipHostRegister of the caller's arrays as an opt-in): the staging
    // copies run at the same 55 GB/s out of pageable memory -- the runtime pins on the fly -- and a registration that outlives the array's owner
    // (an interpreter freeing a numpy array before fy_destroy runs) leaves a page-locked range over freed pages, which is what the rare aborts
    // inside later pageable copies were.  No benefit, one lifetime hazard: removed.

    // ---- parallel Yade: yadeProcs + sendMeshBbox, FoamYade.C:35-45,77-111
    if (has_transport && !serial_yade) {
        if (transport.describe_block) {                 // a transport with wire helpers cuts the block among them before any box goes out
            if (slab.active || !(m->nx > 0 && m->dx > 0) || m->xf) return fail(FY_ERR_UNSUPPORTED, "a zero-copy wire transport needs the uniform block on one domain");
            const int32_t nn[3] = {m->nx, m->ny, m->nz};
            FY_TR(transport.describe_block(transport.user, m->origin, m->dx, nn));
        }
        double bbox[6] = {m->bbox_min[0], m->bbox_min[1], m->bbox_min[2], m->bbox_max[0], m->bbox_max[1], m->bbox_max[2]};
        if (slab.active) {       // every solver rank sends the box of ITS mesh (FoamYade.C:81-95 runs over the rank's own mesh.points()): this slab's planes
            bbox[2] = m->origin[2] + (double)slab.kglob0 * m->dx;
            bbox[5] = m->origin[2] + (double)(slab.kglob0 + slab.nz) * m->dx;
        }
        for (int rnk = 0; rnk != comm_sz_diff; ++rnk) FY_TR(transport.send(transport.user, bbox, 6, FY_T_DOUBLE, rnk, TAG_GRID_BBOX));
    }
    if (!has_transport || serial_yade) set_num_batches(1);     // FoamYade.C:46-50: one YadeProc{yRank = 0}

    FY_TRY(init_fields());
    FY_HIP(hipStreamSynchronize(stream));
    created = true;
    return FY_OK;
}

// per-cell start nodes of the walk and per-(cell, octant) candidate lists (implicit trees, Gaussian mode): built once per mesh, at
// create time (init_fields) so that no step pays for them; see k_build_locate_start / k_build_locate_lists
int Coupling::ensure_locate_tables(double maxdist) {
    if (!use_implicit) return FY_OK;
    if (!d_loc_start.p) {
        FY_TRY(d_loc_start.alloc_exact((size_t)n_cells));
        FY_TRY(launch_build_locate_start(stream, d_tree_packed.p, implicit, n_cells, maxdist, d_loc_start.p));
    }
    // candidate lists: 384 B per cell; valid while the rounding of a coordinate stays far below the builder's margins
    if (!options().no_locate_lists && !loc_lists_tried) {
        loc_lists_tried = true;
        const double ext = std::max({std::fabs(implicit.ox), std::fabs(implicit.oy), std::fabs(implicit.oz), std::fabs(implicit.ox + implicit.nx * implicit.dx),
                                     std::fabs(implicit.oy + implicit.ny * implicit.dx), std::fabs(implicit.oz + implicit.nz * implicit.dx)});
        // a slab only ever places particles whose cell lies in its own planes (SlabOwn): lists for those cells only
        const int64_t pl = (int64_t)implicit.nx * implicit.ny;
        loc_cell0 = slab.active ? (int32_t)((int64_t)slab.kglob0 * pl) : 0;
        loc_n_listed = slab.active ? (int32_t)((int64_t)slab.nz * pl) : n_cells;
        if (ext / implicit.dx <= 1e6 && d_loc_lists.alloc_exact((size_t)loc_n_listed * 8 * kLocateListLen) == FY_OK && d_loc_fb_n.alloc_exact(2) == FY_OK) {        // [0] the counter, [1] the last pass's count (kept when k_tile_caps clears [0])
            FY_TRY(launch_build_locate_lists(stream, d_tree_packed.p, implicit, n_cells, maxdist, d_loc_lists.p, loc_cell0, loc_n_listed));
        } else {
            d_loc_lists.release();               // not enough memory (or far-from-origin coordinates): the walk does all particles
        }
    }
    return FY_OK;
}

// FoamYade::initFields FoamYade.C:56-73
int Coupling::init_fields() {
    FY_TRY(launch_set_source_zero(stream, (int32_t)n_field, gaussian ? 1 : 0, dUSourceDrag, dAlpha, dUSource, dUParticle));
    FY_TRY(launch_fill_f64(stream, dAlpha, (size_t)n_field, 1.0));         // `alpha = 1.0` in both modes, FoamYade.C:68
    interp_range = 4 * std::pow(v0, 1.0 / 3.0);                            // FoamYade.C:69
    sigma_interp = interp_range * 0.42460;                                 // FoamYade.C:70
    interp_range_cu = std::pow(interp_range, 3.0);                         // FoamYade.C:71
    sigma_pi = 1.0 / (std::pow(2 * M_PI * sigma_interp * sigma_interp, 1.5));   // FoamYade.C:72
    if (fields_on_host) FY_TRY(stage_mutable_out());
    if (gaussian) FY_TRY(ensure_locate_tables((interp_range * interp_range) + (0.25 * interp_range * interp_range)));   // maxdist, meshTree.C:155
    return FY_OK;
}

int Coupling::stage_mutable_in() {
    const size_t n = (size_t)n_cells * sizeof(double);
    if (fields.uSourceDrag) FY_HIP(hipMemcpyAsync(own_uSourceDrag.p, fields.uSourceDrag, n, hipMemcpyHostToDevice, stream));
    if (fields.alpha) FY_HIP(hipMemcpyAsync(own_alpha.p, fields.alpha, n, hipMemcpyHostToDevice, stream));
    if (fields.uSource) FY_HIP(hipMemcpyAsync(own_uSource.p, fields.uSource, 3 * n, hipMemcpyHostToDevice, stream));
    if (fields.uParticle) FY_HIP(hipMemcpyAsync(own_uParticle.p, fields.uParticle, 3 * n, hipMemcpyHostToDevice, stream));
    return FY_OK;
}

int Coupling::stage_mutable_out() {
    const size_t n = (size_t)n_cells * sizeof(double);
    if (fields.alpha) FY_HIP(hipMemcpyAsync(fields.alpha, own_alpha.p, n, hipMemcpyDeviceToHost, stream));
    if (fields.uSource) FY_HIP(hipMemcpyAsync(fields.uSource, own_uSource.p, 3 * n, hipMemcpyDeviceToHost, stream));
    if (gaussian) {
        if (fields.uSourceDrag) FY_HIP(hipMemcpyAsync(fields.uSourceDrag, own_uSourceDrag.p, n, hipMemcpyDeviceToHost, stream));
        if (fields.uParticle) FY_HIP(hipMemcpyAsync(fields.uParticle, own_uParticle.p, 3 * n, hipMemcpyDeviceToHost, stream));
    }
    FY_HIP(hipStreamSynchronize(stream));
    return FY_OK;
}

int Coupling::stage_readonly_in() {
    const size_t n = (size_t)n_cells * sizeof(double);
    FY_HIP(hipMemcpyAsync(own_U.p, fields.U, 3 * n, hipMemcpyHostToDevice, stream));
    if (gaussian) {
        FY_HIP(hipMemcpyAsync(own_gradP.p, fields.gradP, 3 * n, hipMemcpyHostToDevice, stream));
        FY_HIP(hipMemcpyAsync(own_divT.p, fields.divT, 3 * n, hipMemcpyHostToDevice, stream));
        if (force_models & FY_FORCE_GAUSSIAN_TORQUE) FY_HIP(hipMemcpyAsync(own_vGrad.p, fields.vGrad, 9 * n, hipMemcpyHostToDevice, stream));
        if (force_models & FY_FORCE_ADDED_MASS) FY_HIP(hipMemcpyAsync(own_ddtU.p, fields.ddtU, 3 * n, hipMemcpyHostToDevice, stream));
    } else {
        FY_HIP(hipMemcpyAsync(own_vGrad.p, fields.vGrad, 9 * n, hipMemcpyHostToDevice, stream));
    }
    return FY_OK;
}

// Opt-in force models without a call site in the reference (see foamyade_hip.h)
int Coupling::set_force_models(unsigned flags) {
    if (!created) return fail(FY_ERR_INVALID, "fy_set_force_models before fy_create");
    if (flags & ~(FY_FORCE_ADDED_MASS | FY_FORCE_GAUSSIAN_TORQUE)) return fail(FY_ERR_INVALID, "fy_set_force_models: unknown flag");
    if (flags && !gaussian) return fail(FY_ERR_INVALID, "fy_set_force_models: Gaussian mode only (point mode always runs stokesDragTorque)");
    if ((flags & FY_FORCE_GAUSSIAN_TORQUE) && !(fields_on_host ? fields.vGrad : dVGrad)) return fail(FY_ERR_INVALID, "Gaussian torque needs vGrad");
    if ((flags & FY_FORCE_ADDED_MASS) && !(fields_on_host ? fields.ddtU : dDdtU)) return fail(FY_ERR_INVALID, "added mass needs ddtU");
    if (fields_on_host && (flags & FY_FORCE_ADDED_MASS) && !own_ddtU.p) {
        FY_TRY(own_ddtU.alloc_exact(3 * (size_t)n_cells));
        dDdtU = own_ddtU.p;
    }
    force_models = flags;
    return FY_OK;
}

// fibreCpl (FoamYade.H:102; FoamYade.C:131-136,161-165,189-198): 15 doubles per particle from Yade
int Coupling::set_fibre_coupling(int on) {
    if (!created) return fail(FY_ERR_INVALID, "fy_set_fibre_coupling before fy_create");
    if (slab.active) return fail(FY_ERR_INVALID, "fy_set_fibre_coupling: not with z-slabs (migration carries 10-double records)");
    fibre = on != 0;
    return FY_OK;
}

void Coupling::set_num_batches(int nb) {
    while ((int)batches.size() < nb) batches.emplace_back(new Batch());
    n_batches = nb;
}

int Coupling::ensure_batch(Batch& b, int64_t n) {
    b.n = n;
    if (n == 0) return FY_OK;
    const size_t cap = (size_t)n;
    FY_TRY(b.force.reserve(6 * cap));
    FY_TRY(b.found.reserve(cap));
    if (gaussian) {
        if (b.cap < cap) {
            const size_t c2 = cap + cap / 8 + 64;
            FY_TRY(b.soa.alloc_exact(7 * c2)); FY_TRY(b.orig.alloc_exact(c2)); FY_TRY(b.chain.alloc_exact(c2)); FY_TRY(b.scan_class.alloc_exact(c2)); FY_HIP(hipMemset(b.scan_class.p, 0, c2));
            FY_TRY(b.ids.alloc_exact((size_t)kMaxK * c2)); FY_TRY(b.w.alloc_exact((size_t)kMaxK * c2));
            FY_TRY(b.key.alloc_exact(c2)); FY_TRY(b.rank.alloc_exact(c2));
            if (tile_flush && structured) {
                const size_t nt = (size_t)tile_grid().n_tiles();
                for (int w = 0; w < 2; ++w) {
                    FY_TRY(b.tb_off[w].alloc_exact(nt)); FY_TRY(b.tb_cap[w].alloc_exact(nt)); FY_TRY(b.tb_fill[w].alloc_exact(nt));
                    FY_HIP(hipMemsetAsync(b.tb_off[w].p, 0, nt * sizeof(uint32_t), stream));
                    FY_HIP(hipMemsetAsync(b.tb_cap[w].p, 0, nt * sizeof(uint32_t), stream));
                    FY_HIP(hipMemsetAsync(b.tb_fill[w].p, 0, nt * sizeof(uint32_t), stream));
                }
                // entry pool: a workgroup's table holds ~1 entry per 10 pairs (measured 8 - 12), i.e. ~0.55 per particle at k = 5.5; two per
                // particle leave room for poorly aggregating clouds, and whatever does not fit simply goes out as atomics
                const size_t pool = 2 * c2 + 136 * nt;
                if (tile_cell.n < pool) { FY_TRY(tile_cell.alloc_exact(pool)); FY_TRY(tile_val.alloc_exact(4 * pool)); }
            }
            b.cap = c2;
            b.binned_n = -1;          // fresh arrays: the old placement is gone
            b.caps_ready = false;
        }
    } else {
        FY_TRY(b.incell.reserve(cap));
    }
    return FY_OK;
}

ParticleSoA Coupling::soa_of(Batch& b) {
    ParticleSoA p;
    double* s = b.soa.p;
    p.px = s; p.py = s + b.cap; p.pz = s + 2 * b.cap; p.vx = s + 3 * b.cap; p.vy = s + 4 * b.cap; p.vz = s + 5 * b.cap; p.rad = s + 6 * b.cap;
    p.orig = b.orig.p; p.chain_len = b.chain.p; p.scan_class = b.scan_class.p; p.ids = b.ids.p; p.w = b.w.p; p.cap = b.cap;
    return p;
}

TileGrid Coupling::tile_grid() const {
    TileGrid tg{};
    tg.nx = mesh.nx; tg.ny = mesh.ny;
    tg.nzs = (int)(n_field / ((int64_t)mesh.nx * mesh.ny));             // planes of the storage block (a slab: owned + ghost planes)
    tg.ntx = (tg.nx + kTileEdge - 1) / kTileEdge; tg.nty = (tg.ny + kTileEdge - 1) / kTileEdge; tg.ntz = (tg.nzs + kTileEdge - 1) / kTileEdge;
    return tg;
}

// which = 0: the void-fraction deposit's flush, 1: the momentum-source back-scatter's
TileBuckets Coupling::buckets_of(Batch& b, int which) {
    TileBuckets tb{};
    if (!tile_flush || !b.tb_off[which].p || !tile_cell.p) return tb;
    tb.cell = tile_cell.p; tb.val = tile_val.p;
    tb.off = b.tb_off[which].p; tb.cap = b.tb_cap[which].p; tb.fill = b.tb_fill[which].p;
    tb.pool = (uint32_t)std::min<size_t>(tile_cell.n, 0xfffffff0u);
    tb.tg = tile_grid();
    return tb;
}

int Coupling::set_particles_host(int bi, const double* rec, int64_t n) {
    if (bi < 0 || bi >= n_batches || n < 0 || (n > 0 && !rec)) return fail(FY_ERR_INVALID, "fy_set_particles_host: bad batch/arguments");
    Batch& b = *batches[bi];
    FY_TRY(b.rec_own.reserve(10 * (size_t)std::max<int64_t>(n, 1)));
    if (fibre) {
        FY_TRY(b.rec_wide.reserve(15 * (size_t)std::max<int64_t>(n, 1)));
        if (n) FY_HIP(hipMemcpyAsync(b.rec_wide.p, rec, 15 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
        FY_TRY(launch_fibre_repack(stream, b.rec_wide.p, b.rec_own.p, n));
    } else if (n) {
        FY_HIP(hipMemcpyAsync(b.rec_own.p, rec, 10 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    b.d_rec = b.rec_own.p;
    // the caller owns `rec` and may free it as soon as this returns: an asynchronous copy out of pageable memory is only staged, not
    // necessarily finished, when hipMemcpyAsync comes back
    if (n) FY_HIP(hipStreamSynchronize(stream));
    return ensure_batch(b, n);
}

int Coupling::ensure_batch_events(Batch& b) {
    if (b.events) return FY_OK;
    FY_TRY(b.t_in.init()); FY_TRY(b.t_out.init());
    FY_HIP(hipEventCreateWithFlags(&b.ev_ready, hipEventDisableTiming));
    b.events = true;
    return FY_OK;
}

// records that the transport has just delivered into the batch's pinned staging buffer: H2D on the copy stream (so that it overlaps the
// kernels of the batches received before, which are already running on the compute stream); the compute stream waits for the event
int Coupling::upload_batch(Batch& b, int64_t n, const double* src) {
    FY_TRY(ensure_batch_events(b));
    FY_TRY(b.rec_own.reserve(10 * (size_t)std::max<int64_t>(n, 1)));
    b.t_in.start(copy_stream);
    if (n) {
        const size_t len = (size_t)rec_len();
        double* dst = b.rec_own.p;
        if (fibre) { FY_TRY(b.rec_wide.reserve(len * (size_t)n)); dst = b.rec_wide.p; }
        FY_HIP(hipMemcpyAsync(dst, src ? src : b.h_rec.p, len * (size_t)n * sizeof(double), hipMemcpyHostToDevice, copy_stream));
        if (fibre) FY_TRY(launch_fibre_repack(copy_stream, b.rec_wide.p, b.rec_own.p, n));
        tm.bytes_in += (int64_t)len * n * (int64_t)sizeof(double);
    }
    b.t_in.stop(copy_stream);
    FY_HIP(hipStreamWaitEvent(stream, b.t_in.b, 0));
    b.d_rec = b.rec_own.p;
    return ensure_batch(b, n);
}

int Coupling::set_particles_device(int bi, const double* d_rec, int64_t n) {
    if (bi < 0 || bi >= n_batches || n < 0 || (n > 0 && !d_rec)) return fail(FY_ERR_INVALID, "fy_set_particles_device: bad batch/arguments");
    Batch& b = *batches[bi];
    if (fibre) {                                  // [n][15] in, the kernels' [n][10] gathered from it (k_fibre_repack)
        FY_TRY(b.rec_own.reserve(10 * (size_t)std::max<int64_t>(n, 1)));
        FY_TRY(launch_fibre_repack(stream, d_rec, b.rec_own.p, n));
        b.d_rec = b.rec_own.p;
    } else {
        b.d_rec = d_rec;
    }
    return ensure_batch(b, n);
}

// Particles that crossed into a neighbour's planes go there over the slab communicator (the same path as the halos: ncclSend/ncclRecv
// in one group under RCCL), the rest is compacted; a particle moves at most one slab per call.  Works on batch 0; afterwards the
// records live in library-owned storage (b.rec_own), whatever they were before.  Collective over the slabs.
int Coupling::migrate(int64_t* d_tags, int64_t tag_capacity, int64_t* n_out) {
    if (!created) return fail(FY_ERR_INVALID, "migrate before create");
    if (n_batches != 1) return fail(FY_ERR_INVALID, "particle migration works on a single batch (direct mode)");
    FY_HIP(hipSetDevice(device));
    Batch& b = *batches[0];
    if (!slab.active) { if (n_out) *n_out = b.n; return FY_OK; }
    const int64_t n = b.n;
    const size_t cap = 11 * (size_t)std::max<int64_t>(n, 1);
    FY_TRY(mig_stay.reserve(cap + cap / 4)); FY_TRY(mig_up.reserve(cap)); FY_TRY(mig_down.reserve(cap));
    FY_TRY(mig_counters.reserve(4)); FY_TRY(mig_cnt.reserve(4));
    FY_HIP(hipMemsetAsync(mig_counters.p, 0, 4 * sizeof(unsigned int), stream));
    FY_TRY(launch_migrate_pack(stream, b.d_rec, d_tags, n, slab_own(), mig_counters.p, mig_stay.p, mig_up.p, mig_down.p));
    unsigned int hc[4] = {0, 0, 0, 0};
    FY_HIP(hipMemcpyAsync(hc, mig_counters.p, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
    FY_HIP(hipStreamSynchronize(stream));
    const size_t n_stay = hc[0], n_up = slab.comm->has_up() ? hc[1] : 0, n_down = slab.comm->has_down() ? hc[2] : 0;
    if ((hc[1] && !slab.comm->has_up()) || (hc[2] && !slab.comm->has_down())) return fail(FY_ERR_INVALID, "migrate: ownership rule sent a particle past the end of the block");
    // sizes first ...
    const double sc[2] = {(double)n_up, (double)n_down};
    FY_HIP(hipMemcpyAsync(mig_cnt.p, sc, sizeof(sc), hipMemcpyHostToDevice, stream));
    FY_HIP(hipMemsetAsync(mig_cnt.p + 2, 0, 2 * sizeof(double), stream));
    FY_TRY(slab.comm->neighbour_exchange(stream, mig_cnt.p, mig_cnt.p + 2, mig_cnt.p + 1, mig_cnt.p + 3, 1));
    double rc[2] = {0, 0};
    FY_HIP(hipMemcpyAsync(rc, mig_cnt.p + 2, sizeof(rc), hipMemcpyDeviceToHost, stream));
    FY_HIP(hipStreamSynchronize(stream));
    const size_t from_down = (size_t)rc[0], from_up = (size_t)rc[1];
    const size_t n_new = n_stay + from_down + from_up;
    if (11 * n_new > mig_stay.n) {                          // grow, keeping the stayers
        DevBuf<double> bigger;
        FY_TRY(bigger.reserve(11 * n_new + 11 * n_new / 4));
        FY_HIP(hipMemcpyAsync(bigger.p, mig_stay.p, 11 * n_stay * sizeof(double), hipMemcpyDeviceToDevice, stream));
        FY_HIP(hipStreamSynchronize(stream));
        std::swap(bigger.p, mig_stay.p); std::swap(bigger.n, mig_stay.n);
    }
    // ... then the payloads, appended behind the stayers
    FY_TRY(slab.comm->neighbour_exchange_sized(stream, mig_up.p, 11 * n_up, mig_stay.p + 11 * n_stay, 11 * from_down, mig_down.p, 11 * n_down,
                                               mig_stay.p + 11 * (n_stay + from_down), 11 * from_up));
    if (d_tags && (int64_t)n_new > tag_capacity) return fail(FY_ERR_INVALID, "migrate: tag array holds %lld entries, %zu needed", (long long)tag_capacity, n_new);
    FY_TRY(b.rec_own.reserve(10 * std::max<size_t>(n_new, 1)));
    FY_TRY(launch_migrate_unpack(stream, mig_stay.p, (int64_t)n_new, b.rec_own.p, d_tags));
    b.d_rec = b.rec_own.p;
    FY_TRY(ensure_batch(b, (int64_t)n_new));
    b.binned_n = -1;                                       // a new population: bin it afresh
    b.torque_zero_buf = nullptr;
    FY_HIP(hipStreamSynchronize(stream));
    if (n_out) *n_out = (int64_t)n_new;
    return FY_OK;
}

int Coupling::ensure_found(Batch& b) {
    if (!b.found_stale) return FY_OK;
    FY_TRY(launch_found_from_chain(stream, soa_of(b), b.n, b.found.p));
    b.found_stale = false;
    return FY_OK;
}

// the device part of setParticleAction for one Yade proc (FoamYade.C:612-628 loop body)
int Coupling::run_batch(Batch& b) {
    if (b.n == 0 && !slab.active) return FY_OK;      // (in slab mode the halo exchanges are collective: every rank walks the same path)
    ForceParams fp{rhoF, nu, 1e-09, rhoP, delta_t, force_models, 0, vol_uniform ? v0 : 0.0};
    if (gaussian) {
        ParticleSoA p = soa_of(b);
        if (timing) marks.mark(0, stream);
        // The binned order only buys locality -- every result is independent of it -- and particles move a fraction of a cell per
        // coupling step, so the placement of an earlier step stays nearly as good: the counting sort runs every rebin_interval
        // steps (or when the particle count changes), in between the records are just gathered through the old permutation.
        // With the candidate lists k_locate_deposit fetches the records through the placement itself (and leaves the SoA copy behind for
        // the force pass): no separate gather pass
        const bool fused_gather = use_implicit && d_loc_lists.p != nullptr;
        const bool have_chains = b.chain_n == b.n && b.binned_n == b.n;       // last step's lengths, in the placement that is still there
        if (b.binned_n != b.n || b.bin_age >= rebin_interval || (have_chains && !b.ordered_by_chain)) {
            if (have_chains) {
                if (b.kwire.n < (size_t)b.n) FY_TRY(b.kwire.alloc_exact((size_t)b.cap));
                FY_TRY(launch_chain_by_wire(stream, b.orig.p, b.chain.p, b.scan_class.p, b.n, b.kwire.p));
            }
            FY_HIP(hipMemsetAsync(d_hist.p, 0, (size_t)bins.nkeys * sizeof(uint32_t), stream));
            FY_TRY(launch_bin_count(stream, b.d_rec, b.n, bins, b.key.p, b.rank.p, d_hist.p));
            FY_TRY(launch_exclusive_scan_u32(stream, d_hist.p, bins.nkeys, d_tile_sums.p));
            FY_TRY(launch_bin_scatter(stream, b.d_rec, b.n, b.key.p, b.rank.p, d_hist.p, d_tile_sums.p, p, false));
            if (have_chains) FY_TRY(launch_order_blocks_by_chain(stream, b.orig.p, b.kwire.p, b.n));
            if (!fused_gather) FY_TRY(launch_bin_gather(stream, b.d_rec, b.n, p));
            b.ordered_by_chain = have_chains;
            b.binned_n = b.n; b.bin_age = 1;
        } else {
            if (!fused_gather) FY_TRY(launch_bin_gather(stream, b.d_rec, b.n, p));
            ++b.bin_age;
        }
        GaussParams gp;
        gp.maxdist = (interp_range * interp_range) + (0.25 * interp_range * interp_range);   // meshTree.C:155
        gp.two_sigma2 = 2 * std::pow(sigma_interp, 2);                                         // FoamYade.C:308
        gp.range_cu = interp_range_cu; gp.sigma_pi = sigma_pi;
        FY_TRY(ensure_locate_tables(gp.maxdist));
        LocateLists ll{};
        if (use_implicit && d_loc_lists.p) {
            if (d_loc_fb.n < (size_t)b.n) FY_TRY(d_loc_fb.alloc_exact((size_t)b.n + (size_t)b.n / 8));
            ll = LocateLists{d_loc_lists.p, d_loc_fb.p, d_loc_fb_n.p, loc_cell0, loc_n_listed, 0, nullptr};
        } else if (!use_implicit) {                        // explicit tree: the list of the walks the short stack gives up (launch_locate)
            if (!d_loc_fb_n.p) FY_TRY(d_loc_fb_n.alloc_exact(1));
            if (d_loc_fb.n < (size_t)b.n) FY_TRY(d_loc_fb.alloc_exact((size_t)b.n + (size_t)b.n / 8));
            if (!d_loc_hwm.p) {
                FY_TRY(d_loc_hwm.alloc_exact(kLocDepthBins)); FY_TRY(h_loc_hwm.reserve(kLocDepthBins + 1));
                FY_HIP(hipMemsetAsync(d_loc_hwm.p, 0, kLocDepthBins * sizeof(unsigned int), stream));
                for (int q = 0; q <= kLocDepthBins; ++q) h_loc_hwm[q] = 0;
            }
            // the stack of this step: the depth that served 99.8 % of the walks sampled in the current window of steps, as last copied back (the first step runs the full
            // depth and measures); the others overflow into the second launch.  The histogram starts anew every 32 steps, so a cloud that moves on is followed; while a
            // window is still short the last window's depth stands; and if more than 1 % of the walks overflowed last step the stack grows by two entries at once
            static const int forced = [] { const char* e = getenv("FOAMYADE_LOCATE_STACK"); return e ? atoi(e) : -1; }();      // (experiments: 0 = always the full depth)
            unsigned long long total = 0, run = 0;
            unsigned int hist[kLocDepthBins];
            for (int q = 0; q < kLocDepthBins; ++q) { hist[q] = ((volatile unsigned int*)h_loc_hwm.p)[q]; total += hist[q]; }
            if (total >= 1000) {
                int d = 0;
                for (d = 0; d < kLocDepthBins - 1; ++d) { run += hist[d]; if (run * 1000 >= total * 998) break; }
                loc_stack_cap = std::max(d, 4);
            }
            const unsigned int overflowed = ((volatile unsigned int*)h_loc_hwm.p)[kLocDepthBins];
            if (loc_stack_used > 0 && loc_last_n > 0 && (unsigned long long)overflowed * 100 > (unsigned long long)loc_last_n) loc_stack_floor = loc_stack_used + 2;      // (kept until the window ends)
            loc_last_n = b.n;
            const int seen = loc_stack_cap > 0 ? std::max(loc_stack_cap, loc_stack_floor) : 0;
            loc_stack_used = seen;
            ll = LocateLists{nullptr, d_loc_fb.p, d_loc_fb_n.p, 0, 0, forced >= 0 ? forced : seen, d_loc_hwm.p};
        }
        const CellWindow cw{slab.active ? slab.base : 0, n_field};
        // the scatters' tables are flushed into per-tile buckets sized from the demand they counted in this batch's last step
        const TileBuckets tbD = buckets_of(b, 0), tbB = buckets_of(b, 1);
        // (the kernel also clears the lists' hand-over counter for the pass below: no 4-byte memset at the head of the phase)
        unsigned int* const fbz = (use_implicit && d_loc_lists.p && tbD.cell) ? d_loc_fb_n.p : nullptr;
        // (the counter is the coupling's, not the batch's: whichever batch's capacities kernel ran last on the side stream cleared it last -- wait for THAT one)
        if (ev_last_caps) { FY_HIP(hipStreamWaitEvent(stream, ev_last_caps, 0)); ev_last_caps = nullptr; }
        if (b.caps_ready && b.caps_key == (const void*)tbD.off) FY_HIP(hipStreamWaitEvent(stream, b.ev_caps, 0));      // formed at the end of the last call, beside the solver's sweeps
        else { FY_TRY(launch_tile_caps(stream, tbD, tbB, fbz)); }
        b.caps_ready = false;
        ll.fb_zeroed = fbz != nullptr;
        if (timing) marks.mark(1, stream);
        FY_TRY(launch_locate_deposit(stream, d_tree.p, use_implicit ? d_tree_packed.p : nullptr, implicit, n_cells, tree_levels, p, b.n, gp,
                                     use_implicit ? d_loc_start.p : nullptr, own_of(b), ll, cw, d_pvol_acc.p, d_up_acc.p, d_touched.p, tbD, side,
                                     fused_gather ? b.d_rec : nullptr));
        if (ll.depth_hwm) {                                // (read at the next step's launch, no wait)
            FY_HIP(hipMemcpyAsync(h_loc_hwm.p, d_loc_hwm.p, kLocDepthBins * sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
            FY_HIP(hipMemcpyAsync(h_loc_hwm.p + kLocDepthBins, d_loc_fb_n.p, sizeof(unsigned int), hipMemcpyDeviceToHost, stream));
            if (++loc_window >= 32) { loc_window = 0; loc_stack_floor = 0; FY_HIP(hipMemsetAsync(d_loc_hwm.p, 0, kLocDepthBins * sizeof(unsigned int), stream)); }
        }
        b.chain_n = b.n;                                   // (what the next placement's runs are ordered by)
        if (timing) marks.mark(2, stream);
        if (mid_hook && !mid_hook_done) {                  // the solver's field sweep: beside the side stream's walk (its own mark pair: it is taken off the phase it falls into)
            if (timing) marks.mark(6, stream);
            FY_TRY(run_mid_hook());
            if (timing) marks.mark(7, stream);
        }
        // The force pass gathers U, alpha and the Archimedes term from one packed record per cell.  U / gradP / divT do not change during the
        // call and the alpha slot follows k_finalize_cells, so the records are built once per call -- here, where the pack runs beside the
        // walk's leftovers on the side stream
        // z-slabs with the communicator's second channel (round 5): the exchanges of this phase run beside work that does not need them --
        //   the reverse halo of the deposit sums beside the pack of the cell records, alpha's ghost planes beside the finalisation of the interior cells
        // (the end planes, which are what the neighbours are sent, are finalised first).  Every cell goes through the same operations as in the serial
        // schedule, in the same order per cell: the same bits.
        const bool ovl = slab_overlap() && slab.nz > 2 * slab.gz;
        auto pack_records = [&]() -> int {
            if (cellrec_fresh && !cellrec_ghosts_stale) return FY_OK;
            if (slab.fields_event) { FY_HIP(hipStreamWaitEvent(stream, slab.fields_event, 0)); slab.fields_event = nullptr; }      // gradP / divT ghost planes have landed
            if (cellrec_fresh) {
                // a slab whose solver's pre-coupling sweep wrote the records of the OWNED cells (fy_solver, as on a single domain): what is left are the
                // ghost planes either side, whose gradP / divT came from the neighbours -- 2 gz planes instead of nz + 2 gz
                const size_t gc = (size_t)slab.gz * slab.plane, hi = (size_t)(slab.gz + slab.nz) * slab.plane;
                FY_TRY(launch_pack_cells(stream, (int64_t)gc, dU, dAlpha, dGradP, dDivT, d_vol.p, nu, rhoF, d_cellrec.p));
                FY_TRY(launch_pack_cells(stream, (int64_t)gc, dU + 3 * hi, dAlpha + hi, dGradP + 3 * hi, dDivT + 3 * hi, d_vol.p + hi, nu, rhoF, d_cellrec.p + 8 * hi));
            } else {
                FY_TRY(launch_pack_cells(stream, n_field, dU, dAlpha, dGradP, dDivT, d_vol.p, nu, rhoF, d_cellrec.p));
            }
            cellrec_fresh = true; cellrec_ghosts_stale = false;
            return FY_OK;
        };
        // z-slabs: the 8-plane tile layers that lie wholly inside the planes no reverse halo reaches ([2 gz, nz) in storage planes) are finished by the
        // reduction itself (k_tile_reduce<3> / <4>); k_finalize_cells / k_fold_sources keep the planes outside [z0, z1)
        int tk_lo = 0, tk_hi = 0;
        if (slab.active && tbD.cell) { tk_lo = (2 * slab.gz + 7) / 8; tk_hi = slab.nz / 8; if (tk_hi <= tk_lo) tk_lo = tk_hi = 0; }
        const bool layers = tk_hi > tk_lo && getenv("FOAMYADE_NO_TILE_LAYERS") == nullptr;
        const int64_t z0 = 8 * (int64_t)tk_lo * (int64_t)slab.plane, z1 = 8 * (int64_t)tk_hi * (int64_t)slab.plane;
        if (!ovl) FY_TRY(pack_records());
        if (side.stream && ll.lists) FY_HIP(hipStreamWaitEvent(stream, side.join, 0));
        if (tbD.cell && !slab.active) {      // single domain: the tile's sums are complete, setCellVolFraction rides on the reduction
            FY_TRY(launch_tile_reduce_finalize(stream, tbD, d_pvol_acc.p, d_up_acc.p, d_touched.p, d_vol.p, dAlpha, dUParticle, d_cellrec.p));
        } else if (!ovl) {
            auto finalize = [&](int64_t c0, int64_t c1) -> int {
                if (c1 <= c0) return FY_OK;
                return launch_finalize_cells(stream, (int32_t)(c1 - c0), d_vol.p + c0, d_pvol_acc.p + c0, d_up_acc.p + 3 * c0, d_touched.p + c0, dAlpha + c0, dUParticle + 3 * c0,
                                             d_cellrec.p + 8 * c0);
            };
            if (layers) FY_TRY(launch_tile_reduce_finalize_layers(stream, tbD, d_pvol_acc.p, d_up_acc.p, d_touched.p, d_vol.p, dAlpha, dUParticle, d_cellrec.p, tk_lo, tk_hi));
            else FY_TRY(launch_tile_reduce(stream, tbD, d_pvol_acc.p, d_up_acc.p, d_touched.p));
            if (slab.active) {      // contributions that landed in ghost planes belong to the neighbours: send them home and add
                FY_TRY(halo_reverse_add2(d_pvol_acc.p, 1, d_touched.p, d_up_acc.p, 3));
            }
            if (layers) { FY_TRY(finalize(0, z0)); FY_TRY(finalize(z1, (int64_t)n_field)); }
            else FY_TRY(finalize(0, (int64_t)n_field));
        } else {
            FY_TRY(slab_events());
            if (layers) FY_TRY(launch_tile_reduce_finalize_layers(stream, tbD, d_pvol_acc.p, d_up_acc.p, d_touched.p, d_vol.p, dAlpha, dUParticle, d_cellrec.p, tk_lo, tk_hi));
            else FY_TRY(launch_tile_reduce(stream, tbD, d_pvol_acc.p, d_up_acc.p, d_touched.p));
            FY_HIP(hipEventRecord(slab.ev_a, stream));
            FY_TRY(pack_records());                                            // (enqueued first: a host-synchronous back-end blocks in the exchange while this runs)
            FY_HIP(hipStreamWaitEvent(slab.aux, slab.ev_a, 0));
            FY_TRY(halo_reverse_start(d_pvol_acc.p, 1, d_up_acc.p, 3, slab.aux));
            FY_HIP(hipEventRecord(slab.ev_b, slab.aux));
            FY_HIP(hipStreamWaitEvent(stream, slab.ev_b, 0));
            FY_TRY(halo_reverse_finish(d_pvol_acc.p, 1, d_touched.p, d_up_acc.p, 3));
            // cells [lo0, lo1) and [hi0, hi1): the ghost planes and the gz owned planes next to them, either end; [lo1, hi0): the interior
            const int64_t pl = (int64_t)slab.plane, lo1 = 2 * (int64_t)slab.gz * pl, hi0 = (int64_t)slab.nz * pl, hi1 = (int64_t)n_field;
            auto finalize = [&](int64_t c0, int64_t c1) -> int {
                if (c1 <= c0) return FY_OK;
                return launch_finalize_cells(stream, (int32_t)(c1 - c0), d_vol.p + c0, d_pvol_acc.p + c0, d_up_acc.p + 3 * c0, d_touched.p + c0, dAlpha + c0, dUParticle + 3 * c0,
                                             d_cellrec.p + 8 * c0);
            };
            FY_TRY(finalize(0, lo1));
            FY_TRY(finalize(hi0, hi1));
            FY_HIP(hipEventRecord(slab.ev_a, stream));
            if (layers) { FY_TRY(finalize(lo1, z0)); FY_TRY(finalize(z1, hi0)); }      // (what the reduction did not finish itself) ...
            else FY_TRY(finalize(lo1, hi0));                                   // ... beside alpha's ghost planes
            FY_HIP(hipStreamWaitEvent(slab.aux, slab.ev_a, 0));
            FY_TRY(halo_fwd(dAlpha, 1, slab.gz, slab.aux));
            FY_HIP(hipEventRecord(slab.ev_b, slab.aux));
            FY_HIP(hipStreamWaitEvent(stream, slab.ev_b, 0));
        }
        if (slab.active) {      // the gathers below reach gz planes into the neighbours (alpha only: uParticle is applied per cell by its owner, k_fold_sources)
            if (!ovl) FY_TRY(halo_fwd(dAlpha, 1, slab.gz));
            const int64_t gcells = (int64_t)slab.gz * (int64_t)slab.plane;
            FY_TRY(launch_patch_rec_alpha(stream, 0, gcells, dAlpha, d_cellrec.p));
            FY_TRY(launch_patch_rec_alpha(stream, (int64_t)(slab.gz + slab.nz) * (int64_t)slab.plane, gcells, dAlpha, d_cellrec.p));
        }
        // Gaussian torque is identically zero unless the opt-in model is on (FoamYade.C:618): zero the records once per buffer,
        // afterwards the kernel stores only the force half of each (permuted, 48-byte) record
        if (force_models & FY_FORCE_GAUSSIAN_TORQUE) {
            b.torque_zero_buf = nullptr;
        } else {
            if (b.torque_zero_buf != b.force.p || b.torque_zero_n < b.n) {
                FY_HIP(hipMemsetAsync(b.force.p, 0, 6 * (size_t)b.n * sizeof(double), stream));
                b.torque_zero_buf = b.force.p; b.torque_zero_n = b.n;
            }
            fp.torque_prezeroed = 1;
        }
        if (timing) marks.mark(3, stream);
        FY_TRY(launch_force_gaussian(stream, p, b.n, fp, cw, d_vol.p, d_cellrec.p, dVGrad, dDdtU, b.d_rec, d_drag_acc.p, dUSource, b.force.p, tbB));
        if (timing) marks.mark(4, stream);
        b.found_stale = true;
        if (tbB.cell && !slab.active) {      // single domain: the fold rides on the reduction
            FY_TRY(launch_tile_reduce_fold(stream, tbB, d_drag_acc.p, dUSource, dUParticle, dUSourceDrag));
        } else {
            const bool layersB = layers && tbB.cell;
            if (layersB) FY_TRY(launch_tile_reduce_fold_layers(stream, tbB, d_drag_acc.p, dUSource, dUParticle, dUSourceDrag, tk_lo, tk_hi));
            else FY_TRY(launch_tile_reduce(stream, tbB, d_drag_acc.p, dUSource, nullptr));
            // FoamYade.C:385-386 per cell: uSourceDrag += D, uSource += uParticle * D (this batch's uParticle: the owner's, after its finalize)
            if (ovl) {
                // the reverse halo of the drag sums and of uSource beside the fold of the interior cells, which receive nothing from the neighbours
                FY_HIP(hipEventRecord(slab.ev_a, stream));
                const int64_t pl = (int64_t)slab.plane, lo1 = 2 * (int64_t)slab.gz * pl, hi0 = (int64_t)slab.nz * pl, hi1 = (int64_t)n_field;
                auto fold = [&](int64_t c0, int64_t c1) -> int {
                    if (c1 <= c0) return FY_OK;
                    return launch_fold_sources(stream, c1 - c0, d_drag_acc.p + c0, dUParticle + 3 * c0, dUSourceDrag + c0, dUSource + 3 * c0);
                };
                if (layersB) { FY_TRY(fold(lo1, z0)); FY_TRY(fold(z1, hi0)); }
                else FY_TRY(fold(lo1, hi0));
                FY_HIP(hipStreamWaitEvent(slab.aux, slab.ev_a, 0));
                FY_TRY(halo_reverse_start(d_drag_acc.p, 1, dUSource, 3, slab.aux));
                FY_HIP(hipEventRecord(slab.ev_b, slab.aux));
                FY_HIP(hipStreamWaitEvent(stream, slab.ev_b, 0));
                FY_TRY(halo_reverse_finish(d_drag_acc.p, 1, nullptr, dUSource, 3));
                FY_TRY(fold(0, lo1));
                FY_TRY(fold(hi0, hi1));
            } else {
                if (slab.active) FY_TRY(halo_reverse_add2(d_drag_acc.p, 1, nullptr, dUSource, 3));
                if (layersB) {
                    if (z0 > 0) FY_TRY(launch_fold_sources(stream, z0, d_drag_acc.p, dUParticle, dUSourceDrag, dUSource));
                    if (n_field > z1) FY_TRY(launch_fold_sources(stream, n_field - z1, d_drag_acc.p + z1, dUParticle + 3 * z1, dUSourceDrag + z1, dUSource + 3 * z1));
                } else {
                    FY_TRY(launch_fold_sources(stream, n_field, d_drag_acc.p, dUParticle, dUSourceDrag, dUSource));
                }
            }
        }
        if (timing) marks.mark(5, stream);
        // the NEXT call's bucket capacities (one workgroup's scan of the demand this call counted: 20 us that used to open the next particle phase) on the side
        // stream now, beside whatever the caller does next; a batch whose arrays are renewed in between forms them again (caps_key)
        static const bool early_caps = getenv("FOAMYADE_NO_EARLY_CAPS") == nullptr;      // (A/B switch)
        if (early_caps && side.stream && tbD.cell) {
            if (!b.ev_caps) FY_HIP(hipEventCreateWithFlags(&b.ev_caps, hipEventDisableTiming));
            FY_HIP(hipEventRecord(side.fork, stream));
            FY_HIP(hipStreamWaitEvent(side.stream, side.fork, 0));
            FY_TRY(launch_tile_caps(side.stream, tbD, tbB, fbz));
            FY_HIP(hipEventRecord(b.ev_caps, side.stream));
            ev_last_caps = b.ev_caps;
            b.caps_ready = true; b.caps_key = (const void*)tbD.off;
        }
    } else {
        BlockGeom g;
        for (int a = 0; a < 3; ++a) { g.bbmin[a] = mesh.bbox_min[a]; g.bbmax[a] = mesh.bbox_max[a]; }
        g.dx = mesh.dx; g.nx = mesh.nx; g.ny = mesh.ny; g.nz = mesh.nz;
        for (int a = 0; a < 3; ++a) g.faces[a] = rectilinear ? d_faces[a].p : nullptr;
        g.cell_of = nullptr;
        if (!structured) {
            // a general mesh (fy_ldu_solver): mesh.findCell (FoamYade.C:251) = the cell whose centre is nearest (the k-d tree), then a walk across
            // the faces the point lies outside of until it is inside every face of a cell (k_ldu_find_cell)
            FY_TRY(b.pos3.reserve(3 * (size_t)std::max<int64_t>(b.n, 1))); FY_TRY(b.cell_hint.reserve((size_t)std::max<int64_t>(b.n, 1)));
            FY_TRY(launch_ldu_positions(stream, b.d_rec, 10, b.n, b.pos3.p));
            FY_TRY(launch_nearest_cell(stream, d_tree.p, nullptr, implicit, n_cells, b.pos3.p, b.n, b.cell_hint.p));
            FY_TRY(launch_ldu_find_cell(stream, *ldu_geo, b.d_rec, 10, b.n, b.cell_hint.p, b.cell_hint.p));
            g.cell_of = b.cell_hint.p;
        }
        if (timing) marks.mark(3, stream);
        const CellWindow cw{slab.active ? slab.base : 0, n_field};
        FY_TRY(launch_point_force(stream, b.d_rec, b.n, g, fp, cw, d_vol.p, dU, dVGrad, dUSource, b.force.p, b.found.p, b.incell.p, own_of(b)));
        if (timing) marks.mark(4, stream);
    }
    return FY_OK;
}

// FoamYade::setParticleAction FoamYade.C:605-632
int Coupling::set_particle_action(double dt) {
    if (!created) return fail(FY_ERR_INVALID, "fy_set_particle_action before fy_create");
    FY_HIP(hipSetDevice(device));
    delta_t = dt;                                                          // FoamYade.C:607
    FY_TRY(collect_timings());                                            // (the previous call's, before its events are recorded again)
    tm = fy_particle_timings{};
    if (timing) timers[T_TOTAL].start(stream);

    // the force pass gathers U, alpha and the Archimedes term from one packed record per cell; U / gradP / divT do not change during
    // the call and alpha follows k_finalize_cells, so the records are built once here
    if (fields_on_host) { FY_TRY(stage_readonly_in()); FY_TRY(stage_mutable_in()); }
    mid_hook_done = false;
    if (!gaussian || n_batches == 0 || fields_on_host) FY_TRY(run_mid_hook());      // (nobody further down would, or the sweep's output is needed at once)
    cellrec_fresh = cellrec_external;       // (fy_solver's pre-coupling sweep may have written the records already)
    cellrec_ghosts_stale = cellrec_external && slab.active;      // (... of the owned cells: a slab's ghost planes are packed here, when the neighbours' fields have landed)
    cellrec_external = false;

    // ---- receive particles, and per Yade proc: locate + deposit + finalize + force (FoamYade.C:609, 612-628).  With a transport every
    // batch is processed as soon as its records have landed, so the kernels of batch q run while the host waits for batch q + 1 on the
    // wire and while its records cross PCIe on the copy stream; results are the reference's (one Yade proc after the other).
    wire_recv_ms = wire_send_ms = 0.0;
    if (has_transport) {
        if (!serial_yade) FY_TRY(recv_yade_intrs());                       // FoamYade.C:609 (runs the batches itself)
        else { FY_TRY(recv_serial()); FY_TRY(run_batch(*batches[0])); }    // FoamYade.C:173-184
    } else {
        for (int bi = 0; bi < n_batches; ++bi) FY_TRY(run_batch(*batches[bi]));
    }
    FY_TRY(run_mid_hook());                                                // (every batch was empty)
    if (slab.fields_event) { FY_HIP(hipStreamWaitEvent(stream, slab.fields_event, 0)); slab.fields_event = nullptr; }      // (no batch gathered anything)
    slab.tail_pending = false;
    if (slab.active && gaussian) {       // UcEqn.H:17-20 interpolates rAUc*uSource across the interface
        if (slab_overlap()) {
            // ... on the second channel: the solver waits for ev_tail where it first reads uSource across a slab face (after its stress and assembly sweeps)
            FY_TRY(slab_events());
            FY_HIP(hipEventRecord(slab.ev_a, stream));
            FY_HIP(hipStreamWaitEvent(slab.aux, slab.ev_a, 0));
            FY_TRY(halo_fwd(dUSource, 3, 1, slab.aux));
            FY_HIP(hipEventRecord(slab.ev_tail, slab.aux));
            slab.tail_pending = true;
        } else {
            FY_TRY(halo_fwd(dUSource, 3, 1));
        }
    }

    if (fields_on_host) FY_TRY(stage_mutable_out());

    // ---- send results
    results_pending = false;
    if (has_transport) {
        if (async_results && !serial_yade && wire_views) {
            for (int bi = 0; bi < n_batches; ++bi) FY_TRY(start_results_copy(*batches[bi]));
            results_pending = true;                                        // (the caller polls and finishes: Coupling::finish_results)
            // The fluid's dt goes out HERE, as the reference sends it from inside setParticleAction (FoamYade.C:630-631, 539-541), not behind the
            // fluid solve: Yade's master waits for it right after the forces and only then starts its DEM sub-steps, which are meant to run
            // beside the fluid solve.  (It is one eager double to the master, who takes no part in the particle exchange; the workers' answers
            // follow as they land.)  Only the blocking half of the handshake -- Yade's dt coming back -- stays with finish_results().
            FY_TRY(send_fluid_dt());
        } else {
            FY_TRY(send_results());                                        // FoamYade.C:228,239-243,487-535
            FY_TRY(exchange_dt());                                         // FoamYade.C:537-553
        }
    }
    async_results = false;
    if (timing) {
        timers[T_TOTAL].stop(stream);
        timings_pending = true;          // read when asked for (fy_get_particle_timings): reading here would drain the stream mid-step
        for (int bi = 0; bi < n_batches; ++bi) tm.n_particles += batches[bi]->n;
    }
    return FY_OK;
}

int Coupling::collect_timings() {
    if (!timings_pending) return FY_OK;
    timings_pending = false;
    FY_HIP(hipStreamSynchronize(stream));
    tm.bin = marks.ms(0, 1); tm.locate_deposit = marks.ms(1, 2); tm.finalize = marks.ms(2, 3) - marks.ms(6, 7); tm.force = marks.ms(3, 4); tm.fold = marks.ms(4, 5);
    tm.total = timers[T_TOTAL].ms();
    marks.clear();
    if (copy_stream) {
        FY_HIP(hipStreamSynchronize(copy_stream));
        FY_HIP(hipStreamSynchronize(copy_out_stream));
        if (getenv("FOAMYADE_WIRE_TRACE") && !piece_clocks.empty() && piece_clocks[0].armed) {
            // development trace: where the pieces' H2D copies and the batches' D2H copies sat on the device's clock, relative to the first piece's copy
            hipEvent_t z = piece_clocks[0].a;
            std::string line = "[wire] H2D pieces (start+dur ms):";
            for (auto& ck : piece_clocks) if (ck.armed) { float a = 0, d = 0; (void)hipEventElapsedTime(&a, z, ck.a); (void)hipEventElapsedTime(&d, ck.a, ck.b); char buf[48]; snprintf(buf, sizeof(buf), " %.1f+%.2f", a, d); line += buf; }
            line += "\n[wire] D2H batches (start+dur ms):";
            for (auto* b : batches) if (b->events && b->t_out.armed) { float a = 0, d = 0; (void)hipEventElapsedTime(&a, z, b->t_out.a); (void)hipEventElapsedTime(&d, b->t_out.a, b->t_out.b); char buf[48]; snprintf(buf, sizeof(buf), " %.1f+%.2f", a, d); line += buf; }
            std::fprintf(stderr, "%s\n", line.c_str());
        }
        for (auto* b : batches) if (b->events) { tm.copy_in += b->t_in.ms(); tm.copy_out += b->t_out.ms(); }
        for (auto& ck : piece_clocks) tm.copy_in += ck.ms();
    }
    tm.h2d = tm.copy_in; tm.d2h = tm.copy_out;              // the PCIe copies themselves (sum over the batches), on the copy stream
    tm.wire_recv = wire_recv_ms; tm.wire_send = wire_send_ms;
    return FY_OK;
}

namespace {
struct WallClock {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

// serial Yade: Bcast N, Bcast 10N doubles (FoamYade.C:176-183)
int Coupling::recv_serial() {
    int N = 0;
    FY_TR(transport.bcast_world(transport.user, &N, 1, FY_T_INT, 0));
    if (N < 0) return fail(FY_ERR_TRANSPORT, "negative particle count from Yade");
    Batch& b = *batches[0];
    b.yrank = 0;
    FY_TRY(b.h_rec.reserve((size_t)rec_len() * (size_t)std::max(N, 1)));
    // the reference broadcasts the record buffer unconditionally (FoamYade.C:181), also when it is empty: a collective has to be
    // matched by every rank, so the zero-count call is issued too
    double none = 0.0;
    const WallClock wc;
    FY_TR(transport.bcast_world(transport.user, N ? b.h_rec.data() : &none, rec_len() * N, FY_T_DOUBLE, 0));
    wire_recv_ms += wc.ms();
    return upload_batch(b, N);
}

// parallel Yade: counts from every worker, then records from the intersecting ones (FoamYade.C:114-155); each batch's kernels are
// enqueued as soon as its records are on their way to the device
int Coupling::recv_yade_intrs() {
    const int W = comm_sz_diff - 1;
    std::vector<int> counts((size_t)transport.local_size);
    std::vector<std::pair<int, int> > in_comm;       // (yRank, n)
    for (int w = 0; w < W; ++w) {
        const int yrank = w + 1;                                            // FoamYade.C:40: the Yade master sends nothing
        FY_TR(transport.recv(transport.user, counts.data(), transport.local_size, FY_T_INT, yrank, TAG_SZ_BUFF));
        // (a slab keeps an empty batch for a worker that sent it nothing: see create())
        if (counts[(size_t)transport.local_rank] > 0 || slab.active) in_comm.emplace_back(yrank, std::max(counts[(size_t)transport.local_rank], 0));
    }
    set_num_batches((int)in_comm.size());
    const bool views = transport.recv_view != nullptr && transport.send_reserve != nullptr && transport.send_commit != nullptr;
    wire_views = views;
    if (views) FY_TRY(lock_view_region());
    if (views && transport.recv_view_layout && transport.recv_view_next && !fibre) return recv_yade_pieces(in_comm);
    for (size_t q = 0; q < in_comm.size(); ++q) {
        Batch& b = *batches[q];
        b.yrank = in_comm[q].first;
        b.committed = false; b.out_found = nullptr; b.out_force = nullptr; b.pieces.n = 0; b.pieces.axis = 2;
        const int n = in_comm[q].second;
        const double* view = nullptr;
        if (n > 0 && views) {
            // the records stay where the transport (its wire helpers) put them: the PCIe copy starts from there
            const WallClock wc;
            const void* vp = nullptr;
            FY_TR(transport.recv_view(transport.user, &vp, rec_len() * n, FY_T_DOUBLE, b.yrank, TAG_YADE_DATA, &b.pieces));
            wire_recv_ms += wc.ms();
            view = static_cast<const double*>(vp);
            if (b.pieces.n < 0 || b.pieces.n > FY_WIRE_MAX_PIECES || b.pieces.axis < 0 || b.pieces.axis > 2) return fail(FY_ERR_TRANSPORT, "recv_view: bad pieces (%d across axis %d)", b.pieces.n, b.pieces.axis);
        } else {
            FY_TRY(b.h_rec.reserve((size_t)rec_len() * (size_t)std::max(n, 1)));
            if (n > 0) {                                                    // FoamYade.C:127-139: only intersecting workers send records
                const WallClock wc;
                FY_TR(transport.recv(transport.user, b.h_rec.data(), rec_len() * n, FY_T_DOUBLE, b.yrank, TAG_YADE_DATA));
                wire_recv_ms += wc.ms();
            }
        }
        FY_TRY(upload_batch(b, n, view));
        FY_TRY(run_batch(b));
        // a batch's forces and found flags are final as soon as ITS kernels are done (a later batch only changes the cell fields): start
        // their way back now, so that the copy runs under the next batch's receive, upload and kernels
        FY_TRY(start_results_copy(b));
        // ... and with a zero-copy wire the results of the batches before it go out as soon as they have landed (FoamYade.C:239-243, 504-507
        // send per Yade proc: nothing orders one proc's answers after another's): the helpers send them while the next records come in
        // (in ascending worker order, stopping at the first batch whose copy has not landed: the wire helpers take the answers worker by worker)
        if (views) FY_TRY(commit_landed(q));
    }
    return FY_OK;
}

// The same with a transport that reports the record messages piece by piece (fy_transport::recv_view_layout / recv_view_next: wire helpers take the
// workers' pieces in whatever order they are delivered -- taking them worker by worker makes the helpers queue behind one sending core): every
// piece's PCIe copy starts when it lands, so the copies hide under the receive; a worker's kernels start when its last piece is on its way to the
// device and every earlier worker's kernels are enqueued -- the reference's order of Yade procs (FoamYade.C:612-628), whatever order the wire kept.
int Coupling::recv_yade_pieces(const std::vector<std::pair<int, int> >& in_comm) {
    const size_t nb = in_comm.size();
    std::vector<const double*> view(nb, nullptr);
    std::vector<int> left(nb, 0);
    for (size_t q = 0; q < nb; ++q) {
        Batch& b = *batches[q];
        b.yrank = in_comm[q].first;
        b.committed = false; b.out_found = nullptr; b.out_force = nullptr; b.pieces.n = 0; b.pieces.axis = 2;
        const int n = in_comm[q].second;
        const void* vp = nullptr;
        FY_TR(transport.recv_view_layout(transport.user, &vp, rec_len() * n, FY_T_DOUBLE, b.yrank, TAG_YADE_DATA, &b.pieces));
        if (b.pieces.n < 0 || b.pieces.n > FY_WIRE_MAX_PIECES || b.pieces.axis < 0 || b.pieces.axis > 2) return fail(FY_ERR_TRANSPORT, "recv_view_layout: bad pieces");
        view[q] = static_cast<const double*>(vp);
        left[q] = std::max(b.pieces.n, 1);
        FY_TRY(ensure_batch_events(b));
        FY_TRY(b.rec_own.reserve(10 * (size_t)std::max(n, 1)));
    }
    // the pieces of different workers interleave on the copy stream: each copy gets an event pair of its own (tm.copy_in is their sum), and a
    // batch's end event b.t_in.b -- what its kernels wait for -- is recorded when its last piece is enqueued
    size_t pc_used = 0;
    size_t next = 0;
    auto run_ready = [&]() -> int {
        while (next < nb && left[next] == 0) {
            Batch& b = *batches[next];
            const int n = in_comm[next].second;
            FY_HIP(hipStreamWaitEvent(stream, b.t_in.b, 0));       // (recorded behind the batch's last piece, below)
            b.d_rec = b.rec_own.p;
            FY_TRY(ensure_batch(b, n));
            FY_TRY(run_batch(b));
            FY_TRY(start_results_copy(b));
            FY_TRY(commit_landed(next));
            ++next;
        }
        return FY_OK;
    };
    size_t outstanding = 0;
    for (size_t q = 0; q < nb; ++q) outstanding += (size_t)left[q];
    static const bool trace = getenv("FOAMYADE_WIRE_TRACE") != nullptr;
    const WallClock wall;
    double t_wait = 0, t_copy = 0, t_run = 0, first_piece = -1, last_piece = 0;
    while (outstanding > 0) {
        int src = -1, pc = -1;
        const WallClock wc;
        FY_TR(transport.recv_view_next(transport.user, &src, &pc));
        wire_recv_ms += wc.ms();
        t_wait += wc.ms();
        if (first_piece < 0) first_piece = wall.ms();
        last_piece = wall.ms();
        size_t q = 0;
        while (q < nb && in_comm[q].first != src) ++q;
        if (q == nb || pc < 0 || pc >= std::max(batches[q]->pieces.n, 1) || left[q] <= 0) return fail(FY_ERR_TRANSPORT, "recv_view_next: unexpected piece %d of worker %d", pc, src);
        Batch& b = *batches[q];
        const int64_t n = in_comm[q].second;
        const int64_t lo = b.pieces.n > 0 ? b.pieces.start[pc] : 0, hi = (b.pieces.n > 0 && pc + 1 < b.pieces.n) ? b.pieces.start[pc + 1] : n;
        if (hi > lo) {
            const WallClock wcc;
            if (pc_used == piece_clocks.size()) { piece_clocks.emplace_back(); FY_TRY(piece_clocks.back().init()); }
            EventTimer& ck = piece_clocks[pc_used++];
            ck.start(copy_stream);
            FY_HIP(hipMemcpyAsync(b.rec_own.p + 10 * lo, view[q] + 10 * lo, 10 * (size_t)(hi - lo) * sizeof(double), hipMemcpyHostToDevice, copy_stream));
            ck.stop(copy_stream);
            tm.bytes_in += 10 * (hi - lo) * (int64_t)sizeof(double);
            t_copy += wcc.ms();
        }
        --left[q]; --outstanding;
        if (left[q] == 0) {
            // the batch's end event goes behind its LAST piece right away, not when its turn to run comes: by then the answers of the batches before it are
            // being copied out, and streams that share a hardware queue would make this event -- and with it the batch's kernels -- wait for those copies
            // (round 5, kernel trace of the drop-in leg: every batch started one D2H copy, 1.4 ms, after the one before it had finished)
            b.t_in.armed = false;
            b.t_in.stop(copy_stream);
        }
        const WallClock wcr;
        FY_TRY(run_ready());
        t_run += wcr.ms();
    }
    const int rc = run_ready();
    if (trace) std::fprintf(stderr, "[wire] pieces: first at %.2f ms, last at %.2f ms; host: waiting %.2f, enqueueing copies %.2f, enqueueing batches %.2f, total %.2f ms\n",
                            first_piece, last_piece, t_wait, t_copy, t_run, wall.ms());
    return rc;
}

// page-lock the memory the transport's views point into (once per generation; a failed registration leaves the copies pageable, which the
// runtime stages at the same rate but synchronously)
int Coupling::lock_view_region() {
    if (!transport.view_region) return FY_OK;
    void* base = nullptr; size_t bytes = 0; uint64_t gen = 0;
    FY_TR(transport.view_region(transport.user, &base, &bytes, &gen));
    if (base == view_base && bytes == view_bytes && gen == view_generation) return FY_OK;
    if (view_locked) { (void)hipHostUnregister(view_base); view_locked = false; }
    view_base = base; view_bytes = bytes; view_generation = gen;
    if (base && bytes) {
        const hipError_t e = hipHostRegister(base, bytes, hipHostRegisterDefault);
        if (e == hipSuccess) view_locked = true;
        else {
            (void)hipGetLastError();
            static bool said = false;
            if (!said) { said = true; std::fprintf(stderr, "libfoamyade_hip: could not page-lock the transport's %zu-byte view region (%s): its copies run pageable (staged, synchronous)\n", bytes, hipGetErrorString(e)); }
        }
    }
    return FY_OK;
}

// hand one batch's results to the transport: found flags (FoamYade.C:239-243), then forces (FoamYade.C:504-507)
int Coupling::commit_results(Batch& b, bool wait) {
    if (b.committed || b.n == 0 || !b.out_started) return FY_OK;
    if (wait) FY_HIP(hipEventSynchronize(b.t_out.b));
    else {
        const hipError_t e = hipEventQuery(b.t_out.b);
        if (e == hipErrorNotReady) return FY_OK;
        if (e != hipSuccess) return fail(FY_ERR_HIP, "results copy failed: %s", hipGetErrorString(e));
    }
    const WallClock wc;
    FY_TR(transport.send_commit(transport.user, b.out_found, (int)b.n, FY_T_INT, b.yrank, TAG_SEARCH_RES));
    FY_TR(transport.send_commit(transport.user, b.out_force, 6 * (int)b.n, FY_T_DOUBLE, b.yrank, TAG_FORCE));
    wire_send_ms += wc.ms();
    b.committed = true;
    return FY_OK;
}

// early hand-over of the batches before `upto` whose results have landed -- strictly in batch order: both D2H events sit on one stream, so two
// batches can complete between two polls, and a helper (transport_mpi.cpp, W_TAG_RESULT) expects worker q's answers before worker q + 1's
int Coupling::commit_landed(size_t upto) {
    for (size_t e = 0; e < upto && e < (size_t)n_batches; ++e) {
        Batch& b = *batches[e];
        if (b.committed || b.n == 0 || !b.out_started) continue;       // (nothing to hand over for this worker: it does not hold the order up)
        FY_TRY(commit_results(b, false));
        if (!b.committed) break;                                        // not landed yet: the later ones wait for it
    }
    return FY_OK;
}

// D2H of one batch's found flags and forces on the outbound copy stream, behind the event that says its results are final
int Coupling::start_results_copy(Batch& b) {
    if (b.out_started) return FY_OK;
    FY_TRY(ensure_batch_events(b));
    int32_t* hf = nullptr; double* hF = nullptr;
    if (b.n && !serial_yade && wire_views) {
        void* v = nullptr;
        FY_TR(transport.send_reserve(transport.user, &v, (int)b.n, FY_T_INT, b.yrank, TAG_SEARCH_RES));
        hf = b.out_found = static_cast<int32_t*>(v);
        FY_TR(transport.send_reserve(transport.user, &v, 6 * (int)b.n, FY_T_DOUBLE, b.yrank, TAG_FORCE));
        hF = b.out_force = static_cast<double*>(v);
    } else {
        FY_TRY(b.h_found.reserve((size_t)std::max<int64_t>(b.n, 1))); FY_TRY(b.h_force.reserve(6 * (size_t)std::max<int64_t>(b.n, 1)));
        hf = b.h_found.data(); hF = b.h_force.data();
        b.out_found = nullptr; b.out_force = nullptr;
    }
    if (b.n) {
        FY_TRY(ensure_found(b));
        FY_HIP(hipEventRecord(b.ev_ready, stream));
        FY_HIP(hipStreamWaitEvent(copy_out_stream, b.ev_ready, 0));
    }
    b.t_out.start(copy_out_stream);
    if (b.n) {
        // FOAMYADE_D2H_STORES=1: into the transport's page-locked arena the answers go by kernel stores (launch_copy_out) instead of DMA copies, so that
        // the copy engine carries the incoming records only.  Measured (round 5, drop-in leg): no gain -- 47 GB/s by stores against 55 GB/s by DMA, and what
        // had delayed the copies was two streams sharing a hardware queue (see the copy streams' priorities), not the engine.  Kept as a switch.
        static const bool dma = getenv("FOAMYADE_D2H_STORES") == nullptr;
        void *df = nullptr, *dF = nullptr;
        const bool by_stores = !dma && wire_views && view_locked && hipHostGetDevicePointer(&df, hf, 0) == hipSuccess && hipHostGetDevicePointer(&dF, hF, 0) == hipSuccess;
        if (by_stores) {
            FY_TRY(launch_copy_out(copy_out_stream, df, b.found.p, (size_t)b.n * sizeof(int32_t)));
            FY_TRY(launch_copy_out(copy_out_stream, dF, b.force.p, 6 * (size_t)b.n * sizeof(double)));
        } else {
            (void)hipGetLastError();
            FY_HIP(hipMemcpyAsync(hf, b.found.p, (size_t)b.n * sizeof(int32_t), hipMemcpyDeviceToHost, copy_out_stream));
            FY_HIP(hipMemcpyAsync(hF, b.force.p, 6 * (size_t)b.n * sizeof(double), hipMemcpyDeviceToHost, copy_out_stream));
        }
        tm.bytes_out += b.n * (int64_t)(sizeof(int32_t) + 6 * sizeof(double));
    }
    b.t_out.stop(copy_out_stream);
    b.out_started = true;
    return FY_OK;
}

int Coupling::send_results() {
    // every batch's results are on their way (parallel Yade: since its kernels were enqueued; else from here); the host then hands batch
    // after batch to the transport as its copy lands (the copy of batch q + 1 overlaps the send of batch q)
    for (int bi = 0; bi < n_batches; ++bi) FY_TRY(start_results_copy(*batches[bi]));
    if (serial_yade) {
        Batch& b = *batches[0];
        FY_HIP(hipEventSynchronize(b.t_out.b));
        const WallClock wc;
        const int N = (int)b.n;
        send_ranks.assign((size_t)N, -1);
        for (int np = 0; np < N; ++np) {                                    // one MAX all-reduce PER PARTICLE, FoamYade.C:202,223,228
            int found = (b.h_found[(size_t)np] == 1) ? transport.world_rank : 0;
            FY_TR(transport.allreduce_world(transport.user, &found, &send_ranks[(size_t)np], 1, FY_T_INT, FY_OP_MAX));
        }
        if (gaussian) {                                                     // 6 SUM all-reduces per particle, FoamYade.C:510-516
            for (int np = 0; np < N; ++np)
                for (int j = 0; j < 6; ++j) {
                    double dummy = 0.0;
                    FY_TR(transport.allreduce_world(transport.user, &b.h_force[6 * (size_t)np + j], &dummy, 1, FY_T_DOUBLE, FY_OP_SUM));
                }
        } else {                                                            // owner sends 6 doubles to rank 0, FoamYade.C:519-531
            for (int np = 0; np < N; ++np)
                if (send_ranks[(size_t)np] == transport.world_rank)
                    FY_TR(transport.send(transport.user, &b.h_force[6 * (size_t)np], 6, FY_T_DOUBLE, 0, TAG_FORCE));
        }
        wire_send_ms += wc.ms();
    } else if (wire_views) {                                                // zero-copy wire: whatever has not gone out yet, in worker order
        for (int bi = 0; bi < n_batches; ++bi) FY_TRY(commit_results(*batches[bi], true));
    } else {
        for (int bi = 0; bi < n_batches; ++bi) {                            // FoamYade.C:239-243
            Batch& b = *batches[bi];
            if (b.n == 0) continue;                                         // (a slab's empty batch: that worker is not in inCommProcs)
            FY_HIP(hipEventSynchronize(b.t_out.b));
            const WallClock wc;
            FY_TR(transport.send(transport.user, b.h_found.data(), (int)b.n, FY_T_INT, b.yrank, TAG_SEARCH_RES));
            wire_send_ms += wc.ms();
        }
        for (int bi = 0; bi < n_batches; ++bi) {                            // FoamYade.C:504-507
            Batch& b = *batches[bi];
            if (b.n == 0) continue;
            const WallClock wc;
            FY_TR(transport.send(transport.user, b.h_force.data(), 6 * (int)b.n, FY_T_DOUBLE, b.yrank, TAG_FORCE));
            wire_send_ms += wc.ms();
        }
    }
    for (int bi = 0; bi < n_batches; ++bi) batches[bi]->out_started = false;
    return FY_OK;
}

int Coupling::finish_results() {
    if (!results_pending) return FY_OK;
    results_pending = false;
    FY_TRY(send_results());
    return recv_yade_dt();                                                 // (the fluid's dt went out from set_particle_action)
}

// FoamYade::exchangeDT FoamYade.C:537-553, in its two halves
int Coupling::exchange_dt() {
    FY_TRY(send_fluid_dt());
    return recv_yade_dt();
}
int Coupling::send_fluid_dt() {                                            // FoamYade.C:539-541
    if (transport.local_rank == 0) FY_TR(transport.send(transport.user, &delta_t, 1, FY_T_DOUBLE, 0, TAG_FLUID_DT));
    return FY_OK;
}
int Coupling::recv_yade_dt() {                                             // FoamYade.C:543-552
    if (!serial_yade) {
        if (transport.local_rank == 0) FY_TR(transport.recv(transport.user, &yade_dt, 1, FY_T_DOUBLE, 0, TAG_YADE_DT));
        FY_TR(transport.bcast_local(transport.user, &yade_dt, 1, FY_T_DOUBLE, 0));
    } else {
        FY_TR(transport.bcast_world(transport.user, &yade_dt, 1, FY_T_DOUBLE, 0));
    }
    return FY_OK;
}

// slab halos of a cell array with ncomp interleaved components: refresh w ghost planes per side from the owners
int Coupling::halo_fwd(double* f, int ncomp, int w, hipStream_t on) {
    const size_t P = slab.plane * (size_t)ncomp;
    return slab.comm->neighbour_exchange(on ? on : stream, f + (size_t)(slab.gz + slab.nz - w) * P, f + (size_t)(slab.gz - w) * P,
                                         f + (size_t)slab.gz * P, f + (size_t)(slab.gz + slab.nz) * P, (size_t)w * P);
}

int Coupling::slab_events() {
    if (slab.ev_a) return FY_OK;
    FY_HIP(hipEventCreateWithFlags(&slab.ev_a, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&slab.ev_b, hipEventDisableTiming));
    FY_HIP(hipEventCreateWithFlags(&slab.ev_tail, hipEventDisableTiming));
    return FY_OK;
}

// ... and the reverse: what this rank accumulated in its ghost planes goes to the owners, who add it; ghost planes are then cleared.
// f1 and f2 (nc1 + nc2 <= 4 components) travel in ONE grouped exchange into tmp buffers (start); the additions and the reset follow (finish)
int Coupling::halo_reverse_start(double* f1, int nc1, double* f2, int nc2, hipStream_t on) {
    struct It { double* f; int nc; } it[2] = {{f1, nc1}, {f2, nc2}};
    double* tmp = halo_tmp.p;
    slab.comm->group_begin();
    for (It& t : it) {
        const size_t P = slab.plane * (size_t)t.nc, cnt = (size_t)slab.gz * P;
        FY_TRY(slab.comm->neighbour_exchange(on, t.f + (size_t)(slab.gz + slab.nz) * P, tmp, t.f, tmp + cnt, cnt));
        tmp += 2 * cnt;
    }
    return slab.comm->group_end(on);
}
int Coupling::halo_reverse_finish(double* f1, int nc1, unsigned char* mark1, double* f2, int nc2) {
    struct It { double* f; int nc; unsigned char* mark; } it[2] = {{f1, nc1, mark1}, {f2, nc2, nullptr}};
    double* tmp = halo_tmp.p;
    for (It& t : it) {
        const size_t P = slab.plane * (size_t)t.nc, cnt = (size_t)slab.gz * P;
        double* tmp_a = tmp; double* tmp_b = tmp + cnt; tmp += 2 * cnt;
        double* own_lo = t.f + (size_t)slab.gz * P;
        double* own_hi = t.f + (size_t)slab.nz * P;                  // the last gz owned planes
        if (slab.comm->has_down()) FY_TRY(launch_add_mark(stream, own_lo, tmp_a, cnt, t.mark ? t.mark + (size_t)slab.gz * slab.plane : nullptr));
        if (slab.comm->has_up()) FY_TRY(launch_add_mark(stream, own_hi, tmp_b, cnt, t.mark ? t.mark + (size_t)slab.nz * slab.plane : nullptr));
        FY_HIP(hipMemsetAsync(t.f, 0, cnt * sizeof(double), stream));
        FY_HIP(hipMemsetAsync(t.f + (size_t)(slab.gz + slab.nz) * P, 0, cnt * sizeof(double), stream));
    }
    return FY_OK;
}

// FoamYade::setSourceZero FoamYade.C:556-566 + clearInCommProcs FoamYade.C:568-580
int Coupling::set_source_zero() {
    if (!created) return fail(FY_ERR_INVALID, "fy_set_source_zero before fy_create");
    FY_HIP(hipSetDevice(device));
    FY_TRY(launch_set_source_zero(stream, (int32_t)n_field, gaussian ? 1 : 0, dUSourceDrag, dAlpha, dUSource, dUParticle));
    if (fields_on_host) FY_TRY(stage_mutable_out());
    if (has_transport && !serial_yade) n_batches = 0;                       // FoamYade.C:577
    return FY_OK;
}

int Coupling::get_forces_host(int bi, double* out) {
    if (bi < 0 || bi >= (int)batches.size() || !out) return fail(FY_ERR_INVALID, "fy_get_forces_host: bad batch");
    Batch& b = *batches[bi];
    if (b.n) FY_HIP(hipMemcpyAsync(out, b.force.p, 6 * (size_t)b.n * sizeof(double), hipMemcpyDeviceToHost, stream));
    FY_HIP(hipStreamSynchronize(stream));
    return FY_OK;
}

int Coupling::get_found_host(int bi, int32_t* out) {
    if (bi < 0 || bi >= (int)batches.size() || !out) return fail(FY_ERR_INVALID, "fy_get_found_host: bad batch");
    Batch& b = *batches[bi];
    FY_TRY(ensure_found(b));
    if (b.n) FY_HIP(hipMemcpyAsync(out, b.found.p, (size_t)b.n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    FY_HIP(hipStreamSynchronize(stream));
    return FY_OK;
}

int Coupling::get_stencils_host(int bi, int32_t* k, int32_t* ids, double* w, int32_t* chain) {
    if (bi < 0 || bi >= (int)batches.size() || !k || !ids || !w || !chain) return fail(FY_ERR_INVALID, "fy_get_stencils_host: bad arguments");
    Batch& b = *batches[bi];
    if (b.n == 0) return FY_OK;
    const size_t n = (size_t)b.n;
    DevBuf<int32_t> dk, dids, dchain; DevBuf<double> dw;
    FY_TRY(dk.alloc_exact(n)); FY_TRY(dids.alloc_exact(kMaxK * n)); FY_TRY(dchain.alloc_exact(n)); FY_TRY(dw.alloc_exact(kMaxK * n));
    if (gaussian) {
        FY_TRY(launch_unpack_stencils(stream, soa_of(b), b.n, dk.p, dids.p, dw.p, dchain.p));
        FY_HIP(hipMemcpyAsync(k, dk.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        FY_HIP(hipMemcpyAsync(ids, dids.p, kMaxK * n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        FY_HIP(hipMemcpyAsync(w, dw.p, kMaxK * n * sizeof(double), hipMemcpyDeviceToHost, stream));
        FY_HIP(hipMemcpyAsync(chain, dchain.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        FY_HIP(hipStreamSynchronize(stream));
    } else {
        std::vector<int32_t> incell(n);
        FY_HIP(hipMemcpyAsync(incell.data(), b.incell.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        FY_HIP(hipStreamSynchronize(stream));
        for (size_t i = 0; i < n; ++i) {
            for (int q = 0; q < kMaxK; ++q) { ids[i * kMaxK + q] = -1; w[i * kMaxK + q] = 0.0; }
            k[i] = incell[i] >= 0 ? 1 : 0; chain[i] = k[i];
            if (k[i]) ids[i * kMaxK] = incell[i];
        }
    }
    return FY_OK;
}

int Coupling::get_tree_preorder(int32_t* out) {
    if (!out) return fail(FY_ERR_INVALID, "null output");
    std::memcpy(out, h_tree_pre.data(), h_tree_pre.size() * sizeof(int32_t));
    return FY_OK;
}

int Coupling::field_by_name(const char* name, double** p, size_t* count) {
    const std::string s = name ? name : "";
    const size_t n = (size_t)n_field;
    if (s == "alpha") { *p = dAlpha; *count = n; }
    else if (s == "uSourceDrag") { *p = dUSourceDrag; *count = n; }
    else if (s == "uSource") { *p = dUSource; *count = 3 * n; }
    else if (s == "uParticle") { *p = dUParticle; *count = 3 * n; }
    else if (s == "U") { *p = const_cast<double*>(dU); *count = 3 * n; }
    else if (s == "gradP") { *p = const_cast<double*>(dGradP); *count = 3 * n; }
    else if (s == "divT") { *p = const_cast<double*>(dDivT); *count = 3 * n; }
    else if (s == "ddtU") { *p = const_cast<double*>(dDdtU); *count = 3 * n; }
    else if (s == "vGrad") { *p = const_cast<double*>(dVGrad); *count = 9 * n; }
    else return fail(FY_ERR_INVALID, "unknown field '%s'", s.c_str());
    if (!*p) return fail(FY_ERR_INVALID, "field '%s' was not supplied", s.c_str());
    return FY_OK;
}

int Coupling::read_field_host(const char* name, double* out) {
    double* p; size_t cnt;
    FY_TRY(field_by_name(name, &p, &cnt));
    FY_HIP(hipMemcpyAsync(out, p, cnt * sizeof(double), hipMemcpyDeviceToHost, stream));
    FY_HIP(hipStreamSynchronize(stream));
    return FY_OK;
}

int Coupling::write_field_host(const char* name, const double* in) {
    double* p; size_t cnt;
    FY_TRY(field_by_name(name, &p, &cnt));
    FY_HIP(hipMemcpyAsync(p, in, cnt * sizeof(double), hipMemcpyHostToDevice, stream));
    FY_HIP(hipStreamSynchronize(stream));
    return FY_OK;
}

Coupling::~Coupling() {
    if (device >= 0) (void)hipSetDevice(device);
    if (view_locked) (void)hipHostUnregister(view_base);
    if (slab.ev_a) (void)hipEventDestroy(slab.ev_a);
    if (slab.ev_b) (void)hipEventDestroy(slab.ev_b);
    if (slab.ev_tail) (void)hipEventDestroy(slab.ev_tail);
    for (auto& ck : piece_clocks) ck.destroy();
    for (auto& t : timers) t.destroy();
    marks.destroy();
    for (auto* b : batches) delete b;
    if (side.fork) (void)hipEventDestroy(side.fork);
    if (side.join) (void)hipEventDestroy(side.join);
    if (side.stream) (void)hipStreamDestroy(side.stream);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (copy_out_stream) (void)hipStreamDestroy(copy_out_stream);
    if (owns_stream && stream) (void)hipStreamDestroy(stream);
}

}  // namespace fy

// ================================================================================================ C ABI
using fy::Coupling;

extern "C" {

const char* fy_last_error(void) { return fy::last_error().c_str(); }
int fy_abi_version(void) { return FY_ABI_VERSION; }
int fy_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fy_create(const fy_mesh_desc* mesh, const fy_field_ptrs* fields, int gaussian_interp, const fy_transport* transport,
              int device_ordinal, fy_ctx** out) {
    if (!out) return fy::fail(FY_ERR_INVALID, "fy_create: null out");
    *out = nullptr;
    fy_ctx* c = new (std::nothrow) fy_ctx();
    if (!c) return fy::fail(FY_ERR_INVALID, "out of host memory");
    int rc = c->c.create(mesh, fields, gaussian_interp, transport, device_ordinal);
    if (rc != FY_OK) { delete c; return rc; }
    *out = c;
    return FY_OK;
}

#define FY_CTX(c) if (!(c)) return fy::fail(FY_ERR_INVALID, "null fy_ctx")

int fy_set_scalar_properties(fy_ctx* c, double rhoP, double rhoF, double nu) { FY_CTX(c); c->c.rhoP = rhoP; c->c.rhoF = rhoF; c->c.nu = nu; return FY_OK; }
int fy_set_force_models(fy_ctx* c, unsigned flags) { FY_CTX(c); return c->c.set_force_models(flags); }
int fy_set_fibre_coupling(fy_ctx* c, int on) { FY_CTX(c); return c->c.set_fibre_coupling(on); }
int fy_set_particle_action(fy_ctx* c, double dt) { FY_CTX(c); return c->c.set_particle_action(dt); }
int fy_set_source_zero(fy_ctx* c) { FY_CTX(c); return c->c.set_source_zero(); }
int fy_finalize_run(fy_ctx* c, int* value_out) {
    FY_CTX(c);
    int value = -1;                                                         // FoamYade.C:596
    if (c->c.has_transport && c->c.transport.bcast_world(c->c.transport.user, &value, 1, FY_T_INT, 0) != 0)
        return fy::fail(FY_ERR_TRANSPORT, "fy_finalize_run: broadcast failed");
    if (value_out) *value_out = value;
    return FY_OK;
}
int fy_destroy(fy_ctx* c) { delete c; return FY_OK; }
int fy_set_num_batches(fy_ctx* c, int nb) {
    FY_CTX(c);
    if (nb < 0) return fy::fail(FY_ERR_INVALID, "negative batch count");
    c->c.set_num_batches(nb);
    return FY_OK;
}
int fy_set_particles_host(fy_ctx* c, int batch, const double* rec, int64_t n) { FY_CTX(c); return c->c.set_particles_host(batch, rec, n); }
int fy_set_particles_device(fy_ctx* c, int batch, const double* rec, int64_t n) { FY_CTX(c); return c->c.set_particles_device(batch, rec, n); }
int fy_get_forces_host(fy_ctx* c, int batch, double* out) { FY_CTX(c); return c->c.get_forces_host(batch, out); }
int fy_migrate_particles(fy_ctx* c, int64_t* d_tags, int64_t tag_capacity, int64_t* n_local_out) { FY_CTX(c); return c->c.migrate(d_tags, tag_capacity, n_local_out); }
int fy_get_particles_host(fy_ctx* c, int batch, double* out, int64_t* n_out) {
    FY_CTX(c);
    if (batch < 0 || batch >= (int)c->c.batches.size()) return fy::fail(FY_ERR_INVALID, "fy_get_particles_host: bad batch");
    fy::Batch& b = *c->c.batches[batch];
    if (n_out) *n_out = b.n;
    if (out && b.n) {
        FY_HIP(hipSetDevice(c->c.device));
        FY_HIP(hipMemcpyAsync(out, b.d_rec, 10 * (size_t)b.n * sizeof(double), hipMemcpyDeviceToHost, c->c.stream));
        FY_HIP(hipStreamSynchronize(c->c.stream));
    }
    return FY_OK;
}
int fy_get_found_host(fy_ctx* c, int batch, int32_t* out) { FY_CTX(c); return c->c.get_found_host(batch, out); }
const double* fy_forces_device(fy_ctx* c, int batch) {
    if (!c || batch < 0 || batch >= (int)c->c.batches.size()) return nullptr;
    return c->c.batches[(size_t)batch]->force.p;
}
int fy_get_stencils_host(fy_ctx* c, int batch, int32_t* k, int32_t* ids, double* w, int32_t* chain) { FY_CTX(c); return c->c.get_stencils_host(batch, k, ids, w, chain); }
int fy_nearest_cells_host(fy_ctx* c, const double* pos, int64_t n, int32_t* out) {
    FY_CTX(c);
    if (n < 0 || (n > 0 && (!pos || !out))) return fy::fail(FY_ERR_INVALID, "fy_nearest_cells_host: bad arguments");
    if (n == 0) return FY_OK;
    Coupling& C = c->c;
    FY_HIP(hipSetDevice(C.device));
    fy::DevBuf<double> dp; fy::DevBuf<int32_t> dout;
    FY_TRY(dp.alloc_exact(3 * (size_t)n)); FY_TRY(dout.alloc_exact((size_t)n));
    FY_HIP(hipMemcpyAsync(dp.p, pos, 3 * (size_t)n * sizeof(double), hipMemcpyHostToDevice, C.stream));
    FY_TRY(fy::launch_nearest_cell(C.stream, C.d_tree.p, C.use_implicit ? C.d_tree_packed.p : nullptr, C.implicit, C.n_cells, dp.p, n, dout.p));
    FY_HIP(hipMemcpyAsync(out, dout.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, C.stream));
    FY_HIP(hipStreamSynchronize(C.stream));
    return FY_OK;
}
int fy_get_tree_preorder(fy_ctx* c, int32_t* out) { FY_CTX(c); return c->c.get_tree_preorder(out); }
int fy_read_field_host(fy_ctx* c, const char* name, double* out) { FY_CTX(c); return c->c.read_field_host(name, out); }
int fy_write_field_host(fy_ctx* c, const char* name, const double* in) { FY_CTX(c); return c->c.write_field_host(name, in); }
double fy_yade_dt(fy_ctx* c) { return c ? c->c.yade_dt : 0.0; }
double fy_interp_range(fy_ctx* c) { return c ? c->c.interp_range : 0.0; }
long long fy_locate_walk_count(fy_ctx* c) {
    if (!c || !c->c.d_loc_lists.p || !c->c.d_loc_fb_n.p) return -1;
    unsigned int n = 0;
    // (the end-of-batch capacities kernel, if it has run since, moved the count to slot 1 and cleared the counter)
    unsigned int both[2] = {0, 0};
    if (hipStreamSynchronize(c->c.stream) != hipSuccess || (c->c.side.stream && hipStreamSynchronize(c->c.side.stream) != hipSuccess) ||
        hipMemcpy(both, c->c.d_loc_fb_n.p, sizeof(both), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    n = c->c.ev_last_caps ? both[1] : both[0];
    return (long long)n;
}
int fy_locate_stack_depth(fy_ctx* c) {
    if (!c || !c->c.d_loc_hwm.p) return -1;
    return c->c.loc_stack_used;
}
int fy_get_particle_timings(fy_ctx* c, fy_particle_timings* out) {
    FY_CTX(c);
    if (!out) return fy::fail(FY_ERR_INVALID, "null out");
    FY_TRY(c->c.collect_timings());
    *out = c->c.tm;
    return FY_OK;
}
int fy_enable_timing(fy_ctx* c, int on) { FY_CTX(c); c->c.timing = on != 0; return FY_OK; }

}  // extern "C"
