// Launchers of the particle-half HIP kernels (bin -> locate+weights+deposit -> finalize -> force+scatter).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kdtree.hpp"

namespace fy {

constexpr int kMaxK = 16;          // stencil slots kept per particle (reference nominal bound 12, observed max 14)

// spatial binning grid used only to give waves spatially coherent particles (results do not depend on it)
struct BinGrid {
    double ox, oy, oz, inv_h;      // bin = floor((p - o) * inv_h), clamped
    int nbx, nby, nbz;             // bins per axis
    int bx4, by4;                  // bricks (4x4x4 bins) per axis in x and y
    uint32_t nkeys;                // padded key space
};

struct GaussParams {
    double maxdist;                // 1.25 * range^2            meshTree.C:155
    double two_sigma2;             // 2 * pow(sigmaInterp, 2)   FoamYade.C:308
    double range_cu;               // interpRangeCu             FoamYade.C:71
    double sigma_pi;               // sigmaPi                   FoamYade.C:72
};

struct ForceParams {
    double rhoF, nu, small;        // FoamYade.H:67,83-85
    double rhoP, delta_t;          // FoamYade.H:83,94 (added mass only)
    unsigned models;               // FY_FORCE_* : the reference's call-site-less models, off by default
    int torque_prezeroed;          // the torque half of every force record is already zero (and stays so): store the force half only
    double uniform_vol;            // > 0: every cell has this volume (structured block) -- the back-scatter does not gather V[c]
};

// sorted SoA particle arrays + per-particle stencil storage for one batch
struct ParticleSoA {
    double *px, *py, *pz, *vx, *vy, *vz, *rad;
    int32_t* orig;                 // original (wire) index
    int32_t* chain_len;            // pushes into the improvement chain (k = min(chain_len, 16))
    unsigned char* scan_class;     // (nullable) how far the locate's list scan ran: 0 one chunk of 8 codes, 1 two, 2 three or the walk -- second ordering key of the placement's runs
    int32_t* ids;                  // [16][cap]  chain order, slot = push index & 15
    double* w;                     // [16][cap]  normalised weights, same slots
    size_t cap;                    // leading dimension
};

int launch_bin_count(hipStream_t s, const double* rec, int64_t n, BinGrid g, uint32_t* key, uint32_t* rank, uint32_t* hist);
int launch_exclusive_scan_u32(hipStream_t s, uint32_t* data, uint32_t n, uint32_t* block_sums /* >= ceil(n/2048)+1 */);
// after the scan `data` holds tile-local exclusive offsets and `block_sums` the exclusive tile offsets:
// start(key) = data[key] + block_sums[key >> 11]
// gather = false: only the placement (p.orig) is formed; whoever runs next fills the SoA arrays (launch_locate_deposit with rec_gather)
int launch_bin_scatter(hipStream_t s, const double* rec, int64_t n, const uint32_t* key, const uint32_t* rank,
                       const uint32_t* start, const uint32_t* tile_off, ParticleSoA p, bool gather = true);

// implicit-coordinate tree (uniform hex block verified at create time): node = packed (i | j << 10 | k << 20)
// z-slab ownership: a rank that is handed particles of the whole block -- what the reference's serial-Yade broadcast
// does (FoamYade.C:176-183) -- locates only those whose containing (nearest) cell lies in its own planes [k0, k1); every other
// particle is "not found" here (found = -1, zero force), exactly one rank owns each particle.  active = 0: single domain.
// npieces > 0 (records that came in as several wire pieces, fy_transport::recv_view): record `w` (wire index) belongs to the last piece
// whose pstart <= w and is located only within that piece's cell layers [pk0, pk1) across paxis -- a particle near a cut arrives in both pieces, one finds it.
struct SlabOwn {
    int active, k0, k1, nzglob;
    double oz, dx;
    int npieces, paxis, pn;        // pieces cut the block across axis paxis (pn cell layers, first one at po)
    double po;
    int pstart[8], pk0[8], pk1[8];
};

struct ImplicitGeom {
    double ox, oy, oz, dx;
    int nx, ny, nz;
};
// the chain lengths of the step before, filed under the wire index (before the placement they belong to is overwritten), and the new placement's
// runs of 512 slots put in order of them (stable): see k_order_blocks_by_chain
int launch_chain_by_wire(hipStream_t s, const int32_t* orig, const int32_t* chain_len, const unsigned char* scan_class /* nullable */, int64_t n, unsigned char* kwire);
int launch_order_blocks_by_chain(hipStream_t s, int32_t* orig, const unsigned char* kwire, int64_t n);
// re-use the placement (p.orig) of an earlier step: only gather the records into the SoA arrays
int launch_bin_gather(hipStream_t s, const double* rec, int64_t n, ParticleSoA p);
// packed == nullptr selects the explicit 32-byte-node path.  Leaves chain ids and squared distances (in the weight slots).
// start (implicit trees only, may be nullptr): per-cell traversal start built by launch_build_locate_start
// ll.lists (implicit trees only, may be nullptr): per-(cell, octant) candidate lists built by launch_build_locate_lists; the walk then
// only takes the particles the lists do not cover (ll.fb_list: one int32 per particle, ll.fb_count: one counter)
struct LocateLists {
    const unsigned short* lists;
    int32_t* fb_list;
    unsigned int* fb_count;
    int32_t cell0, n_listed;             // the lists cover cells [cell0, cell0 + n_listed): the whole block, or a slab's own planes
    // explicit trees (lists == nullptr): the walk's short LDS stack -- stack_cap entries per lane (0: the full depth), the walks that need more go to fb_list and a
    // second launch; depth_hwm (device, nullable): kLocDepthBins counters, a histogram of the deepest stack of one walk in 64, from which the caller picks the next
    // step's stack_cap
    int32_t stack_cap;
    unsigned int* depth_hwm;
    bool fb_zeroed = false;              // fb_count is zero already (launch_tile_caps did it): launch_locate_deposit skips its 4-byte memset
};
constexpr int kLocDepthBins = 32;
constexpr int kLocateListLen = 24;       // codes (2 B) per (cell, octant)
int launch_locate(hipStream_t s, const KdNode* tree, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, int levels,
                  ParticleSoA p, int64_t n, GaussParams gp, const unsigned long long* start = nullptr, SlabOwn own = SlabOwn{},
                  LocateLists ll = LocateLists{});
// meshTree::nearestCell (meshTree.C:66-135) for n query points [n][3]: the id of the nearest cell centre, ties to the first in DFS order
int launch_nearest_cell(hipStream_t s, const KdNode* tree, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, const double* pos, int64_t n, int32_t* out);
// lists: n_listed * 8 * kLocateListLen codes, for the cells [cell0, cell0 + n_listed).  See k_build_locate_lists for what a list is and why scanning it reproduces the walk.
int launch_build_locate_lists(hipStream_t s, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, double maxdist, unsigned short* lists,
                              int32_t cell0, int32_t n_listed);
// For every cell of the block: the deepest tree node a walk for a query inside that cell is guaranteed to reach with an empty
// stack and an empty chain (entry: offset | size << 25 | axis << 50).  See k_build_locate_start.
int launch_build_locate_start(hipStream_t s, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, double maxdist, unsigned long long* start);
// cell arrays are indexed with (global cell id - cell_base) and hold n_field cells (slab storage); ids outside are skipped
struct CellWindow { int64_t base; int64_t n_field; };

// ---- cell-tile ownership of the scatters' flush.  Global FP64 atomics run at ~23 G/s on MI355X whatever their locality or packing
// (tools/micro/atomic_rate.hip): the 21.8 M flush atomics of ONE LDS-hashed scatter at C3 are 0.93 ms -- that, not the gathers, is what
// bounded k_locate_deposit and k_force_gaussian in round 1 -- while LDS FP64 atomics run at ~850 G/s (tools/micro/lds_atomic_rate.hip) and
// plain stores at > 200 G/s.  So a workgroup no longer adds its aggregation table to the cell arrays itself: it hands every table
// entry {cell, 4 partial sums} to the TILE (8 x 8 x 8 cells of the storage block) the cell lies in -- the entries are counted per tile
// in LDS, ONE returning global atomic per (workgroup, tile) claims a run in the tile's bucket, plain stores fill it -- and
// k_tile_reduce, one workgroup per tile and the tile's only writer, sums its bucket into a dense 16 KiB LDS accumulator and adds it to
// the cell arrays with plain read-modify-writes.  ~0.5 M global atomics per scatter instead of 21.8 M.
// A bucket's capacity is last step's demand x 1.25 + 128 entries (k_tile_caps); an entry that does not fit is added with the four
// global atomics of round 1, so the first step of a population runs on atomics and counts, and nothing is ever dropped.
constexpr int kTileEdge = 8, kTileCells = 512;
struct TileGrid { int nx, ny, nzs, ntx, nty, ntz; __host__ __device__ int n_tiles() const { return ntx * nty * ntz; } };
struct TileBuckets {               // all null: flush with global atomics (round-1 behaviour)
    uint32_t* cell;                // [pool] cell index inside the tile
    double* val;                   // [pool][4] the workgroup's partial sums for that cell
    uint32_t* off;                 // [tiles] first entry of the tile's bucket
    uint32_t* cap;                 // [tiles] its capacity
    uint32_t* fill;                // [tiles] entries asked for this step (may exceed cap: the excess went out as atomics)
    uint32_t pool;
    TileGrid tg;
};
// capacities / offsets for this step from the demand counted last step, demand counters reset (two bucket sets in one launch)
int launch_tile_caps(hipStream_t s, TileBuckets a, TileBuckets b, unsigned int* also_zero = nullptr /* a counter the next locate pass wants cleared */);
// dst0[c] += sum of the tile's entries' first value, dst3[c][0..2] += the other three; touched[c] = 1 where something arrived (nullable)
// z-slabs: the same with the finish of launch_tile_reduce_finalize / _fold for the tiles of the z-layers [tk_lo, tk_hi) only (planes no reverse halo reaches)
int launch_tile_reduce_finalize_layers(hipStream_t s, TileBuckets tb, double* pvol_acc, double* up_acc, unsigned char* touched, const double* vol, double* alpha,
                                       double* uParticle, double* R, int tk_lo, int tk_hi);
int launch_tile_reduce_fold_layers(hipStream_t s, TileBuckets tb, double* drag_acc, double* uSource, const double* uParticle, double* uSourceDrag, int tk_lo, int tk_hi);
int launch_tile_reduce(hipStream_t s, TileBuckets tb, double* dst0, double* dst3, unsigned char* touched);
// single domain: the reduction and what follows it per cell in ONE pass over the tiles -- launch_finalize_cells, resp. launch_fold_sources
int launch_tile_reduce_finalize(hipStream_t s, TileBuckets tb, double* pvol_acc, double* up_acc, unsigned char* touched, const double* vol, double* alpha,
                                double* uParticle, double* R);
int launch_tile_reduce_fold(hipStream_t s, TileBuckets tb, double* drag_acc, double* uSource, const double* uParticle, double* uSourceDrag);

// a second stream for work that may run beside the caller's next launches: forked from the main stream at `fork`, done at `join`
struct SideStream { hipStream_t stream; hipEvent_t fork, join; };
// forms the normalised Gaussian weights from the parked squared distances, then deposits (LDS-aggregated)
int launch_deposit(hipStream_t s, ParticleSoA p, int64_t n, GaussParams gp, CellWindow cw, double* pvol_acc, double* up_acc, unsigned char* touched, TileBuckets tb = TileBuckets{});
// launch_locate + launch_deposit; with candidate lists (ll.lists) both run as ONE pass (k_locate_deposit) and the chain's squared
// distances never reach memory
int launch_locate_deposit(hipStream_t s, const KdNode* tree, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, int levels, ParticleSoA p, int64_t n,
                          GaussParams gp, const unsigned long long* start, SlabOwn own, LocateLists ll, CellWindow cw, double* pvol_acc, double* up_acc,
                          unsigned char* touched, TileBuckets tb = TileBuckets{}, SideStream side = SideStream{},
                          const double* rec_gather = nullptr /* wire records: k_locate_deposit fetches them through p.orig and fills the SoA arrays itself */);
// device buffer -> mapped host memory by 16-byte stores of a kernel (a D2H that does not occupy the copy engine)
int launch_copy_out(hipStream_t s, void* dst_mapped, const void* src, size_t bytes);
int launch_add_mark(hipStream_t s, double* y, const double* x, size_t n, unsigned char* mark /* nullable; set where x != 0 */);
int launch_finalize_cells(hipStream_t s, int32_t n_cells, const double* vol, double* pvol_acc, double* up_acc,
                          unsigned char* touched, double* alpha, double* uParticle, double* R /* nullable: cell records whose alpha slot follows */);
// the force pass gathers one 64-byte record per stencil cell: {U[3], alpha, A[3] = 2 nu rho_f divT - gradP, V} (k_pack_cells)
int launch_pack_cells(hipStream_t s, int64_t n_field, const double* U, const double* alpha, const double* gradP, const double* divT, const double* vol,
                      double nu, double rhoF, double* R);
int launch_patch_rec_alpha(hipStream_t s, int64_t c0, int64_t n, const double* alpha, double* R);    // alpha slot of cells [c0, c0 + n)
// scatters D[c] = sum -coeff w / rho_f into drag_acc and the Archimedes reaction into uSource; launch_fold_sources then adds D to
// uSourceDrag and uParticle * D to uSource and clears D
int launch_force_gaussian(hipStream_t s, ParticleSoA p, int64_t n, ForceParams fp, CellWindow cw, const double* vol, const double* R,
                          const double* vGrad, const double* ddtU, const double* rec, double* drag_acc, double* uSource,
                          double* force_out, TileBuckets tb = TileBuckets{});
int launch_fold_sources(hipStream_t s, int64_t n_field, double* drag_acc, const double* uParticle, double* uSourceDrag, double* uSource);
// z-slab migration: classify by owner slab and pack (11 doubles per particle: record + tag bits); counters = {stay, up, down}
int launch_fibre_repack(hipStream_t s, const double* wide, double* rec, int64_t n);   // 15-double fibre records -> [n][10]
int launch_migrate_pack(hipStream_t s, const double* rec, const int64_t* tags, int64_t n, SlabOwn own, unsigned int* counters, double* stay, double* up, double* down);
int launch_migrate_unpack(hipStream_t s, const double* packed, int64_t n, double* rec, int64_t* tags);
// Gaussian mode: found flags in wire order from the chain lengths (launch_force_gaussian no longer writes found_out)
int launch_found_from_chain(hipStream_t s, ParticleSoA p, int64_t n, int32_t* found);
int launch_unpack_stencils(hipStream_t s, ParticleSoA p, int64_t n, int32_t* k, int32_t* ids, double* w, int32_t* chain);

// point-force mode (icoFoamYade): findCell on the uniform block + Stokes drag/torque + source scatter
struct BlockGeom {
    double bbmin[3], bbmax[3], dx;
    int nx, ny, nz;
    const double* faces[3];        // graded block: coordinates of the n + 1 face planes per axis (device); null = uniform block
    const int32_t* cell_of;        // general mesh (fy_ldu_solver): the containing cell of every record, located beforehand (-1: outside); null = block arithmetic
};
int launch_point_force(hipStream_t s, const double* rec, int64_t n, BlockGeom g, ForceParams fp, CellWindow cw, const double* vol,
                       const double* U, const double* vGrad, double* uSource, double* force_out, int32_t* found_out,
                       int32_t* incell_out, SlabOwn own = SlabOwn{});

int launch_fill_f64(hipStream_t s, double* p, size_t n, double v);
int launch_set_source_zero(hipStream_t s, int32_t n_cells, int gaussian, double* uSourceDrag, double* alpha,
                           double* uSource, double* uParticle);

}  // namespace fy
