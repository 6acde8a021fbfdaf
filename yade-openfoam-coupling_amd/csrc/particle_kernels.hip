// Particle half of the hot path as hand-written HIP for gfx950 (MI355X, wave64).
//
//   bin_count / scan / bin_scatter   AoS wire records -> spatially binned SoA (locality only; results independent)
//   locate_deposit                   meshTree::nnearestCellsRange (meshTree.C:148-238) as a flat per-lane DFS over the
//                                    preorder tree with an LDS stack, fused with calcInterpWeightGaussian
//                                    (FoamYade.C:293-316) and buildCellPartList's deposition (FoamYade.C:261-290)
//   finalize_cells                   setCellVolFraction (FoamYade.C:318-328)
//   force_gaussian                   hydroDragForce + archimedesForce + source back-scatter (FoamYade.C:354-389,415-435)
//   point_force                      findCell + stokesDragForce + stokesDragTorque (FoamYade.C:248-253,437-453)
//
// All arithmetic is FP64 in the reference's operation order; the file is compiled with -ffp-contract=off so that the
// distance comparisons that steer the tree walk are bit-identical to the CPU reference (index work must be exact).
// Everything here is HBM/L2-latency bound: no MFMA (there is no dense contraction on this path).
#include "particle_kernels.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "common.hpp"

namespace fy {
namespace {

constexpr int kWave = 64;

// Block order: plain round-robin.  An XCD-aware remap (guide T1: block b runs on XCD b % 8, give each XCD a contiguous run of the
// binned particle order) was measured on MI355X and LOSES for the particle kernels: locate+deposit 14.5 -> 16.1 ms, force 10.0 ->
// 10.8 ms at 10 M particles, because the FP64 atomics of a spatially compact run pile onto few memory channels.

// pow(dia, 3.0) of FoamYade.H:36 as two multiplications in the two hot kernels (k_locate_deposit, k_force_gaussian): the library pow is ~200 FP64
// instructions per call, which showed in the VALU-bound list scan (1.02 -> 0.98 ms); x*x*x is within one ulp of the correctly rounded cube, the
// same order as the difference between the device's and glibc's pow that the golden tolerances already carry.  The cold sites keep pow: with
// the cheap form the compiler if-converts the opt-in torque branch of force_law and the force kernel loses 0.2 ms to register pressure
__device__ __forceinline__ double cube3(double x) { return (x * x) * x; }

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
    // global_atomic_add_f64, no return value, device (agent) scope.  (Workgroup-scope atomics were measured: no faster.)
    unsafeAtomicAdd(p, v);
}

// ------------------------------------------------------------------------------------------------ binning
__device__ __forceinline__ uint32_t bin_key(const BinGrid& g, double x, double y, double z) {
    int bx = (int)floor((x - g.ox) * g.inv_h);
    int by = (int)floor((y - g.oy) * g.inv_h);
    int bz = (int)floor((z - g.oz) * g.inv_h);
    bx = min(max(bx, 0), g.nbx - 1);
    by = min(max(by, 0), g.nby - 1);
    bz = min(max(bz, 0), g.nbz - 1);
    const uint32_t brick = ((uint32_t)(bz >> 2) * g.by4 + (uint32_t)(by >> 2)) * g.bx4 + (uint32_t)(bx >> 2);
    const uint32_t local = ((uint32_t)(bz & 3) << 4) | ((uint32_t)(by & 3) << 2) | (uint32_t)(bx & 3);
    return brick * 64u + local;
}

__global__ __launch_bounds__(256) void k_bin_count(const double* __restrict__ rec, int64_t n, BinGrid g,
                                                   uint32_t* __restrict__ key, uint32_t* __restrict__ rank,
                                                   uint32_t* __restrict__ hist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = rec + 10 * i;
    const uint32_t k = bin_key(g, r[0], r[1], r[2]);
    key[i] = k;
    rank[i] = atomicAdd(&hist[k], 1u);
}

// exclusive scan of 2048-element tiles; tile totals go to block_sums (scanned in place by k_scan_sums)
__global__ __launch_bounds__(256) void k_scan_tiles(uint32_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t wave_tot[4];
    const uint32_t base = blockIdx.x * 2048u + threadIdx.x * 8u;
    uint32_t v[8];
    uint32_t t = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t x = (base + j < n) ? data[base + j] : 0u;
        v[j] = t;
        t += x;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 64);
        if (lane >= d) inc += y;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int q = 0; q < wv; ++q) woff += wave_tot[q];
    const uint32_t excl = woff + inc - t;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (base + j < n) data[base + j] = excl + v[j];
    if (threadIdx.x == 255) block_sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t* __restrict__ sums, uint32_t nb) {
    __shared__ uint32_t wave_tot[16];
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per;
    uint32_t t = 0;
    for (uint32_t j = 0; j < per; ++j)
        if (lo + j < nb) t += sums[lo + j];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d, 64);
        if (lane >= d) inc += y;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int q = 0; q < wv; ++q) woff += wave_tot[q];
    uint32_t run = woff + inc - t;
    for (uint32_t j = 0; j < per; ++j)
        if (lo + j < nb) {
            const uint32_t x = sums[lo + j];
            sums[lo + j] = run;
            run += x;
        }
}

// Counting-sort placement in two steps: the scatter writes ONLY the 4-byte source index to the sorted position (random 4-B
// writes), then a gather pass reads whole 80-byte records at random and writes the SoA arrays coalesced.  Scattering the seven
// doubles directly (10 M x 8 random 8-byte writes) measured 1.28 ms at 10 M particles; random reads are far cheaper than random writes.
__global__ __launch_bounds__(256) void k_bin_scatter(int64_t n, const uint32_t* __restrict__ key, const uint32_t* __restrict__ rank,
                                                     const uint32_t* __restrict__ start, const uint32_t* __restrict__ tile_off,
                                                     int32_t* __restrict__ orig) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = key[i];
    const size_t d = (size_t)start[k] + tile_off[k >> 11] + rank[i];
    orig[d] = (int32_t)i;
}

// The chain length of a particle barely changes from one coupling step to the next (at rest: not at all), and the force pass's three loops
// over a stencil run to the LONGEST chain of a wave: with a wave's lanes holding a 0 .. 14 mix around the mean of 5.46 that is ~9.5
// iterations, with chains of one length (two at a class boundary) ~5.6.  So when the placement is recomputed, every run of 512 slots (one
// workgroup of the locate and of the force pass: the same particles, the same cells in reach) is put in order of the chain lengths of the
// step before -- stable, so the cell order survives inside a class.  k_chain_by_wire files the lengths under the wire index first (the
// placement they were computed in is about to be overwritten).  Costs two small launches per rebin_interval steps.
constexpr int kOrderClasses = 3 * (kMaxK + 1);
__global__ __launch_bounds__(256) void k_chain_by_wire(const int32_t* __restrict__ orig, const int32_t* __restrict__ chain_len,
                                                       const unsigned char* __restrict__ scan_class, int64_t n, unsigned char* __restrict__ kwire) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = chain_len[i];
    const int k = c < 0 ? 0 : (c > kMaxK ? kMaxK : c);
    const int sc = scan_class ? (int)scan_class[i] : 0;
    kwire[orig[i]] = (unsigned char)(3 * k + sc);        // chain length first, then how far the list scan ran (measured: +0.5 - 1 % of the particle phase over the length alone; the other way round is no better)
}
constexpr int kRun = 512;      // = kDepThreads: the locate's workgroup (1024 loses 4 % of the step to locality, 256 is no better)
__global__ __launch_bounds__(kRun) void k_order_blocks_by_chain(int32_t* __restrict__ orig, const unsigned char* __restrict__ kwire, int64_t n) {
    __shared__ uint32_t cnt[kOrderClasses][kRun / 64];             // [class][wave]
    const int64_t i = (int64_t)blockIdx.x * kRun + threadIdx.x;
    const bool have = i < n;
    const int32_t w = have ? orig[i] : 0;
    const int k = have ? (int)kwire[w] : -1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t rank = 0;
    for (int c = 0; c < kOrderClasses; ++c) {
        const unsigned long long m = __ballot(k == c);
        if (k == c) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) cnt[c][wv] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (!have) return;
    uint32_t before = 0;
    for (int c = 0; c < k; ++c) for (int q = 0; q < kRun / 64; ++q) before += cnt[c][q];
    for (int q = 0; q < wv; ++q) before += cnt[k][q];
    orig[(int64_t)blockIdx.x * kRun + before + rank] = w;
}

__global__ __launch_bounds__(256) void k_bin_gather(const double* __restrict__ rec, int64_t n, ParticleSoA p) {
    const int64_t d = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (d >= n) return;
    const double* r = rec + 10 * (size_t)p.orig[d];
    p.px[d] = r[0]; p.py[d] = r[1]; p.pz[d] = r[2];
    p.vx[d] = r[3]; p.vy[d] = r[4]; p.vz[d] = r[5];
    p.rad[d] = r[9];
}

// ------------------------------------------------------------------------------------------------ locate + deposit
// One lane per particle, one wave per 64-particle chunk of the binned order.  Per-lane DFS stack lives in LDS as
// [level][lane] uint4 = {offset, size | axis<<30, df2 (2 dwords)}: 16 B * 64 lanes = one 1 KiB row per level, and the bank of an
// entry depends on the lane only, so lanes at different levels never conflict.
// Node fetch.  Explicit: 32-byte {x,y,z,id} records (any mesh).  Implicit: for a uniform hex block whose centres were verified
// at create time to equal origin + (i + 0.5) * dx bit for bit, a node is just the packed (i,j,k) of its cell -- 4 bytes, 16 nodes
// per 64-byte line instead of 2 -- and the centre is recomputed with the same two IEEE operations (contraction is off).
// The divergent bottom-of-tree loads are what bound this kernel (one L1 tag lookup per distinct line per instruction), so
// trading ~12 FP64 ALU ops for 8x fewer lines is the MI355X-shaped choice.
struct NodeVal { double x, y, z; int32_t id; };

// may this record be located here: its containing cell lies in the slab's planes and -- records that came in wire pieces -- in its piece's layers
__device__ __forceinline__ bool own_planes(const SlabOwn& own, int wire_index, int kz, double qx, double qy, double qz) {
    if (kz < own.k0 || kz >= own.k1) return false;
    if (own.npieces == 0) return true;
    const double qa = own.paxis == 0 ? qx : (own.paxis == 1 ? qy : qz);
    int ka = (int)floor((qa - own.po) / own.dx);
    ka = min(max(ka, 0), own.pn - 1);
    int a0 = 0, a1 = 0;
    for (int q = 0; q < own.npieces; ++q)
        if (wire_index >= own.pstart[q]) { a0 = own.pk0[q]; a1 = own.pk1[q]; }
    return ka >= a0 && ka < a1;
}

#ifndef FY_NODE_LOAD2
#define FY_NODE_LOAD2 1
#endif
template <bool IMPLICIT>
__device__ __forceinline__ NodeVal fetch_node(const KdNode* __restrict__ tree, const uint32_t* __restrict__ packed, const ImplicitGeom& ig, uint32_t o) {
    NodeVal v;
    if constexpr (IMPLICIT) {
        const uint32_t q = packed[o];
        const int i = (int)(q & 1023u), j = (int)((q >> 10) & 1023u), k = (int)(q >> 20);     // 10 + 10 + 12 bits
        v.x = ig.ox + ((double)i + 0.5) * ig.dx;
        v.y = ig.oy + ((double)j + 0.5) * ig.dx;
        v.z = ig.oz + ((double)k + 0.5) * ig.dx;
        v.id = i + ig.nx * (j + ig.ny * k);
    } else {
#if FY_NODE_LOAD2
        // the 32-byte node as two 16-byte loads: the compiler's own split of the struct copy is 16 + 8 + 4, three passes of the address unit over one line -- the explicit walk of
        // 10 M particles through 4.1 M nodes 4.45 -> 3.80 ms.  (One line access per visit -- neighbouring lanes fetching both their nodes together, a half each, and swapping
        // halves by DPP -- was built too: same chains, 7 % SLOWER.  Past two accesses the walk is bound by the latency of its slowest lane's fetch, not by the address unit.)
        const uint4* q = reinterpret_cast<const uint4*>(tree + o);
        const uint4 lo = q[0], hi = q[1];
        v.x = __hiloint2double((int)lo.y, (int)lo.x); v.y = __hiloint2double((int)lo.w, (int)lo.z); v.z = __hiloint2double((int)hi.y, (int)hi.x); v.id = (int32_t)hi.z;
#else
        const KdNode nd = tree[o];
        v.x = nd.x; v.y = nd.y; v.z = nd.z; v.id = nd.id;
#endif
    }
    return v;
}

// Persistent lanes: a wave owns kLocPPB consecutive particles of the binned order and every lane pulls its next particle
// from the wave's range as soon as its walk ends (ballot + prefix count), so lanes do not idle while the slowest walk of a
// 64-particle chunk finishes (measured before this change: 36 % VALU lane utilisation, SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU).
// One loop iteration = at most one pop and one node visit per lane; the Gaussian weights are formed later, at full width,
// by k_deposit -- here the squared distance of every chain member is parked in its weight slot.
constexpr int kLocPPB = 1024;
#ifndef FY_LOC_REFILL
#define FY_LOC_REFILL 12
#endif
constexpr int kLocRefill = FY_LOC_REFILL;    // refill once this many lanes are idle

// DFS stack entry.  Implicit nodes: 8 B -- the far side's df2 is
// NOT stored; the entry carries the parent's 10-bit lattice index along the split axis and df2 = (o + (idx + 0.5) dx - q)^2 is
// recomputed at pop time with the same IEEE operations, bit for bit.  Halving the entry doubles the waves a CU can hold
// (LDS: levels x 64 lanes x entry), and this kernel is latency bound.
//   implicit entry: bits 0..24 far offset | 25..49 far size | 50..51 axis of the far child | 52..63 parent index on the split axis
//   (offsets and sizes < 2^25 = 33.5 M nodes, lattice index < 4096: covers the 32.8 M-cell C5 block and 1280-plane weak-scaling boxes)
template <bool IMPLICIT, bool WIDE> struct StackEntry { typedef unsigned long long type; };
// explicit nodes, 8 B: offset | size << 25 | axis << 50 | t << 52, t = a 12-bit LOWER bound of df2 in units of maxdist / 4095 (round 6; 16 B with the exact df2 before).  A far side is
// visited when the bound is below `best`: a superset of what the exact test admits, and what it adds cannot change a chain -- every centre behind the plane has d >= df2 >= best in
// floating point too (the candidate lists' note below).  Half the LDS per wave: 13 -> 24 waves per CU, the walk of 10 M particles through 4.1 M nodes 3.76 -> 3.30 ms.
// WIDE (trees of 2^25 nodes and more): the 16-byte entry {offset, size | axis << 30, df2}.
template <> struct StackEntry<false, true> { typedef uint4 type; };

// Per-cell traversal start.  For a query q inside cell (ci,cj,ck) the top of the reference's walk is fully determined:
//  (a) an ancestor whose centre is farther than sqrt(maxdist) from every point of the cell can improve `best` but can never enter
//      the chain (meshTree.C:185-192 only queues d < maxdist), and any `best` >= maxdist it leaves behind is indistinguishable, for
//      the chain, from best = +inf: nodes it would have rejected are >= maxdist away and are not queued either;
//  (b) when the ancestor's lattice index on its split axis differs from the cell's by >= 2, the near side is the side holding the
//      cell (no exact comparison involved) and the far side is >= 1.5 dx away; by the time the reference comes back to it the near
//      subtree has been searched, best <= |q - own cell centre|^2 <= 0.75 dx^2 < 2.25 dx^2, so `df2 < best` (meshTree.C:225) fails.
// While both hold the walk just descends; the first node where one fails is where k_locate starts, with an empty stack.
// 160^3: 8.8 levels skipped on average, 47.1 -> 38.4 visits and 21.9 -> 14.6 stack entries per particle, chains bit-identical.
__global__ __launch_bounds__(256) void k_build_locate_start(const uint32_t* __restrict__ packed, ImplicitGeom ig, int32_t n_cells, double md_cells,
                                                            unsigned long long* __restrict__ start) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    const int cc[3] = {c % ig.nx, (c / ig.nx) % ig.ny, c / (ig.nx * ig.ny)};
    uint32_t o = 0, nn = (uint32_t)n_cells, axis = 0;
    while (nn > 0) {
        const uint32_t pk = packed[o];
        const int nd[3] = {(int)(pk & 1023u), (int)((pk >> 10) & 1023u), (int)(pk >> 20)};
        double g2 = 0.0;                                  // squared gap between the node centre and the (slightly inflated) cell box, in dx
        for (int a = 0; a < 3; ++a) {
            const double gap = fabs((double)(nd[a] - cc[a])) - 0.5 - 1e-6;
            if (gap > 0) g2 += gap * gap;
        }
        const int ds = nd[axis] - cc[axis];
        if (!(g2 >= md_cells) || (ds < 2 && ds > -2)) break;
        const uint32_t nl = nn >> 1, nr = nn - nl - 1u;
        if (ds > 0) { o = o + 1u; nn = nl; } else { o = o + 1u + nl; nn = nr; }
        axis = (axis == 2u ? 0u : axis + 1u);
    }
    if (nn == 0) { o = 0; axis = 0; nn = (uint32_t)n_cells; }   // cannot happen (a leaf's own cell fails (a)); fall back to the root
    start[c] = (unsigned long long)o | ((unsigned long long)nn << 25) | ((unsigned long long)axis << 50);
}

// ------------------------------------------------------------------------------------------------ candidate lists
// What the walk below produces is, exactly, the sequence of strict running minima of d(node, q) over the tree's nodes in near-first
// DFS order, kept where d < maxdist (meshTree.C:192-196): a far side is skipped only when df2 >= best (meshTree.C:225), every node
// behind it has d >= df2 in floating point too (the squared axis term is one of d's three non-negative addends and rounding is
// monotone), so a skipped subtree holds no improvement.  On the lattice that order is a function of (cell, octant) alone: the side
// taken at an ancestor whose lattice index on its split axis differs from the query cell's is decided by the cell, and where the
// index is the same the ancestor's coordinate IS the cell centre's (same expression, same bits), so the side is the octant bit
// `!(q - centre < 0)`.  Hence, per (cell, octant), ONE list of the nodes that can be a running minimum for some q of that octant, in
// DFS order; a particle evaluates d for ~7 listed nodes with the reference's operations instead of walking ~38 tree nodes with a
// stack.  k_build_locate_lists simulates the walk over the octant box B:
//   * a node enters the list unless it is outside the range for every q in B, or an earlier listed node Y is closer for every q in B
//     (d(Y,q) - d(X,q) is linear in q: its maximum over B is a sum of per-axis endpoint values).  A node left out this way never
//     lowers the running minimum below what Y already did and is never pushed, so the scan's `best` and chain are the walk's;
//   * a far side is skipped when it is out of range for all of B or some listed Y has d(Y,q) <= (q_a - plane)^2 on all of B -- a
//     superset of what the reference visits for any single q; extra nodes have d >= best and change nothing.
// All comparisons carry a margin (kListMargin, lattice units^2) far above the rounding of d (<= 1e-9 for |coordinate| / dx <= 1e6,
// enforced by the caller), and B is the half cell shrunk by kListEps at the faces: particles closer than 2 kListEps dx to a cell face
// (where floor() and the nearest centre may disagree by an ulp), outside the block, or in an octant whose list overflowed, are
// handed to the plain walk (k_locate with a work list).  The root is listed with a no-emit flag (it sets `best` but is never pushed,
// meshTree.C:156).  Verified against the walk on every golden case and by tools/proto/locate_lists.cpp (host prototype).
constexpr int kListLen = kLocateListLen;              // 24 codes per (cell, octant): 48 B = three 16-byte loads
constexpr double kListEps = 4e-6, kListMargin = 1e-8;
constexpr uint32_t kListEnd = 0xffffu, kListOverflow = 0xfffeu, kListNoEmit = 0x1000u;

__device__ __forceinline__ double ax_min2(double lo, double hi, double x) { const double g = x < lo ? lo - x : (x > hi ? x - hi : 0.0); return g * g; }
__device__ __forceinline__ double ax_max2(double lo, double hi, double x) { const double g = fmax(fabs(lo - x), fabs(hi - x)); return g * g; }

__global__ __launch_bounds__(256) void k_build_locate_lists(const uint32_t* __restrict__ packed, ImplicitGeom ig, int32_t n_cells, double md,
                                                            unsigned short* __restrict__ lists, int32_t cell0, int32_t n_listed) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;       // list number: (cell - cell0) * 8 + octant, cells [cell0, cell0 + n_listed)
    if (t >= (size_t)n_listed * 8) return;
    const int cell = cell0 + (int)(t >> 3), oct = (int)(t & 7);
    const int c[3] = {cell % ig.nx, (cell / ig.nx) % ig.ny, cell / (ig.nx * ig.ny)};
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        if ((oct >> a) & 1) { lo[a] = c[a] - kListEps; hi[a] = c[a] + 0.5 - kListEps; }
        else { lo[a] = c[a] - 0.5 + kListEps; hi[a] = c[a] + kListEps; }
    }
    unsigned long long st[28];                       // offset | size << 25 | axis << 50 | parent index on its split axis << 52
    uint32_t kept[kListLen];
    int nk = 0, sp = 0;
    bool overflow = false;
    uint32_t o = 0, nn = (uint32_t)n_cells, axis = 0;
    for (;;) {
        if (nn == 0) {
            if (sp == 0) break;
            const unsigned long long e = st[--sp];
            const uint32_t ea = (uint32_t)((e >> 50) & 3ull), pa = (ea == 0 ? 2u : ea - 1u);
            const int P = (int)(e >> 52);
            bool prune = ax_min2(lo[pa], hi[pa], (double)P) >= md + kListMargin;
            for (int y = 0; y < nk && !prune; ++y) {
                const int Y[3] = {(int)(kept[y] & 1023u), (int)((kept[y] >> 10) & 1023u), (int)(kept[y] >> 20)};
                double sdiff = 0.0;                  // max over B of d(Y,q) - (q_pa - P)^2
                for (int b = 0; b < 3; ++b) {
                    if (b == (int)pa) { const double k = (double)(P - Y[b]); sdiff += fmax(k * (2 * lo[b] - P - Y[b]), k * (2 * hi[b] - P - Y[b])); }
                    else sdiff += ax_max2(lo[b], hi[b], (double)Y[b]);
                }
                prune = sdiff <= -kListMargin;
            }
            if (!prune) { o = (uint32_t)(e & 0x1ffffffull); nn = (uint32_t)((e >> 25) & 0x1ffffffull); axis = ea; }
            continue;
        }
        const uint32_t pk = packed[o];
        const int X[3] = {(int)(pk & 1023u), (int)((pk >> 10) & 1023u), (int)(pk >> 20)};
        double dmin = 0.0;
        for (int a = 0; a < 3; ++a) dmin += ax_min2(lo[a], hi[a], (double)X[a]);
        if (dmin < md + kListMargin) {
            bool dom = false;
            for (int y = 0; y < nk && !dom; ++y) {
                const int Y[3] = {(int)(kept[y] & 1023u), (int)((kept[y] >> 10) & 1023u), (int)(kept[y] >> 20)};
                double sdiff = 0.0;                  // max over B of d(Y,q) - d(X,q); per axis (q-Y)^2 - (q-X)^2 = (X-Y)(2q - X - Y)
                for (int a = 0; a < 3; ++a) {
                    const double k = (double)(X[a] - Y[a]);
                    sdiff += fmax(k * (2 * lo[a] - X[a] - Y[a]), k * (2 * hi[a] - X[a] - Y[a]));
                }
                dom = sdiff <= -kListMargin;
            }
            if (!dom) {
                if (nk < kListLen) {
                    const int di = X[0] - c[0] + 8, dj = X[1] - c[1] + 8, dk = X[2] - c[2] + 8;
                    if ((unsigned)di > 15u || (unsigned)dj > 15u || (unsigned)dk > 15u) overflow = true;      // cannot happen for a range < 7.5 cells
                    kept[nk] = pk;
                    lists[t * kListLen + nk] = (unsigned short)((uint32_t)di | ((uint32_t)dj << 4) | ((uint32_t)dk << 8) | (o == 0 ? kListNoEmit : 0u));
                    ++nk;
                } else {
                    overflow = true;
                }
            }
        }
        const bool left_near = X[axis] != c[axis] ? c[axis] < X[axis] : !((oct >> axis) & 1);
        const uint32_t nl = nn >> 1, nr = nn - nl - 1u;
        uint32_t near_o, near_n, far_o, far_n;
        if (left_near) { near_o = o + 1u; near_n = nl; far_o = o + 1u + nl; far_n = nr; }
        else           { near_o = o + 1u + nl; near_n = nr; far_o = o + 1u; far_n = nl; }
        const unsigned long long P = (unsigned long long)X[axis];
        axis = (axis == 2u ? 0u : axis + 1u);
        if (far_n > 0 && sp < 28) st[sp++] = (unsigned long long)far_o | ((unsigned long long)far_n << 25) | ((unsigned long long)axis << 50) | (P << 52);
        else if (far_n > 0) overflow = true;
        o = near_o; nn = near_n;
    }
    if (nk < kListLen) lists[t * kListLen + nk] = (unsigned short)kListEnd;
    if (overflow) lists[t * kListLen] = (unsigned short)kListOverflow;
}

// What k_deposit does for one particle, straight after its walk and with plain global atomics: the candidate lists' leftovers are a few
// hundred particles (within 8e-6 dx of a cell face), for which a second, latency-bound launch with its own aggregation table cost more
// than the walk itself.  pvol_acc == nullptr: the walk only parks the squared distances (k_deposit follows).
struct WalkDeposit { GaussParams gp; CellWindow cw; double* pvol_acc; double* up_acc; unsigned char* touched; };
__device__ __attribute__((noinline)) void walk_deposit(const ParticleSoA& p, int64_t i, int chain, const WalkDeposit& wd) {
    const int k = chain < kMaxK ? chain : kMaxK;
    if (k <= 0) return;
    const double dia = 2 * p.rad[i];                                  // FoamYade.C:219
    const double pVol = M_PI * cube3(dia) / 6.0;                      // FoamYade.H:36 (as k_locate_deposit forms it)
    const double vx = p.vx[i], vy = p.vy[i], vz = p.vz[i];
    double allwt = 0.0;                                               // calcInterpWeightGaussian FoamYade.C:301-314, ascending-d2 order
#pragma unroll 1
    for (int t = 0; t < k; ++t) {
        const size_t slot = (size_t)((chain - 1 - t) & (kMaxK - 1)) * p.cap + (size_t)i;
        const double wt = exp(-p.w[slot] * (1.0 / wd.gp.two_sigma2)) * wd.gp.range_cu * wd.gp.sigma_pi;      // (reciprocals as in k_locate_deposit: one particle, one set of bits whichever path places it)
        p.w[slot] = wt;
        allwt += wt;
    }
    const double rallwt_ = 1.0 / allwt;
#pragma unroll 1
    for (int t = 0; t < k; ++t) {
        const size_t slot = (size_t)((chain - 1 - t) & (kMaxK - 1)) * p.cap + (size_t)i;
        const double weight = p.w[slot] * rallwt_;                    // FoamYade.C:312-314
        p.w[slot] = weight;
        const int64_t cl = (int64_t)p.ids[slot] - wd.cw.base;         // storage index (slab window)
        if (cl < 0 || cl >= wd.cw.n_field) continue;
        atomic_add_f64(&wd.pvol_acc[cl], pVol * weight);              // buildCellPartList FoamYade.C:265-288
        atomic_add_f64(&wd.up_acc[3 * (size_t)cl + 0], (weight * vx) * pVol);
        atomic_add_f64(&wd.up_acc[3 * (size_t)cl + 1], (weight * vy) * pVol);
        atomic_add_f64(&wd.up_acc[3 * (size_t)cl + 2], (weight * vz) * pVol);
        wd.touched[cl] = 1;
    }
}

// WD: the walk also deposits for the particle it has placed (the leftovers of the candidate lists); without it the instance carries neither the 16 weights' scratch (192 B per
// lane) nor their registers: the explicit walk 97 -> 42 VGPRs
template <bool IMPLICIT, bool WD, bool WIDE = false>
__global__ __launch_bounds__(kWave) void k_locate(const KdNode* __restrict__ tree, const uint32_t* __restrict__ packed, ImplicitGeom ig,
                                                  int32_t n_cells, ParticleSoA p, int64_t n, double maxdist,
                                                  const unsigned long long* __restrict__ start, SlabOwn own,
                                                  const int32_t* __restrict__ work, const unsigned int* __restrict__ work_n, WalkDeposit wd,
                                                  int stack_cap, int32_t* __restrict__ ovf_list, unsigned int* __restrict__ ovf_count, unsigned int* __restrict__ depth_hwm) {
    // ovf_list != null: the LDS stack holds stack_cap entries per lane only (more waves per CU: the kernel is latency bound); a walk that would need more is given up and its
    // particle filed in ovf_list for a second launch with the full depth -- every particle is walked from its start by exactly one of the two, so the chains are the same
    static_assert(!(IMPLICIT && WIDE), "implicit entries are 8 bytes");
    typedef typename StackEntry<IMPLICIT, WIDE>::type entry_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char stack_raw[];
    entry_t* stack = reinterpret_cast<entry_t*>(stack_raw);
#define STK(sp_) stack[(sp_) * kWave + lane]
    const int lane = threadIdx.x;
    // work list (k_locate_lists' leftovers): the waves share its entries evenly; otherwise a wave owns kLocPPB consecutive particles
    int64_t per = (n + gridDim.x - 1) / gridDim.x, base = (int64_t)blockIdx.x * per;      // (kLocPPB for a large cloud; fewer per wave when the cloud is small: locate_grid)
    if (work) { n = (int64_t)*work_n; per = (n + gridDim.x - 1) / gridDim.x; base = (int64_t)blockIdx.x * per; }
    const int64_t end = (base + per < n) ? base + per : n;
    int64_t next = base;                               // wave-uniform cursor into [base, end)
    const NodeVal root = fetch_node<IMPLICIT>(tree, packed, ig, 0u);
    const double hdx = 0.5 * ig.dx;

    bool active = false;
    int64_t i = 0;
    double qx = 0, qy = 0, qz = 0, best = 0;
    int chain = 0, sp = 0, spmax = 0;
    uint32_t o = 0, nn = 0, axis = 0;
    for (;;) {
        // ---- hand new particles to idle lanes (batched: the refill code is wave-wide, so wait until it pays)
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (next < end && (n_idle >= kLocRefill || n_idle == kWave)) {
            const int rank = __popcll(idle & ((1ull << lane) - 1ull));
            const int64_t cand = next + rank;
            if (!active && cand < end) {
                i = work ? (int64_t)work[cand] : cand;
                qx = p.px[i]; qy = p.py[i]; qz = p.pz[i];
                // meshTree.C:156: dist = distance(root->p, px); the root itself can never enter the queue (x < x is false)
                const double a = qx - root.x, b = qy - root.y, c = qz - root.z;
                best = a * a;
                best += b * b;
                best += c * c;
                // Round 4: the walk starts from best = min(|q - root|^2, maxdist).  The chain is the same: a centre is queued iff it is nearer than maxdist AND than every
                // centre visited before it (meshTree.C:192-196), and centres at or beyond maxdist -- the only ones a smaller starting bound hides, in the subtrees whose
                // split plane is at least sqrt(maxdist) away -- can neither be queued nor outbid one that is.  What changes is the stack: the far sides of the top levels,
                // pushed while `best` is still the distance to some far ancestor and dropped again at their pop, are not pushed at all
                best = fmin(best, maxdist);
                chain = 0; sp = 0; spmax = 0; o = 0; nn = (uint32_t)n_cells; axis = 0;
                if constexpr (IMPLICIT) {
                    if (start) {                                 // skip the levels the query's cell determines (k_build_locate_start)
                        const double fx = floor((qx - ig.ox) / ig.dx), fy_ = floor((qy - ig.oy) / ig.dx), fz = floor((qz - ig.oz) / ig.dx);
                        if (fx >= 0 && fx < ig.nx && fy_ >= 0 && fy_ < ig.ny && fz >= 0 && fz < ig.nz) {
                            const unsigned long long e = start[(size_t)fx + (size_t)ig.nx * ((size_t)fy_ + (size_t)ig.ny * (size_t)fz)];
                            o = (uint32_t)(e & 0x1ffffffull); nn = (uint32_t)((e >> 25) & 0x1ffffffull); axis = (uint32_t)((e >> 50) & 3ull);
                            if (o != 0) best = maxdist;          // no ancestor was within maxdist: same chain as any best >= maxdist, and no far side beyond it is stacked
                        }
                    }
                }
                active = true;
                if (own.active) {                                // another slab's particle: not located here (k = 0)
                    int kz = (int)floor((qz - own.oz) / own.dx);
                    kz = min(max(kz, 0), own.nzglob - 1);
                    if (!(qz == qz) || !own_planes(own, p.orig[i], kz, qx, qy, qz)) { p.chain_len[i] = 0; active = false; }
                }
            }
            next += n_idle;
        } else if (n_idle == kWave) {
            break;                                       // nothing running, nothing left
        }
        if (active) {
            if (nn == 0) {
                // up to two pops per iteration: most popped far sides fail df2 < best and would waste the visit slot
                for (int attempt = 0; attempt < 2; ++attempt) {
                    if (nn != 0) break;
                    if (sp == 0) {
                        if (active) {                                          // walk finished; k = min(chain, 16)
                            p.chain_len[i] = chain; active = false;
                            if constexpr (WD) walk_deposit(p, i, chain, wd);
                            if (depth_hwm && (i & 63) == 0) atomicAdd(&depth_hwm[min(spmax, kLocDepthBins - 1)], 1u);      // (one walk in 64: a histogram of the stack depths)
                        }
                        break;
                    }
                    --sp;
                    const entry_t e = STK(sp);
                    double df2;
                    uint32_t eo, en, ea;
                    if constexpr (IMPLICIT) {
                        eo = (uint32_t)(e & 0x1ffffffull); en = (uint32_t)((e >> 25) & 0x1ffffffull); ea = (uint32_t)((e >> 50) & 3ull);
                        const int idx = (int)(e >> 52);
                        const uint32_t pa = (ea == 0 ? 2u : ea - 1u);               // the parent's split axis
                        const double org = (pa == 0 ? ig.ox : (pa == 1 ? ig.oy : ig.oz));
                        const double qq = (pa == 0 ? qx : (pa == 1 ? qy : qz));
                        const double df = (org + (double)(2 * idx + 1) * hdx) - qq;
                        df2 = df * df;
                    } else if constexpr (WIDE) {
                        eo = e.x; en = e.y & 0x3fffffffu; ea = e.y >> 30;
                        df2 = __hiloint2double((int)e.w, (int)e.z);
                    } else {
                        eo = (uint32_t)(e & 0x1ffffffull); en = (uint32_t)((e >> 25) & 0x1ffffffull); ea = (uint32_t)((e >> 50) & 3ull);
                        df2 = (double)(uint32_t)(e >> 52) * (maxdist * (1.0 / 4095.0));      // the lower bound
                    }
                    if (df2 < best) { o = eo; nn = en; axis = ea; }                  // meshTree.C:225, evaluated when the near subtree has returned
                }
            }
            if (active && nn != 0) {
                uint32_t pk = 0;
                NodeVal nd;
                if constexpr (IMPLICIT) {
                    pk = packed[o];
                    const int ci = (int)(pk & 1023u), cj = (int)((pk >> 10) & 1023u), ck = (int)(pk >> 20);
                    // (2i + 1) * (dx / 2) is the same real number as (i + 0.5) * dx and both factors are exact, so the rounded
                    // product is the same double: one FP64 add less per axis
                    nd.x = ig.ox + (double)(2 * ci + 1) * hdx;
                    nd.y = ig.oy + (double)(2 * cj + 1) * hdx;
                    nd.z = ig.oz + (double)(2 * ck + 1) * hdx;
                    nd.id = ci + ig.nx * (cj + ig.ny * ck);
                } else {
                    nd = fetch_node<false>(tree, packed, ig, o);
                }
                const double a = qx - nd.x, b = qy - nd.y, c = qz - nd.z;
                const double aa = a * a, bb = b * b, cc = c * c;
                double d = aa;                       // meshTree.C:54-64: dist += ds*ds over x, y, z
                d += bb;
                d += cc;
                if (d < best) {                      // meshTree.C:192 (and the re-push on return is a no-op: same id)
                    best = d;
                    if (d < maxdist) {               // meshTree.C:195
                        const size_t slot = (size_t)(chain & (kMaxK - 1)) * p.cap + (size_t)i;
                        p.ids[slot] = nd.id;
                        p.w[slot] = d;               // squared distance parked here until k_deposit forms the weights
                        ++chain;
                    }
                }
                // meshTree.C:200: df = node[axis] - q[axis] = -(q[axis] - node[axis]) exactly, so df*df is the squared term already
                // formed for the distance and df > 0 <=> (q - node)[axis] < 0 -- same bits, three subtractions and a multiply fewer
                const double mdf = (axis == 0 ? a : (axis == 1 ? b : c));
                const double df2 = mdf * mdf;        // == aa / bb / cc of that axis, one multiply instead of a second three-way select
                const uint32_t nl = nn >> 1, nr = nn - nl - 1;
                uint32_t near_o, near_n, far_o, far_n;
                if (mdf < 0.0) { near_o = o + 1; near_n = nl; far_o = o + 1 + nl; far_n = nr; }     // meshTree.C:206-208
                else          { near_o = o + 1 + nl; near_n = nr; far_o = o + 1; far_n = nl; }      // meshTree.C:209-212
                const uint32_t paxis = axis;
                axis = (axis == 2 ? 0 : axis + 1);
                // best only decreases, so a far side that already fails df2 < best can never pass later
                const bool push = far_n > 0 && df2 < best;
                const bool ovf = push && ovf_list && sp >= stack_cap;
                if (ovf_list) {                          // (kernel-uniform) the lanes that give up file their particles with ONE atomic per wave and iteration
                    const unsigned long long m = __ballot(ovf);
                    if (m) {
                        const int leader = __ffsll((long long)m) - 1;
                        unsigned int at = 0;
                        if (lane == leader) at = atomicAdd(ovf_count, (unsigned int)__popcll(m));
                        at = __shfl(at, leader, kWave);
                        if (ovf) { ovf_list[at + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i; active = false; }
                    }
                }
                if (push && !ovf) {
                    if constexpr (IMPLICIT) {
                        const unsigned long long idx = (paxis == 0 ? (pk & 1023u) : (paxis == 1 ? ((pk >> 10) & 1023u) : (pk >> 20)));
                        STK(sp) = (unsigned long long)far_o | ((unsigned long long)far_n << 25) | ((unsigned long long)axis << 50) | (idx << 52);
                    } else if constexpr (WIDE) {
                        uint4 e;
                        e.x = far_o; e.y = far_n | (axis << 30);
                        e.z = (uint32_t)__double2loint(df2); e.w = (uint32_t)__double2hiint(df2);
                        STK(sp) = e;
                    } else {
                        const int t = max((int)(df2 * (4095.0 / maxdist)) - 1, 0);          // df2 < best <= maxdist: t <= 4094; one unit of slack against the rounding of the product
                        STK(sp) = (unsigned long long)far_o | ((unsigned long long)far_n << 25) | ((unsigned long long)axis << 50) | ((unsigned long long)t << 52);
                    }
                    ++sp;
                    spmax = max(spmax, sp);
                }
                o = near_o; nn = near_n;
            }
        }
    }
#undef STK
}

// ------------------------------------------------------------------------------------------------ nearest cell
// meshTree::nearestCell (meshTree.C:66-135): plain nearest-neighbour descent of the same tree -- near side first, far side iff
// df^2 < best, strict improvements only, so among equidistant centres the first one in depth-first order wins.  The reference never calls
// it; it is the natural findCell stand-in on meshes where the nearest centre is the containing cell.  One lane per
// query, explicit stack in scratch (tree depth <= 25); far sides that cannot pass `df2 < best` any more are dropped at push time.
template <bool IMPLICIT>
__global__ __launch_bounds__(256) void k_nearest_cell(const KdNode* __restrict__ tree, const uint32_t* __restrict__ packed, ImplicitGeom ig, int32_t n_cells,
                                                      const double* __restrict__ pos, int64_t n, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double qx = pos[3 * i], qy = pos[3 * i + 1], qz = pos[3 * i + 2];
    uint32_t st_o[28], st_n[28];
    double st_d[28];
    unsigned char st_a[28];
    int sp = 0;
    const NodeVal root = fetch_node<IMPLICIT>(tree, packed, ig, 0u);
    double best;
    {
        const double a = qx - root.x, b = qy - root.y, c = qz - root.z;
        best = a * a; best += b * b; best += c * c;                 // meshTree.C:71: dist = distance(root->p, v)
    }
    int32_t best_id = root.id;
    uint32_t o = 0, nn = (uint32_t)n_cells, axis = 0;
    for (;;) {
        if (nn == 0) {
            if (sp == 0) break;
            --sp;
            if (!(st_d[sp] < best)) continue;                       // meshTree.C:121, evaluated when the near subtree has returned
            o = st_o[sp]; nn = st_n[sp]; axis = st_a[sp];
        }
        const NodeVal nd = fetch_node<IMPLICIT>(tree, packed, ig, o);
        const double a = qx - nd.x, b = qy - nd.y, c = qz - nd.z;
        double d = a * a;                                           // meshTree.C:54-64
        d += b * b;
        d += c * c;
        if (d < best) { best = d; best_id = nd.id; }                // meshTree.C:90-93 (and the comparisons on return: same node, same distance)
        const double mdf = axis == 0 ? a : (axis == 1 ? b : c);     // -(node[axis] - q[axis])
        const double df2 = mdf * mdf;
        const uint32_t nl = nn >> 1, nr = nn - nl - 1;
        uint32_t near_o, near_n, far_o, far_n;
        if (mdf < 0.0) { near_o = o + 1; near_n = nl; far_o = o + 1 + nl; far_n = nr; }     // df > 0: left first (meshTree.C:104-110)
        else          { near_o = o + 1 + nl; near_n = nr; far_o = o + 1; far_n = nl; }
        axis = (axis == 2 ? 0 : axis + 1);
        if (far_n > 0 && df2 < best && sp < 28) { st_o[sp] = far_o; st_n[sp] = far_n; st_d[sp] = df2; st_a[sp] = (unsigned char)axis; ++sp; }
        o = near_o; nn = near_n;
    }
    out[i] = best_id;
}

// ------------------------------------------------------------------------------------------------ LDS aggregation of scatters
// Binned particles that share a workgroup also share most of their stencil cells (measured on the C3 cloud: 8.4 (particle, cell)
// pairs per distinct cell in a 256-particle block, 12 in a 1024-particle block).  FP64 global atomics are what bounds the scatter
// phases (force kernel: 10.0 ms with, 2.1 ms without them at 10 M particles), so every workgroup first sums its contributions
// per cell in an LDS hash table (ds_add_f64) and only the per-cell totals go out as global_atomic_add_f64.
constexpr uint32_t kAggEmpty = 0xffffffffu;

template <int LOG2SLOTS>
__device__ __forceinline__ int agg_slot(uint32_t* keys, uint32_t cell) {
    uint32_t h = (cell * 2654435761u) >> (32 - LOG2SLOTS);
#pragma unroll 1
    for (int pr = 0; pr < 24; ++pr) {
        const uint32_t old = atomicCAS(&keys[h], kAggEmpty, cell);
        if (old == kAggEmpty || old == cell) return (int)h;
        h = (h + 1) & ((1u << LOG2SLOTS) - 1u);
    }
    return -1;      // table crowded: caller falls back to a direct global atomic
}

__device__ __forceinline__ void lds_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }   // ds_add_f64

// Where component j of table slot h lives.  Component-major (round 4): an 8-byte LDS access is banked by (byte address / 4) mod 32 within
// groups of 16 lanes, so slot-major storage (4 h + j: a 32-byte stride) lets the random slots of a group reach only 4 of the 16 bank pairs
// for a given j -- at least 4-way conflicts by construction; component-major (j N + h) spreads them over all 16.
// Measured (profiles/r04_lds_table_layout.txt): the micro-benchmark's random-slot ds_add_f64 rate doubles (1.4 -> 2.9 lanes per clock per CU),
// k_force_gaussian loses a quarter of its conflict cycles and a third of its LDS wait cycles but only 0.03 ms -- what is left are the
// 16 random slots of a lane group on 16 bank pairs (~3 deep), same-cell adds (an atomic cannot broadcast) and the 4-byte CAS probes.
template <int NSLOTS> __device__ __forceinline__ int agg_at(int h, int j) { return j * NSLOTS + h; }

// ---- quad-lane exchange without the LDS: v_mov_b32 under a DPP quad_perm control -- every lane reads lane J of its own quad.  All four lanes of the
// quad must be active at the call (a disabled source lane reads as zero).
#ifndef FY_FORCE_QUAD
#define FY_FORCE_QUAD 0
#endif
// register budgets of the two hot kernels (build-time constants; tools/build_variant.sh overrides them for A/B runs)
//   k_force_gaussian: room for 6 waves per SIMD = 80 VGPRs (the compiler's own choice is 82: five waves, i.e. TWO 512-thread workgroups per CU; with 80 a third one
//   fits: 1.05 -> 0.93 ms at C3, no spills; 8 waves = 64 VGPRs spills 68 bytes per lane and is slower again)
//   k_locate_deposit: room for 8 waves per SIMD (56 VGPRs as it stands; the cap also keeps the scalar registers at 78 -- with its natural 100 the hardware admits three
//   workgroups per CU instead of four)
#ifndef FY_FORCE_ATTR
#define FY_FORCE_ATTR __attribute__((amdgpu_waves_per_eu(6)))
#endif
#ifndef FY_LD_ATTR
#define FY_LD_ATTR __attribute__((amdgpu_waves_per_eu(8)))
#endif
#ifndef FY_LD_QUAD
#define FY_LD_QUAD 1
#endif
#ifndef FY_LD_QUADROW
#define FY_LD_QUADROW 1
#endif
template <int J> __device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xf, 0xf, true); }
template <int J> __device__ __forceinline__ double quad_bcast(double v) {
    return __hiloint2double(quad_bcast<J>(__double2hiint(v)), quad_bcast<J>(__double2loint(v)));
}
template <int J> __device__ __forceinline__ double2 quad_bcast2(double2 v) { return make_double2(quad_bcast<J>(v.x), quad_bcast<J>(v.y)); }
// 4 x 4 transpose inside every quad, one dword per (lane, register): on entry register r of lane q holds M[q][r], on return register w of lane j
// holds M[w][j].  Two butterfly stages (lane ^ 1 on the register pairs (0,1), (2,3); lane ^ 2 on (0,2), (1,3)), each a select of what to send, one DPP
// move and two selects of what to keep: 16 VALU operations per dword against 28 for "broadcast each source lane, pick mine".
__device__ __forceinline__ void quad_transpose(uint32_t& m0, uint32_t& m1, uint32_t& m2, uint32_t& m3, bool b0, bool b1) {
    constexpr int kXor1 = 1 | (0 << 2) | (3 << 4) | (2 << 6), kXor2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)(b0 ? m0 : m1), kXor1, 0xf, 0xf, true); m0 = b0 ? t : m0; m1 = b0 ? m1 : t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)(b0 ? m2 : m3), kXor1, 0xf, 0xf, true); m2 = b0 ? t : m2; m3 = b0 ? m3 : t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)(b1 ? m0 : m2), kXor2, 0xf, 0xf, true); m0 = b1 ? t : m0; m2 = b1 ? m2 : t;
    t = (uint32_t)__builtin_amdgcn_mov_dpp((int)(b1 ? m1 : m3), kXor2, 0xf, 0xf, true); m1 = b1 ? t : m1; m3 = b1 ? m3 : t;
}
__device__ __forceinline__ void quad_transpose(uint4& m0, uint4& m1, uint4& m2, uint4& m3, int lq) {
    const bool b0 = lq & 1, b1 = lq & 2;
    quad_transpose(m0.x, m1.x, m2.x, m3.x, b0, b1); quad_transpose(m0.y, m1.y, m2.y, m3.y, b0, b1);
    quad_transpose(m0.z, m1.z, m2.z, m3.z, b0, b1); quad_transpose(m0.w, m1.w, m2.w, m3.w, b0, b1);
}
#if FY_FORCE_QUAD
__device__ __forceinline__ double2 quad_pick(double2 v0, double2 v1, double2 v2, double2 v3, int lq) {
    const double2 lo = lq & 1 ? v1 : v0, hi = lq & 1 ? v3 : v2;
    return lq & 2 ? hi : lo;
}
#endif
__device__ __forceinline__ double2 as_double2(uint4 v) {
    return make_double2(__hiloint2double((int)v.y, (int)v.x), __hiloint2double((int)v.w, (int)v.z));
}

// ------------------------------------------------------------------------------------------------ tile buckets (see particle_kernels.hpp)
// storage cell index -> (tile, cell inside the tile)
__device__ __forceinline__ void tile_of(const TileGrid& tg, uint32_t cl, uint32_t* tile, uint32_t* local) {
    const uint32_t i = cl % (uint32_t)tg.nx, r = cl / (uint32_t)tg.nx;
    const uint32_t j = r % (uint32_t)tg.ny, k = r / (uint32_t)tg.ny;
    *tile = (i >> 3) + (uint32_t)tg.ntx * ((j >> 3) + (uint32_t)tg.nty * (k >> 3));
    *local = (i & 7u) | ((j & 7u) << 3) | ((k & 7u) << 6);
}

// k_tile_caps: block b serves bucket set b.  cap[t] = demand x 1.25 + 128, offsets by an exclusive scan, clamped to the pool; demand reset
__global__ __launch_bounds__(1024) void k_tile_caps(TileBuckets s0, TileBuckets s1, unsigned int* __restrict__ also_zero) {
    if (also_zero && blockIdx.x == 0 && threadIdx.x == 0) { also_zero[1] = also_zero[0]; also_zero[0] = 0u; }      // (slot 1 keeps the last pass's count for the diagnostics)
    const TileBuckets tb = blockIdx.x == 0 ? s0 : s1;
    if (!tb.cell) return;
    const int n_tiles = tb.tg.n_tiles();
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int base = 0; base < n_tiles; base += 1024) {
        const int t = base + (int)threadIdx.x;
        uint32_t want = 0;
        if (t < n_tiles) {
            const uint32_t need = tb.fill[t];
            want = need ? ((need + (need >> 2) + 128u + 7u) & ~7u) : 0u;      // a tile nobody wrote to last step keeps no room (its first entries go out as atomics)
            tb.fill[t] = 0;
        }
        uint32_t inc = want;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d, 64);
            if (lane >= d) inc += y;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        uint32_t woff = carry;
        for (int q = 0; q < wv; ++q) woff += wave_tot[q];
        uint32_t start = woff + inc - want;
        if (t < n_tiles) {
            // the pool is finite: a tile that no longer fits gets what is left (its excess goes out as atomics)
            if (start > tb.pool) start = tb.pool;
            const uint32_t room = tb.pool - start;
            tb.off[t] = start;
            tb.cap[t] = want < room ? want : room;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = woff + inc;
        __syncthreads();
    }
}

// A workgroup's aggregation table -> the tiles' buckets.  Per table entry: which tile, and the entry's rank among this workgroup's
// entries for that tile (a 128-slot LDS map: a workgroup's stencils reach a few dozen tiles); one returning global atomic per
// (workgroup, tile) claims the run; plain stores.  What finds no room (bucket full, map full, or no buckets at all) is added with
// the four global atomics.  Every thread of the workgroup calls this (it synchronises).
constexpr int kTileMap = 128;
struct TileMapLds { uint32_t tile[kTileMap], cnt[kTileMap], base[kTileMap]; };

template <int NSLOTS, int NTHREADS>
__device__ __forceinline__ void flush_table(const uint32_t* keys, const double* vals, TileMapLds& m, const TileBuckets& tb, double* __restrict__ dst0,
                                            double* __restrict__ dst3, unsigned char* __restrict__ touched) {
    static_assert(NSLOTS % NTHREADS == 0, "table slots per thread");
    constexpr int R = NSLOTS / NTHREADS;
    uint32_t pr[R];                                         // map slot << 16 | rank; 0xffffffff: empty table slot; 0xfffffffe: no room in the map
#pragma unroll
    for (int r = 0; r < R; ++r) pr[r] = keys[threadIdx.x + r * NTHREADS] == kAggEmpty ? 0xffffffffu : 0xfffffffeu;
    if (tb.cell) {                                          // (uniform)
        for (int q = threadIdx.x; q < kTileMap; q += NTHREADS) { m.tile[q] = kAggEmpty; m.cnt[q] = 0; }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (pr[r] == 0xffffffffu) continue;
            uint32_t tile, local;
            tile_of(tb.tg, keys[threadIdx.x + r * NTHREADS], &tile, &local);
            uint32_t h = (tile * 2654435761u) >> 25;        // 7 bits
#pragma unroll 1
            for (int probe = 0; probe < kTileMap; ++probe) {
                const uint32_t old = atomicCAS(&m.tile[h], kAggEmpty, tile);
                if (old == kAggEmpty || old == tile) { pr[r] = (h << 16) | atomicAdd(&m.cnt[h], 1u); break; }
                h = (h + 1) & (kTileMap - 1);
            }
        }
        __syncthreads();
        if (threadIdx.x < kTileMap && m.tile[threadIdx.x] != kAggEmpty)
            m.base[threadIdx.x] = atomicAdd(&tb.fill[m.tile[threadIdx.x]], m.cnt[threadIdx.x]);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (pr[r] == 0xffffffffu) continue;
        const int q = threadIdx.x + r * NTHREADS;
        const uint32_t c = keys[q];
        const double v0 = vals[agg_at<NSLOTS>(q, 0)], v1 = vals[agg_at<NSLOTS>(q, 1)], v2 = vals[agg_at<NSLOTS>(q, 2)], v3 = vals[agg_at<NSLOTS>(q, 3)];
        bool placed = false;
        if (pr[r] != 0xfffffffeu) {
            const uint32_t slot = pr[r] >> 16, rank = pr[r] & 0xffffu;
            const uint32_t tile = m.tile[slot], pos = m.base[slot] + rank;
            if (pos < tb.cap[tile]) {
                uint32_t t2, local;
                tile_of(tb.tg, c, &t2, &local);
                const size_t at = (size_t)tb.off[tile] + pos;
                tb.cell[at] = local;
                double2* o = reinterpret_cast<double2*>(tb.val + 4 * at);
                o[0] = make_double2(v0, v1);
                o[1] = make_double2(v2, v3);
                placed = true;
            }
        }
        if (!placed) {
            atomic_add_f64(&dst0[c], v0);
            atomic_add_f64(&dst3[3 * (size_t)c + 0], v1);
            atomic_add_f64(&dst3[3 * (size_t)c + 1], v2);
            atomic_add_f64(&dst3[3 * (size_t)c + 2], v3);
            if (touched) touched[c] = 1;
        }
    }
}

// One workgroup per tile of 8 x 8 x 8 cells: the tile's bucket is summed into a dense LDS accumulator (ds_add_f64: ~37 x the global
// atomics' rate), then the tile goes to the cell arrays with plain read-modify-writes -- this workgroup is the tile's only writer.
// FINISH 0: that is all (z-slabs: the reverse-halo sums of the neighbours come in before the cells are finished).
// FINISH 1 (single domain, void-fraction deposit): the cell's sums are complete here, so setCellVolFraction (k_finalize_cells) happens in the
//          same pass -- no round trip of the accumulators through memory, no second sweep over the touched flags.
// FINISH 2 (single domain, momentum-source back-scatter): likewise k_fold_sources (uSourceDrag += D, uSource += uParticle D).
// Same operations on the same operands in the same order as the separate kernels: identical bits.  With FINISH != 0 every tile runs,
// also one whose bucket is empty (its cells may have been reached by the fallback atomics).
// FINISH 3 / 4 (z-slabs): FINISH 1 / 2 for the tiles of the z-layers [fin.tk_lo, fin.tk_hi) -- planes that receive nothing from a neighbour's reverse halo,
//          whose cells' sums are therefore complete here -- and FINISH 0 for the others, in ONE launch: the finish of the interior no longer costs a
//          pass of its own over the accumulators (k_finalize_cells / k_fold_sources keep the few planes at the slab's ends).
struct TileFinish { const double* vol; double* alpha; double* uParticle; double* R; const double* uParticleC; double* uSourceDrag; int tk_lo, tk_hi; };
template <int FINISH>
__global__ __launch_bounds__(256) void k_tile_reduce(TileBuckets tb, double* __restrict__ dst0, double* __restrict__ dst3, unsigned char* __restrict__ touched, TileFinish fin) {
    __shared__ double acc[kTileCells * 4];
    const uint32_t tile = blockIdx.x;
    uint32_t cnt = tb.fill[tile];
    const uint32_t cap = tb.cap[tile];
    if (cnt > cap) cnt = cap;
    // (block-uniform; a compile-time constant for FINISH 0 .. 2)
    const int tkz = (int)(tile / (uint32_t)(tb.tg.ntx * tb.tg.nty));
    const int mode = FINISH >= 3 ? ((tkz >= fin.tk_lo && tkz < fin.tk_hi) ? FINISH - 2 : 0) : FINISH;
    if (mode == 0 && cnt == 0) return;
    if (cnt) {
        for (int q = threadIdx.x; q < kTileCells * 4; q += 256) acc[q] = 0.0;
        __syncthreads();
        const size_t off = tb.off[tile];
        for (uint32_t j = threadIdx.x; j < cnt; j += 256) {
            const uint32_t l = tb.cell[off + j];
            const double2* v = reinterpret_cast<const double2*>(tb.val + 4 * (off + j));
            const double2 v0 = v[0], v1 = v[1];
            lds_add_f64(&acc[agg_at<kTileCells>(l, 0)], v0.x); lds_add_f64(&acc[agg_at<kTileCells>(l, 1)], v0.y);
            lds_add_f64(&acc[agg_at<kTileCells>(l, 2)], v1.x); lds_add_f64(&acc[agg_at<kTileCells>(l, 3)], v1.y);
        }
        __syncthreads();
    }
    const TileGrid& tg = tb.tg;
    const int ti = (int)(tile % (uint32_t)tg.ntx), tj = (int)((tile / (uint32_t)tg.ntx) % (uint32_t)tg.nty), tk = (int)(tile / (uint32_t)(tg.ntx * tg.nty));
    for (int l = threadIdx.x; l < kTileCells; l += 256) {
        const int i = ti * 8 + (l & 7), j = tj * 8 + ((l >> 3) & 7), k = tk * 8 + (l >> 6);
        if (i >= tg.nx || j >= tg.ny || k >= tg.nzs) continue;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        if (cnt) { a0 = acc[agg_at<kTileCells>(l, 0)]; a1 = acc[agg_at<kTileCells>(l, 1)]; a2 = acc[agg_at<kTileCells>(l, 2)]; a3 = acc[agg_at<kTileCells>(l, 3)]; }
        const bool any = !(a0 == 0.0 && a1 == 0.0 && a2 == 0.0 && a3 == 0.0);
        const size_t c = (size_t)i + (size_t)tg.nx * ((size_t)j + (size_t)tg.ny * (size_t)k);
        if (mode == 0) {
            if (!any) continue;
            dst0[c] += a0;
            double* d = dst3 + 3 * c;
            const double d0 = d[0] + a1, d1 = d[1] + a2, d2 = d[2] + a3;
            d[0] = d0; d[1] = d1; d[2] = d2;
            if (touched) touched[c] = 1;
        } else if (mode == 1) {
            // the accumulators hold what the fallback atomics left (touched says so); they go back to zero, the cell is finished
            const bool t = touched[c] != 0;
            if (!any && !t) continue;
            double pv = a0, u0 = a1, u1 = a2, u2 = a3;
            if (t) {
                double* d = dst3 + 3 * c;
                if (any) { pv = dst0[c] + a0; u0 = d[0] + a1; u1 = d[1] + a2; u2 = d[2] + a3; }
                else { pv = dst0[c]; u0 = d[0]; u1 = d[1]; u2 = d[2]; }
                dst0[c] = 0.0; d[0] = 0.0; d[1] = 0.0; d[2] = 0.0;
                touched[c] = 0;
            }
            const double V = fin.vol[c];                      // setCellVolFraction FoamYade.C:318-328
            const double pvolC = 1.0 - (pv / V);
            const double al = ((pvolC > 0.10) ? pvolC : 0.10);
            fin.alpha[c] = al;
            if (fin.R) fin.R[8 * c + 3] = al;
            double* o = fin.uParticle + 3 * c;
            o[0] = u0 / V; o[1] = u1 / V; o[2] = u2 / V;
        } else {
            const double D0 = dst0[c];                        // what the fallback atomics added to the drag accumulator (usually nothing)
            double D = D0;
            double* sp = dst3 + 3 * c;
            if (!any && D0 == 0.0) continue;
            double s0 = sp[0], s1 = sp[1], s2 = sp[2];
            if (any) { D += a0; s0 += a1; s1 += a2; s2 += a3; }
            if (D0 != 0.0) dst0[c] = 0.0;                     // the accumulator is empty again either way
            if (D != 0.0) {                                   // k_fold_sources, FoamYade.C:385-386
                fin.uSourceDrag[c] += D;
                const double* up = fin.uParticleC + 3 * c;
                s0 = s0 + (up[0] * D); s1 = s1 + (up[1] * D); s2 = s2 + (up[2] * D);
            }
            sp[0] = s0; sp[1] = s1; sp[2] = s2;
        }
    }
}

// buildCellPartList FoamYade.C:265-288: pVol*w and (w*v)*pVol per (particle, cell) pair into the per-batch accumulators
#ifndef FY_DEP_THREADS
#define FY_DEP_THREADS 512
#endif
#ifndef FY_DEP_LOG2
#define FY_DEP_LOG2 10
#endif
constexpr int kDepThreads = FY_DEP_THREADS, kDepLog2 = FY_DEP_LOG2;      // 1024 slots x (4 + 32) B = 36 KiB of LDS (2048: 1.46 vs 1.36 ms for k_locate_deposit -- occupancy)
// table crowded (practically never): straight to memory.  Out of line -- k_locate_deposit inlines deposit_pair 24 times
__device__ __attribute__((noinline)) void deposit_direct(int32_t cid, double c0, double c1, double c2, double c3, double* __restrict__ pvol_acc,
                                                         double* __restrict__ up_acc, unsigned char* __restrict__ touched) {
    atomic_add_f64(&pvol_acc[cid], c0);
    atomic_add_f64(&up_acc[3 * (size_t)cid + 0], c1);
    atomic_add_f64(&up_acc[3 * (size_t)cid + 1], c2);
    atomic_add_f64(&up_acc[3 * (size_t)cid + 2], c3);
    touched[cid] = 1;
}

// one (particle, cell) contribution into the workgroup's table (or straight to memory when the table is crowded)
__device__ __forceinline__ void deposit_pair(uint32_t* keys, double* vals, int32_t cid, double c0, double c1, double c2, double c3,
                                             double* __restrict__ pvol_acc, double* __restrict__ up_acc, unsigned char* __restrict__ touched) {
    const int h = agg_slot<kDepLog2>(keys, (uint32_t)cid);
    if (h >= 0) {
        // the LDS atomic unit (~1.4 ds_add_f64 lanes per clock per CU, tools/micro/lds_atomic_rate.hip) is what bounds this half of the
        // kernel: adding an exact zero is the identity, so the momentum terms of a particle at rest are not issued at all
        constexpr int N = 1 << kDepLog2;
        lds_add_f64(&vals[agg_at<N>(h, 0)], c0);
        if (c1 != 0.0) lds_add_f64(&vals[agg_at<N>(h, 1)], c1);
        if (c2 != 0.0) lds_add_f64(&vals[agg_at<N>(h, 2)], c2);
        if (c3 != 0.0) lds_add_f64(&vals[agg_at<N>(h, 3)], c3);
    } else {
        deposit_direct(cid, c0, c1, c2, c3, pvol_acc, up_acc, touched);
    }
}
// work != nullptr: the particles listed there (k_locate_deposit's leftovers, placed by the walk), shared by a fixed grid
#ifndef FY_DEP_ATTR
#define FY_DEP_ATTR
#endif
#ifndef FY_DEP_REEXP
#define FY_DEP_REEXP 1      // the deposit loop forms exp(-d2 ...) a second time instead of reading back a stored unnormalised weight: one store per pair less (1.12 -> 1.06 ms)
#endif
__global__ __launch_bounds__(kDepThreads) FY_DEP_ATTR void k_deposit(ParticleSoA p, int64_t n, GaussParams gp, CellWindow cw, double* __restrict__ pvol_acc,
                                                          double* __restrict__ up_acc, unsigned char* __restrict__ touched,
                                                          const int32_t* __restrict__ work, const unsigned int* __restrict__ work_n, TileBuckets tb) {
    __shared__ uint32_t keys[1 << kDepLog2];
    __shared__ double vals[(1 << kDepLog2) * 4];
    __shared__ TileMapLds tmap;
    if (work) { n = (int64_t)*work_n; if ((int64_t)blockIdx.x * kDepThreads >= n) return; }      // (block-uniform)
    for (int q = threadIdx.x; q < (1 << kDepLog2); q += kDepThreads) {
        keys[q] = kAggEmpty;
        vals[q] = 0.0; vals[q + (1 << kDepLog2)] = 0.0; vals[q + 2 * (1 << kDepLog2)] = 0.0; vals[q + 3 * (1 << kDepLog2)] = 0.0;
    }
    __syncthreads();
    for (int64_t idx = (int64_t)blockIdx.x * kDepThreads + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * kDepThreads) {
        const int64_t i = work ? (int64_t)work[idx] : idx;
        const int chain = p.chain_len[i];
        const int k = chain < kMaxK ? chain : kMaxK;
        if (k > 0) {
            const double dia = 2 * p.rad[i];                                  // FoamYade.C:219
            const double pVol = M_PI * cube3(dia) / 6.0;                      // FoamYade.H:36 (as k_locate_deposit forms it)
            const double vx = p.vx[i], vy = p.vy[i], vz = p.vz[i];
            // calcInterpWeightGaussian FoamYade.C:301-314, slots visited in ascending-d2 order (= reverse push order).  Round 6: the unnormalised weights no longer stay in 16
            // register pairs across the deposit loop (113 -> 70 VGPRs: three workgroups per CU instead of two, and the kernel waits on memory 71 % of its wave cycles: 1.18 ->
            // 1.09 ms); the loop below reads each squared distance again (cached), forms the same weight again, scales it and stores the normalised weight -- the same products.
            // (waves_per_eu(8) -- 64 VGPRs, four workgroups -- spills 32 B and is slower: 1.16.)
            double allwt = 0.0;
#pragma unroll
            for (int t = 0; t < kMaxK; ++t) {
                if (t < k) {
                    const size_t slot = (size_t)((chain - 1 - t) & (kMaxK - 1)) * p.cap + (size_t)i;
                    const double wt = exp(-p.w[slot] * (1.0 / gp.two_sigma2)) * gp.range_cu * gp.sigma_pi;      // (the expressions of k_locate_deposit's list path, so that
#if !FY_DEP_REEXP
                    p.w[slot] = wt;                                                                             //  a particle's weights do not depend on which path placed it)
#endif
                    allwt += wt;
                }
            }
            const double rallwt_ = 1.0 / allwt;
            size_t slot = (size_t)((chain - 1) & (kMaxK - 1)) * p.cap + (size_t)i;
            double w_next = p.w[slot];
            int32_t id_next = p.ids[slot];
#pragma unroll 1
            for (int t = 0; t < k; ++t) {
#if FY_DEP_REEXP
                const double weight = (exp(-w_next * (1.0 / gp.two_sigma2)) * gp.range_cu * gp.sigma_pi) * rallwt_;
#else
                const double weight = w_next * rallwt_;                       // FoamYade.C:312-314
#endif
                const int64_t cl = (int64_t)id_next - cw.base;                // storage index (slab window)
                p.w[slot] = weight;
                if (t + 1 < k) {
                    slot = (size_t)((chain - 2 - t) & (kMaxK - 1)) * p.cap + (size_t)i;
                    w_next = p.w[slot]; id_next = p.ids[slot];
                }
                if (cl < 0 || cl >= cw.n_field) continue;
                const int32_t cid = (int32_t)cl;
                const double c0 = pVol * weight, c1 = (weight * vx) * pVol, c2 = (weight * vy) * pVol, c3 = (weight * vz) * pVol;
                deposit_pair(keys, vals, cid, c0, c1, c2, c3, pvol_acc, up_acc, touched);
            }
        }
    }
    __syncthreads();
    flush_table<(1 << kDepLog2), kDepThreads>(keys, vals, tmap, tb, pvol_acc, up_acc, touched);
}

// k_locate_lists and k_deposit in one pass over the particles: the list scan keeps the unnormalised Gaussian weight of every listed
// node in a register (zero where the node did not enter the chain; slots are static because the scan is unrolled), so the chain's
// squared distances never travel through memory: ids and NORMALISED weights are stored once, for k_force_gaussian.  Same arithmetic
// as the two kernels: allwt adds the weights from the last push to the first (adding the zeros in between is exact).
// rec != nullptr: the binned SoA arrays are not filled yet -- the lane fetches its wire record through the placement (p.orig) itself and
// leaves the SoA copy behind for the force pass (k_bin_gather folded into this kernel: one launch and one coalesced read of the SoA less;
// the record fetch is the same random 80-byte read either way)
__global__ __launch_bounds__(kDepThreads) FY_LD_ATTR void k_locate_deposit(LocateLists ll, ImplicitGeom ig, ParticleSoA p, int64_t n,
                                                                 GaussParams gp, SlabOwn own, CellWindow cw, double* __restrict__ pvol_acc,
                                                                 double* __restrict__ up_acc, unsigned char* __restrict__ touched, TileBuckets tb,
                                                                 const double* __restrict__ rec) {
    const unsigned short* __restrict__ lists = ll.lists;
    int32_t* __restrict__ fb_list = ll.fb_list;
    unsigned int* __restrict__ fb_count = ll.fb_count;
    __shared__ uint32_t keys[1 << kDepLog2];
    __shared__ double vals[(1 << kDepLog2) * 4];
    __shared__ TileMapLds tmap;
    for (int q = threadIdx.x; q < (1 << kDepLog2); q += kDepThreads) {
        keys[q] = kAggEmpty;
        vals[q] = 0.0; vals[q + (1 << kDepLog2)] = 0.0; vals[q + 2 * (1 << kDepLog2)] = 0.0; vals[q + 3 * (1 << kDepLog2)] = 0.0;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kDepThreads + threadIdx.x;
    // an 80-byte record of a 16-byte aligned array is five aligned 16-byte words, of which this kernel wants words 0, 1, 2 and 4
    const bool rec16 = rec && (reinterpret_cast<uintptr_t>(rec) & 15) == 0;      // (uniform)
#if FY_LD_QUAD
    // the four lanes of a quad fetch one record per instruction (lane q: word q, lane 3: word 4) -- a wave instruction looks up 16 record
    // starts instead of 64 -- and the words go home to the record's lane by DPP quad_perm moves (same values, no arithmetic involved)
    double2 qa = make_double2(0, 0), qb = qa, qc = qa, qe = qa;
    if (rec16) {
        const int lq = (int)(threadIdx.x & 3u);
        const int32_t o = i < n ? p.orig[i] : 0;
        const uint4* r4 = reinterpret_cast<const uint4*>(rec) + (lq < 3 ? lq : 4);
        uint4 g0 = r4[5 * (size_t)quad_bcast<0>(o)], g1 = r4[5 * (size_t)quad_bcast<1>(o)], g2 = r4[5 * (size_t)quad_bcast<2>(o)],
              g3 = r4[5 * (size_t)quad_bcast<3>(o)];      // round j: the quad's four words of lane j's record
        quad_transpose(g0, g1, g2, g3, lq);              // -> register w: word w (lane 3's: word 4) of MY record
        qa = as_double2(g0); qb = as_double2(g1); qc = as_double2(g2); qe = as_double2(g3);
    }
#endif
    // Straight-line (predicated) down to the list row, so that the four lanes of a quad are still together when the rows are fetched
    const bool live = i < n;
    double qx = 0, qy = 0, qz = 0, pvx = 0, pvy = 0, pvz = 0, prad = 0;
    if (live) {
        if (rec) {
            const double* r = rec + 10 * (size_t)p.orig[i];
            if (rec16) {
#if FY_LD_QUAD
                const double2 a = qa, b = qb, cc = qc, e = qe;
#else
                const double2* r2 = reinterpret_cast<const double2*>(r);
                const double2 a = r2[0], b = r2[1], cc = r2[2], e = r2[4];
#endif
                qx = a.x; qy = a.y; qz = b.x; pvx = b.y; pvy = cc.x; pvz = cc.y; prad = e.y;
            } else {
                qx = r[0]; qy = r[1]; qz = r[2]; pvx = r[3]; pvy = r[4]; pvz = r[5]; prad = r[9];
            }
            p.px[i] = qx; p.py[i] = qy; p.pz[i] = qz; p.vx[i] = pvx; p.vy[i] = pvy; p.vz[i] = pvz; p.rad[i] = prad;
        } else {
            qx = p.px[i]; qy = p.py[i]; qz = p.pz[i]; pvx = p.vx[i]; pvy = p.vy[i]; pvz = p.vz[i]; prad = p.rad[i];
        }
    }
    bool mine = live;
    if (live && own.active) {                                // another slab's particle: not located here (k = 0)
        int kz = (int)floor((qz - own.oz) / own.dx);
        kz = min(max(kz, 0), own.nzglob - 1);
        if (!(qz == qz) || !own_planes(own, p.orig[i], kz, qx, qy, qz)) { p.chain_len[i] = 0; mine = false; }
    }
    const double hdx = 0.5 * ig.dx;
    // Reciprocals instead of FP64 divisions (round 5; ~35 instructions each, 22 per particle: same-box A/B 1.167 -> 1.127 ms).  What they may move by an
    // ulp cannot change an answer: the lattice coordinate only picks the (cell, octant) list, and a particle within 2 kListEps of a cell face goes to
    // the walk anyway; the Gaussian's argument and the normalisation feed weights that are held to 1e-10, not to the bit.  Every comparison that steers
    // the chain (d < best, d < maxdist) is formed exactly as the reference forms it.
    const double rdx_ = 1.0 / ig.dx;
    const double sx = (qx - ig.ox) * rdx_, sy = (qy - ig.oy) * rdx_, sz = (qz - ig.oz) * rdx_;
    const double fx = floor(sx), fy_ = floor(sy), fz = floor(sz);
    const double tx = sx - fx, ty = sy - fy_, tz = sz - fz;
    const double tlo = 2 * kListEps, thi = 1.0 - 2 * kListEps;
    int sclass = 0;
    bool ok = mine && fx >= 0 && fx < ig.nx && fy_ >= 0 && fy_ < ig.ny && fz >= 0 && fz < ig.nz &&
              tx >= tlo && tx <= thi && ty >= tlo && ty <= thi && tz >= tlo && tz <= thi;      // (false for NaN)
    const int ci = ok ? (int)fx : 0, cj = ok ? (int)fy_ : 0, ck = ok ? (int)fz : 0;
    uint4 v[kListLen / 8];
#pragma unroll
    for (int ch = 0; ch < kListLen / 8; ++ch) v[ch] = make_uint4(kListEnd | (kListEnd << 16), 0, 0, 0);
    int32_t rowi = 0;                                        // (cell, octant) row of the lists; row 0 where there is none to fetch
    {
        const double cx = ig.ox + (double)(2 * ci + 1) * hdx, cy = ig.oy + (double)(2 * cj + 1) * hdx, cz = ig.oz + (double)(2 * ck + 1) * hdx;
        const int oct = (qx - cx < 0.0 ? 0 : 1) | (qy - cy < 0.0 ? 0 : 2) | (qz - cz < 0.0 ? 0 : 4);
        const int64_t cell = (int64_t)ci + (int64_t)ig.nx * ((int64_t)cj + (int64_t)ig.ny * (int64_t)ck) - ll.cell0;
        ok = ok && cell >= 0 && cell < ll.n_listed;          // (a slab lists its own planes only)
        if (ok) rowi = (int32_t)(cell * 8 + oct);
    }
    {
        // all three 16-byte chunks of the row at once (one or two lines): fetched only where the list goes on, the second and the third each cost
        // the wave a dependent memory round trip (round 5); what lies behind a list's end mark is never looked at
        const uint4* __restrict__ L4 = reinterpret_cast<const uint4*>(lists);
        static_assert(kListLen * sizeof(unsigned short) == 3 * sizeof(uint4), "a list row is three 16-byte words");
#if FY_LD_QUADROW
        // quad-cooperative as the record fetch above: round j, lanes 0..2 fetch their word of quad lane j's row (48 contiguous bytes per quad and
        // instruction instead of 16 bytes from each of four rows), lane 3 fetches nothing; then the 4 x 4 transpose
        const int lq = (int)(threadIdx.x & 3u);
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, c2 = c0, c3 = c0;
        const int32_t w0 = quad_bcast<0>(rowi), w1 = quad_bcast<1>(rowi), w2 = quad_bcast<2>(rowi), w3 = quad_bcast<3>(rowi);
        if (lq < 3) { c0 = L4[3 * (size_t)w0 + lq]; c1 = L4[3 * (size_t)w1 + lq]; c2 = L4[3 * (size_t)w2 + lq]; c3 = L4[3 * (size_t)w3 + lq]; }
        quad_transpose(c0, c1, c2, c3, lq);              // -> c0, c1, c2: the three words of MY row (c3: lane 3's, nothing)
#else
        uint4 c0 = make_uint4(0, 0, 0, 0), c1 = c0, c2 = c0;
        if (ok) { c0 = L4[3 * (size_t)rowi]; c1 = L4[3 * (size_t)rowi + 1]; c2 = L4[3 * (size_t)rowi + 2]; }
#endif
        if (ok) {
            v[0] = c0;
            ok = (c0.x & 0xffffu) != kListOverflow;
            if (ok && (c0.w >> 16) != kListEnd) { v[1] = c1; sclass = 1; if ((c1.w >> 16) != kListEnd) { v[2] = c2; sclass = 2; } }
        }
    }
    if (mine) {
        {
            if (p.scan_class) p.scan_class[i] = (unsigned char)(ok ? sclass : 2);
            if (!ok) {
                const unsigned int at = atomicAdd(fb_count, 1u);
                fb_list[at] = (int32_t)i;
            } else {
                // The scan keeps nothing per list position: an entry that enters the chain has its id and its UNNORMALISED weight stored at once, in the row of its push
                // index; the sum (last push first, FoamYade.C:301-311: the same additions as over the 24 positions with their zeros) and the normalisation + deposit
                // then run as loops over the CHAIN (k entries) that read the rows back (this lane's own stores: L2) -- not as 24 unrolled, mostly idle copies
                // of the deposit, and without 24 weights held in registers
                double best = 1e300;
                int pos = 0;
                bool done = false;
#pragma unroll
                for (int h = 0; h < kListLen; ++h) {
                    const uint4 vv = v[h >> 3];
                    const uint32_t wd = ((h >> 1) & 3) == 0 ? vv.x : (((h >> 1) & 3) == 1 ? vv.y : (((h >> 1) & 3) == 2 ? vv.z : vv.w));
                    const uint32_t code = (wd >> ((h & 1) * 16)) & 0xffffu;
                    if (code == kListEnd) done = true;
                    if (!done) {
                        const int ni = ci + (int)(code & 15u) - 8, nj = cj + (int)((code >> 4) & 15u) - 8, nk = ck + (int)((code >> 8) & 15u) - 8;
                        const double a = qx - (ig.ox + (double)(2 * ni + 1) * hdx), b = qy - (ig.oy + (double)(2 * nj + 1) * hdx),
                                     c = qz - (ig.oz + (double)(2 * nk + 1) * hdx);
                        double d = a * a;                    // meshTree.C:54-64
                        d += b * b;
                        d += c * c;
                        if (d < best) {                      // meshTree.C:192
                            best = d;
                            if (d < gp.maxdist && !(code & kListNoEmit)) {       // meshTree.C:195; the root is never pushed
                                const size_t slot = (size_t)(pos & (kMaxK - 1)) * p.cap + (size_t)i;
                                p.ids[slot] = ni + ig.nx * (nj + ig.ny * nk);
                                const double wu = exp(-d * (1.0 / gp.two_sigma2)) * gp.range_cu * gp.sigma_pi;     // FoamYade.C:308
                                p.w[slot] = wu;
                                ++pos;
                            }
                        }
                    }
                }
                const int chain = pos;
                p.chain_len[i] = chain;
                if (chain > 0) {
                    const int k = chain < kMaxK ? chain : kMaxK;      // (a chain longer than the ring has lost its oldest rows, as in the walk: k newest)
                    double allwt = 0.0;
                    {   // all the rows at once (one round trip), then the reference's order: last push first (FoamYade.C:301-311); rows beyond the chain read as zero,
                        // and adding them is exact -- the same additions as round 5's sum over the 24 list positions
                        double u[kMaxK];
#pragma unroll
                        for (int t = 0; t < kMaxK; ++t) u[t] = t < k ? p.w[(size_t)((chain - k + t) & (kMaxK - 1)) * p.cap + (size_t)i] : 0.0;
#pragma unroll
                        for (int t = kMaxK - 1; t >= 0; --t) allwt += u[t];
                    }
                    const double rallwt_ = 1.0 / allwt;
                    const double dia = 2 * prad;                                      // FoamYade.C:219
                    const double pVol = M_PI * cube3(dia) / 6.0;                      // FoamYade.H:36
                    const double vx = pvx, vy = pvy, vz = pvz;
                    size_t slot = (size_t)((chain - k) & (kMaxK - 1)) * p.cap + (size_t)i;
                    double w_n = p.w[slot];
                    int32_t id_n = p.ids[slot];
#pragma unroll 1
                    for (int t = chain - k; t < chain; ++t) {
                        const double weight = w_n * rallwt_;                          // FoamYade.C:312-314
                        const int32_t id_t = id_n;
                        p.w[slot] = weight;
                        if (t + 1 < chain) {                                          // (the next entry travels while this one goes through the table)
                            slot = (size_t)((t + 1) & (kMaxK - 1)) * p.cap + (size_t)i;
                            w_n = p.w[slot]; id_n = p.ids[slot];
                        }
                        const int64_t cl = (int64_t)id_t - cw.base;                   // storage index (slab window)
                        if (cl >= 0 && cl < cw.n_field) {
                            const double c0 = pVol * weight, c1 = (weight * vx) * pVol, c2 = (weight * vy) * pVol, c3 = (weight * vz) * pVol;
                            deposit_pair(keys, vals, (int32_t)cl, c0, c1, c2, c3, pvol_acc, up_acc, touched);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    flush_table<(1 << kDepLog2), kDepThreads>(keys, vals, tmap, tb, pvol_acc, up_acc, touched);
}

// setCellVolFraction FoamYade.C:318-328: assignment on the cells this batch touched; accumulators reset for the next batch
__global__ __launch_bounds__(256) void k_finalize_cells(int32_t n_cells, const double* __restrict__ vol, double* __restrict__ pvol_acc,
                                                        double* __restrict__ up_acc, unsigned char* __restrict__ touched,
                                                        double* __restrict__ alpha, double* __restrict__ uParticle, double* __restrict__ R) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    if (!touched[c]) return;
    touched[c] = 0;
    const double V = vol[c];
    const double pvolC = 1.0 - (pvol_acc[c] / V);
    const double a = ((pvolC > 0.10) ? pvolC : 0.10);
    alpha[c] = a;
    if (R) R[8 * (size_t)c + 3] = a;       // the force pass reads alpha out of the packed cell record (k_pack_cells)
    double* up = up_acc + 3 * (size_t)c;
    double* o = uParticle + 3 * (size_t)c;
    o[0] = up[0] / V; o[1] = up[1] / V; o[2] = up[2] / V;
    pvol_acc[c] = 0.0; up[0] = 0.0; up[1] = 0.0; up[2] = 0.0;
}

// ------------------------------------------------------------------------------------------------ force + back-scatter
// What bounded the round-1 force pass was the vector-memory pipeline: 13 eight-byte gathers per (particle, cell) pair out of five
// cell arrays (U, alpha, gradP, divT and -- in the back-scatter loop -- uParticle), 632 M L1 accesses per launch.  Two changes:
//   * the interpolated cell data comes from ONE 64-byte record per cell, CellRec = {U[3], alpha, A[3], V} with
//     A = 2 nu rho_f divT - gradP (the two Archimedes terms of FoamYade.C:416-426 are both interpolated with the same weights, so their
//     per-cell combination is interpolated instead): four 16-byte loads per pair, all from one line, instead of ten 8-byte ones from four;
//   * the drag share of the momentum source, uSource[c] += -coeff w uParticle[c] / rho_f (FoamYade.C:386), has the per-CELL factor
//     uParticle[c]; it is pulled out of the per-pair loop: the pairs only accumulate D[c] = sum -coeff w / rho_f (which is the
//     reference's uSourceDrag contribution, FoamYade.C:385) and k_fold_sources adds D to uSourceDrag and uParticle[c] * D[c] to uSource once
//     per cell -- no uParticle gather per pair.  The Archimedes share -f w / (V rho_f) (FoamYade.C:433) goes into uSource directly.
// Rounding differs from the per-pair form in the last bits only (same terms, summed in another association): covered by the 1e-10 bar.
constexpr int kRecDoubles = 8;      // CellRec: Ux Uy Uz alpha | Ax Ay Az V

__global__ __launch_bounds__(256) void k_pack_cells(int64_t n_field, const double* __restrict__ U, const double* __restrict__ alpha,
                                                    const double* __restrict__ gradP, const double* __restrict__ divT,
                                                    const double* __restrict__ vol, double two_nu, double rhoF, double* __restrict__ R) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_field) return;
    const double* u = U + 3 * (size_t)c;
    const double* g = gradP + 3 * (size_t)c;
    const double* d = divT + 3 * (size_t)c;
    const double u0 = u[0], u1 = u[1], u2 = u[2];
    // FoamYade.C:421-426: -gradP w + ((2 nu divT) w) rho_f, per cell
    const double a0 = ((two_nu * d[0]) * rhoF) - g[0], a1 = ((two_nu * d[1]) * rhoF) - g[1], a2 = ((two_nu * d[2]) * rhoF) - g[2];
    double2* r = reinterpret_cast<double2*>(R + kRecDoubles * (size_t)c);
    r[0] = make_double2(u0, u1);
    r[1] = make_double2(u2, alpha[c]);
    r[2] = make_double2(a0, a1);
    r[3] = make_double2(a2, vol[c]);
}

// alpha of the listed cells only (ghost planes after their halo exchange)
__global__ __launch_bounds__(256) void k_patch_rec_alpha(int64_t c0, int64_t n, const double* __restrict__ alpha, double* __restrict__ R) {
    const int64_t c = c0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c < c0 + n) R[kRecDoubles * (size_t)c + 3] = alpha[c];
}

// after all pairs of a batch: uSourceDrag += D, uSource += uParticle * D (FoamYade.C:385-386), D reset for the next batch
__global__ __launch_bounds__(256) void k_fold_sources(int64_t n_field, double* __restrict__ drag_acc, const double* __restrict__ uParticle,
                                                      double* __restrict__ uSourceDrag, double* __restrict__ uSource) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_field) return;
    const double D = drag_acc[c];
    if (D == 0.0) return;
    drag_acc[c] = 0.0;
    uSourceDrag[c] += D;
    const double* up = uParticle + 3 * (size_t)c;
    double* s = uSource + 3 * (size_t)c;
    const double s0 = s[0] + (up[0] * D), s1 = s[1] + (up[1] * D), s2 = s[2] + (up[2] * D);
    s[0] = s0; s[1] = s1; s[2] = s2;
}

// per-particle part of hydroDragForce + archimedesForce (+ the opt-in models) from the interpolated values (FoamYade.C:366-382, 426-427,
// 402-404, 477-478); returns the drag coefficient and the Archimedes (+ added-mass) force whose reaction is scattered, stores F
struct Interp { double ufx, ufy, ufz, alpha_f, pv, sax, say, saz; };
struct ModelSums { double t1, t2, t3, dux, duy, duz, pva; };
struct ParticleForce { double coeff, bx, by, bz; };

__device__ __forceinline__ ParticleForce force_law(const ForceParams& fp, const Interp& s, const ModelSums& ms, int k, double dia, double lvx, double lvy,
                                                   double lvz, const double* __restrict__ rec_orig, double* __restrict__ F) {
    const double rhoF = fp.rhoF, nu = fp.nu;
    const double alpha_p = 1 - s.alpha_f;                                             // FoamYade.C:366
    const double urx = s.ufx - lvx, ury = s.ufy - lvy, urz = s.ufz - lvz;
    const double magUR = sqrt(urx * urx + ury * ury + urz * urz);
    const double Re = fp.small + ((magUR * dia) / nu);                                // FoamYade.C:370
    const double cd = Re < 1000 ? (24 / (Re)) * (1 + (0.15 * pow(Re, 0.687))) : 0.44;
    double coeff;
    if (s.alpha_f > 0.8) {                                                            // FoamYade.C:373-374
        coeff = 0.75 * cd * s.alpha_f * alpha_p * rhoF * magUR * pow(s.alpha_f, -2.65);
    } else {                                                                          // FoamYade.C:376-378
        const double cf1 = 150 * ((alpha_p * alpha_p) / s.alpha_f) * ((nu * rhoF) / (dia * dia));
        const double cf2 = 1.75 * alpha_p * rhoF * (1 / dia) * magUR;
        coeff = cf1 + cf2;
    }
    const double s1 = s.pv * coeff, ia = 1 / (alpha_p);                               // FoamYade.C:381
    const double hfx = (s1 * urx) * ia, hfy = (s1 * ury) * ia, hfz = (s1 * urz) * ia;
    const double afx = s.pv * s.sax, afy = s.pv * s.say, afz = s.pv * s.saz;          // FoamYade.C:426
    double fx = (0.0 + hfx) + afx, fy_ = (0.0 + hfy) + afy, fz = (0.0 + hfz) + afz;   // FoamYade.C:382,427
    double tqx = 0.0, tqy = 0.0, tqz = 0.0;                                           // Gaussian torque disabled, FoamYade.C:618
    ParticleForce r{coeff, afx, afy, afz};
    if (fp.models & FY_FORCE_GAUSSIAN_TORQUE) {                                       // FoamYade.C:477-478
        const double c3 = M_PI * (pow(dia, 3.0));
        tqx = 0.0 + (((c3 * (ms.t1 - rec_orig[6])) * nu) * rhoF);
        tqy = 0.0 + (((c3 * (ms.t2 - rec_orig[7])) * nu) * rhoF);
        tqz = 0.0 + (((c3 * (ms.t3 - rec_orig[8])) * nu) * rhoF);
    }
    if (fp.models & FY_FORCE_ADDED_MASS) {                                            // FoamYade.C:402-404
        const double pva = ms.pva / (double)(unsigned)k;
        const double amx = (pva * (ms.dux - (lvx / fp.delta_t))) * fp.rhoP;
        const double amy = (pva * (ms.duy - (lvy / fp.delta_t))) * fp.rhoP;
        const double amz = (pva * (ms.duz - (lvz / fp.delta_t))) * fp.rhoP;
        fx = fx + amx; fy_ = fy_ + amy; fz = fz + amz;
        r.bx = r.bx + amx; r.by = r.by + amy; r.bz = r.bz + amz;                      // FoamYade.C:406-411: same -f w / (V rho_f) reaction
    }
    F[0] = fx; F[1] = fy_; F[2] = fz;
    if (!fp.torque_prezeroed) { F[3] = tqx; F[4] = tqy; F[5] = tqz; }    // permuted 48-byte records: half the store traffic when the torque is identically zero
    return r;
}

#if FY_FORCE_QUAD
// round J of a quad's cooperative gather, in two halves so that the four rounds' loads are all in flight before the first is consumed:
// the load of chunk lq of the 64-byte record of cell cj (quad lane J's; < 0: none -- record 0 is fetched and dropped) ...
__device__ __forceinline__ double2 quad_fetch(const double2* __restrict__ R2, int32_t cj, int lq) {
    return R2[(size_t)(cj > 0 ? cj : 0) * (kRecDoubles / 2) + (size_t)lq];
}
// ... and, times lane J's weight, onto the running pair of sums this lane keeps for lane J's particle (FoamYade.C:361-365, 421-424: product
// first, then the addition, as interp_add)
__device__ __forceinline__ void quad_accumulate(double2& acc, double2 r, int32_t cj, double wj) {
    if (cj >= 0) { acc.x += (r.x * wj); acc.y += (r.y * wj); }
}
#else
__device__ __forceinline__ void interp_add(Interp& s, const double* __restrict__ R, int64_t cl, double w, double volp) {
    const double2* r = reinterpret_cast<const double2*>(R + kRecDoubles * (size_t)cl);
    const double2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    s.ufx += (r0.x * w); s.ufy += (r0.y * w); s.ufz += (r1.x * w);                    // FoamYade.C:361-365
    s.alpha_f += (r1.y * w);
    s.pv += (volp * w);
    s.sax += (r2.x * w); s.say += (r2.y * w); s.saz += (r3.x * w);                    // FoamYade.C:421-424 (per-cell combination, k_pack_cells)
}

#endif

__device__ __forceinline__ void model_add(ModelSums& ms, const ForceParams& fp, const double* __restrict__ vGrad, const double* __restrict__ ddtU, int64_t cl,
                                          double w, double volp) {
    if (fp.models & FY_FORCE_GAUSSIAN_TORQUE) {                                       // calcHydroTorque FoamYade.C:468-476
        const double* G = vGrad + 9 * (size_t)cl;                                     // xx xy xz yx yy yz zx zy zz
        ms.t1 += ((G[5] - G[7]) * w); ms.t2 += ((G[6] - G[2]) * w); ms.t3 += ((G[3] - G[1]) * w);
    }
    if (fp.models & FY_FORCE_ADDED_MASS) {                                            // addedMassForce FoamYade.C:396-401
        const double* d = ddtU + 3 * (size_t)cl;
        ms.pva += (volp * w);
        ms.dux = ms.dux + (d[0] * w); ms.duy = ms.duy + (d[1] * w); ms.duz = ms.duz + (d[2] * w);
    }
}

// gather + force law + back-scatter in one kernel, one lane per particle (LDS aggregation table per workgroup, flushed to the tile buckets).
// (As two kernels -- gathers without LDS, then the back-scatter -- it measured 1.01 + 1.02 ms against 1.50 fused; timing-only variants of this
// kernel are built from a copy with tools/build_variant.sh, the shipped source carries none.)
#ifndef FY_FORCE_THREADS
#define FY_FORCE_THREADS 512
#endif
#ifndef FY_FORCE_LOG2
#define FY_FORCE_LOG2 10
#endif
constexpr int kForceThreads = FY_FORCE_THREADS, kForceLog2 = FY_FORCE_LOG2;
__global__ __launch_bounds__(kForceThreads) FY_FORCE_ATTR void k_force_gaussian(
        ParticleSoA p, int64_t n, ForceParams fp, CellWindow cw, const double* __restrict__ vol, const double* __restrict__ R,
        const double* __restrict__ vGrad, const double* __restrict__ ddtU, const double* __restrict__ rec,
        double* __restrict__ drag_acc, double* __restrict__ uSource, double* __restrict__ force_out, TileBuckets tb) {
    constexpr int kSlots = 1 << kForceLog2;
    __shared__ uint32_t keys[kSlots];
    __shared__ double vals[kSlots * 4];
    __shared__ TileMapLds tmap;
    for (int q = threadIdx.x; q < kSlots; q += kForceThreads) {
        keys[q] = kAggEmpty;
        vals[q] = 0.0; vals[q + kSlots] = 0.0; vals[q + 2 * kSlots] = 0.0; vals[q + 3 * kSlots] = 0.0;
    }
    __syncthreads();
#if FY_FORCE_QUAD
    // Quad-cooperative record gathers.  A lane that pulls its own 64-byte cell record issues four 16-byte loads whose 64 lanes touch 64 different
    // lines each: 256 line look-ups per wave and stencil entry, and the line look-up, not the bytes, is what the L1 charges for (round 5's counters:
    // 4.6 L1 accesses per pair, texture-data unit 94 % busy at 0.34 lines per clock).  Here the four lanes of a quad fetch ONE record per instruction:
    // in round j lane q loads 16-byte chunk q of the record quad lane j asks for (id and weight come over by DPP quad_perm moves, no LDS), so a wave
    // instruction touches 16 lines, contiguous 64 bytes each.  No transpose per pair: lane q keeps, for each of the quad's four particles, the running
    // sums of ITS chunk (two doubles) -- the same products added in the same order as the owner lane would add them -- and the four particles' sums
    // go home to their owners once, after the loop.  Identical bits.
    const int64_t i = (int64_t)blockIdx.x * kForceThreads + threadIdx.x;
    const bool live = i < n;
    const size_t ii = live ? (size_t)i : 0;
    const int chain = live ? p.chain_len[ii] : 0;
    const int k = chain < kMaxK ? chain : kMaxK;
    const int first = chain - k;                        // oldest first: see the note on the row order below
    const int lq = (int)(threadIdx.x & 3u);
    const int kq = max(max(quad_bcast<0>(k), quad_bcast<1>(k)), max(quad_bcast<2>(k), quad_bcast<3>(k)));
    const double dia = 2 * p.rad[ii];
    const double volp = M_PI * cube3(dia) / 6.0;
    Interp s{0, 0, 0, 0, 0, 0, 0, 0};
    {
        const double2* __restrict__ R2 = reinterpret_cast<const double2*>(R);
        double2 a0 = make_double2(0, 0), a1 = a0, a2 = a0, a3 = a0;
        size_t slot = (size_t)(first & (kMaxK - 1)) * p.cap + ii;
        int32_t id_n = 0;
        double w_n = 0.0;
        if (k > 0) { id_n = p.ids[slot]; w_n = p.w[slot]; }
        for (int t = 0; t < kq; ++t) {                   // (quad-uniform trip count: the four lanes stay together for the DPP moves)
            const int64_t cl = (int64_t)id_n - cw.base;
            const int32_t c32 = (t < k && cl >= 0 && cl < cw.n_field) ? (int32_t)cl : -1;
            const double w = w_n;
            if (t + 1 < k) {                             // the next entry's id and weight travel while this entry's records do
                slot = (size_t)((first + t + 1) & (kMaxK - 1)) * p.cap + ii;
                id_n = p.ids[slot]; w_n = p.w[slot];
            }
            if (c32 >= 0) s.pv += (volp * w);
            const int32_t c0 = quad_bcast<0>(c32), c1 = quad_bcast<1>(c32), c2 = quad_bcast<2>(c32), c3 = quad_bcast<3>(c32);
            const double w0 = quad_bcast<0>(w), w1 = quad_bcast<1>(w), w2 = quad_bcast<2>(w), w3 = quad_bcast<3>(w);
            const double2 r0 = quad_fetch(R2, c0, lq), r1 = quad_fetch(R2, c1, lq), r2 = quad_fetch(R2, c2, lq), r3 = quad_fetch(R2, c3, lq);
            quad_accumulate(a0, r0, c0, w0);
            quad_accumulate(a1, r1, c1, w1);
            quad_accumulate(a2, r2, c2, w2);
            quad_accumulate(a3, r3, c3, w3);
        }
        // home: owner lane j takes chunk c of its sums from quad lane c (which holds it as its a_j)
        const double2 o0 = quad_pick(quad_bcast2<0>(a0), quad_bcast2<0>(a1), quad_bcast2<0>(a2), quad_bcast2<0>(a3), lq);
        const double2 o1 = quad_pick(quad_bcast2<1>(a0), quad_bcast2<1>(a1), quad_bcast2<1>(a2), quad_bcast2<1>(a3), lq);
        const double2 o2 = quad_pick(quad_bcast2<2>(a0), quad_bcast2<2>(a1), quad_bcast2<2>(a2), quad_bcast2<2>(a3), lq);
        const double2 o3 = quad_pick(quad_bcast2<3>(a0), quad_bcast2<3>(a1), quad_bcast2<3>(a2), quad_bcast2<3>(a3), lq);
        s.ufx = o0.x; s.ufy = o0.y; s.ufz = o1.x; s.alpha_f = o1.y; s.sax = o2.x; s.say = o2.y; s.saz = o3.x;
    }
    if (live) {
        ParticleForce pf{0.0, 0.0, 0.0, 0.0};
        const int32_t orig = p.orig[i];
        double* F = force_out + 6 * (size_t)orig;
        if (k == 0) {                                   // zeros for particles nobody located (FoamYade.C:142)
            F[0] = F[1] = F[2] = 0.0;
            if (!fp.torque_prezeroed) { F[3] = F[4] = F[5] = 0.0; }
        } else {
            ModelSums ms{0, 0, 0, 0, 0, 0, 0};
            if (fp.models)                              // uniform: off in the shipped reference
                for (int t = 0; t < k; ++t) {
                    const size_t slot = (size_t)((first + t) & (kMaxK - 1)) * p.cap + (size_t)i;
                    const int64_t cl = (int64_t)p.ids[slot] - cw.base;
                    if (cl < 0 || cl >= cw.n_field) continue;
                    model_add(ms, fp, vGrad, ddtU, cl, p.w[slot], volp);
                }
            pf = force_law(fp, s, ms, k, dia, p.vx[i], p.vy[i], p.vz[i], rec + 10 * (size_t)orig, F);
            const double irho = 1 / fp.rhoF;
            // a uniform block's cell volume is a constant, not a gather
            const double ooUniform = fp.uniform_vol > 0 ? 1. / (fp.uniform_vol * fp.rhoF) : 0.0;
            size_t slot_b = (size_t)(first & (kMaxK - 1)) * p.cap + (size_t)i;
            int32_t idb_n = p.ids[slot_b];
            double wb_n = p.w[slot_b];
            for (int t = 0; t < k; ++t) {
                const int64_t cl = (int64_t)idb_n - cw.base;
                const double w = wb_n;
                if (t + 1 < k) {                         // (prefetched as in the gather loop)
                    slot_b = (size_t)((first + t + 1) & (kMaxK - 1)) * p.cap + (size_t)i;
                    idb_n = p.ids[slot_b]; wb_n = p.w[slot_b];
                }
                if (cl < 0 || cl >= cw.n_field) continue;
                const int32_t c = (int32_t)cl;
                const double ooCellVol = fp.uniform_vol > 0 ? ooUniform : 1. / (vol[c] * fp.rhoF);  // FoamYade.C:432
                const double c0 = (-pf.coeff * w) * irho;                                      // FoamYade.C:385 (and, times uParticle[c], :386)
                const double c1 = (-pf.bx * w) * ooCellVol, c2 = (-pf.by * w) * ooCellVol, c3 = (-pf.bz * w) * ooCellVol;   // FoamYade.C:433, 406-411
                const int h = agg_slot<kForceLog2>(keys, (uint32_t)c);
                if (h >= 0) {
                    lds_add_f64(&vals[agg_at<kSlots>(h, 0)], c0); lds_add_f64(&vals[agg_at<kSlots>(h, 1)], c1);
                    lds_add_f64(&vals[agg_at<kSlots>(h, 2)], c2); lds_add_f64(&vals[agg_at<kSlots>(h, 3)], c3);
                } else {
                    atomic_add_f64(&drag_acc[c], c0);
                    atomic_add_f64(&uSource[3 * (size_t)c + 0], c1);
                    atomic_add_f64(&uSource[3 * (size_t)c + 1], c2);
                    atomic_add_f64(&uSource[3 * (size_t)c + 2], c3);
                }
            }
        }
    }
#else
    const int64_t i = (int64_t)blockIdx.x * kForceThreads + threadIdx.x;
    if (i < n) {
        const int chain = p.chain_len[i];
        const int k = chain < kMaxK ? chain : kMaxK;
        // The stencil rows are a ring indexed by push order (slot = push index & 15).  The loops below walk the k newest entries OLDEST
        // FIRST: row (chain - k + t) & 15, which is row t for every chain that never wrapped (chain <= 12: all of them in practice) -- so
        // in iteration t all lanes of a wave read the same row, whatever their chain lengths, and the id / weight loads are coalesced
        // (walking newest first, row (chain - 1 - t) & 15, spread each load over as many rows as the wave has chain lengths: ~8).
        // The sums are the reference's sums taken in the opposite order (FoamYade.C:358-365 runs over the container newest first);
        // the difference is rounding in the last bits (covered by the 1e-10 bar of the golden tests).
        const int first = chain - k;
        ParticleForce pf{0.0, 0.0, 0.0, 0.0};
        const int32_t orig = p.orig[i];
        double* F = force_out + 6 * (size_t)orig;
        if (k == 0) {                                   // zeros for particles nobody located (FoamYade.C:142)
            F[0] = F[1] = F[2] = 0.0;
            if (!fp.torque_prezeroed) { F[3] = F[4] = F[5] = 0.0; }
        } else {
            const double dia = 2 * p.rad[i];
            const double volp = M_PI * cube3(dia) / 6.0;
            // hydroDragForce FoamYade.C:358-365 and archimedesForce FoamYade.C:416-424 share one gather pass over the cell records
            Interp s{0, 0, 0, 0, 0, 0, 0, 0};
            ModelSums ms{0, 0, 0, 0, 0, 0, 0};
            {
                // the next entry's id and weight are fetched while this entry's record gathers are in flight: one memory round trip per entry instead of
                // two dependent ones (id -> record); same operations in the same order (round 5: 1.11 -> 1.03 ms).  One stage deeper -- the NEXT entry's
                // record gathers in flight as well -- was measured too and loses (1.15 ms): not used
                size_t slot = (size_t)(first & (kMaxK - 1)) * p.cap + (size_t)i;
                int32_t id_n = p.ids[slot];
                double w_n = p.w[slot];
                for (int t = 0; t < k; ++t) {
                    const int64_t cl = (int64_t)id_n - cw.base;
                    const double w = w_n;
                    if (t + 1 < k) {
                        slot = (size_t)((first + t + 1) & (kMaxK - 1)) * p.cap + (size_t)i;
                        id_n = p.ids[slot]; w_n = p.w[slot];
                    }
                    if (cl < 0 || cl >= cw.n_field) continue;
                    interp_add(s, R, cl, w, volp);
                }
            }
            if (fp.models)                              // uniform: off in the shipped reference
                for (int t = 0; t < k; ++t) {
                    const size_t slot = (size_t)((first + t) & (kMaxK - 1)) * p.cap + (size_t)i;
                    const int64_t cl = (int64_t)p.ids[slot] - cw.base;
                    if (cl < 0 || cl >= cw.n_field) continue;
                    model_add(ms, fp, vGrad, ddtU, cl, p.w[slot], volp);
                }
            pf = force_law(fp, s, ms, k, dia, p.vx[i], p.vy[i], p.vz[i], rec + 10 * (size_t)orig, F);
            const double irho = 1 / fp.rhoF;
            // a uniform block's cell volume is a constant, not a gather
            const double ooUniform = fp.uniform_vol > 0 ? 1. / (fp.uniform_vol * fp.rhoF) : 0.0;
            size_t slot_b = (size_t)(first & (kMaxK - 1)) * p.cap + (size_t)i;
            int32_t idb_n = p.ids[slot_b];
            double wb_n = p.w[slot_b];
            for (int t = 0; t < k; ++t) {
                const int64_t cl = (int64_t)idb_n - cw.base;
                const double w = wb_n;
                if (t + 1 < k) {                         // (prefetched as in the gather loop)
                    slot_b = (size_t)((first + t + 1) & (kMaxK - 1)) * p.cap + (size_t)i;
                    idb_n = p.ids[slot_b]; wb_n = p.w[slot_b];
                }
                if (cl < 0 || cl >= cw.n_field) continue;
                const int32_t c = (int32_t)cl;
                const double ooCellVol = fp.uniform_vol > 0 ? ooUniform : 1. / (vol[c] * fp.rhoF);  // FoamYade.C:432
                const double c0 = (-pf.coeff * w) * irho;                                      // FoamYade.C:385 (and, times uParticle[c], :386)
                const double c1 = (-pf.bx * w) * ooCellVol, c2 = (-pf.by * w) * ooCellVol, c3 = (-pf.bz * w) * ooCellVol;   // FoamYade.C:433, 406-411
                const int h = agg_slot<kForceLog2>(keys, (uint32_t)c);
                if (h >= 0) {
                    lds_add_f64(&vals[agg_at<kSlots>(h, 0)], c0); lds_add_f64(&vals[agg_at<kSlots>(h, 1)], c1);
                    lds_add_f64(&vals[agg_at<kSlots>(h, 2)], c2); lds_add_f64(&vals[agg_at<kSlots>(h, 3)], c3);
                } else {
                    atomic_add_f64(&drag_acc[c], c0);
                    atomic_add_f64(&uSource[3 * (size_t)c + 0], c1);
                    atomic_add_f64(&uSource[3 * (size_t)c + 1], c2);
                    atomic_add_f64(&uSource[3 * (size_t)c + 2], c3);
                }
            }
        }
    }
#endif
    __syncthreads();
    flush_table<kSlots, kForceThreads>(keys, vals, tmap, tb, drag_acc, uSource, nullptr);
}

// ---- particle migration between z-slabs: records whose containing cell now lies in a neighbour's planes are packed for
// that neighbour, the others compacted; 11 doubles per particle = the wire record + its tag (an int64 carried as a bit pattern)
__global__ __launch_bounds__(256) void k_migrate_pack(const double* __restrict__ rec, const int64_t* __restrict__ tags, int64_t n, SlabOwn own,
                                                      unsigned int* __restrict__ counters, double* __restrict__ stay, double* __restrict__ up,
                                                      double* __restrict__ down) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = rec + 10 * i;
    const double z = r[2];
    int kz = (int)floor((z - own.oz) / own.dx);
    kz = min(max(kz, 0), own.nzglob - 1);
    const int dest = !(z == z) ? 0 : (kz >= own.k1 ? 1 : (kz < own.k0 ? 2 : 0));      // NaN stays (and is never located)
    double* out = dest == 0 ? stay : (dest == 1 ? up : down);
    const unsigned int pos = atomicAdd(&counters[dest], 1u);
    double* o = out + 11 * (size_t)pos;
    for (int q = 0; q < 10; ++q) o[q] = r[q];
    o[10] = __longlong_as_double(tags ? (long long)tags[i] : (long long)-1);
}

__global__ __launch_bounds__(256) void k_migrate_unpack(const double* __restrict__ packed, int64_t n, double* __restrict__ rec, int64_t* __restrict__ tags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = packed + 11 * i;
    for (int q = 0; q < 10; ++q) rec[10 * i + q] = r[q];
    if (tags) tags[i] = (int64_t)__double_as_longlong(r[10]);
}

// found flags in wire order (FoamYade.C:141,204,222): 1 if the particle has a stencil, -1 otherwise.  Only the parallel-Yade
// protocol and the tests read them, so they are formed on demand instead of as 10 M scattered 4-byte stores in every force pass.
__global__ __launch_bounds__(256) void k_found_from_chain(ParticleSoA p, int64_t n, int32_t* __restrict__ found_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) found_out[p.orig[i]] = p.chain_len[i] > 0 ? 1 : -1;
}

// sorted chain-order stencil storage -> [n][16] ascending-d2 rows in wire order (parity tests only)
__global__ __launch_bounds__(256) void k_unpack_stencils(ParticleSoA p, int64_t n, int32_t* __restrict__ k_out, int32_t* __restrict__ ids,
                                                         double* __restrict__ w, int32_t* __restrict__ chain_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int chain = p.chain_len[i];
    const int k = chain < kMaxK ? chain : kMaxK;
    const size_t orig = (size_t)p.orig[i];
    k_out[orig] = chain;       // the reference's container also grows past 12 (meshTree.H:74-77)
    chain_out[orig] = chain;
    for (int t = 0; t < kMaxK; ++t) {
        if (t < k) {
            const size_t slot = (size_t)((chain - 1 - t) & (kMaxK - 1)) * p.cap + (size_t)i;
            ids[orig * kMaxK + t] = p.ids[slot];
            w[orig * kMaxK + t] = p.w[slot];
        } else {
            ids[orig * kMaxK + t] = -1;
            w[orig * kMaxK + t] = 0.0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ point force
// index q with f[q] <= x < f[q + 1] by bisection over the n + 1 face planes of one axis (x inside [f[0], f[n]] is the caller's business)
__device__ __forceinline__ int axis_cell(const double* __restrict__ f, int n, double x) {
    int lo = 0, hi = n;                        // invariant: f[lo] <= x, and x < f[hi] or hi == n
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (f[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_point_force(const double* __restrict__ rec, int64_t n, BlockGeom g, ForceParams fp, CellWindow cw,
                                                     const double* __restrict__ vol, const double* __restrict__ U,
                                                     const double* __restrict__ vGrad, double* __restrict__ uSource,
                                                     double* __restrict__ force_out, int32_t* __restrict__ found_out,
                                                     int32_t* __restrict__ incell_out, SlabOwn own) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* r = rec + 10 * i;
    const double x = r[0], y = r[1], z = r[2];
    double* F = force_out + 6 * (size_t)i;
    int cglob;
    if (g.cell_of) {
        // general mesh: mesh.findCell (FoamYade.C:251) was stood in for by the nearest centre + a walk across faces (k_ldu_find_cell)
        cglob = g.cell_of[i];
        if (cglob < 0) {
            F[0] = F[1] = F[2] = F[3] = F[4] = F[5] = 0.0;
            found_out[i] = -1;
            incell_out[i] = -1;
            return;
        }
    } else {
        // uniform-block stand-in for mesh.findCell (FoamYade.C:251): inside the closed bounding box, floor((p-min)/dx) clamped
        const bool outside = (x < g.bbmin[0] || y < g.bbmin[1] || z < g.bbmin[2] || x > g.bbmax[0] || y > g.bbmax[1] || z > g.bbmax[2]);
        if (outside || !(x == x) || !(y == y) || !(z == z)) {
            F[0] = F[1] = F[2] = F[3] = F[4] = F[5] = 0.0;
            found_out[i] = -1;
            incell_out[i] = -1;
            return;
        }
        int ci, cj, ck;
        if (g.faces[0]) {              // graded block: the cell whose [low, high) face planes hold the coordinate (the last cell takes its high plane too)
            ci = axis_cell(g.faces[0], g.nx, x); cj = axis_cell(g.faces[1], g.ny, y); ck = axis_cell(g.faces[2], g.nz, z);
        } else {
            ci = min(g.nx - 1, (int)((x - g.bbmin[0]) / g.dx));
            cj = min(g.ny - 1, (int)((y - g.bbmin[1]) / g.dx));
            ck = min(g.nz - 1, (int)((z - g.bbmin[2]) / g.dx));
        }
        if (own.active && !own_planes(own, (int)i, ck, x, y, z)) {           // in another slab's (or wire piece's) planes: that one owns it
            F[0] = F[1] = F[2] = F[3] = F[4] = F[5] = 0.0;
            found_out[i] = -1;
            incell_out[i] = -1;
            return;
        }
        cglob = ci + g.nx * (cj + g.ny * ck);
    }
    found_out[i] = 1;
    incell_out[i] = cglob;
    const int64_t cl = (int64_t)cglob - cw.base;
    if (cl < 0 || cl >= cw.n_field) { F[0] = F[1] = F[2] = F[3] = F[4] = F[5] = 0.0; return; }   // not in this slab's window
    const int c = (int)cl;
    const double dia = 2 * r[9];
    const double rhoF = fp.rhoF, nu = fp.nu;
    // stokesDragForce FoamYade.C:437-444
    const double coeff = 3 * M_PI * (dia)*nu * rhoF;
    const double ooCellVol = 1. / (vol[c] * rhoF);
    const double* u = U + 3 * (size_t)c;
    const double hx = coeff * (u[0] - r[3]), hy = coeff * (u[1] - r[4]), hz = coeff * (u[2] - r[5]);
    const double m = -1 * ooCellVol;
    atomic_add_f64(&uSource[3 * (size_t)c + 0], m * hx);
    atomic_add_f64(&uSource[3 * (size_t)c + 1], m * hy);
    atomic_add_f64(&uSource[3 * (size_t)c + 2], m * hz);
    F[0] = hx; F[1] = hy; F[2] = hz;
    // stokesDragTorque FoamYade.C:446-453: wfluid = (zy - yz, zx - xz, yx - xy)
    const double* G = vGrad + 9 * (size_t)c;
    const double s1 = G[7] - G[5], s2 = G[6] - G[2], s3 = G[3] - G[1];
    const double pd3 = M_PI * (pow(dia, 3.0));
    F[3] = ((pd3 * (s1 - r[6])) * nu) * rhoF;
    F[4] = ((pd3 * (s2 - r[7])) * nu) * rhoF;
    F[5] = ((pd3 * (s3 - r[8])) * nu) * rhoF;
}

// reverse-halo accumulation: y += x, and flag the cells that received something (stands in for the neighbour's touched flags)
__global__ __launch_bounds__(256) void k_add_mark(double* __restrict__ y, const double* __restrict__ x, size_t n, unsigned char* __restrict__ mark) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    y[i] += v;
    if (mark && v != 0.0) mark[i] = 1;
}

__global__ __launch_bounds__(256) void k_fill_f64(double* __restrict__ p, size_t n, double v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// setSourceZero FoamYade.C:556-564
__global__ __launch_bounds__(256) void k_set_source_zero(int32_t n_cells, int gaussian, double* __restrict__ uSourceDrag,
                                                         double* __restrict__ alpha, double* __restrict__ uSource,
                                                         double* __restrict__ uParticle) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cells) return;
    double* s = uSource + 3 * (size_t)c;
    s[0] = 0.0; s[1] = 0.0; s[2] = 0.0;
    if (gaussian) {
        alpha[c] = 1.0;
        uSourceDrag[c] = 0.0;
        double* u = uParticle + 3 * (size_t)c;
        u[0] = 0.0; u[1] = 0.0; u[2] = 0.0;
    }
}

#define FY_LAUNCH_CHECK()                                                                                     \
    do {                                                                                                      \
        hipError_t _e = hipGetLastError();                                                                    \
        if (_e != hipSuccess) return fail(FY_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

}  // namespace

int launch_bin_count(hipStream_t s, const double* rec, int64_t n, BinGrid g, uint32_t* key, uint32_t* rank, uint32_t* hist) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_bin_count, dim3(div_up(n, 256)), dim3(256), 0, s, rec, n, g, key, rank, hist);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_exclusive_scan_u32(hipStream_t s, uint32_t* data, uint32_t n, uint32_t* block_sums) {
    if (n == 0) return FY_OK;
    const uint32_t nb = (n + 2047u) / 2048u;
    hipLaunchKernelGGL(k_scan_tiles, dim3(nb), dim3(256), 0, s, data, n, block_sums);
    FY_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, block_sums, nb);
    FY_LAUNCH_CHECK();
    return FY_OK;   // the tile offsets stay separate: consumers add block_sums[i >> 11]
}

int launch_bin_scatter(hipStream_t s, const double* rec, int64_t n, const uint32_t* key, const uint32_t* rank,
                       const uint32_t* start, const uint32_t* tile_off, ParticleSoA p, bool gather) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_bin_scatter, dim3(div_up(n, 256)), dim3(256), 0, s, n, key, rank, start, tile_off, p.orig);
    FY_LAUNCH_CHECK();
    if (gather) {
        hipLaunchKernelGGL(k_bin_gather, dim3(div_up(n, 256)), dim3(256), 0, s, rec, n, p);
        FY_LAUNCH_CHECK();
    }
    return FY_OK;
}

int launch_chain_by_wire(hipStream_t s, const int32_t* orig, const int32_t* chain_len, const unsigned char* scan_class, int64_t n, unsigned char* kwire) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_chain_by_wire, dim3(div_up(n, 256)), dim3(256), 0, s, orig, chain_len, scan_class, n, kwire);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_order_blocks_by_chain(hipStream_t s, int32_t* orig, const unsigned char* kwire, int64_t n) {
    if (n <= 0) return FY_OK;
    static_assert(kDepThreads == 512, "the runs that are ordered are the locate's workgroups");
    hipLaunchKernelGGL(k_order_blocks_by_chain, dim3(div_up(n, kRun)), dim3(kRun), 0, s, orig, kwire, n);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_bin_gather(hipStream_t s, const double* rec, int64_t n, ParticleSoA p) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_bin_gather, dim3(div_up(n, 256)), dim3(256), 0, s, rec, n, p);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_build_locate_start(hipStream_t s, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, double maxdist, unsigned long long* start) {
    const double md_cells = (maxdist / (ig.dx * ig.dx)) * (1.0 + 1e-6);
    hipLaunchKernelGGL(k_build_locate_start, dim3(div_up(n_cells, 256)), dim3(256), 0, s, packed, ig, n_cells, md_cells, start);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

// waves of the walk: kLocPPB particles each for a large cloud; a small one is spread over up to 8192 waves (at least 64 particles each) instead of leaving most CUs idle --
// 300 000 particles on 293 waves took 0.74 ms, 2.5 us per particle against 0.41 at 10 M; particle phase on the explicit tree, same box, 1024 per wave / at most 2048 / 4096 /
// 8192 / 16384 waves: 300 k 0.92 / 0.42 / 0.43 / 0.41 / 0.41 ms, 1 M 1.69 / 1.25 / 1.24 / 1.17 / 1.26, 2.5 M 2.42 / 2.42 / 2.56 / 2.37 / 2.39
static unsigned locate_grid(int64_t n) {
    static const int64_t cap = [] { const char* e = getenv("FOAMYADE_LOCATE_WAVES"); return (int64_t)(e ? atoi(e) : 8192); }();      // (experiments)
    const int64_t big = div_up(n, kLocPPB), spread = std::min<int64_t>(div_up(n, 64), cap);
    return (unsigned)std::max<int64_t>(std::max(big, spread), 1);
}
int launch_locate(hipStream_t s, const KdNode* tree, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, int levels,
                  ParticleSoA p, int64_t n, GaussParams gp, const unsigned long long* start, SlabOwn own, LocateLists ll) {
    if (n <= 0) return FY_OK;
    // 8-byte entries need offsets and sizes < 2^25; an explicit tree past that takes the instance with 16-byte entries (an implicit one has no other)
    if (packed && n_cells >= (1 << 25)) return fail(FY_ERR_UNSUPPORTED, "implicit-coordinate tree limited to 2^25 cells");
    static const bool force_wide = [] { const char* e = getenv("FOAMYADE_LOCATE_WIDE"); return e && atoi(e) != 0; }();      // (tests: the 16-byte entries on a small tree)
    const bool wide = !packed && (n_cells >= (1 << 25) || force_wide);
    const size_t esz = wide ? sizeof(uint4) : sizeof(unsigned long long);
    const size_t lds = (size_t)(levels + 1) * kWave * esz;
    const dim3 grid(locate_grid(n));
    auto explicit_walk = [&](size_t lds_bytes, const int32_t* work, const unsigned int* work_n, int cap, int32_t* ovf_list, unsigned int* ovf_count) {
        if (wide)
            hipLaunchKernelGGL((k_locate<false, false, true>), grid, dim3(kWave), lds_bytes, s, tree, packed, ig, n_cells, p, n, gp.maxdist, nullptr, own, work, work_n, WalkDeposit{}, cap, ovf_list,
                               ovf_count, ll.depth_hwm);
        else
            hipLaunchKernelGGL((k_locate<false, false, false>), grid, dim3(kWave), lds_bytes, s, tree, packed, ig, n_cells, p, n, gp.maxdist, nullptr, own, work, work_n, WalkDeposit{}, cap, ovf_list,
                               ovf_count, ll.depth_hwm);
    };
    if (packed) {
        hipLaunchKernelGGL((k_locate<true, false>), grid, dim3(kWave), lds, s, tree, packed, ig, n_cells, p, n, gp.maxdist, start, own, nullptr, nullptr, WalkDeposit{}, 0, nullptr, nullptr, nullptr);
    } else if (ll.fb_list && ll.fb_count && ll.stack_cap > 0 && ll.stack_cap < levels + 1) {
        // (levels + 1) entries per lane are 12 KB per wave at 4 M cells (23 KB with the 16-byte entries of rounds 4 - 5: 6 waves per CU), and the kernel is latency bound.  The stack
        // never gets that deep (starting from best <= maxdist only the split planes within the search radius of the query are stacked): a stack of the depth the walks have been
        // seen to need (ll.depth_hwm) serves them all, and a walk that needs more than that takes the second launch
        const int cap = ll.stack_cap;
        FY_HIP(hipMemsetAsync(ll.fb_count, 0, sizeof(unsigned int), s));
        explicit_walk((size_t)cap * kWave * esz, nullptr, nullptr, cap, ll.fb_list, ll.fb_count);
        FY_LAUNCH_CHECK();
        explicit_walk(lds, ll.fb_list, ll.fb_count, 0, nullptr, nullptr);
    } else {
        explicit_walk(lds, nullptr, nullptr, 0, nullptr, nullptr);
    }
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_nearest_cell(hipStream_t s, const KdNode* tree, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, const double* pos, int64_t n, int32_t* out) {
    if (n <= 0) return FY_OK;
    if (packed) hipLaunchKernelGGL(k_nearest_cell<true>, dim3(div_up(n, 256)), dim3(256), 0, s, tree, packed, ig, n_cells, pos, n, out);
    else hipLaunchKernelGGL(k_nearest_cell<false>, dim3(div_up(n, 256)), dim3(256), 0, s, tree, packed, ig, n_cells, pos, n, out);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_build_locate_lists(hipStream_t s, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, double maxdist, unsigned short* lists,
                              int32_t cell0, int32_t n_listed) {
    const double md_cells = maxdist / (ig.dx * ig.dx);
    hipLaunchKernelGGL(k_build_locate_lists, dim3(div_up((int64_t)n_listed * 8, 256)), dim3(256), 0, s, packed, ig, n_cells, md_cells, lists, cell0, n_listed);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_locate_deposit(hipStream_t s, const KdNode* tree, const uint32_t* packed, ImplicitGeom ig, int32_t n_cells, int levels, ParticleSoA p, int64_t n,
                          GaussParams gp, const unsigned long long* start, SlabOwn own, LocateLists ll, CellWindow cw, double* pvol_acc, double* up_acc,
                          unsigned char* touched, TileBuckets tb, SideStream side, const double* rec_gather) {
    if (n <= 0) return FY_OK;
    if (rec_gather && !(packed && ll.lists)) return fail(FY_ERR_INVALID, "launch_locate_deposit: the fused record gather needs the candidate lists");
    if (!(packed && ll.lists)) {
        FY_TRY(launch_locate(s, tree, packed, ig, n_cells, levels, p, n, gp, start, own, packed ? LocateLists{} : ll));      // (explicit tree: ll carries the overflow list)
        return launch_deposit(s, p, n, gp, cw, pvol_acc, up_acc, touched, tb);
    }
    if (n_cells >= (1 << 25)) return fail(FY_ERR_UNSUPPORTED, "implicit-coordinate tree limited to 2^25 cells");
    // the lists place and deposit almost every particle; the walk + k_deposit pair takes what is left (usually nothing: zero count, exit)
    if (!ll.fb_zeroed) FY_HIP(hipMemsetAsync(ll.fb_count, 0, sizeof(unsigned int), s));
    hipLaunchKernelGGL(k_locate_deposit, dim3(div_up(n, kDepThreads)), dim3(kDepThreads), 0, s, ll, ig, p, n, gp, own, cw, pvol_acc, up_acc, touched, tb, rec_gather);
    FY_LAUNCH_CHECK();
    // The leftovers (~5e-5 of the particles: within 8e-6 dx of a cell face) are one latency-bound launch (the walk, which also deposits for them).  With a side stream they run beside whatever the caller enqueues next on `s` (the cell-record pack); the caller waits
    // for side.join before anything reads the deposit.  Their few thousand contributions go out as plain atomics (no tile buckets).
    hipStream_t w = s;
    if (side.stream) {
        FY_HIP(hipEventRecord(side.fork, s));
        FY_HIP(hipStreamWaitEvent(side.stream, side.fork, 0));
        w = side.stream;
    }
    const size_t lds = (size_t)(levels + 1) * kWave * sizeof(unsigned long long);
    const dim3 wgrid((unsigned)std::min<int64_t>(div_up(n, kLocPPB), 2048));
    hipLaunchKernelGGL((k_locate<true, true>), wgrid, dim3(kWave), lds, w, tree, packed, ig, n_cells, p, n, gp.maxdist, start, SlabOwn{}, ll.fb_list, ll.fb_count,
                       WalkDeposit{gp, cw, pvol_acc, up_acc, touched}, 0, nullptr, nullptr, nullptr);
    FY_LAUNCH_CHECK();
    if (side.stream) FY_HIP(hipEventRecord(side.join, side.stream));
    return FY_OK;
}

int launch_deposit(hipStream_t s, ParticleSoA p, int64_t n, GaussParams gp, CellWindow cw, double* pvol_acc, double* up_acc, unsigned char* touched, TileBuckets tb) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_deposit, dim3(div_up(n, kDepThreads)), dim3(kDepThreads), 0, s, p, n, gp, cw, pvol_acc, up_acc, touched, nullptr, nullptr, tb);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_finalize_cells(hipStream_t s, int32_t n_cells, const double* vol, double* pvol_acc, double* up_acc,
                          unsigned char* touched, double* alpha, double* uParticle, double* R) {
    hipLaunchKernelGGL(k_finalize_cells, dim3(div_up(n_cells, 256)), dim3(256), 0, s, n_cells, vol, pvol_acc, up_acc, touched, alpha, uParticle, R);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_pack_cells(hipStream_t s, int64_t n_field, const double* U, const double* alpha, const double* gradP, const double* divT, const double* vol,
                      double nu, double rhoF, double* R) {
    if (n_field <= 0) return FY_OK;
    hipLaunchKernelGGL(k_pack_cells, dim3(div_up(n_field, 256)), dim3(256), 0, s, n_field, U, alpha, gradP, divT, vol, 2.0 * nu, rhoF, R);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_patch_rec_alpha(hipStream_t s, int64_t c0, int64_t n, const double* alpha, double* R) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_patch_rec_alpha, dim3(div_up(n, 256)), dim3(256), 0, s, c0, n, alpha, R);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_fold_sources(hipStream_t s, int64_t n_field, double* drag_acc, const double* uParticle, double* uSourceDrag, double* uSource) {
    if (n_field <= 0) return FY_OK;
    hipLaunchKernelGGL(k_fold_sources, dim3(div_up(n_field, 256)), dim3(256), 0, s, n_field, drag_acc, uParticle, uSourceDrag, uSource);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_force_gaussian(hipStream_t s, ParticleSoA p, int64_t n, ForceParams fp, CellWindow cw, const double* vol, const double* R,
                          const double* vGrad, const double* ddtU, const double* rec, double* drag_acc, double* uSource,
                          double* force_out, TileBuckets tb) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_force_gaussian, dim3(div_up(n, kForceThreads)), dim3(kForceThreads), 0, s, p, n, fp, cw, vol, R, vGrad, ddtU, rec, drag_acc, uSource, force_out, tb);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_tile_caps(hipStream_t s, TileBuckets a, TileBuckets b, unsigned int* also_zero) {
    if (!a.cell && !b.cell) return also_zero ? fail(FY_ERR_INVALID, "launch_tile_caps: nothing to launch for the counter") : FY_OK;
    hipLaunchKernelGGL(k_tile_caps, dim3(2), dim3(1024), 0, s, a, b, also_zero);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_tile_reduce(hipStream_t s, TileBuckets tb, double* dst0, double* dst3, unsigned char* touched) {
    if (!tb.cell) return FY_OK;
    hipLaunchKernelGGL(k_tile_reduce<0>, dim3((unsigned)tb.tg.n_tiles()), dim3(256), 0, s, tb, dst0, dst3, touched, TileFinish{});
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_tile_reduce_finalize(hipStream_t s, TileBuckets tb, double* pvol_acc, double* up_acc, unsigned char* touched, const double* vol, double* alpha,
                                double* uParticle, double* R) {
    if (!tb.cell) return fail(FY_ERR_INVALID, "launch_tile_reduce_finalize without tile buckets");
    hipLaunchKernelGGL(k_tile_reduce<1>, dim3((unsigned)tb.tg.n_tiles()), dim3(256), 0, s, tb, pvol_acc, up_acc, touched, TileFinish{vol, alpha, uParticle, R, nullptr, nullptr, 0, 0});
    FY_LAUNCH_CHECK();
    return FY_OK;
}

// z-slabs: the tiles of the z-layers [tk_lo, tk_hi) are finished in the reduction (setCellVolFraction resp. the fold of the sources), the others only summed
int launch_tile_reduce_finalize_layers(hipStream_t s, TileBuckets tb, double* pvol_acc, double* up_acc, unsigned char* touched, const double* vol, double* alpha,
                                       double* uParticle, double* R, int tk_lo, int tk_hi) {
    if (!tb.cell) return fail(FY_ERR_INVALID, "launch_tile_reduce_finalize_layers without tile buckets");
    hipLaunchKernelGGL(k_tile_reduce<3>, dim3((unsigned)tb.tg.n_tiles()), dim3(256), 0, s, tb, pvol_acc, up_acc, touched, TileFinish{vol, alpha, uParticle, R, nullptr, nullptr, tk_lo, tk_hi});
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_tile_reduce_fold_layers(hipStream_t s, TileBuckets tb, double* drag_acc, double* uSource, const double* uParticle, double* uSourceDrag, int tk_lo, int tk_hi) {
    if (!tb.cell) return fail(FY_ERR_INVALID, "launch_tile_reduce_fold_layers without tile buckets");
    hipLaunchKernelGGL(k_tile_reduce<4>, dim3((unsigned)tb.tg.n_tiles()), dim3(256), 0, s, tb, drag_acc, uSource, nullptr, TileFinish{nullptr, nullptr, nullptr, nullptr, uParticle, uSourceDrag, tk_lo, tk_hi});
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_tile_reduce_fold(hipStream_t s, TileBuckets tb, double* drag_acc, double* uSource, const double* uParticle, double* uSourceDrag) {
    if (!tb.cell) return fail(FY_ERR_INVALID, "launch_tile_reduce_fold without tile buckets");
    hipLaunchKernelGGL(k_tile_reduce<2>, dim3((unsigned)tb.tg.n_tiles()), dim3(256), 0, s, tb, drag_acc, uSource, nullptr, TileFinish{nullptr, nullptr, nullptr, nullptr, uParticle, uSourceDrag, 0, 0});
    FY_LAUNCH_CHECK();
    return FY_OK;
}

// Fibre coupling (`fibreCpl`, FoamYade.H:102): Yade sends 15 doubles per particle (FoamYade.C:131-136,161-165) and the position is read
// with that stride (FoamYade.C:194-198), but velocity, spin and radius are still taken from `buf[np*10 + 3..9]` of the SAME buffer
// (FoamYade.C:211-221).  The reference ships it that way; the narrow records the kernels consume are gathered with exactly that indexing.
__global__ void k_fibre_repack(const double* __restrict__ wide, double* __restrict__ rec, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int q = 0; q < 3; ++q) rec[10 * i + q] = wide[15 * i + q];
#pragma unroll
    for (int q = 3; q < 10; ++q) rec[10 * i + q] = wide[10 * i + q];
}

int launch_fibre_repack(hipStream_t s, const double* wide, double* rec, int64_t n) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_fibre_repack, dim3(div_up(n, 256)), dim3(256), 0, s, wide, rec, n);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_migrate_pack(hipStream_t s, const double* rec, const int64_t* tags, int64_t n, SlabOwn own, unsigned int* counters, double* stay, double* up, double* down) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_migrate_pack, dim3(div_up(n, 256)), dim3(256), 0, s, rec, tags, n, own, counters, stay, up, down);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_migrate_unpack(hipStream_t s, const double* packed, int64_t n, double* rec, int64_t* tags) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_migrate_unpack, dim3(div_up(n, 256)), dim3(256), 0, s, packed, n, rec, tags);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_found_from_chain(hipStream_t s, ParticleSoA p, int64_t n, int32_t* found) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_found_from_chain, dim3(div_up(n, 256)), dim3(256), 0, s, p, n, found);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_unpack_stencils(hipStream_t s, ParticleSoA p, int64_t n, int32_t* k, int32_t* ids, double* w, int32_t* chain) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_unpack_stencils, dim3(div_up(n, 256)), dim3(256), 0, s, p, n, k, ids, w, chain);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_point_force(hipStream_t s, const double* rec, int64_t n, BlockGeom g, ForceParams fp, CellWindow cw, const double* vol,
                       const double* U, const double* vGrad, double* uSource, double* force_out, int32_t* found_out,
                       int32_t* incell_out, SlabOwn own) {
    if (n <= 0) return FY_OK;
    hipLaunchKernelGGL(k_point_force, dim3(div_up(n, 256)), dim3(256), 0, s, rec, n, g, fp, cw, vol, U, vGrad, uSource, force_out, found_out, incell_out, own);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

// D2H by STORES: the answers (found flags, forces) written into mapped host memory by a kernel instead of a DMA copy.  The copy engine then carries one
// direction only -- the records still coming in -- and PCIe runs both ways at once (round 5: the H2D and D2H DMA copies of the drop-in leg executed one
// after the other, 16 + 9.5 ms per step).  16-byte words; a modest grid: the link, not the shader, bounds it
template <class W>
__global__ __launch_bounds__(256) void k_copy_out(W* __restrict__ dst, const W* __restrict__ src, size_t nw, unsigned char* __restrict__ dtail,
                                                  const unsigned char* __restrict__ stail, int ntail) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nw; q += (size_t)gridDim.x * 256) dst[q] = src[q];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dtail[threadIdx.x] = stail[threadIdx.x];
}
template <class W>
static int copy_out_words(hipStream_t s, void* dst, const void* src, size_t bytes) {
    const size_t nw = bytes / sizeof(W);
    const int ntail = (int)(bytes - sizeof(W) * nw);
    const unsigned blocks = (unsigned)std::min<size_t>(256, std::max<size_t>(1, (nw + 255) / 256));
    hipLaunchKernelGGL(k_copy_out<W>, dim3(blocks), dim3(256), 0, s, static_cast<W*>(dst), static_cast<const W*>(src), nw,
                       static_cast<unsigned char*>(dst) + sizeof(W) * nw, static_cast<const unsigned char*>(src) + sizeof(W) * nw, ntail);
    FY_LAUNCH_CHECK();
    return FY_OK;
}
int launch_copy_out(hipStream_t s, void* dst_mapped, const void* src, size_t bytes) {
    if (!bytes) return FY_OK;
    const uintptr_t al = reinterpret_cast<uintptr_t>(dst_mapped) | reinterpret_cast<uintptr_t>(src);
    if (!(al & 15)) return copy_out_words<uint4>(s, dst_mapped, src, bytes);
    if (!(al & 7)) return copy_out_words<uint2>(s, dst_mapped, src, bytes);
    if (!(al & 3)) return copy_out_words<uint32_t>(s, dst_mapped, src, bytes);
    return fail(FY_ERR_INVALID, "launch_copy_out: buffers must be 4-byte aligned");
}

int launch_add_mark(hipStream_t s, double* y, const double* x, size_t n, unsigned char* mark) {
    if (n == 0) return FY_OK;
    hipLaunchKernelGGL(k_add_mark, dim3(div_up(n, 256)), dim3(256), 0, s, y, x, n, mark);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_fill_f64(hipStream_t s, double* p, size_t n, double v) {
    if (n == 0) return FY_OK;
    hipLaunchKernelGGL(k_fill_f64, dim3(div_up(n, 256)), dim3(256), 0, s, p, n, v);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

int launch_set_source_zero(hipStream_t s, int32_t n_cells, int gaussian, double* uSourceDrag, double* alpha, double* uSource,
                           double* uParticle) {
    hipLaunchKernelGGL(k_set_source_zero, dim3(div_up(n_cells, 256)), dim3(256), 0, s, n_cells, gaussian, uSourceDrag, alpha, uSource, uParticle);
    FY_LAUNCH_CHECK();
    return FY_OK;
}

}  // namespace fy
