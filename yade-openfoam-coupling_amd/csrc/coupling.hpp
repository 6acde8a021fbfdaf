// Coupling = the state behind fy_ctx (mirror of Foam::FoamYade, FoamYade/FoamYade.H:57-161).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "comm.hpp"
#include "common.hpp"
#include "kdtree.hpp"
#include "particle_kernels.hpp"

namespace fy {

// one "Yade proc" worth of particles (FoamYade.H:41-55 YadeProc)
struct Batch {
    int64_t n = 0;
    int yrank = 0;                       // world rank of the Yade proc
    const double* d_rec = nullptr;       // device records [n][10] (borrowed, or rec_own)
    DevBuf<double> rec_own;
    DevBuf<double> rec_wide;             // fibre coupling: the 15-double records as they arrived (repacked into rec_own)
    size_t cap = 0;                      // leading dimension of the SoA / stencil arrays
    DevBuf<double> soa;                  // 7 * cap : px py pz vx vy vz rad (binned order)
    DevBuf<int32_t> orig, chain, ids, found, incell;
    DevBuf<double> w, force;
    DevBuf<uint32_t> key, rank;
    const double* torque_zero_buf = nullptr;   // force buffer whose torque slots [0, torque_zero_n) are known to be zero
    int64_t torque_zero_n = 0;
    bool found_stale = false;            // Gaussian mode: `found` is formed lazily from the chain lengths (ensure_found)
    int64_t binned_n = -1;               // particle count the current placement (orig) was computed for
    int bin_age = 0;                     // steps since it was computed
    DevBuf<unsigned char> scan_class;    // per slot: how far the locate's list scan ran (ParticleSoA::scan_class)
    DevBuf<unsigned char> kwire;         // chain length of the step before, by wire index (what the placement's runs are ordered by)
    int64_t chain_n = -1;                // particle count chain_len holds last step's lengths for (-1: none)
    bool ordered_by_chain = false;       // the current placement was ordered with them
    // tile buckets of the two scatters' flushes (TileBuckets): per-tile offsets / capacities / demand counters of this batch's population;
    // [0] void-fraction deposit, [1] momentum-source back-scatter.  The entry pool itself is shared (Coupling::tile_cell / tile_val).
    DevBuf<uint32_t> tb_off[2], tb_cap[2], tb_fill[2];
    HostBuf<double> h_rec, h_force;      // wire staging (pinned: H2D / D2H at PCIe rate, asynchronous on the copy stream)
    HostBuf<int32_t> h_found;
    EventTimer t_in, t_out;              // the batch's H2D / D2H copies on the copy stream (their end events are what the compute stream / the host wait for)
    hipEvent_t ev_ready = nullptr;       // results final on the compute stream
    hipEvent_t ev_caps = nullptr;        // the next call's tile-bucket capacities are formed (side stream, end of run_batch)
    bool caps_ready = false;
    const void* caps_key = nullptr;      // ... for the bucket arrays at this address
    bool events = false;
    DevBuf<double> pos3;                 // general mesh, point force: the records' positions [n][3] and the located cells
    DevBuf<int32_t> cell_hint;
    bool out_started = false;            // this step's D2H of force / found is already on the outbound copy stream
    // zero-copy wire (fy_transport::recv_view / send_reserve): the records stay where the transport keeps them and the results are copied
    // straight into the memory the transport sends from
    fy_wire_pieces pieces{};             // cuts of the record message (wire helpers), for the ownership test
    int32_t* out_found = nullptr;        // reserved send buffers of this step (nullptr: the pinned h_found / h_force above)
    double* out_force = nullptr;
    bool committed = false;              // this step's results were handed to the transport already
    ~Batch() { t_in.destroy(); t_out.destroy(); if (ev_ready) (void)hipEventDestroy(ev_ready); if (ev_caps) (void)hipEventDestroy(ev_caps); }
};

// z-slab mode (set by fy_solver before create()): the k-d tree spans the GLOBAL block, every cell array is this rank's slab
// storage (owned planes + gz ghost planes per side); a global cell id maps to storage index id - base.
struct SlabInfo {
    bool active = false;
    Comm* comm = nullptr;
    int gz = 0, nz = 0;
    size_t plane = 0, n_store = 0;
    int64_t base = 0;
    int kglob0 = 0, nzglob = 0;          // global k of the first owned plane, global plane count (particle ownership)
    // round 5: exchanges beside independent work (fy_solver sets these; FOAMYADE_HALO_OVERLAP=0 leaves aux null: exchange, then consume)
    hipStream_t aux = nullptr;           // the communicator's second channel (the solver's comm_stream)
    hipEvent_t fields_event = nullptr;   // recorded by the solver behind the exchange of gradP / divT ghost planes: waited for before the first gather of fluid fields
    hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_tail = nullptr;     // created on first use
    bool tail_pending = false;           // uSource's ghost planes are still on their way (ev_tail): the solver waits before it interpolates rAUc uSource
};

struct Coupling {
    SlabInfo slab;
    int64_t n_field = 0;                 // length of the cell arrays (n_cells, or slab.n_store)
    DevBuf<double> halo_tmp;             // 2 x gz x plane x 3 staging for reverse-halo sums
    // ---- configuration
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t ext_stream = nullptr;    // set before create() to run on a caller-owned stream (fy_solver does)
    bool owns_stream = false;
    bool created = false;
    fy_mesh_desc mesh{};
    int32_t n_cells = 0;
    bool gaussian = false, structured = false;
    bool has_transport = false;
    // fy_solver's field sweep that feeds the force pass (gradP, divT, cell records) reads nothing the locate writes and vice versa: handed over as a
    // hook, it is launched right AFTER the locate + deposit of the first batch, so that it runs while the few particles the candidate lists hand to the
    // tree walk are walked on the side stream (a latency-bound ~90 us that otherwise sits alone between the locate and the cells' finalisation).
    // Called at the top of set_particle_action instead when nothing would call it later (point-force mode, no batch)
    int (*mid_hook)(void*) = nullptr;
    void* mid_hook_user = nullptr;
    bool mid_hook_done = true;
    int run_mid_hook() { if (mid_hook && !mid_hook_done) { mid_hook_done = true; return mid_hook(mid_hook_user); } return 0; }
    fy_transport transport{};
    const struct LduGeo* ldu_geo = nullptr;   // set before create() by fy_ldu_solver: the general mesh's face addressing on the device (point-force locate)
    int comm_sz_diff = 0;                // FoamYade.H:74
    bool serial_yade = true;             // FoamYade.H:91
    double rhoP = 0, rhoF = 0, nu = 0;   // FoamYade.H:83-85
    double delta_t = 0, yade_dt = 0;     // FoamYade.H:94-95
    bool vol_uniform = false;            // every cell volume equals v0 (checked at create)
    double v0 = 0, interp_range = 0, sigma_interp = 0, interp_range_cu = 0, sigma_pi = 0;   // FoamYade.H:96-99
    std::vector<int> send_ranks;         // FoamYade.H:70

    // ---- device state
    DevBuf<KdNode> d_tree;
    DevBuf<uint32_t> d_tree_packed;      // implicit-coordinate nodes, only when the block's centres are exactly o + (i+0.5)*dx
    DevBuf<unsigned long long> d_loc_start;   // per-cell traversal start (implicit trees; launch_build_locate_start), built at the first Gaussian step
    DevBuf<unsigned short> d_loc_lists;       // per-(cell, octant) candidate lists (implicit trees; launch_build_locate_lists), built with it
    DevBuf<int32_t> d_loc_fb;                 // particles the lists do not cover (work list of the walk) + their count
    DevBuf<unsigned int> d_loc_fb_n;
    DevBuf<unsigned int> d_loc_hwm;           // explicit tree: histogram of the walks' stack depths (k_locate, one walk in 64), and its pinned host copy
    HostBuf<unsigned int> h_loc_hwm;          // [kLocDepthBins] the histogram, [kLocDepthBins] + 1: the walks that overflowed last step
    int loc_stack_floor = 0, loc_stack_used = 0;      // raised by two entries when more than 1 % of a step's walks overflowed; the depth the last step ran with
    int loc_stack_cap = 0, loc_window = 0;    // the explicit walk's stack depth in use (0: the full depth), steps into the histogram's window
    int64_t loc_last_n = 0;
    bool loc_lists_tried = false;
    int32_t loc_cell0 = 0, loc_n_listed = 0;  // cells the lists cover (a slab: its own planes)
    int ensure_locate_tables(double maxdist);
    DevBuf<uint32_t> tile_cell;          // entry pool of the tile buckets (one flush at a time uses it)
    DevBuf<double> tile_val;
    SideStream side{};                   // the walk's leftovers run here, beside the cell-record pack
    bool tile_flush = true;              // the scatters' tables are flushed into per-tile buckets (false: global atomics, round-1 behaviour)
    TileGrid tile_grid() const;
    TileBuckets buckets_of(Batch& b, int which);
    bool rectilinear = false;            // graded block (fy_mesh_desc.xf / yf / zf): lattice indexing, explicit tree, findCell by search
    DevBuf<double> d_faces[3];
    ImplicitGeom implicit{};
    bool use_implicit = false;
    int tree_levels = 0;
    std::vector<int32_t> h_tree_pre;     // preorder cell ids (host copy; fy_get_tree_preorder, tree cache)
    DevBuf<double> d_vol;
    fy_field_ptrs fields{};
    bool fields_on_host = false;
    DevBuf<double> own_U, own_gradP, own_vGrad, own_divT, own_ddtU, own_uSourceDrag, own_alpha, own_uSource, own_uParticle;
    const double *dU = nullptr, *dGradP = nullptr, *dVGrad = nullptr, *dDivT = nullptr, *dDdtU = nullptr;
    bool fibre = false;                  // fibreCpl (FoamYade.H:102)
    unsigned force_models = 0;           // FY_FORCE_*: the reference's call-site-less models (off = shipped behaviour)
    double *dUSourceDrag = nullptr, *dAlpha = nullptr, *dUSource = nullptr, *dUParticle = nullptr;
    DevBuf<double> d_pvol_acc, d_up_acc;           // per-batch deposit accumulators (pVolContrib / uParticleContrib)
    bool cellrec_fresh = false;                    // d_cellrec was packed in this setParticleAction call
    hipEvent_t ev_last_caps = nullptr;            // the latest end-of-batch capacities kernel on the side stream (it also clears d_loc_fb_n); not owned
    bool cellrec_ghosts_stale = false;             // slab: the caller's sweep wrote the OWNED cells' records only
    bool cellrec_external = false;                 // ... by the caller (fy_solver's pre-coupling sweep), for the NEXT setParticleAction only
    DevBuf<double> d_cellrec;                      // 8 doubles per cell: what the force pass gathers (k_pack_cells), rebuilt every setParticleAction
    DevBuf<double> d_drag_acc;                     // per-batch sum of -coeff w / rho_f per cell, folded into uSourceDrag / uSource by k_fold_sources
    DevBuf<unsigned char> d_touched;
    BinGrid bins{};
    int rebin_interval = 32;             // full counting sort every this many steps (options().rebin_interval; 1 = every step)
    DevBuf<uint32_t> d_hist, d_tile_sums;
    std::vector<Batch*> batches;
    int n_batches = 0;

    // ---- drop-in path (host-resident peer): pinned staging, copies on their own stream, overlapped with the kernels of the other batches
    hipStream_t copy_stream = nullptr, copy_out_stream = nullptr;      // H2D of the records / D2H of the results
    double wire_recv_ms = 0, wire_send_ms = 0;  // host wall time inside the transport's data calls (the MPI side)
    int ensure_batch_events(Batch& b);
    int start_results_copy(Batch& b);           // D2H of one batch's forces + found flags, as soon as its kernels are enqueued
    int upload_batch(Batch& b, int64_t n, const double* src = nullptr);      // pinned h_rec (or the transport's view) -> rec_own on the copy stream; the compute stream waits for it
    int recv_yade_pieces(const std::vector<std::pair<int, int> >& in_comm);      // recv_yade_intrs with a transport that reports the record messages piece by piece
    int commit_landed(size_t upto);             // the early hand-over of batches [0, upto), in ascending order, stopping at the first one still in flight
    int commit_results(Batch& b, bool wait);    // hand a batch's found flags and forces to the transport once their D2H has landed (wait = false: only if it has)
    int lock_view_region();                     // page-lock the memory the transport's views point into (fy_transport::view_region)
    void* view_base = nullptr; size_t view_bytes = 0; uint64_t view_generation = 0; bool view_locked = false;
    std::vector<EventTimer> piece_clocks;       // one event pair per piece copy of the piece-wise receive (recv_yade_pieces)
    bool wire_views = false;                    // this step's records came as views and its results go out through send_reserve / send_commit

    // ---- timing
    enum { T_TOTAL = 0, T_COUNT };
    EventTimer timers[T_COUNT];
    PhaseMarks marks;                    // 0 | bin (+ tile capacities) | 1 | k_locate_deposit | 2 | pack, reduce, finalize | 3 | k_force_gaussian | 4 | reduce, fold | 5;  6 | the hook (mid_hook) | 7 lies inside 2 .. 3 and is taken off it
    bool timing = false;
    fy_particle_timings tm{};

    ~Coupling();
    int create(const fy_mesh_desc* m, const fy_field_ptrs* f, int gaussian_interp, const fy_transport* tr, int device_ordinal);
    int init_fields();
    int stage_mutable_in();
    int stage_mutable_out();
    int stage_readonly_in();
    void set_num_batches(int nb);
    int ensure_batch(Batch& b, int64_t n);
    ParticleSoA soa_of(Batch& b);
    int set_particles_host(int bi, const double* rec, int64_t n);
    int set_particles_device(int bi, const double* d_rec, int64_t n);
    SlabOwn slab_own() const { return SlabOwn{slab.active ? 1 : 0, slab.kglob0, slab.kglob0 + slab.nz, slab.nzglob, mesh.origin[2], mesh.dx, 0, 0, 0, 0.0, {0}, {0}, {0}}; }
    // ... of one batch: its slab's planes, cut further by the wire pieces its records arrived in
    SlabOwn own_of(const Batch& b) const {
        SlabOwn o = slab_own();
        if (b.pieces.n > 0) {
            if (!o.active) { o.active = 1; o.k0 = 0; o.k1 = mesh.nz; o.nzglob = mesh.nz; }
            o.npieces = b.pieces.n;
            o.paxis = b.pieces.axis; o.po = mesh.origin[b.pieces.axis]; o.pn = b.pieces.axis == 0 ? mesh.nx : (b.pieces.axis == 1 ? mesh.ny : mesh.nz);
            for (int q = 0; q < b.pieces.n; ++q) { o.pstart[q] = b.pieces.start[q]; o.pk0[q] = b.pieces.k0[q]; o.pk1[q] = b.pieces.k1[q]; }
        }
        return o;
    }
    int migrate(int64_t* d_tags, int64_t tag_capacity, int64_t* n_out);
    DevBuf<double> mig_stay, mig_up, mig_down, mig_cnt;
    DevBuf<unsigned int> mig_counters;
    int ensure_found(Batch& b);
    int run_batch(Batch& b);
    int set_force_models(unsigned flags);
    int set_fibre_coupling(int on);
    int rec_len() const { return fibre ? 15 : 10; }   // doubles per particle on the wire (FoamYade.C:131-136)
    int set_particle_action(double dt);
    bool timings_pending = false;        // the last call's phase events have not been read yet
    int collect_timings();
    int recv_serial();
    int recv_yade_intrs();
    int send_results();
    int exchange_dt();
    int send_fluid_dt();                        // the two halves of exchange_dt (FoamYade.C:539-541 / 543-552)
    int recv_yade_dt();
    // round 5: fy_solver lets the fluid solve run while the answers cross PCIe.  async_results (set by the solver for one call): with a zero-copy wire
    // setParticleAction returns as soon as every batch's D2H copy is enqueued; poll_results() hands over, in worker order, whatever has landed
    // (called from the solver's host waits), finish_results() the rest + the dt handshake.  Yade gets each worker's answers as early as before
    // -- when their copy lands -- while the PISO / PIMPLE kernels are already running (nothing in pimpleFoamYade.C:83-105 reads the forces)
    bool async_results = false, results_pending = false;
    int poll_results() { return (results_pending && wire_views) ? commit_landed((size_t)n_batches) : FY_OK; }
    int finish_results();
    int set_source_zero();
    int halo_fwd(double* f, int ncomp, int w, hipStream_t on = nullptr);
    // the exchange (start: ghost-plane sums travel to tmp buffers, on `on`) and the owners' additions + ghost reset (finish: on the main stream) of a reverse halo
    int halo_reverse_start(double* f1, int nc1, double* f2, int nc2, hipStream_t on);
    int halo_reverse_finish(double* f1, int nc1, unsigned char* mark1, double* f2, int nc2);
    int halo_reverse_add2(double* f1, int nc1, unsigned char* mark1, double* f2, int nc2) { FY_TRY(halo_reverse_start(f1, nc1, f2, nc2, stream)); return halo_reverse_finish(f1, nc1, mark1, f2, nc2); }
    bool slab_overlap() const { return slab.active && slab.aux != nullptr; }
    int slab_events();                   // create ev_a / ev_b / ev_tail on first use
    int get_forces_host(int bi, double* out);
    int get_found_host(int bi, int32_t* out);
    int get_stencils_host(int bi, int32_t* k, int32_t* ids, double* w, int32_t* chain);
    int get_tree_preorder(int32_t* out);
    int field_by_name(const char* name, double** p, size_t* count);
    int read_field_host(const char* name, double* out);
    int write_field_host(const char* name, const double* in);
};

}  // namespace fy

struct fy_ctx { fy::Coupling c; };
