// engine.hip -- host side of libmcrat_hip.so: the C ABI of include/mcrat_hip.h on top of the kernels.
//
// Owns the HBM residency of one rank's photon list (SoA columns) and of the current hydro frame
// (packed cell records + the exact cell-lookup grid), and drives the per-iteration kernel pair.
// There is no CPU compute path in this file: if HIP is unavailable every entry point fails.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/mcrat_hip.h"
#include "device_types.hpp"
#include "launch.hpp"
#include "rng.hpp"

using namespace mcrat;

static_assert(sizeof(mcrat_hip_photon) == 176, "struct photon layout (Src/mcrat.h:142-171) must be 176 bytes");
static_assert(offsetof(mcrat_hip_photon, num_scatt) == 128 && offsetof(mcrat_hip_photon, recalc_properties) == 136 &&
              offsetof(mcrat_hip_photon, weight) == 144 && offsetof(mcrat_hip_photon, nearest_block_index) == 152 &&
              offsetof(mcrat_hip_photon, time_to_scatter) == 160 && offsetof(mcrat_hip_photon, total_optical_depth) == 168,
              "struct photon field offsets");
static_assert(sizeof(LoopState) == 256, "LoopState is one 256-B record");

struct mcrat_hip_ctx {
    mcrat_hip_config cfg;
    KernelConfig kc;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string last_error;

    // photons
    PhotonDev ph{};
    void *ph_buf = nullptr;
    size_t ph_bytes = 0;
    void *ph_snap = nullptr;          // mcrat_hip_snapshot_photons
    void *ph_cap = nullptr;           // mcrat_hip_pool_run_frames with capture_frames: the lists as frames 0 .. n_frames - 2 left them, ph_bytes each
    size_t ph_cap_bytes = 0;
    int cap_frames = 0;               // how many of them the last plan filled
    int selected_frame = -1;          // mcrat_hip_pool_select_frame: the read entry points look at that capture (c->ph shifted), -1: the live lists
    PhotonDev ph_live{};
    size_t ph_snap_bytes = 0;
    bool have_photons = false;
    void *aos_buf = nullptr;          // device copy of the caller's struct photon records (mcrat_hip_set_photons / get_photons)
    void *pin_ring[2] = {nullptr, nullptr};   // pinned staging for uploads gathered from many host buffers (mcrat_hip_pool_set_photons)
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    size_t aos_bytes = 0;
    int step_blocks = 0;
    Cand *partials = nullptr;
    int partials_cap = 0;
    Shortlist *shortlist = nullptr;

    // hydro
    HydroDev hy{};
    void *hy_buf = nullptr;
    size_t hy_bytes = 0;
    bool have_hydro = false;
    void *grid_buf = nullptr;          // bucket records, bucket lists (+ build scratch)
    size_t grid_bytes = 0;
    unsigned *grid_count = nullptr;    // per-bucket counters of the device build
    size_t grid_count_cap = 0;
    unsigned long long *d_grid_total = nullptr;
    void *d_cs_hook = nullptr;                 // CsFrame of the cyclo-synchrotron frame driver
    void *d_cs_args = nullptr;                 // CsHookArgs of the pool's frame driver
    HydroCols hcol{};                  // the frame as struct hydro_dataframe's columns (set_hydro / ingest), kept for get_hydro
    void *hcol_buf = nullptr;
    size_t hcol_bytes = 0;
    int hcol_M = 0;
    void *raw_buf = nullptr;           // device copy of a reader's buffers (mcrat_hip_ingest_*)
    size_t raw_bytes = 0;

    // TAU_CALCULATION == TABLE
    double *d_hot_table = nullptr;
    int hot_n_ph_e = 0, hot_n_t = 0;
    int hot_fallback_calls = 500000;   // hot_x_section.c:348
    double hot_grid[4] = {0, 0, 0, 0};          // log10 photon energy min/max, log10 theta min/max
    int *d_table_fallbacks = nullptr;

    // loop
    LoopState *d_state = nullptr;
    LoopState *h_state = nullptr;     // pinned
    bool frame_open = false;
    int find_switch = 1;
    RngKey key{0, 0, 0};
    long long frame_photon_steps = 0;
    mcrat_hip_ctx *hydro_owner = nullptr;   // mcrat_hip_share_hydro: the staged frame (and cross-section table) are another context's
    std::vector<mcrat_hip_ctx *> hydro_sharers;   // ... and, on that context, who reads its frame: they are cut loose before it changes or goes
    void *d_fast = nullptr;           // FAST mode's counters (FastCounts)
    bool pending_applied = false;     // step_locate_sample has applied the pending advance that LoopState still lists
    bool rank_current = false;        // a view whose frame rank_loop_kernel has run: it leaves no pending advance (until the next begin_frame)

    // virtual ranks (cfg.virtual_rank_photons > 0): one LoopState per list
    int n_ranks = 0;
    int rank_block = 256;             // threads per list of the next launches (choose_rank_block)
    bool rank_fuse = true;            // ... and whether they use the build with the fused pass
    // the random stream as an input (mcrat_hip_set_rng_tape): device copy of the caller's uniforms, the position reached, an error word
    double *d_tape = nullptr;
    long long tape_n = 0;
    long long *d_tape_cursor = nullptr;   // {cursor, error word} in one 16-byte block
    int fast_auto_windows = 32;       // FAST mode with fast_windows <= 0: the refresh cadence, from what the last FAST frame looked like (fast_cadence)
    bool rank_block_fixed = false;
    double rank_passes_per_list = 0;  // of the last completed frame
    LoopState *d_rstates = nullptr;
    LoopState *h_rstates = nullptr;   // pinned
    int rstates_cap = 0;

    // rank pool (mcrat_hip_pool_*): the photon arrays hold n_ranks lists of up to rank_stride slots, each with its own length,
    // seed, stream and clock -- the reference's MPI ranks, adopted by one GPU.  List r is reached through a *view*: a context
    // whose columns, LoopState and scratch are windows into this one, so that every per-list entry point (injection, reductions,
    // output columns, checkpoint records, ...) works on one rank's list unchanged.
    bool is_pool = false;
    int rank_stride = 0;              // slots reserved per list (pool), or cfg.virtual_rank_photons
    std::vector<mcrat_hip_ctx *> views;
    std::vector<int> snap_lens;       // the lists' lengths when mcrat_hip_snapshot_photons was taken
    RankDesc *d_desc = nullptr;
    RankDesc *h_desc = nullptr;       // pinned
    void *d_fq = nullptr, *h_fq = nullptr;   // the frame queue's block (mcrat_hip_pool_run_frames): ticket, frames_done, order, items, records; pinned mirror
    size_t fq_bytes = 0;
    mcrat_hip_ctx *parent = nullptr;  // this context is the view of list view_rank of `parent`
    int view_rank = -1;
    uint32_t view_stream = 0;

    // one list over several GPUs, one clock (mcrat_hip_shared_clock_*)
    int sc_world = 0, sc_rank = 0;
    ScState *d_sc = nullptr;
    ScProposal *sc_send = nullptr, *sc_recv = nullptr;
    // device-initiated exchange (mcrat_hip_shared_clock_attach_device): fine-grained receive buffer (2 x world proposals) and stamps
    bool sc_device = false, sc_peers_set = false;
    bool sc_fold = true;              // device exchange: push and wait run inside the propose and resolve kernels (MCRAT_HIP_SC_FOLD=0: as launches of their own)
    unsigned long long *sc_flags = nullptr;
    ScProposal *sc_gather = nullptr;   // the round's proposals of all ranks, copied out of the receive buffer's current half: what resolve reads
    ScPeers sc_peers{};
    bool sc_own_send = false, sc_own_recv = false;

    // scratch
    ReducePartial *d_red = nullptr;
    ReducePartial *h_red = nullptr;   // pinned
    static constexpr int RED_BLOCKS = 512;

    // profiling
    std::vector<hipEvent_t> ev;
    double prof_step_ms = 0, prof_event_ms = 0;
    long long prof_launches = 0;

    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_batch = 0;
};

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return (e_ == hipErrorOutOfMemory) ? MCRAT_HIP_ENOMEM : MCRAT_HIP_EHIP;                \
        }                                                                                          \
    } while (0)

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static void drop_graph(mcrat_hip_ctx *c)
{
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    c->graph_batch = 0;
}

static void sync_views(mcrat_hip_ctx *c);
static int view_refuses(mcrat_hip_ctx *c, const char *what);
static void release_shared_hydro(mcrat_hip_ctx *c);
static int ensure_counts(mcrat_hip_ctx *c, size_t n);

extern "C" const char *mcrat_hip_version(void) { return "mcrat_hip 0.1 (gfx950, abi 1)"; }

extern "C" const char *mcrat_hip_strerror(int code)
{
    switch (code) {
    case MCRAT_HIP_OK: return "ok";
    case MCRAT_HIP_EINVAL: return "invalid argument or unsupported switch combination";
    case MCRAT_HIP_ENODEV: return "no usable HIP device";
    case MCRAT_HIP_ENOMEM: return "out of memory";
    case MCRAT_HIP_EHIP: return "HIP runtime error";
    case MCRAT_HIP_ESTATE: return "call out of order";
    case MCRAT_HIP_EREFUSED: return "refused as the reference refuses it (nothing was changed)";
    default: return "unknown error";
    }
}

extern "C" const char *mcrat_hip_last_error(const mcrat_hip_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

static bool geometry_supported(int dims, int geom)
{
    if (dims == DIM_TWO || dims == DIM_TWO_POINT_FIVE)
        return geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL || geom == GEOM_SPHERICAL;
    if (dims == DIM_THREE) return geom == GEOM_CARTESIAN || geom == GEOM_SPHERICAL || geom == GEOM_POLAR;
    return false;
}

extern "C" int mcrat_hip_init(mcrat_hip_ctx **out, const mcrat_hip_config *cfg)
{
    if (!out || !cfg) return MCRAT_HIP_EINVAL;
    *out = nullptr;
    if (cfg->abi_version != MCRAT_HIP_ABI_VERSION) return MCRAT_HIP_EINVAL;
    if (!geometry_supported(cfg->dimensions, cfg->geometry)) return MCRAT_HIP_EINVAL;
    if (cfg->tau_calculation != MCRAT_HIP_TAU_DIRECT && cfg->tau_calculation != MCRAT_HIP_TAU_TABLE) return MCRAT_HIP_EINVAL;
    if (cfg->virtual_rank_photons < 0) return MCRAT_HIP_EINVAL;
    // with the switch on the list changes length inside the loop; one list per context, driven by
    // mcrat_hip_scatter_frame_cyclosynch
    if (cfg->cyclosynchrotron_switch != 0 && (cfg->cyclosynchrotron_switch != 1 || cfg->virtual_rank_photons != 0)) return MCRAT_HIP_EINVAL;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) return MCRAT_HIP_ENODEV;
    if (hipSetDevice(cfg->device) != hipSuccess) return MCRAT_HIP_ENODEV;

    mcrat_hip_ctx *c = new (std::nothrow) mcrat_hip_ctx();
    if (!c) return MCRAT_HIP_ENOMEM;
    c->cfg = *cfg;
    if (c->cfg.iterations_per_sync <= 0) c->cfg.iterations_per_sync = 256;
    c->kc.dimensions = cfg->dimensions;
    c->kc.geometry = cfg->geometry;
    c->kc.stokes = cfg->stokes_switch ? 1 : 0;
    c->kc.table = cfg->tau_calculation == MCRAT_HIP_TAU_TABLE ? 1 : 0;
    c->key.stream = cfg->rng_stream & 0xffffffu;

    auto fail = [&](int code) { mcrat_hip_destroy(c); return code; };
    if (cfg->stream) {
        c->stream = (hipStream_t)cfg->stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(MCRAT_HIP_ENODEV);
        c->own_stream = true;
    }
    if (hipMalloc((void **)&c->d_state, sizeof(LoopState)) != hipSuccess) return fail(MCRAT_HIP_ENOMEM);
    if (hipMalloc((void **)&c->d_table_fallbacks, sizeof(int)) != hipSuccess) return fail(MCRAT_HIP_ENOMEM);
    if (hipMemset(c->d_table_fallbacks, 0, sizeof(int)) != hipSuccess) return fail(MCRAT_HIP_ENODEV);
    if (hipHostMalloc((void **)&c->h_state, sizeof(LoopState), hipHostMallocDefault) != hipSuccess) return fail(MCRAT_HIP_ENOMEM);
    if (hipMalloc((void **)&c->d_red, sizeof(ReducePartial) * mcrat_hip_ctx::RED_BLOCKS) != hipSuccess) return fail(MCRAT_HIP_ENOMEM);
    if (hipHostMalloc((void **)&c->h_red, sizeof(ReducePartial) * mcrat_hip_ctx::RED_BLOCKS, hipHostMallocDefault) != hipSuccess)
        return fail(MCRAT_HIP_ENOMEM);
    memset(c->h_state, 0, sizeof(LoopState));
    c->h_state->done = 1;
    c->h_state->skip_idx = -1;
    if (hipMemcpy(c->d_state, c->h_state, sizeof(LoopState), hipMemcpyHostToDevice) != hipSuccess) return fail(MCRAT_HIP_ENODEV);
    *out = c;
    return MCRAT_HIP_OK;
}

static void destroy_view(mcrat_hip_ctx *v)
{
    // a view owns nothing but its cyclo-synchrotron hook state, a snapshot and its events; the rest are windows into the pool
    if (v->stream) (void)hipStreamSynchronize(v->stream);
    if (v->d_cs_hook) (void)hipFree(v->d_cs_hook);
    if (v->d_fast) (void)hipFree(v->d_fast);
    if (v->ph_snap) (void)hipFree(v->ph_snap);
    for (hipEvent_t e : v->ev) (void)hipEventDestroy(e);
    if (v->parent && v->view_rank >= 0 && v->view_rank < (int)v->parent->views.size() && v->parent->views[v->view_rank] == v)
        v->parent->views[v->view_rank] = nullptr;
    delete v;
}

// Contexts that read another one's staged frame (mcrat_hip_share_hydro) hold raw device pointers into its buffers.  Before the owner
// re-stages (the buffers are rewritten in place, or freed when the new frame is larger), replaces its cross-section table or is destroyed,
// every sharer is cut loose: its stream is drained -- a launch of its own may still be reading the frame -- and it is left WITHOUT a frame
// (have_hydro = false: begin_frame / run / share answer MCRAT_HIP_ESTATE until it shares or stages again).  One mutex guards the
// bookkeeping, so that pools driven by their own host threads may share and re-stage; a sharer must not be inside a call of its own
// while its owner re-stages (stage once per hydro frame behind a barrier and share again after it: INTEGRATION.md).
static std::mutex g_share_mutex;
static void sync_views(mcrat_hip_ctx *c);

static void forget_shared_frame(mcrat_hip_ctx *s)          // (g_share_mutex held)
{
    s->hydro_owner = nullptr;
    s->hy = HydroDev{};
    s->hcol = HydroCols{};
    s->hcol_buf = nullptr; s->hcol_M = 0; s->hcol_bytes = 0;
    s->d_hot_table = nullptr;
    s->have_hydro = false;
    s->frame_open = false;
    sync_views(s);
}

static void detach_sharers(mcrat_hip_ctx *owner, const char *why)
{
    std::lock_guard<std::mutex> lock(g_share_mutex);
    for (mcrat_hip_ctx *s : owner->hydro_sharers) {
        if (!s || s->hydro_owner != owner) continue;
        if (s->stream) (void)hipStreamSynchronize(s->stream);
        forget_shared_frame(s);
        s->last_error = std::string("the hydro frame this context shared is gone (its owner ") + why + "): share or stage a frame again";
    }
    owner->hydro_sharers.clear();
}

extern "C" void mcrat_hip_destroy(mcrat_hip_ctx *c)
{
    if (!c) return;
    if (c->parent) { destroy_view(c); return; }
    detach_sharers(c, "was destroyed");
    for (mcrat_hip_ctx *v : c->views)
        if (v) { v->parent = nullptr; destroy_view(v); }
    c->views.clear();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_graph(c);
    if (c->hydro_owner) {                                    // another context's frame: nothing of it is ours to free
        std::lock_guard<std::mutex> lock(g_share_mutex);
        auto &v = c->hydro_owner->hydro_sharers;
        v.erase(std::remove(v.begin(), v.end(), c), v.end());
        c->hcol_buf = nullptr; c->d_hot_table = nullptr; c->hydro_owner = nullptr;
    }
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    if (c->d_desc) (void)hipFree(c->d_desc);
    if (c->h_desc) (void)hipHostFree(c->h_desc);
    if (c->d_fq) (void)hipFree(c->d_fq);
    if (c->h_fq) (void)hipHostFree(c->h_fq);
    if (c->ph_buf) (void)hipFree(c->ph_buf);
    for (int k = 0; k < 2; ++k) { if (c->pin_ring[k]) (void)hipHostFree(c->pin_ring[k]); if (c->pin_ev[k]) (void)hipEventDestroy(c->pin_ev[k]); }
    if (c->ph_snap) (void)hipFree(c->ph_snap);
    if (c->ph_cap) (void)hipFree(c->ph_cap);
    if (c->aos_buf) (void)hipFree(c->aos_buf);
    if (c->hy_buf) (void)hipFree(c->hy_buf);
    if (c->grid_buf) (void)hipFree(c->grid_buf);
    if (c->hcol_buf) (void)hipFree(c->hcol_buf);
    if (c->raw_buf) (void)hipFree(c->raw_buf);
    if (c->grid_count) (void)hipFree(c->grid_count);
    if (c->d_grid_total) (void)hipFree(c->d_grid_total);
    if (c->d_cs_hook) (void)hipFree(c->d_cs_hook);
    if (c->d_cs_args) (void)hipFree(c->d_cs_args);
    if (c->d_fast) (void)hipFree(c->d_fast);
    if (c->partials) (void)hipFree(c->partials);
    if (c->shortlist) (void)hipFree(c->shortlist);
    if (c->d_hot_table) (void)hipFree(c->d_hot_table);
    if (c->d_tape) (void)hipFree(c->d_tape);
    if (c->d_tape_cursor) (void)hipFree(c->d_tape_cursor);
    if (c->d_table_fallbacks) (void)hipFree(c->d_table_fallbacks);
    if (c->d_sc) (void)hipFree(c->d_sc);
    if (c->sc_own_send && c->sc_send) (void)hipFree(c->sc_send);
    if (c->sc_own_recv && c->sc_recv) (void)hipFree(c->sc_recv);
    if (c->sc_flags) (void)hipFree(c->sc_flags);
    if (c->sc_gather) (void)hipFree(c->sc_gather);
    if (c->d_rstates) (void)hipFree(c->d_rstates);
    if (c->h_rstates) (void)hipHostFree(c->h_rstates);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->h_state) (void)hipHostFree(c->h_state);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->h_red) (void)hipHostFree(c->h_red);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// HIP's current device is a property of the host thread: a thread other than the one that created the context must select the context's
// device before it calls into the library (allocations made on the way would otherwise land on the thread's default device)
extern "C" int mcrat_hip_bind_thread(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (hipSetDevice(c->cfg.device) != hipSuccess) return MCRAT_HIP_ENODEV;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_synchronize(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCRAT_HIP_OK;
}

extern "C" size_t mcrat_hip_device_bytes(const mcrat_hip_ctx *c)
{
    if (!c) return 0;
    return c->ph_bytes + c->hy_bytes + c->grid_bytes + c->grid_count_cap * sizeof(unsigned) + sizeof(Shortlist) + sizeof(LoopState) + (size_t)c->partials_cap * sizeof(Cand) +
           sizeof(ReducePartial) * mcrat_hip_ctx::RED_BLOCKS;
}

extern "C" int mcrat_hip_create_hot_cross_section(mcrat_hip_ctx *c, double *thermal_table, int n_ph_e, int n_t, double log_ph_e_min,
                                                  double log_ph_e_max, double log_t_min, double log_t_max, long long calls, uint64_t seed)
{
    if (!c || !thermal_table || n_ph_e < 1 || n_t < 1 || n_ph_e > 100000 || n_t > 100000 || calls < 1 || !(log_ph_e_max > log_ph_e_min) ||
        !(log_t_max > log_t_min))
        return MCRAT_HIP_EINVAL;
    const size_t count = (size_t)(n_ph_e + 1) * (size_t)(n_t + 1);
    if (count > 0x7fffffffull) return MCRAT_HIP_EINVAL;
    double *d = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, count * sizeof(double)));
    HotTableParams p{n_ph_e, n_t, log_ph_e_min, log_ph_e_max, log_t_min, log_t_max, calls, (unsigned long long)seed};
    hipError_t e = launch_hot_table(p, d, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(thermal_table, d, count * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
    return MCRAT_HIP_OK;
}

// ---------------------------------------------------------------------------------------------- hydro staging
namespace {

struct GridHost {
    std::vector<int> start, cells;
    std::vector<unsigned> hints;      // per bucket, see BucketDir
    double ext_lo[3] = {0, 0, 0}, ext_hi[3] = {0, 0, 0}, ncell[3] = {1, 1, 1}, f0 = 1.0;   // grid_plan()
    double org[3] = {0, 0, 0}, inv[3] = {0, 0, 0};
    int dim[3] = {1, 1, 1}, logmap[3] = {0, 0, 0}, naxes = 2;
};

inline int bucket_of(double x, int logmap, double org, double inv, int dim)
{
    double u = logmap ? std::log(x) : x;
    double f = std::floor((u - org) * inv);
    if (!(f == f)) return 0;
    if (f < 0.0) return 0;
    if (f > (double)(dim - 1)) return dim - 1;
    return (int)f;
}

// Exact accelerator for the reference's linear findContainingBlock (geometry.c:350-391).  Every cell is
// entered into all buckets its closed extent, widened by 1e-9 relative, touches; cells are visited in
// ascending index so each bucket list is ascending.  The device walks the list of the bucket holding the
// point and applies the reference's own closed-interval test, so it returns the lowest-index containing
// cell exactly as the linear scan does.  (The reference's own buildSpatialGrid, geometry.c:526-676, is
// disabled at HEAD and tests DIMENSIONS against the wrong constants; it is not reproduced.)
// extents, log mapping and typical cell widths of the mesh: what fixes the bucket grid up to a scale factor f
struct MeshStats {      // what the plan needs from the mesh: per axis the extent, the smallest and largest width, and every
    double lo[3], hi[3], smin[3], smax[3];       // stride-th cell's centre and width (stride = max(1, M / 4096))
    std::vector<double> sc[3], ss[3];
};
inline int plan_stride(int M) { return std::max(1, M / 4096); }

bool grid_plan_from_stats(const MeshStats &ms, int M, int naxes, GridHost &g)
{
    g.naxes = naxes;
    double *ext_lo = g.ext_lo, *ext_hi = g.ext_hi, *ncell = g.ncell;
    for (int k = 0; k < naxes; ++k) {
        const double lo = ms.lo[k], hi = ms.hi[k], smin = ms.smin[k], smax = ms.smax[k];
        if (!(hi > lo) || !(smin > 0)) return false;
        g.logmap[k] = (lo > 0 && smax / smin > 4.0) ? 1 : 0;
        // typical cell width in the mapped coordinate: median over a sample
        std::vector<double> w;
        for (size_t i = 0; i < ms.sc[k].size(); ++i) {
            const double a = ms.sc[k][i] - 0.5 * ms.ss[k][i], b = ms.sc[k][i] + 0.5 * ms.ss[k][i];
            w.push_back(g.logmap[k] ? std::log(b) - std::log(std::max(a, 1e-300)) : b - a);
        }
        if (w.empty()) return false;
        // bucket width = a small typical cell (lower quartile): in a mesh with two refinement levels the fine cells,
        // where the photons are, then get buckets of their own size instead of lists of nine
        std::nth_element(w.begin(), w.begin() + w.size() / 4, w.end());
        const double med = w[w.size() / 4];
        ext_lo[k] = g.logmap[k] ? std::log(lo) : lo;
        ext_hi[k] = g.logmap[k] ? std::log(hi) : hi;
        ncell[k] = std::max(1.0, (ext_hi[k] - ext_lo[k]) / med);
    }
    double prod = 1;
    for (int k = 0; k < naxes; ++k) prod *= ncell[k];
    const double target = std::min(std::max(4.0 * (double)M, 1.0), 16777216.0);
    g.f0 = (prod > target) ? std::pow(target / prod, 1.0 / naxes) : 1.0;
    return true;
}

bool grid_plan(const mcrat_hip_hydro *h, int naxes, GridHost &g)
{
    const int M = h->num_elements;
    const double *c[3] = {h->r0, h->r1, h->r2};
    const double *s[3] = {h->r0_size, h->r1_size, h->r2_size};
    MeshStats ms;
    for (int k = 0; k < naxes; ++k) {
        double lo = INFINITY, hi = -INFINITY, smin = INFINITY, smax = 0;
        for (int i = 0; i < M; ++i) {
            lo = std::min(lo, c[k][i] - 0.5 * s[k][i]);
            hi = std::max(hi, c[k][i] + 0.5 * s[k][i]);
            smin = std::min(smin, s[k][i]);
            smax = std::max(smax, s[k][i]);
        }
        ms.lo[k] = lo; ms.hi[k] = hi; ms.smin[k] = smin; ms.smax[k] = smax;
        for (int i = 0; i < M; i += plan_stride(M)) { ms.sc[k].push_back(c[k][i]); ms.ss[k].push_back(s[k][i]); }
    }
    return grid_plan_from_stats(ms, M, naxes, g);
}

// the bucket grid for scale factor f; returns the number of buckets
long long grid_dims(GridHost &g, int naxes, double f)
{
    long long nb = 1;
    for (int k = 0; k < 3; ++k) {
        g.dim[k] = 1; g.org[k] = 0; g.inv[k] = 0;
        if (k < naxes) {
            // buckets of about one cell, shifted by half a bucket against the mesh: on a regular mesh a bucket then
            // straddles 2 cells per axis (4 in 2-D); aligned buckets would each touch 3 per axis because cell faces lie
            // on bucket faces
            const int nbk = (int)std::max(1.0, std::min(65536.0, std::floor(g.ncell[k] * f)));
            const double width = (g.ext_hi[k] - g.ext_lo[k]) / nbk;
            g.dim[k] = nbk + 1;
            g.org[k] = g.ext_lo[k] - 0.5 * width;
            g.inv[k] = 1.0 / width;
        }
        nb *= g.dim[k];
    }
    return nb;
}

// the host build (MCRAT_HIP_HOST_GRID=1): the cross-check of grid_build.hip
bool build_grid(const mcrat_hip_hydro *h, int naxes, GridHost &g)
{
    const int M = h->num_elements;
    const double *c[3] = {h->r0, h->r1, h->r2};
    const double *s[3] = {h->r0_size, h->r1_size, h->r2_size};
    if (!grid_plan(h, naxes, g)) return false;
    double f = g.f0;

    for (int attempt = 0; attempt < 12; ++attempt, f *= 0.5) {
        const long long nb = grid_dims(g, naxes, f);
        std::vector<long long> count((size_t)nb + 1, 0);
        auto range = [&](int i, int k, int &b0, int &b1) {
            const double m = 1e-9 * (std::fabs(c[k][i]) + s[k][i]);
            double a = c[k][i] - 0.5 * s[k][i] - m, b = c[k][i] + 0.5 * s[k][i] + m;
            if (g.logmap[k] && a <= 0) a = 1e-300;
            b0 = bucket_of(a, g.logmap[k], g.org[k], g.inv[k], g.dim[k]);
            b1 = bucket_of(b, g.logmap[k], g.org[k], g.inv[k], g.dim[k]);
        };
        long long total = 0;
        for (int i = 0; i < M; ++i) {
            int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            for (int k = 0; k < naxes; ++k) range(i, k, lo[k], hi[k]);
            for (int z = lo[2]; z <= hi[2]; ++z)
                for (int y = lo[1]; y <= hi[1]; ++y)
                    for (int x = lo[0]; x <= hi[0]; ++x) count[((size_t)z * g.dim[1] + y) * g.dim[0] + x + 1]++;
            total += (long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
            if (total > 64LL * M + 1024) break;
        }
        if (total > 64LL * M + 1024) continue;   // too fine for this mesh: coarsen and retry
        if (total > 2000000000LL) continue;
        for (size_t b = 0; b < (size_t)nb; ++b) count[b + 1] += count[b];
        g.start.resize((size_t)nb + 1);
        for (size_t b = 0; b <= (size_t)nb; ++b) g.start[b] = (int)count[b];
        if (getenv("MCRAT_HIP_VERBOSE"))
            fprintf(stderr, "mcrat_hip: cell-lookup grid %d x %d x %d buckets, %lld entries for %d cells (%.2f per bucket)\n",
                    g.dim[0], g.dim[1], g.dim[2], total, M, (double)total / (double)nb);
        g.cells.assign((size_t)total, -1);
        std::vector<int> fill(g.start.begin(), g.start.end() - 1);
        for (int i = 0; i < M; ++i) {
            int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            for (int k = 0; k < naxes; ++k) range(i, k, lo[k], hi[k]);
            for (int z = lo[2]; z <= hi[2]; ++z)
                for (int y = lo[1]; y <= hi[1]; ++y)
                    for (int x = lo[0]; x <= hi[0]; ++x) g.cells[(size_t)fill[((size_t)z * g.dim[1] + y) * g.dim[0] + x]++] = i;
        }
        if (nb > (long long)GRID_CODE_BUCKET_MASK) continue;      // bucket codes keep 27 bits for the index
        // hints: an octant of a bucket gets the entry whose cell is the ONLY one reaching into the octant's interior
        // (extents shrunk by the 1e-9 margin the lists were widened by, so cells that merely abut do not count)
        g.hints.assign((size_t)nb, 0);
        const int nocts = 1 << naxes;
        auto mapped = [&](double x, int k) { return g.logmap[k] ? std::log(std::max(x, 1e-300)) : x; };
        for (long long b = 0; b < nb; ++b) {
            int bi[3] = {(int)(b % g.dim[0]), (int)((b / g.dim[0]) % g.dim[1]), (int)(b / ((long long)g.dim[0] * g.dim[1]))};
            const int e0 = g.start[(size_t)b], n = g.start[(size_t)b + 1] - e0;
            unsigned h = 0;
            for (int o = 0; o < 8; ++o) {
                unsigned pick = GRID_NO_HINT;
                if (o < nocts) {
                    int found = 0;
                    for (int e = 0; e < n && found < 2; ++e) {
                        const int ci = g.cells[(size_t)e0 + e];
                        bool reaches = true;
                        for (int k = 0; k < naxes && reaches; ++k) {
                            const double w = 1.0 / g.inv[k];
                            const double olo = g.org[k] + (bi[k] + 0.5 * ((o >> k) & 1)) * w, ohi = olo + 0.5 * w;
                            const double m = 1e-9 * (std::fabs(c[k][ci]) + s[k][ci]);
                            const double clo = mapped(c[k][ci] - 0.5 * s[k][ci] + m, k), chi = mapped(c[k][ci] + 0.5 * s[k][ci] - m, k);
                            reaches = (clo < ohi) && (chi > olo);
                        }
                        if (reaches) { found += 1; if (e < (int)GRID_NO_HINT) pick = (unsigned)e; else found = 2; }
                    }
                    if (found != 1) pick = GRID_NO_HINT;
                }
                h |= pick << (4 * o);
            }
            g.hints[(size_t)b] = h;
        }
        if (getenv("MCRAT_HIP_VERBOSE")) {
            long long hinted = 0;
            for (long long b = 0; b < nb; ++b)
                for (int o = 0; o < nocts; ++o) hinted += ((g.hints[(size_t)b] >> (4 * o)) & 15u) != GRID_NO_HINT;
            fprintf(stderr, "mcrat_hip: %.1f %% of the bucket octants have a single-cell hint\n", 100.0 * hinted / ((double)nb * nocts));
        }
        return true;
    }
    return false;
}

}  // namespace

static void apply_hot_table(mcrat_hip_ctx *c)
{
    HydroDev &hy = c->hy;
    const bool table = c->cfg.tau_calculation == MCRAT_HIP_TAU_TABLE && c->d_hot_table;
    hy.hot_table = table ? c->d_hot_table : nullptr;
    hy.hot_n_ph_e = c->hot_n_ph_e; hy.hot_n_t = c->hot_n_t;
    hy.hot_e0 = c->hot_grid[0]; hy.hot_t0 = c->hot_grid[2];
    // the grid steps as hot_x_section.c:464 forms them
    hy.hot_de = table ? (c->hot_grid[1] - c->hot_grid[0]) / c->hot_n_ph_e : 0.0;
    hy.hot_dt = table ? (c->hot_grid[3] - c->hot_grid[2]) / c->hot_n_t : 0.0;
    hy.table_fallbacks = c->d_table_fallbacks;
    hy.hot_fallback_calls = c->hot_fallback_calls;
}

extern "C" int mcrat_hip_table_fallback_calls(mcrat_hip_ctx *c, int calls)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (calls > 0) {
        c->hot_fallback_calls = calls;
        apply_hot_table(c);
        drop_graph(c);
        sync_views(c);
    }
    return c->hot_fallback_calls;
}

extern "C" int mcrat_hip_set_hot_cross_section(mcrat_hip_ctx *c, const double *thermal_table, int n_ph_e, int n_t,
                                               double log_ph_e_min, double log_ph_e_max, double log_t_min, double log_t_max)
{
    if (!c || !thermal_table || n_ph_e < 1 || n_t < 1 || !(log_ph_e_max > log_ph_e_min) || !(log_t_max > log_t_min)) return MCRAT_HIP_EINVAL;
    if (c->parent) return view_refuses(c, "set_hot_cross_section");
    if (c->cfg.tau_calculation != MCRAT_HIP_TAU_TABLE) { c->last_error = "the context was created with TAU_CALCULATION == DIRECT"; return MCRAT_HIP_ESTATE; }
    const size_t count = (size_t)(n_ph_e + 1) * (size_t)(n_t + 1);
    for (size_t k = 0; k < count; ++k)
        if (!(thermal_table[k] == thermal_table[k])) { c->last_error = "NaN in the cross-section table"; return MCRAT_HIP_EINVAL; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    release_shared_hydro(c);
    detach_sharers(c, "replaced its cross-section table");   // they read this table too
    if (c->d_hot_table) { (void)hipFree(c->d_hot_table); c->d_hot_table = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->d_hot_table, count * sizeof(double)));
    HIPCHK(c, hipMemcpy(c->d_hot_table, thermal_table, count * sizeof(double), hipMemcpyHostToDevice));
    c->hot_n_ph_e = n_ph_e; c->hot_n_t = n_t;
    c->hot_grid[0] = log_ph_e_min; c->hot_grid[1] = log_ph_e_max; c->hot_grid[2] = log_t_min; c->hot_grid[3] = log_t_max;
    apply_hot_table(c);
    drop_graph(c);
    sync_views(c);
    return MCRAT_HIP_OK;
}

static int ensure_aos(mcrat_hip_ctx *c, size_t bytes);
static int flush_pending(mcrat_hip_ctx *c);

// rank pool: the views read the pool's hydro frame (and cross-section table) through copies of its descriptors
static void sync_views(mcrat_hip_ctx *c)
{
    for (mcrat_hip_ctx *v : c->views) {
        if (!v) continue;
        v->hy = c->hy;
        v->hcol = c->hcol; v->hcol_buf = c->hcol_buf; v->hcol_M = c->hcol_M;
        v->have_hydro = c->have_hydro;
        v->d_hot_table = c->d_hot_table; v->hot_n_ph_e = c->hot_n_ph_e; v->hot_n_t = c->hot_n_t;
        v->hot_fallback_calls = c->hot_fallback_calls;
        for (int k = 0; k < 4; ++k) v->hot_grid[k] = c->hot_grid[k];
        drop_graph(v);
    }
}

// per-context scratch: `n` counters and the 8-byte total.  A view borrows its pool's (one stream, one call at a time): a thousand
// views must not hold a thousand cell-sized arrays.
static int ensure_counts(mcrat_hip_ctx *c, size_t n)
{
    if (c->parent) {
        int rc = ensure_counts(c->parent, n);
        c->grid_count = c->parent->grid_count; c->grid_count_cap = c->parent->grid_count_cap; c->d_grid_total = c->parent->d_grid_total;
        return rc;
    }
    if (!c->d_grid_total) HIPCHK(c, hipMalloc((void **)&c->d_grid_total, sizeof(unsigned long long)));
    if (c->grid_count_cap < n) {
        if (c->grid_count) { HIPCHK(c, hipFree(c->grid_count)); c->grid_count = nullptr; c->grid_count_cap = 0; }
        HIPCHK(c, hipMalloc((void **)&c->grid_count, sizeof(unsigned) * n));
        c->grid_count_cap = n;
    }
    return MCRAT_HIP_OK;
}

static int view_refuses(mcrat_hip_ctx *c, const char *what)
{
    c->last_error = std::string(what) + ": a rank view shares its pool's hydro frame; call this on the pool context";
    return MCRAT_HIP_ESTATE;
}

// The frame's per-cell records and cell-lookup grid.  h == nullptr (the product path): from the device columns c->hcol,
// on the device -- stage_cells_kernel (ingest.hip) + grid_build.hip; the host only plans the bucket grid from the mesh
// statistics the kernel reduces.  h != nullptr (MCRAT_HIP_HOST_GRID=1): staging loop and grid build on the host, the
// cross-check of the device path (tests/test_gpu_parity.py).
static int stage_hydro(mcrat_hip_ctx *c, const mcrat_hip_hydro *h, int M, const double *dom0, const double *dom1, const double *dom2)
{
    if (c->hydro_owner) { c->last_error = "stage_hydro on a context that reads another one's frame"; return MCRAT_HIP_ESTATE; }   // (ensure_hcol released it)
    const bool three = c->kc.dimensions == DIM_THREE, two = c->kc.dimensions == DIM_TWO;
    const int naxes = three ? 3 : 2;
    const bool host_grid = h != nullptr;
    GridHost g;
    if (host_grid && !build_grid(h, naxes, g)) { c->last_error = "cell-lookup grid: degenerate mesh"; return MCRAT_HIP_EINVAL; }
    bool any_hot = false;
    if (host_grid)
        for (int i = 0; i < M; ++i) any_hot = any_hot || (h->temp[i] >= 1e7);

    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_geom = take(sizeof(CellGeom) * M);
    const size_t o_geom2 = three ? take(sizeof(CellGeom2) * M) : 0;
    const size_t o_fluid = take(sizeof(CellFluid) * M);
    const size_t o_temp = take(sizeof(double) * M);
    const size_t o_fc = !two ? take(sizeof(double) * M) : 0;
    const size_t o_k2e = (any_hot || !host_grid) ? take(sizeof(double) * M) : 0;     // device path: known only after staging
    const size_t o_gamma = take(sizeof(double) * M);
    const size_t total = off;

    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->hy_buf && c->hy_bytes < total) { HIPCHK(c, hipFree(c->hy_buf)); c->hy_buf = nullptr; c->hy_bytes = 0; }
    if (!c->hy_buf) { HIPCHK(c, hipMalloc(&c->hy_buf, total)); c->hy_bytes = total; }
    char *base = static_cast<char *>(c->hy_buf);

    if (!host_grid) {
        const int nblk = stage_cells_blocks(M), stride = plan_stride(M), nsamp = (M + stride - 1) / stride;
        const size_t scratch_bytes = sizeof(StagePartial) * (size_t)nblk + sizeof(double) * 2 * naxes * (size_t)nsamp;
        int rc = ensure_aos(c, scratch_bytes);
        if (rc) return rc;
        StagePartial *d_part = static_cast<StagePartial *>(c->aos_buf);
        double *d_samp = reinterpret_cast<double *>(d_part + nblk);
        HIPCHK(c, launch_stage_cells(c->kc.dimensions, c->kc.geometry, c->hcol, M, reinterpret_cast<CellGeom *>(base + o_geom),
                                     three ? reinterpret_cast<CellGeom2 *>(base + o_geom2) : nullptr, reinterpret_cast<CellFluid *>(base + o_fluid),
                                     !two ? reinterpret_cast<double *>(base + o_fc) : nullptr, reinterpret_cast<double *>(base + o_temp),
                                     reinterpret_cast<double *>(base + o_gamma), d_part, d_samp, stride, nsamp, c->stream));
        std::vector<char> hs(scratch_bytes);
        HIPCHK(c, hipMemcpyAsync(hs.data(), c->aos_buf, scratch_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const StagePartial *part = reinterpret_cast<const StagePartial *>(hs.data());
        const double *samp = reinterpret_cast<const double *>(part + nblk);
        MeshStats ms;
        for (int k = 0; k < naxes; ++k) {
            double lo = INFINITY, hi = -INFINITY, smin = INFINITY, smax = 0;
            for (int b = 0; b < nblk; ++b) {
                lo = std::min(lo, part[b].lo[k]);
                hi = std::max(hi, part[b].hi[k]);
                smin = (part[b].smin[k] < smin || !(part[b].smin[k] == part[b].smin[k])) ? part[b].smin[k] : smin;
                smax = std::max(smax, part[b].smax[k]);
            }
            ms.lo[k] = lo; ms.hi[k] = hi; ms.smin[k] = smin; ms.smax[k] = smax;
            ms.sc[k].assign(samp + (size_t)(2 * k) * nsamp, samp + (size_t)(2 * k + 1) * nsamp);
            ms.ss[k].assign(samp + (size_t)(2 * k + 1) * nsamp, samp + (size_t)(2 * k + 2) * nsamp);
        }
        for (int b = 0; b < nblk; ++b) any_hot = any_hot || part[b].any_hot;
        if (!grid_plan_from_stats(ms, M, naxes, g)) { c->last_error = "cell-lookup grid: degenerate mesh"; return MCRAT_HIP_EINVAL; }
    }

    std::vector<char> host(host_grid ? total : 0, 0);
    CellGeom *geom = reinterpret_cast<CellGeom *>(host.data() + o_geom);
    CellFluid *fluid = reinterpret_cast<CellFluid *>(host.data() + o_fluid);
    // the per-cell part of hydroVectorToCartesian (geometry.c:189-253) is applied here, once per frame; the device adds the
    // photon-azimuth part (physics.hpp, cell_beta)
    double *fc = !two ? reinterpret_cast<double *>(host.data() + o_fc) : nullptr;
    const int geomv = c->kc.geometry;
    for (int i = 0; host_grid && i < M; ++i) {
        geom[i].c0 = h->r0[i]; geom[i].c1 = h->r1[i]; geom[i].s0 = h->r0_size[i]; geom[i].s1 = h->r1_size[i];
        const double v0 = h->v0[i], v1 = h->v1[i], v2 = two ? 0.0 : h->v2[i];
        double fa, fb, fcv = 0.0;
        if (!three) {
            if (geomv == GEOM_SPHERICAL) {
                const double th = h->r1[i];
                fa = v0 * std::sin(th) + v1 * std::cos(th);
                fb = v0 * std::cos(th) - v1 * std::sin(th);
            } else {
                fa = v0;
                fb = v1;
            }
            fcv = v2;
        } else if (geomv == GEOM_CARTESIAN) {
            fa = v0; fb = v1; fcv = v2;
        } else if (geomv == GEOM_SPHERICAL) {
            const double x1 = h->r1[i], x2 = h->r2[i];
            fa = v0 * std::sin(x1) * std::cos(x2) + v1 * std::cos(x1) * std::cos(x2) - v2 * std::sin(x2);
            fb = v0 * std::sin(x1) * std::sin(x2) + v1 * std::cos(x1) * std::sin(x2) + v2 * std::cos(x2);
            fcv = v0 * std::cos(x1) - v1 * std::sin(x1);
        } else {   // POLAR
            const double x1 = h->r1[i];
            fa = v0 * std::cos(x1) - v1 * std::sin(x1);
            fb = v0 * std::sin(x1) + v1 * std::cos(x1);
            fcv = v2;
        }
        if (fc) fc[i] = fcv;
        cell_staged_operands(fa, fb, fcv, h->gamma[i], h->dens_lab[i], fluid[i]);
    }
    if (host_grid) {
        if (three) {
            CellGeom2 *g2 = reinterpret_cast<CellGeom2 *>(host.data() + o_geom2);
            for (int i = 0; i < M; ++i) { g2[i].c2 = h->r2[i]; g2[i].s2 = h->r2_size[i]; }
        }
        memcpy(host.data() + o_temp, h->temp, sizeof(double) * M);
        memcpy(host.data() + o_gamma, h->gamma, sizeof(double) * M);
        HIPCHK(c, hipMemcpy(c->hy_buf, host.data(), total, hipMemcpyHostToDevice));
    }

    HydroDev &hy = c->hy;
    hy.geom = reinterpret_cast<const CellGeom *>(base + o_geom);
    hy.geom2 = three ? reinterpret_cast<const CellGeom2 *>(base + o_geom2) : nullptr;
    hy.fluid = reinterpret_cast<const CellFluid *>(base + o_fluid);
    hy.temp = reinterpret_cast<const double *>(base + o_temp);
    hy.fluid_c = !two ? reinterpret_cast<const double *>(base + o_fc) : nullptr;
    hy.k2e = any_hot ? reinterpret_cast<const double *>(base + o_k2e) : nullptr;
    hy.gamma = reinterpret_cast<const double *>(base + o_gamma);
    hy.M = M;
    hy.dom0[0] = dom0[0]; hy.dom0[1] = dom0[1];
    hy.dom1[0] = dom1[0]; hy.dom1[1] = dom1[1];
    hy.dom2[0] = dom2[0]; hy.dom2[1] = dom2[1];

    // ---- the grid
    long long nb = 0, entries_total = 0;
    if (host_grid) {
        nb = (long long)g.start.size() - 1;
        entries_total = (long long)g.cells.size();
    } else {
        { int rc = ensure_counts(c, 1); if (rc) return rc; }
        double f = g.f0;
        bool ok = false;
        for (int attempt = 0; attempt < 12 && !ok; ++attempt, f *= 0.5) {
            nb = grid_dims(g, naxes, f);
            if (nb > (long long)GRID_CODE_BUCKET_MASK) continue;              // bucket codes keep 27 bits for the index
            { int rc = ensure_counts(c, (size_t)nb); if (rc) return rc; }
            GridPlan plan;
            for (int k = 0; k < 3; ++k) { plan.org[k] = g.org[k]; plan.inv[k] = g.inv[k]; plan.dim[k] = g.dim[k]; plan.logmap[k] = g.logmap[k]; }
            plan.naxes = naxes;
            HIPCHK(c, grid_count(plan, hy.geom, hy.geom2, M, c->grid_count, nb, c->d_grid_total, c->stream));
            unsigned long long tot = 0;
            HIPCHK(c, hipMemcpyAsync(&tot, c->d_grid_total, sizeof tot, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (tot > (unsigned long long)(64LL * M + 1024) || tot > 2000000000ull) continue;   // too fine for this mesh: coarsen and retry
            entries_total = (long long)tot;
            ok = true;
            if (getenv("MCRAT_HIP_VERBOSE"))
                fprintf(stderr, "mcrat_hip: cell-lookup grid %d x %d x %d buckets, %lld entries for %d cells (%.2f per bucket), built on the device\n",
                        g.dim[0], g.dim[1], g.dim[2], entries_total, M, (double)entries_total / (double)nb);
        }
        if (!ok) { c->last_error = "cell-lookup grid: no bucket size fits this mesh"; return MCRAT_HIP_EINVAL; }
    }
    size_t goff = 0;
    auto gtake = [&](size_t bytes) { size_t o = goff; goff = align_up(goff + bytes, 256); return o; };
    const size_t o_dir = gtake(sizeof(BucketDir) * (size_t)nb);
    const size_t o_cells = gtake(sizeof(FatCell) * std::max<size_t>((size_t)entries_total, 1));
    const size_t o_start = gtake(sizeof(int) * ((size_t)nb + 1));                // device build only: scratch
    const size_t o_scan = gtake(sizeof(int) * grid_scan_scratch_ints(nb));
    const size_t o_entries = gtake(sizeof(int) * std::max<size_t>((size_t)entries_total, 1));
    const size_t gtotal = goff;
    if (c->grid_buf && c->grid_bytes < gtotal) { HIPCHK(c, hipFree(c->grid_buf)); c->grid_buf = nullptr; c->grid_bytes = 0; }
    if (!c->grid_buf) { HIPCHK(c, hipMalloc(&c->grid_buf, gtotal)); c->grid_bytes = gtotal; }
    char *gbase = static_cast<char *>(c->grid_buf);
    if (host_grid) {
        std::vector<char> gh(o_start, 0);
        BucketDir *dir = reinterpret_cast<BucketDir *>(gh.data() + o_dir);
        for (size_t b = 0; b < (size_t)nb; ++b) {
            dir[b].e0 = g.start[b]; dir[b].n = g.start[b + 1] - g.start[b]; dir[b].hints = g.hints[b]; dir[b].pad = 0;
        }
        // bucket lists as complete copies of the member cells' records (device_types.hpp, FatCell)
        FatCell *fat = reinterpret_cast<FatCell *>(gh.data() + o_cells);
        for (size_t e = 0; e < g.cells.size(); ++e) {
            const int ci = g.cells[e];
            FatCell &f = fat[e];
            f.c0 = geom[ci].c0; f.c1 = geom[ci].c1; f.s0 = geom[ci].s0; f.s1 = geom[ci].s1;
            f.a = fluid[ci].a; f.b = fluid[ci].b; f.c = fluid[ci].c; f.w = fluid[ci].w;
            f.nsig = fluid[ci].nsig; f.gam = fluid[ci].gam;
            f.c2 = three ? h->r2[ci] : 0.0; f.s2 = three ? h->r2_size[ci] : 0.0;
            f.cell = ci; f.pad = 0; f.pad2[0] = f.pad2[1] = f.pad2[2] = 0.0;
        }
        HIPCHK(c, hipMemcpy(gbase, gh.data(), o_start, hipMemcpyHostToDevice));
    } else {
        GridPlan plan;
        for (int k = 0; k < 3; ++k) { plan.org[k] = g.org[k]; plan.inv[k] = g.inv[k]; plan.dim[k] = g.dim[k]; plan.logmap[k] = g.logmap[k]; }
        plan.naxes = naxes;
        HIPCHK(c, grid_build(plan, hy.geom, hy.geom2, hy.fluid, hy.fluid_c, M, c->grid_count, reinterpret_cast<int *>(gbase + o_start),
                             reinterpret_cast<int *>(gbase + o_scan), reinterpret_cast<int *>(gbase + o_entries),
                             reinterpret_cast<FatCell *>(gbase + o_cells), reinterpret_cast<BucketDir *>(gbase + o_dir), nb, entries_total, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    hy.grid.dir = reinterpret_cast<const BucketDir *>(gbase + o_dir);
    hy.grid.cells = reinterpret_cast<const FatCell *>(gbase + o_cells);
    for (int k = 0; k < 3; ++k) {
        hy.grid.org[k] = g.org[k]; hy.grid.inv[k] = g.inv[k]; hy.grid.dim[k] = g.dim[k]; hy.grid.logmap[k] = g.logmap[k];
    }
    hy.grid.naxes = g.naxes;
    apply_hot_table(c);
    if (any_hot) {
        HIPCHK(c, launch_k2e(hy.temp, reinterpret_cast<double *>(base + o_k2e), M, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    c->have_hydro = true;
    drop_graph(c);
    sync_views(c);
    return MCRAT_HIP_OK;
}

// a context that reads another one's staged frame (mcrat_hip_share_hydro) holds that context's pointers: forget them before staging its own
static void release_shared_hydro(mcrat_hip_ctx *c)
{
    std::lock_guard<std::mutex> lock(g_share_mutex);
    if (!c->hydro_owner) return;
    auto &v = c->hydro_owner->hydro_sharers;
    v.erase(std::remove(v.begin(), v.end(), c), v.end());
    forget_shared_frame(c);
}

// Several contexts on one GPU that are in the same hydro frame (rank pools on their own streams, bench.py --pools) need
// one copy of the frame, its per-cell records and its lookup grid, not one each: `ctx` reads `owner`'s staged frame (and cross-section table)
// from now on.  The owner must keep that frame (not re-stage, not be destroyed) while others read it; staging a frame on `ctx` itself, or
// sharing again, ends the arrangement.  Both contexts must be of the same DIMENSIONS / GEOMETRY / TAU_CALCULATION on the same device.
extern "C" int mcrat_hip_share_hydro(mcrat_hip_ctx *c, mcrat_hip_ctx *owner)
{
    if (!c || !owner || c == owner) return MCRAT_HIP_EINVAL;
    if (c->parent) return view_refuses(c, "share_hydro");
    if (owner->parent) owner = owner->parent;
    if (!owner->have_hydro || owner->hydro_owner == c) return MCRAT_HIP_ESTATE;
    if (owner->hydro_owner) owner = owner->hydro_owner;
    if (c->kc.dimensions != owner->kc.dimensions || c->kc.geometry != owner->kc.geometry || c->kc.table != owner->kc.table ||
        c->cfg.device != owner->cfg.device) {
        c->last_error = "share_hydro: the two contexts differ in DIMENSIONS / GEOMETRY / TAU_CALCULATION or device";
        return MCRAT_HIP_EINVAL;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(owner->stream));            // the owner's staging kernels have finished
    release_shared_hydro(c);
    detach_sharers(c, "shares another context's frame now");   // its own buffers go
    if (c->hy_buf) { HIPCHK(c, hipFree(c->hy_buf)); c->hy_buf = nullptr; c->hy_bytes = 0; }
    if (c->grid_buf) { HIPCHK(c, hipFree(c->grid_buf)); c->grid_buf = nullptr; c->grid_bytes = 0; }
    if (c->hcol_buf) { HIPCHK(c, hipFree(c->hcol_buf)); c->hcol_buf = nullptr; c->hcol_bytes = 0; }
    if (c->d_hot_table) { HIPCHK(c, hipFree(c->d_hot_table)); c->d_hot_table = nullptr; }
    c->hy = owner->hy;
    c->hcol = owner->hcol; c->hcol_buf = owner->hcol_buf; c->hcol_M = owner->hcol_M;
    c->d_hot_table = owner->d_hot_table; c->hot_n_ph_e = owner->hot_n_ph_e; c->hot_n_t = owner->hot_n_t;
    for (int k = 0; k < 4; ++k) c->hot_grid[k] = owner->hot_grid[k];
    apply_hot_table(c);                                        // (the count of lookups outside the table stays this context's own)
    c->have_hydro = true;
    {
        std::lock_guard<std::mutex> lock(g_share_mutex);
        c->hydro_owner = owner;
        owner->hydro_sharers.push_back(c);
    }
    c->frame_open = false;
    drop_graph(c);
    sync_views(c);
    return MCRAT_HIP_OK;
}

// struct hydro_dataframe's columns on the device: 16 arrays of M doubles
static int ensure_hcol(mcrat_hip_ctx *c, int M)
{
    release_shared_hydro(c);
    detach_sharers(c, "staged another frame");              // the buffers they point into are about to be rewritten or freed
    const size_t stride = align_up(sizeof(double) * (size_t)M, 256), total = 19 * stride;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->hcol_buf && c->hcol_bytes < total) { HIPCHK(c, hipFree(c->hcol_buf)); c->hcol_buf = nullptr; c->hcol_bytes = 0; }
    if (!c->hcol_buf) { HIPCHK(c, hipMalloc(&c->hcol_buf, total)); c->hcol_bytes = total; }
    char *b = static_cast<char *>(c->hcol_buf);
    double **cols[19] = {&c->hcol.r0, &c->hcol.r1, &c->hcol.r2, &c->hcol.s0, &c->hcol.s1, &c->hcol.s2, &c->hcol.v0, &c->hcol.v1, &c->hcol.v2,
                         &c->hcol.dens, &c->hcol.dens_lab, &c->hcol.pres, &c->hcol.temp, &c->hcol.gamma, &c->hcol.r, &c->hcol.theta,
                         &c->hcol.B0, &c->hcol.B1, &c->hcol.B2};
    for (int k = 0; k < 19; ++k) *cols[k] = reinterpret_cast<double *>(b + k * stride);
    c->hcol_M = M;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_set_hydro(mcrat_hip_ctx *c, const mcrat_hip_hydro *h)
{
    if (!c || !h || h->num_elements <= 0) return MCRAT_HIP_EINVAL;
    if (c->parent) return view_refuses(c, "set_hydro");
    const int M = h->num_elements;
    const bool three = c->kc.dimensions == DIM_THREE, two = c->kc.dimensions == DIM_TWO;
    if (!h->r0 || !h->r1 || !h->r0_size || !h->r1_size || !h->v0 || !h->v1 || !h->dens_lab || !h->temp || !h->gamma) return MCRAT_HIP_EINVAL;
    if (three && (!h->r2 || !h->r2_size)) return MCRAT_HIP_EINVAL;
    if (!two && !h->v2) return MCRAT_HIP_EINVAL;
    c->have_hydro = false;
    sync_views(c);
    // the columns cross PCIe as they lie in the caller's memory; per-cell records and the cell-lookup grid are produced on
    // the device (stage_hydro)
    int rc = ensure_hcol(c, M);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->hcol_buf, 0, c->hcol_bytes, c->stream));
    const struct { double *dst; const double *src; } copy[12] = {
        {c->hcol.r0, h->r0}, {c->hcol.r1, h->r1}, {c->hcol.r2, three ? h->r2 : nullptr}, {c->hcol.s0, h->r0_size}, {c->hcol.s1, h->r1_size},
        {c->hcol.s2, three ? h->r2_size : nullptr}, {c->hcol.v0, h->v0}, {c->hcol.v1, h->v1}, {c->hcol.v2, !two ? h->v2 : nullptr},
        {c->hcol.dens_lab, h->dens_lab}, {c->hcol.temp, h->temp}, {c->hcol.gamma, h->gamma}};
    for (const auto &cp : copy)
        if (cp.src) HIPCHK(c, hipMemcpyAsync(cp.dst, cp.src, sizeof(double) * (size_t)M, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_fill_spherical(c->kc.dimensions, c->kc.geometry, c->hcol, M, c->stream));
    return stage_hydro(c, getenv("MCRAT_HIP_HOST_GRID") ? h : nullptr, M, h->r0_domain, h->r1_domain, h->r2_domain);
}

extern "C" void mcrat_hip_outflow_defaults(int simulation_type, mcrat_hip_outflow *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->simulation_type = simulation_type;
    if (simulation_type == MCRAT_HIP_CYLINDRICAL_OUTFLOW) { o->gamma_infinity = 100; o->t_comov = 1e5; o->ddensity = 3e-7; }                      // analytic_outflows.c:5
    if (simulation_type == MCRAT_HIP_SPHERICAL_OUTFLOW) { o->gamma_infinity = 100; o->lumi = 1e54; o->r00 = 1e8; }                                // :65
    if (simulation_type == MCRAT_HIP_STRUCTURED_SPHERICAL_OUTFLOW) { o->gamma_infinity = 100; o->lumi = 1e52; o->r00 = 1e8; o->theta_j = 1e-2; o->p = 4; }   // :140
}

namespace {

// device copies of a reader's buffers, packed into c->raw_buf
struct RawPacker {
    mcrat_hip_ctx *c;
    size_t off = 0;
    std::vector<std::pair<size_t, std::pair<const void *, size_t>>> items;
    size_t add(const void *src, size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); items.push_back({o, {src, bytes}}); return o; }
    int upload()
    {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->raw_buf && c->raw_bytes < off) { HIPCHK(c, hipFree(c->raw_buf)); c->raw_buf = nullptr; c->raw_bytes = 0; }
        if (!c->raw_buf) { HIPCHK(c, hipMalloc(&c->raw_buf, off)); c->raw_bytes = off; }
        for (const auto &it : items)
            if (it.second.first) HIPCHK(c, hipMemcpyAsync(static_cast<char *>(c->raw_buf) + it.first, it.second.first, it.second.second, hipMemcpyHostToDevice, c->stream));
        return MCRAT_HIP_OK;
    }
    template <class T> const T *at(size_t o) const { return reinterpret_cast<const T *>(static_cast<const char *>(c->raw_buf) + o); }
};

// the slab for one elem_factor (mclib_flash.c:84-85,309 == mclib_pluto.c:1081-1082,1276)
SlabDev slab_for(const mcrat_hip_ctx *c, const mcrat_hip_slab *s, int elem_factor)
{
    SlabDev d{};
    d.dimensions = c->kc.dimensions; d.geometry = c->kc.geometry; d.ph_inj_switch = s->ph_inj_switch;
    d.r_inj_095 = 0.95 * s->r_inj;
    if (s->ph_inj_switch == 0) {
        d.r_lo = s->min_r - elem_factor * C_LIGHT / s->fps;
        d.r_hi = s->max_r + elem_factor * C_LIGHT / s->fps;
        d.th_lo = s->min_theta - 2 * 0.017453292519943295;
        d.th_hi = s->max_theta + 2 * 0.017453292519943295;
    }
    return d;
}

// the part of getHydroData both readers share: count with a growing elem_factor, scan, write, fill r/theta, overwrite
// with the analytic outflow, stage
template <class Count, class Write>
int ingest_common(mcrat_hip_ctx *c, long long n_virtual, const mcrat_hip_slab *slab, const mcrat_hip_outflow *outflow,
                  mcrat_hip_ingest_result *result, Count count, Write write)
{
    const long long nblk = ingest_blocks(n_virtual);
    if (nblk <= 0 || nblk > 0x7fffffffLL) return MCRAT_HIP_EINVAL;
    { int rc_ = ensure_counts(c, (size_t)nblk); if (rc_) return rc_; }
    // CYCLOSYNCHROTRON_SWITCH is OFF in this engine (mcrat_hip_init refuses it): elem_factor starts at 0 (mclib_flash.c:275-279)
    int elem_factor = 0;
    unsigned long long total = 0;
    SlabDev sd{};
    while (total == 0) {
        elem_factor++;
        if (elem_factor > 1000) {
            c->last_error = "hydro ingest: no cell lies in the requested slab for any elem_factor up to 1000 (the reference would not return)";
            return MCRAT_HIP_EINVAL;
        }
        sd = slab_for(c, slab, elem_factor);
        HIPCHK(c, count(sd));
        HIPCHK(c, hipMemcpyAsync(&total, c->d_grid_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (total > 0x7fffffffull) { c->last_error = "hydro ingest: more than INT_MAX cells selected"; return MCRAT_HIP_EINVAL; }
    const int M = (int)total;
    int rc = ensure_hcol(c, M);
    if (rc) return rc;
    HIPCHK(c, hipMemsetAsync(c->hcol_buf, 0, c->hcol_bytes, c->stream));
    const size_t scan_bytes = sizeof(int) * ((size_t)nblk + 1 + grid_scan_scratch_ints(nblk));
    if ((rc = ensure_aos(c, scan_bytes))) return rc;
    int *start = static_cast<int *>(c->aos_buf), *scratch = start + nblk + 1;
    HIPCHK(c, launch_exclusive_scan(c->grid_count, nblk, start, scratch, (long long)total, c->stream));
    HIPCHK(c, write(sd, start));
    HIPCHK(c, launch_fill_spherical(c->kc.dimensions, c->kc.geometry, c->hcol, M, c->stream));          // mcrat_io.c:1962
    if (outflow && outflow->simulation_type != MCRAT_HIP_SCIENCE) {                                      // mcrat_io.c:1967-1973
        OutflowDev o{outflow->simulation_type, outflow->gamma_infinity, outflow->lumi, outflow->r00, outflow->t_comov, outflow->ddensity,
                     outflow->theta_j, outflow->p};
        HIPCHK(c, launch_outflow_prep(c->kc.dimensions, c->kc.geometry, o, c->hcol, M, c->stream));
    }
    if (result) { result->num_elements = M; result->elem_factor = elem_factor; }
    return stage_hydro(c, nullptr, M, slab->r0_domain, slab->r1_domain, slab->r2_domain);
}

bool slab_ok(const mcrat_hip_slab *s) { return s && s->fps > 0 && (s->ph_inj_switch == 0 || s->ph_inj_switch == 1); }
bool outflow_ok(const mcrat_hip_outflow *o) { return !o || (o->simulation_type >= MCRAT_HIP_SCIENCE && o->simulation_type <= MCRAT_HIP_STRUCTURED_SPHERICAL_OUTFLOW); }

}  // namespace

extern "C" int mcrat_hip_ingest_flash(mcrat_hip_ctx *c, const mcrat_hip_flash_blocks *b, const mcrat_hip_slab *slab, const mcrat_hip_outflow *outflow,
                                      mcrat_hip_ingest_result *result)
{
    if (!c || !b || !slab_ok(slab) || !outflow_ok(outflow)) return MCRAT_HIP_EINVAL;
    if (c->parent) return view_refuses(c, "ingest_flash");
    if (b->n_blocks <= 0 || b->coord_stride < 2 || b->bsize_stride < 2 || !b->coordinates || !b->block_size || !b->node_type || !b->velx || !b->vely ||
        !b->dens || !b->pres)
        return MCRAT_HIP_EINVAL;
    if (c->kc.dimensions != DIM_TWO) { c->last_error = "FLASH frames are two-dimensional (mcrat_io.c:1938: 3D FLASH is not supported)"; return MCRAT_HIP_EINVAL; }
    c->have_hydro = false;
    sync_views(c);
    const size_t nb = (size_t)b->n_blocks;
    RawPacker pk{c};
    const size_t o_coord = pk.add(b->coordinates, sizeof(double) * nb * b->coord_stride), o_bs = pk.add(b->block_size, sizeof(double) * nb * b->bsize_stride);
    const size_t o_node = pk.add(b->node_type, sizeof(int) * nb);
    const size_t o_vx = pk.add(b->velx, sizeof(double) * nb * 64), o_vy = pk.add(b->vely, sizeof(double) * nb * 64);
    const size_t o_d = pk.add(b->dens, sizeof(double) * nb * 64), o_p = pk.add(b->pres, sizeof(double) * nb * 64);
    int rc = pk.upload();
    if (rc) return rc;
    FlashDev f{};
    f.coord = pk.at<double>(o_coord); f.bsize = pk.at<double>(o_bs); f.node = pk.at<int>(o_node);
    f.velx = pk.at<double>(o_vx); f.vely = pk.at<double>(o_vy); f.dens = pk.at<double>(o_d); f.pres = pk.at<double>(o_p);
    f.coord_stride = b->coord_stride; f.bsize_stride = b->bsize_stride; f.n_blocks = b->n_blocks;
    f.L = b->l_scale; f.D = b->d_scale; f.P = b->p_scale;
    if (result) { memset(result, 0, sizeof *result); }
    rc = ingest_common(c, (long long)nb * 64, slab, outflow, result,
                       [&](const SlabDev &sd) { return ingest_count_flash(f, sd, c->grid_count, c->d_grid_total, c->stream); },
                       [&](const SlabDev &sd, const int *start) { return ingest_write_flash(f, sd, start, c->hcol, c->stream); });
    if (rc == MCRAT_HIP_OK && result) {
        long long leaves = 0;
        for (size_t i = 0; i < nb; ++i) leaves += b->node_type[i] == 1;
        result->cells_read = leaves * 64;
    }
    return rc;
}

extern "C" int mcrat_hip_ingest_pluto(mcrat_hip_ctx *c, const mcrat_hip_pluto_grid *g, const mcrat_hip_slab *slab, const mcrat_hip_outflow *outflow,
                                      mcrat_hip_ingest_result *result)
{
    if (!c || !g || !slab_ok(slab) || !outflow_ok(outflow)) return MCRAT_HIP_EINVAL;
    if (c->parent) return view_refuses(c, "ingest_pluto");
    const bool three = c->kc.dimensions == DIM_THREE, two = c->kc.dimensions == DIM_TWO;
    if (g->nx <= 0 || g->ny <= 0 || (three && g->nz <= 0) || !g->x1 || !g->dx1 || !g->x2 || !g->dx2 || !g->rho || !g->vx1 || !g->vx2 || !g->prs)
        return MCRAT_HIP_EINVAL;
    if (three && (!g->x3 || !g->dx3)) return MCRAT_HIP_EINVAL;
    if (!two && !g->vx3) return MCRAT_HIP_EINVAL;
    const int nz = three ? g->nz : 1;
    const size_t cells = (size_t)g->nx * g->ny * nz;
    if (cells > 0x7fffffffull) return MCRAT_HIP_EINVAL;            // the reference's grid_size is an int (mclib_pluto.c:1061)
    c->have_hydro = false;
    sync_views(c);
    RawPacker pk{c};
    const size_t o_x1 = pk.add(g->x1, sizeof(double) * g->nx), o_dx1 = pk.add(g->dx1, sizeof(double) * g->nx);
    const size_t o_x2 = pk.add(g->x2, sizeof(double) * g->ny), o_dx2 = pk.add(g->dx2, sizeof(double) * g->ny);
    const size_t o_x3 = three ? pk.add(g->x3, sizeof(double) * nz) : 0, o_dx3 = three ? pk.add(g->dx3, sizeof(double) * nz) : 0;
    const size_t o_rho = pk.add(g->rho, sizeof(double) * cells), o_v1 = pk.add(g->vx1, sizeof(double) * cells), o_v2 = pk.add(g->vx2, sizeof(double) * cells);
    const size_t o_v3 = !two ? pk.add(g->vx3, sizeof(double) * cells) : 0, o_p = pk.add(g->prs, sizeof(double) * cells);
    int rc = pk.upload();
    if (rc) return rc;
    PlutoDev d{};
    d.nx = g->nx; d.ny = g->ny; d.nz = nz;
    d.x1 = pk.at<double>(o_x1); d.dx1 = pk.at<double>(o_dx1); d.x2 = pk.at<double>(o_x2); d.dx2 = pk.at<double>(o_dx2);
    d.x3 = three ? pk.at<double>(o_x3) : nullptr; d.dx3 = three ? pk.at<double>(o_dx3) : nullptr;
    d.rho = pk.at<double>(o_rho); d.vx1 = pk.at<double>(o_v1); d.vx2 = pk.at<double>(o_v2); d.vx3 = !two ? pk.at<double>(o_v3) : nullptr;
    d.prs = pk.at<double>(o_p);
    d.L = g->l_scale; d.D = g->d_scale; d.P = g->p_scale;
    if (result) { memset(result, 0, sizeof *result); }
    rc = ingest_common(c, (long long)cells, slab, outflow, result,
                       [&](const SlabDev &sd) { return ingest_count_pluto(d, sd, c->grid_count, c->d_grid_total, c->stream); },
                       [&](const SlabDev &sd, const int *start) { return ingest_write_pluto(d, sd, start, c->hcol, c->stream); });
    if (rc == MCRAT_HIP_OK && result) result->cells_read = (long long)cells;
    return rc;
}

extern "C" int mcrat_hip_ingest_chombo(mcrat_hip_ctx *c, const mcrat_hip_chombo *h, const mcrat_hip_slab *slab, const mcrat_hip_outflow *outflow,
                                       mcrat_hip_ingest_result *result)
{
    if (!c || !h || !slab_ok(slab) || !outflow_ok(outflow)) return MCRAT_HIP_EINVAL;
    if (c->parent) return view_refuses(c, "ingest_chombo");
    if (h->num_levels <= 0 || h->num_vars <= 0 || !h->levels || !h->var_names || !h->data) return MCRAT_HIP_EINVAL;
    const bool three = c->kc.dimensions == DIM_THREE;
    const int nd = three ? 3 : 2, bi = 2 * nd, nl = h->num_levels, nv = h->num_vars;
    // the box table in the reader's cell numbering, and the per-level coordinate arrays (mclib_pluto.c:446-517)
    std::vector<ChomboBox> boxes;
    std::vector<int> level_first_box(nl + 1, 0);
    std::vector<double> xs[3], dxs[3];
    long long total = 0;                                         // doubles of all levels: start_displacement (:151-155)
    for (int i = 0; i < nl; ++i) {
        const mcrat_hip_chombo_level &L = h->levels[i];
        if (L.n_boxes < 0 || (L.n_boxes > 0 && (!L.boxes || !L.box_offsets)) || L.data_len < 0 || L.ref_ratio <= 0) return MCRAT_HIP_EINVAL;
        int ext[3] = {1, 1, 1}, cb[3] = {0, 0, 0};
        for (int a = 0; a < nd; ++a) {
            ext[a] = L.prob_domain[nd + a] - L.prob_domain[a] + 1;
            if (ext[a] <= 0) return MCRAT_HIP_EINVAL;
            cb[a] = (int)xs[a].size();
        }
        for (int j = 0; j < ext[0]; ++j) {
            const int g = L.prob_domain[0] + j;
            if (L.logr == 0) { xs[0].push_back(L.dombeg1 + L.dx * (g + 0.5)); dxs[0].push_back(L.dx); }
            else {
                xs[0].push_back(L.dombeg1 * 0.5 * (std::exp(L.dx * (g + 1)) + std::exp(L.dx * g)));
                dxs[0].push_back(L.dombeg1 * (std::exp(L.dx * (g + 1)) - std::exp(L.dx * g)));
            }
        }
        for (int j = 0; j < ext[1]; ++j) { xs[1].push_back(L.dombeg2 + L.dx * L.g_x2stretch * (L.prob_domain[1] + j + 0.5)); dxs[1].push_back(L.dx * L.g_x2stretch); }
        for (int j = 0; three && j < ext[2]; ++j) { xs[2].push_back(L.dombeg3 + L.dx * L.g_x3stretch * (L.prob_domain[2] + j + 0.5)); dxs[2].push_back(L.dx * L.g_x3stretch); }
        level_first_box[i] = (int)boxes.size();
        for (int j = 0; j < L.n_boxes; ++j) {
            const int *b = L.boxes + (size_t)j * bi;
            ChomboBox r{};
            r.level = i;
            long long ncell = 1;
            for (int a = 0; a < 3; ++a) {
                r.lo[a] = a < nd ? b[a] : 0;
                r.n[a] = a < nd ? b[nd + a] - b[a] + 1 : 1;
                r.cb[a] = cb[a];
                // the reader indexes its coordinate arrays with the box's own indices (:541): they must exist
                if (r.n[a] <= 0 || (a < nd && (r.lo[a] < 0 || r.lo[a] + r.n[a] > ext[a]))) { c->last_error = "PLUTO-Chombo ingest: a box lies outside its level's prob_domain"; return MCRAT_HIP_EINVAL; }
                ncell *= r.n[a];
            }
            r.data_off = total + L.box_offsets[j];
            r.first_cell = r.data_off / nv;
            if (L.box_offsets[j] < 0 || L.box_offsets[j] + ncell * nv > L.data_len) { c->last_error = "PLUTO-Chombo ingest: a box's data lies outside its level's data"; return MCRAT_HIP_EINVAL; }
            if (!boxes.empty() && r.first_cell != boxes.back().first_cell + (long long)boxes.back().n[0] * boxes.back().n[1] * boxes.back().n[2]) {
                c->last_error = "PLUTO-Chombo ingest: box data do not follow one another in data:offsets order";
                return MCRAT_HIP_EINVAL;
            }
            boxes.push_back(r);
        }
        total += L.data_len;
    }
    level_first_box[nl] = (int)boxes.size();
    const long long cells = total / nv;
    if (boxes.empty() || cells <= 0 || cells > 0x7fffffffLL || boxes.front().first_cell != 0) return MCRAT_HIP_EINVAL;
    int kv[5] = {-1, -1, -1, -1, -1};
    static const char *want[5] = {"rho", "vx1", "vx2", "vx3", "prs"};
    for (int k = 0; k < nv; ++k)
        for (int w = 0; w < 5; ++w)
            if (h->var_names[k] && strcmp(h->var_names[k], want[w]) == 0) kv[w] = k;
    if (kv[0] < 0 || kv[1] < 0 || kv[2] < 0 || kv[4] < 0 || (c->kc.dimensions != DIM_TWO && kv[3] < 0)) { c->last_error = "PLUTO-Chombo ingest: a component (rho, vx1, vx2, [vx3], prs) is missing"; return MCRAT_HIP_EINVAL; }

    c->have_hydro = false;
    sync_views(c);
    const bool masked = slab->ph_inj_switch != 0;
    RawPacker pk{c};
    const size_t o_box = pk.add(boxes.data(), sizeof(ChomboBox) * boxes.size());
    size_t o_x[3] = {0, 0, 0}, o_dx[3] = {0, 0, 0};
    for (int a = 0; a < nd; ++a) { o_x[a] = pk.add(xs[a].data(), sizeof(double) * xs[a].size()); o_dx[a] = pk.add(dxs[a].data(), sizeof(double) * dxs[a].size()); }
    const size_t o_data = pk.add(h->data, sizeof(double) * (size_t)total);
    const size_t o_mask = masked ? pk.add(nullptr, (size_t)cells) : 0;
    int rc = pk.upload();
    if (rc) return rc;
    ChomboDev d{};
    d.boxes = pk.at<ChomboBox>(o_box); d.n_boxes = (int)boxes.size(); d.cells = cells;
    d.data = pk.at<double>(o_data);
    for (int a = 0; a < nd; ++a) { d.x[a] = pk.at<double>(o_x[a]); d.dx[a] = pk.at<double>(o_dx[a]); }
    for (int w = 0; w < 5; ++w) d.kv[w] = kv[w];
    d.L = h->l_scale; d.D = h->d_scale; d.P = h->p_scale;
    if (masked) {
        unsigned char *mask = static_cast<unsigned char *>(c->raw_buf) + o_mask;
        HIPCHK(c, hipMemsetAsync(mask, 0, (size_t)cells, c->stream));
        for (int i = nl - 2; i >= 0; --i)                      // :206-345
            HIPCHK(c, launch_chombo_mask(d.boxes, level_first_box[i], level_first_box[i + 1], level_first_box[i + 1], level_first_box[i + 2],
                                         h->levels[i].ref_ratio, three ? 1 : 0, mask, c->stream));
        d.covered = mask;
    }
    if (result) { memset(result, 0, sizeof *result); }
    rc = ingest_common(c, cells, slab, outflow, result,
                       [&](const SlabDev &sd) { return ingest_count_chombo(d, sd, c->grid_count, c->d_grid_total, c->stream); },
                       [&](const SlabDev &sd, const int *start) { return ingest_write_chombo(d, sd, start, c->hcol, c->stream); });
    if (rc == MCRAT_HIP_OK && result) result->cells_read = cells;
    return rc;
}

extern "C" int mcrat_hip_set_hydro_extras(mcrat_hip_ctx *c, const double *dens, const double *B0, const double *B1, const double *B2)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->parent) return view_refuses(c, "set_hydro_extras");
    if (!c->have_hydro || !c->hcol_buf) return MCRAT_HIP_ESTATE;
    const size_t bytes = sizeof(double) * (size_t)c->hcol_M;
    const struct { double *dst; const double *src; } copy[4] = {{c->hcol.dens, dens}, {c->hcol.B0, B0}, {c->hcol.B1, B1}, {c->hcol.B2, B2}};
    for (const auto &cp : copy)
        if (cp.src) HIPCHK(c, hipMemcpyAsync(cp.dst, cp.src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_absorb_cyclosynch(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, int *num_abs_ph, int *scatt_cyclosynch_num_ph,
                                           double *abs_weight)
{
    if (!c || !cs || cs->b_field_calc < 0 || cs->b_field_calc > 2) return MCRAT_HIP_EINVAL;
    if (!c->have_hydro || !c->have_photons || !c->hcol_buf) return MCRAT_HIP_ESTATE;
    int rc = flush_pending(c);
    if (rc) return rc;
    const int nblk = cs_absorb_blocks(c->ph.n);
    if ((rc = ensure_aos(c, sizeof(CsAbsPartial) * (size_t)nblk))) return rc;
    CsParams p{c->kc.dimensions, cs->b_field_calc, cs->epsilon_b};
    HIPCHK(c, launch_cs_absorb(p, c->ph, c->hy.temp, c->hcol, static_cast<CsAbsPartial *>(c->aos_buf), c->stream));
    std::vector<CsAbsPartial> part((size_t)nblk);
    HIPCHK(c, hipMemcpyAsync(part.data(), c->aos_buf, sizeof(CsAbsPartial) * (size_t)nblk, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double w = 0;
    long long a = 0, s = 0;
    for (const auto &q : part) { w += q.abs_weight; a += q.abs_count; s += q.scatt_count; }
    if (num_abs_ph) *num_abs_ph = (int)a;
    if (scatt_cyclosynch_num_ph) *scatt_cyclosynch_num_ph = (int)s;
    if (abs_weight) *abs_weight = w;
    drop_graph(c);
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_get_hydro(mcrat_hip_ctx *c, mcrat_hip_hydro_columns *out)
{
    if (!c || !out) return MCRAT_HIP_EINVAL;
    if (!c->have_hydro || !c->hcol_buf) return MCRAT_HIP_ESTATE;
    const int M = c->hcol_M;
    if (out->num_elements < M) return MCRAT_HIP_EINVAL;
    const struct { double *dst; const double *src; } copy[16] = {
        {out->r0, c->hcol.r0}, {out->r1, c->hcol.r1}, {out->r2, c->hcol.r2}, {out->r0_size, c->hcol.s0}, {out->r1_size, c->hcol.s1},
        {out->r2_size, c->hcol.s2}, {out->v0, c->hcol.v0}, {out->v1, c->hcol.v1}, {out->v2, c->hcol.v2}, {out->dens, c->hcol.dens},
        {out->dens_lab, c->hcol.dens_lab}, {out->pres, c->hcol.pres}, {out->temp, c->hcol.temp}, {out->gamma, c->hcol.gamma},
        {out->r, c->hcol.r}, {out->theta, c->hcol.theta}};
    for (const auto &cp : copy)
        if (cp.dst) HIPCHK(c, hipMemcpyAsync(cp.dst, cp.src, sizeof(double) * (size_t)M, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out->num_elements = M;
    return MCRAT_HIP_OK;
}

// ---------------------------------------------------------------------------------------------- photons
// a view's list: a window of `n` slots into the pool's columns (the window spans the pool's slots per rank; slots beyond n stay invalid)
static int alloc_view_photons(mcrat_hip_ctx *c, int n, bool clear_window = true)
{
    mcrat_hip_ctx *P = c->parent;
    if (n > P->rank_stride) {
        c->last_error = "the list is longer than the pool's slots per rank (mcrat_hip_pool_create)";
        return MCRAT_HIP_ENOMEM;
    }
    const size_t o = (size_t)c->view_rank * (size_t)P->rank_stride;
    PhotonDev p = P->ph;
    double **cols[24] = {&p.r0, &p.r1, &p.r2, &p.p0, &p.p1, &p.p2, &p.p3, &p.c0, &p.c1, &p.c2, &p.c3, &p.s0, &p.s1, &p.s2, &p.s3,
                         &p.num_scatt, &p.weight, &p.tau, &p.tts, &p.u0, &p.u1, &p.u2, &p.ntau, &p.tau_next};
    for (int k = 0; k < 24; ++k) *cols[k] += o;
    p.idx += o; p.flags += o; p.type += o;
    p.n = n;
    p.n_pad = P->rank_stride;                 // (col_stride stays the pool's: the columns of a view are windows into the pool's)
    c->ph = p;
    if (clear_window) HIPCHK(c, launch_clear_slots(c->ph, 0, P->rank_stride, c->stream));
    c->step_blocks = step_grid_blocks(p.n_pad);
    c->partials = P->partials;                // list-mode scratch is the pool's: one stream, one list at a time
    c->partials_cap = P->partials_cap;
    c->shortlist = P->shortlist;
    if (clear_window) HIPCHK(c, hipMemsetAsync(c->shortlist, 0, sizeof(Shortlist), c->stream));
    c->n_ranks = 0;
    drop_graph(c);
    return MCRAT_HIP_OK;
}

static int alloc_photons(mcrat_hip_ctx *c, int n)
{
    if (c->parent) return alloc_view_photons(c, n);
    if (c->is_pool) {
        c->last_error = "this context is a rank pool: its photons are set through the views (mcrat_hip_pool_rank)";
        return MCRAT_HIP_ESTATE;
    }
    const int n_pad = (int)align_up((size_t)std::max(n, 1), 2 * STEP_BLOCK);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    size_t o_d[25];                                                           // the 24 columns of PhotonDev + rank_loop_kernel's scratch column
    for (int k = 0; k < 25; ++k) o_d[k] = take(sizeof(double) * n_pad);     // (photon_cols.hpp, ListCols::draw_log)
    const size_t o_idx = take(sizeof(int) * n_pad);
    const size_t o_flags = take(n_pad);
    const size_t o_type = take(n_pad);
    const size_t total = off;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->ph_buf && c->ph_bytes < total) { HIPCHK(c, hipFree(c->ph_buf)); c->ph_buf = nullptr; c->ph_bytes = 0; }
    if (!c->ph_buf) { HIPCHK(c, hipMalloc(&c->ph_buf, total)); c->ph_bytes = total; }
    HIPCHK(c, hipMemsetAsync(c->ph_buf, 0, total, c->stream));
    char *b = static_cast<char *>(c->ph_buf);
    PhotonDev &p = c->ph;
    double **cols[19] = {&p.r0, &p.r1, &p.r2, &p.p0, &p.p1, &p.p2, &p.p3, &p.c0, &p.c1, &p.c2, &p.c3,
                         &p.s0, &p.s1, &p.s2, &p.s3, &p.num_scatt, &p.weight, &p.tau, &p.tts};
    for (int k = 0; k < 19; ++k) *cols[k] = reinterpret_cast<double *>(b + o_d[k]);
    p.u0 = reinterpret_cast<double *>(b + o_d[19]);
    p.u1 = reinterpret_cast<double *>(b + o_d[20]);
    p.u2 = reinterpret_cast<double *>(b + o_d[21]);
    p.ntau = reinterpret_cast<double *>(b + o_d[22]);
    p.tau_next = reinterpret_cast<double *>(b + o_d[23]);
    p.idx = reinterpret_cast<int *>(b + o_idx);
    p.flags = reinterpret_cast<unsigned char *>(b + o_flags);
    p.type = b + o_type;
    p.n = n;
    p.n_pad = n_pad;
    if ((o_d[1] - o_d[0]) / sizeof(double) * 25 > 0xffffffffull) { c->last_error = "photon list: more than 2^32 / 25 slots"; return MCRAT_HIP_EINVAL; }
    p.col_stride = (unsigned)((o_d[1] - o_d[0]) / sizeof(double));
    for (int k = 1; k < 25; ++k)
        if (o_d[k] - o_d[k - 1] != o_d[1] - o_d[0]) { c->last_error = "photon columns are not equally spaced"; return MCRAT_HIP_EHIP; }
    c->step_blocks = step_grid_blocks(n_pad);
    const int need = c->step_blocks;                          // one candidate per workgroup of the step kernel
    if (c->partials_cap < need) {
        if (c->partials) HIPCHK(c, hipFree(c->partials));
        c->partials = nullptr;
        HIPCHK(c, hipMalloc((void **)&c->partials, sizeof(Cand) * need));
        c->partials_cap = need;
    }
    c->n_ranks = 0;
    if (c->cfg.virtual_rank_photons > 0) {
        c->rank_stride = c->cfg.virtual_rank_photons;
        c->n_ranks = (n + c->cfg.virtual_rank_photons - 1) / c->cfg.virtual_rank_photons;
        if (c->rstates_cap < c->n_ranks) {
            if (c->d_rstates) HIPCHK(c, hipFree(c->d_rstates));
            if (c->h_rstates) HIPCHK(c, hipHostFree(c->h_rstates));
            c->d_rstates = nullptr; c->h_rstates = nullptr;
            HIPCHK(c, hipMalloc((void **)&c->d_rstates, sizeof(LoopState) * c->n_ranks));
            HIPCHK(c, hipHostMalloc((void **)&c->h_rstates, sizeof(LoopState) * c->n_ranks, hipHostMallocDefault));
            c->rstates_cap = c->n_ranks;
        }
    }
    if (!c->shortlist) HIPCHK(c, hipMalloc((void **)&c->shortlist, sizeof(Shortlist)));
    HIPCHK(c, hipMemsetAsync(c->shortlist, 0, sizeof(Shortlist), c->stream));
    drop_graph(c);
    return MCRAT_HIP_OK;
}

static inline unsigned char make_flags(char type, double weight, int recalc)
{
    unsigned f = FLAG_VALID;
    if (type != 'p' && weight != 0) f |= FLAG_MOVES;      // mclib.c:1070
    if (recalc == 1) f |= FLAG_RECALC;
    return (unsigned char)f;
}

static int upload_columns(mcrat_hip_ctx *c, int n, const std::vector<const double *> &src, const int *idx,
                          const unsigned char *flags, const char *type)
{
    PhotonDev &p = c->ph;
    double *cols[19] = {p.r0, p.r1, p.r2, p.p0, p.p1, p.p2, p.p3, p.c0, p.c1, p.c2, p.c3,
                        p.s0, p.s1, p.s2, p.s3, p.num_scatt, p.weight, p.tau, p.tts};
    for (int k = 0; k < 19; ++k)
        if (src[k]) HIPCHK(c, hipMemcpyAsync(cols[k], src[k], sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    // derived columns (device_types.hpp): same operations, in the same order, as mclib.c:1074-1080 and :680
    std::vector<double> der((size_t)4 * n, 0.0);
    for (int i = 0; i < n; ++i) {
        const double p0 = src[3][i];
        if (p0 != 0) {
            const double d = 1.0 / p0;
            der[i] = src[4][i] * d * C_LIGHT;
            der[(size_t)n + i] = src[5][i] * d * C_LIGHT;
            der[(size_t)2 * n + i] = src[6][i] * d * C_LIGHT;
        }
        const double tau = src[17] ? src[17][i] : 0.0;
        der[(size_t)3 * n + i] = -1.0 / tau;
    }
    HIPCHK(c, hipMemcpyAsync(p.u0, der.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p.u1, der.data() + n, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p.u2, der.data() + (size_t)2 * n, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p.ntau, der.data() + (size_t)3 * n, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p.idx, idx, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p.flags, flags, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(p.type, type, n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_photons = true;
    c->frame_open = false;
    return MCRAT_HIP_OK;
}

static int ensure_aos(mcrat_hip_ctx *c, size_t bytes)
{
    if (c->parent) {                               // a view borrows its pool's staging buffer (ensure_counts)
        int rc = ensure_aos(c->parent, bytes);
        c->aos_buf = c->parent->aos_buf; c->aos_bytes = c->parent->aos_bytes;
        return rc;
    }
    if (c->aos_buf && c->aos_bytes < bytes) { HIPCHK(c, hipFree(c->aos_buf)); c->aos_buf = nullptr; c->aos_bytes = 0; }
    if (!c->aos_buf) { HIPCHK(c, hipMalloc(&c->aos_buf, bytes)); c->aos_bytes = bytes; }
    return MCRAT_HIP_OK;
}

// Page-lock the caller's memory (its struct photon array, its hydro columns) so that the copies of set_photons / get_photons /
// set_hydro run as direct DMA at the link's rate instead of through the runtime's staging of pageable memory.
extern "C" int mcrat_hip_register_host(mcrat_hip_ctx *c, void *ptr, size_t bytes)
{
    if (!c || !ptr || bytes == 0) return MCRAT_HIP_EINVAL;
    HIPCHK(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_unregister_host(mcrat_hip_ctx *c, void *ptr)
{
    if (!c || !ptr) return MCRAT_HIP_EINVAL;
    HIPCHK(c, hipHostUnregister(ptr));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_set_photons(mcrat_hip_ctx *c, const mcrat_hip_photon_list *l)
{
    if (!c || !l || !l->photons || l->list_capacity <= 0) return MCRAT_HIP_EINVAL;
    const int n = l->list_capacity;
    int rc = alloc_photons(c, n);
    if (rc) return rc;
    // the records cross PCIe as they lie in the caller's memory; staging.hip transposes them on the device
    const size_t bytes = sizeof(mcrat_hip_photon) * (size_t)n;
    if ((rc = ensure_aos(c, bytes))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->aos_buf, l->photons, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_aos_to_soa(c->aos_buf, c->ph, n, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_photons = true;
    c->frame_open = false;
    return MCRAT_HIP_OK;
}

// mcrat_hip_set_photons for many lists of a rank pool at once (a CONTINUE run's restart: every adopted rank's checkpoint, mcrat.c:235-330 /
// readCheckpoint): the lists' records are gathered into one staging buffer, cross PCIe in ONE copy and are transposed into their windows by ONE
// launch, instead of a copy, two launches and a wait per list.  rank_of[j] is the pool rank list j goes to (its view must exist:
// mcrat_hip_pool_rank); every list exactly as mcrat_hip_set_photons(view, list) would leave it.
extern "C" int mcrat_hip_pool_set_photons(mcrat_hip_ctx *c, int count, const int *rank_of, const mcrat_hip_photon_list *lists)
{
    if (!c || count < 0 || (count > 0 && (!rank_of || !lists))) return MCRAT_HIP_EINVAL;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    if (count == 0) return MCRAT_HIP_OK;
    struct Desc { int rank, n; long long first; };
    static_assert(sizeof(Desc) == 16, "PoolSetDesc of staging.hip");
    std::vector<Desc> desc((size_t)count);
    std::vector<char> seen((size_t)c->n_ranks, 0);
    size_t total = 0;
    for (int j = 0; j < count; ++j) {                                       // everything is checked before any window is touched
        const int r = rank_of[j];
        if (r < 0 || r >= c->n_ranks || !lists[j].photons || lists[j].list_capacity <= 0 || seen[(size_t)r]) return MCRAT_HIP_EINVAL;
        seen[(size_t)r] = 1;
        if (!c->views[r]) { c->last_error = "pool_set_photons: a list without a view (mcrat_hip_pool_rank)"; return MCRAT_HIP_ESTATE; }
        if (lists[j].list_capacity > c->rank_stride) { c->last_error = "a list is longer than the pool's slots per rank (mcrat_hip_pool_create)"; return MCRAT_HIP_ENOMEM; }
        desc[(size_t)j] = Desc{r, lists[j].list_capacity, (long long)total};
        total += (size_t)lists[j].list_capacity;
    }
    const size_t rec_bytes = sizeof(mcrat_hip_photon) * total, desc_bytes = sizeof(Desc) * (size_t)count;
    int rc = ensure_aos(c, align_up(rec_bytes, 256) + desc_bytes);
    if (rc) return rc;
    char *dev = static_cast<char *>(c->aos_buf);
    {   // the lists' records, gathered through two pinned pieces: the copy into one overlaps the transfer of the other (a pageable source
        // of this size moves at 3 GB/s; the caller's lists are wherever readCheckpoint put them)
        constexpr size_t PIECE = 16u << 20;
        for (int k = 0; k < 2; ++k) {
            if (!c->pin_ring[k]) HIPCHK(c, hipHostMalloc(&c->pin_ring[k], PIECE, hipHostMallocDefault));
            if (!c->pin_ev[k]) HIPCHK(c, hipEventCreateWithFlags(&c->pin_ev[k], hipEventDisableTiming));
        }
        size_t done = 0;            // bytes of the concatenated records already on their way
        int j = 0, k = 0;
        size_t in_list = 0;         // bytes of list j already taken
        bool used[2] = {false, false};
        while (done < rec_bytes) {
            if (used[k]) HIPCHK(c, hipEventSynchronize(c->pin_ev[k]));
            char *dst = static_cast<char *>(c->pin_ring[k]);
            size_t fill = 0;
            while (fill < PIECE && j < count) {
                const size_t list_bytes = sizeof(mcrat_hip_photon) * (size_t)lists[j].list_capacity;
                const size_t take = std::min(PIECE - fill, list_bytes - in_list);
                memcpy(dst + fill, reinterpret_cast<const char *>(lists[j].photons) + in_list, take);
                fill += take; in_list += take;
                if (in_list == list_bytes) { ++j; in_list = 0; }
            }
            HIPCHK(c, hipMemcpyAsync(dev + done, dst, fill, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipEventRecord(c->pin_ev[k], c->stream));
            used[k] = true;
            done += fill;
            k ^= 1;
        }
    }
    HIPCHK(c, hipMemcpyAsync(dev + align_up(rec_bytes, 256), desc.data(), desc_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_pool_aos_to_soa(dev, c->ph, c->rank_stride, dev + align_up(rec_bytes, 256), count, c->stream));
    for (int j = 0; j < count; ++j) {
        mcrat_hip_ctx *v = c->views[desc[(size_t)j].rank];
        if ((rc = alloc_view_photons(v, desc[(size_t)j].n, false))) { c->last_error = v->last_error; return rc; }    // the view's columns in place; the launch above filled and cleared the window
        v->have_photons = true;
        v->frame_open = false;
    }
    HIPCHK(c, hipMemsetAsync(c->shortlist, 0, sizeof(Shortlist), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_inject_photons(mcrat_hip_ctx *c, double r_inj, double ph_weight, int min_photons, int max_photons, char spect,
                                        double theta_min, double theta_max, double fps, uint64_t seed, int *num_photons,
                                        double *ph_weight_adjusted)
{
    if (!c || !(ph_weight > 0) || !(fps > 0) || min_photons < 0 || max_photons < min_photons || (spect != 'b' && spect != 'w')) return MCRAT_HIP_EINVAL;
    if (!c->have_hydro) return MCRAT_HIP_ESTATE;
    const int M = c->hy.M;
    InjectParams p;
    p.dimensions = c->kc.dimensions; p.geometry = c->kc.geometry;
    p.rmin = r_inj - 0.5 * C_LIGHT / fps;                      // mclib.c:34-35
    p.rmax = r_inj + 0.5 * C_LIGHT / fps;
    p.theta_min = theta_min; p.theta_max = theta_max;
    p.num_dens_coeff = (spect == 'w') ? (double)8.44f : (double)20.29f;   // a float in the reference, mclib.c:17,23-32
    p.wien = spect == 'w';
    RngKey key = c->key;
    key.seed = seed;
    { int rc_ = ensure_counts(c, (size_t)M); if (rc_) return rc_; }
    // mclib.c:87-136: draw the per-cell counts; too many photons -> weight x 10, too few -> weight x 0.5, draw again
    double weight = ph_weight;
    unsigned long long total = 0;
    bool ok = false;
    for (unsigned long long attempt = 0; attempt <= 200; ++attempt) {
        HIPCHK(c, launch_inject_count(p, c->hy, weight, attempt, key, c->grid_count, c->d_grid_total, c->stream));
        HIPCHK(c, hipMemcpyAsync(&total, c->d_grid_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (total > (unsigned long long)max_photons) weight *= 10;
        else if (total < (unsigned long long)min_photons) weight *= 0.5;
        else { ok = true; break; }
    }
    if (!ok) { c->last_error = "photon injection: no weight puts the photon count between min_photons and max_photons"; return MCRAT_HIP_EINVAL; }
    if (total == 0) { c->last_error = "photon injection: no photons (no cell of the frame touches the injection slab?)"; return MCRAT_HIP_EINVAL; }
    const int n = (int)total;
    int rc = alloc_photons(c, n);
    if (rc) return rc;
    // cell -> first photon: exclusive scan of the counts (scratch: the record staging buffer)
    const size_t scan_bytes = sizeof(int) * ((size_t)M + 1 + grid_scan_scratch_ints(M));
    if ((rc = ensure_aos(c, scan_bytes))) return rc;
    int *start = static_cast<int *>(c->aos_buf), *scratch = start + M + 1;
    HIPCHK(c, launch_exclusive_scan(c->grid_count, M, start, scratch, (long long)total, c->stream));
    HIPCHK(c, launch_inject_generate(p, c->hy, weight, key, start, c->ph, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_photons = true;
    c->frame_open = false;
    if (num_photons) *num_photons = n;
    if (ph_weight_adjusted) *ph_weight_adjusted = weight;
    return MCRAT_HIP_OK;
}

// photonInjection (mclib.c:9-300) for the lists of a rank pool that ask for it, a handful of launches for all of them: the injection slab
// of each group of lists that share one (same radius, angles and spectrum: the ranks of an angle bin) is found once, then one workgroup per
// list runs the list's weight loop and writes its photons into the list's window (inject.hip).  Every list gets exactly the photons
// mcrat_hip_inject_photons gives its view (same keys); a list with more photons than the kernel's table holds takes that path.
extern "C" int mcrat_hip_pool_inject_photons(mcrat_hip_ctx *c, double fps, mcrat_hip_pool_inject_list *lists)
{
    if (!c || !lists || !(fps > 0)) return MCRAT_HIP_EINVAL;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    if (!c->have_hydro) return MCRAT_HIP_ESTATE;
    const int R = c->n_ranks, M = c->hy.M;
    struct Slab { double rmin, rmax, tmin, tmax; int wien; };
    std::vector<Slab> slabs;
    std::vector<PoolInject> pi((size_t)R);
    bool any = false;
    int rc;
    // every list's arguments are looked at before any window is touched: a bad request for list 7 must not leave lists 0-6 emptied
    for (int r = 0; r < R; ++r) {
        mcrat_hip_pool_inject_list &q = lists[r];
        q.status = MCRAT_HIP_OK;
        if (!q.inject) continue;
        if (!(q.ph_weight > 0) || q.min_photons < 0 || q.max_photons < q.min_photons || (q.spect != 'b' && q.spect != 'w')) {
            q.status = MCRAT_HIP_EINVAL;
            c->last_error = "pool_inject_photons: a list's weight, photon-count range or spectrum is not valid";
            return MCRAT_HIP_EINVAL;
        }
        if (!c->views[r]) { q.status = MCRAT_HIP_ESTATE; c->last_error = "pool_inject_photons: a list without a view (mcrat_hip_pool_rank)"; return MCRAT_HIP_ESTATE; }
    }
    for (int r = 0; r < R; ++r) {
        pi[(size_t)r] = PoolInject{};
        mcrat_hip_pool_inject_list &q = lists[r];
        if (!q.inject) continue;
        mcrat_hip_ctx *v = c->views[r];
        if ((rc = alloc_view_photons(v, 0))) { c->last_error = v->last_error; return rc; }      // the window cleared, the view's columns in place
        v->have_photons = false;
        Slab sl{q.r_inj - 0.5 * C_LIGHT / fps, q.r_inj + 0.5 * C_LIGHT / fps, q.theta_min, q.theta_max, q.spect == 'w'};   // mclib.c:34-35
        int g = -1;
        for (size_t k = 0; k < slabs.size(); ++k)
            if (slabs[k].rmin == sl.rmin && slabs[k].rmax == sl.rmax && slabs[k].tmin == sl.tmin && slabs[k].tmax == sl.tmax && slabs[k].wien == sl.wien) { g = (int)k; break; }
        if (g < 0) { g = (int)slabs.size(); slabs.push_back(sl); }
        PoolInject &e = pi[(size_t)r];
        e.inject = 1; e.group = g; e.seed = q.seed; e.stream = v->key.stream; e.weight_in = q.ph_weight; e.min_photons = q.min_photons; e.max_photons = q.max_photons;
        any = true;
    }
    if (!any) return MCRAT_HIP_OK;
    if ((rc = ensure_counts(c, (size_t)M + 8))) return rc;
    const size_t scan_ints = (size_t)M + 1 + grid_scan_scratch_ints(M);
    int *d_start = nullptr;
    PoolInject *d_pi = nullptr;
    InjectSlabCell *d_slab = nullptr;
    size_t slab_cap = 0;
    auto done = [&](int code) { (void)hipFree(d_start); (void)hipFree(d_pi); (void)hipFree(d_slab); return code; };
    if (hipMalloc((void **)&d_start, sizeof(int) * scan_ints) != hipSuccess || hipMalloc((void **)&d_pi, sizeof(PoolInject) * (size_t)R) != hipSuccess)
        return done(MCRAT_HIP_ENOMEM);
    if (hipMemcpyAsync(d_pi, pi.data(), sizeof(PoolInject) * (size_t)R, hipMemcpyHostToDevice, c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
    for (size_t g = 0; g < slabs.size(); ++g) {
        InjectParams p;
        p.dimensions = c->kc.dimensions; p.geometry = c->kc.geometry;
        p.rmin = slabs[g].rmin; p.rmax = slabs[g].rmax; p.theta_min = slabs[g].tmin; p.theta_max = slabs[g].tmax;
        p.num_dens_coeff = slabs[g].wien ? (double)8.44f : (double)20.29f;   // a float in the reference, mclib.c:17,23-32
        p.wien = slabs[g].wien;
        int n_slab = 0;
        if (launch_inject_slab_flag(p, c->hy, c->grid_count, c->d_grid_total, &n_slab, c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
        if ((size_t)n_slab > slab_cap) {
            (void)hipFree(d_slab); d_slab = nullptr;
            if (hipMalloc((void **)&d_slab, sizeof(InjectSlabCell) * (size_t)n_slab) != hipSuccess) return done(MCRAT_HIP_ENOMEM);
            slab_cap = (size_t)n_slab;
        }
        if (n_slab > 0 && launch_inject_slab_write(p, c->hy, c->grid_count, n_slab, d_start, d_start + M + 1, d_slab, c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
        if (launch_inject_pool(p, c->hy, c->ph, c->rank_stride, R, d_slab, n_slab, d_pi, (int)g, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { c->last_error = "photon injection of the pool's lists failed"; return done(MCRAT_HIP_EHIP); }
    }
    if (hipMemcpyAsync(pi.data(), d_pi, sizeof(PoolInject) * (size_t)R, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
    (void)done(0);
    // every list that was injected is committed; a list that failed says so in its own status (its window stays empty) and the call returns
    // the first such code after all lists have been looked at
    int first_error = MCRAT_HIP_OK;
    for (int r = 0; r < R; ++r) {
        const PoolInject &e = pi[(size_t)r];
        if (!e.inject) continue;
        mcrat_hip_ctx *v = c->views[r];
        mcrat_hip_pool_inject_list &q = lists[r];
        if (e.error == 3) {                                                 // more photons than the kernel's table: the one-list path (it says what is wrong, if anything)
            if ((rc = mcrat_hip_inject_photons(v, q.r_inj, q.ph_weight, q.min_photons, q.max_photons, q.spect, q.theta_min, q.theta_max, fps, q.seed,
                                               &q.num_photons, &q.ph_weight_adjusted))) {
                q.status = rc;
                if (!first_error) { first_error = rc; c->last_error = v->last_error; }
            }
            continue;
        }
        if (e.error == 1 || e.error == 2) {
            q.status = MCRAT_HIP_EINVAL; q.num_photons = 0;
            if (!first_error) {
                first_error = MCRAT_HIP_EINVAL;
                c->last_error = e.error == 1 ? "photon injection: no weight puts the photon count between min_photons and max_photons"
                                             : "photon injection: no photons (no cell of the frame touches the injection slab?)";
            }
            continue;
        }
        v->ph.n = e.n;
        v->have_photons = true;
        v->frame_open = false;
        q.num_photons = e.n;
        q.ph_weight_adjusted = e.weight_out;
    }
    return first_error;
}

// reallocatePhotonListMemory (photons.c:37-80) on the device: a larger set of columns, the old slots copied, the new ones null
static int grow_photons(mcrat_hip_ctx *c, int new_n)
{
    const PhotonDev old = c->ph;
    void *old_buf = c->ph_buf;
    const int old_n = old.n;
    if (new_n <= old_n) return MCRAT_HIP_EINVAL;
    if (c->parent) {                                           // a view grows inside its window of the pool
        if (new_n > c->parent->rank_stride) {
            c->last_error = "the list cannot double: it would outgrow the pool's slots per rank (mcrat_hip_pool_create)";
            return MCRAT_HIP_ENOMEM;
        }
        HIPCHK(c, launch_null_fill(c->ph, old_n, new_n - old_n, c->stream));
        c->ph.n = new_n;
        return MCRAT_HIP_OK;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->ph_buf = nullptr; c->ph_bytes = 0;                     // a fresh, zeroed allocation
    int rc = alloc_photons(c, new_n);
    if (rc) { if (c->ph_buf) (void)hipFree(c->ph_buf); c->ph_buf = old_buf; c->ph = old; return rc; }
    const double *src[24] = {old.r0, old.r1, old.r2, old.p0, old.p1, old.p2, old.p3, old.c0, old.c1, old.c2, old.c3, old.s0, old.s1, old.s2, old.s3,
                             old.num_scatt, old.weight, old.tau, old.tts, old.u0, old.u1, old.u2, old.ntau, old.tau_next};
    double *dst[24] = {c->ph.r0, c->ph.r1, c->ph.r2, c->ph.p0, c->ph.p1, c->ph.p2, c->ph.p3, c->ph.c0, c->ph.c1, c->ph.c2, c->ph.c3, c->ph.s0, c->ph.s1,
                       c->ph.s2, c->ph.s3, c->ph.num_scatt, c->ph.weight, c->ph.tau, c->ph.tts, c->ph.u0, c->ph.u1, c->ph.u2, c->ph.ntau, c->ph.tau_next};
    for (int k = 0; k < 24; ++k) HIPCHK(c, hipMemcpyAsync(dst[k], src[k], sizeof(double) * (size_t)old_n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ph.idx, old.idx, sizeof(int) * (size_t)old_n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ph.flags, old.flags, (size_t)old_n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->ph.type, old.type, (size_t)old_n, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, launch_null_fill(c->ph, old_n, new_n - old_n, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(old_buf));
    if (c->ph_snap) { HIPCHK(c, hipFree(c->ph_snap)); c->ph_snap = nullptr; c->ph_snap_bytes = 0; }     // a snapshot of the smaller list
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_emit_cyclosynch_pool(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, double r_inj, double ph_weight, int maximum_photons,
                                              double theta_min, double theta_max, double fps, uint64_t seed, int *num_emitted,
                                              double *ph_weight_adjusted, int *integrals_not_converged)
{
    if (!c || !cs || cs->b_field_calc < 0 || cs->b_field_calc > 2 || !(ph_weight > 0) || !(fps > 0) || maximum_photons < 0) return MCRAT_HIP_EINVAL;
    if (!c->have_hydro || !c->hcol_buf || !c->have_photons) return MCRAT_HIP_ESTATE;
    int rc = flush_pending(c);
    if (rc) return rc;
    const int M = c->hy.M;
    CsEmitParams p{};
    p.dimensions = c->kc.dimensions; p.geometry = c->kc.geometry; p.b_field_calc = cs->b_field_calc; p.epsilon_b = cs->epsilon_b;
    p.rmin = r_inj + (C_LIGHT * (cs->scatt_frame_number - cs->inj_frame_number) / fps - 0.5 * C_LIGHT / fps);      // calcCyclosynchRLimits :225-244
    p.rmax = r_inj + (C_LIGHT * (cs->scatt_frame_number - cs->inj_frame_number) / fps + 0.5 * C_LIGHT / fps);
    p.theta_min = theta_min; p.theta_max = theta_max;
    RngKey key = c->key;
    key.seed = seed;
    const size_t need_counts = (size_t)std::max(M, (c->ph.n + 255) / 256 + 8);
    { int rc_ = ensure_counts(c, need_counts); if (rc_) return rc_; }
    unsigned *d_flags = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_flags, 4 * sizeof(unsigned)));
    auto fail = [&](int code) { (void)hipFree(d_flags); return code; };
    // :1244-1296: the weight loop on the device's totals
    const double max_photons = cs->rebin_e_perc * maximum_photons;
    double weight = ph_weight;
    unsigned long long total = 0;
    unsigned flags[4] = {0, 0, 0, 0}, cells_in_shell = 0;
    bool ok = false;
    for (unsigned long long attempt = 0; attempt <= 400 && !ok; ++attempt) {
        if (hipMemsetAsync(d_flags, 0, 4 * sizeof(unsigned), c->stream) != hipSuccess) return fail(MCRAT_HIP_EHIP);
        if (launch_cs_emit_count(p, c->hy, c->hcol, weight, attempt, key, c->grid_count, c->d_grid_total, d_flags, c->stream) != hipSuccess ||
            hipMemcpyAsync(&total, c->d_grid_total, sizeof total, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { c->last_error = "cyclo-synchrotron emission: count pass failed"; return fail(MCRAT_HIP_EHIP); }
        if (attempt == 0) cells_in_shell = flags[1];                                // (the kernel counts them on its first pass only)
        if (flags[0] != 0) {  // gsl_integration_qags (:1276) past its first rule AND past the device's bisection (inject.hip, qags_planck): not a result to go on with
            if (integrals_not_converged) *integrals_not_converged = (int)flags[0];
            c->last_error = "cyclo-synchrotron emission: the photon-density integral of " + std::to_string(flags[0]) + " cell(s) did not converge within the device's interval limit";
            return fail(MCRAT_HIP_EREFUSED);
        }
        const int min_photons = cells_in_shell ? 1 : 0;                             // no cell in the shell: nothing to emit (:1236-1239)
        if ((double)total > max_photons) weight *= 10;
        else if ((long long)total < min_photons) weight *= 0.5;
        else ok = true;
    }
    if (!ok) { c->last_error = "cyclo-synchrotron emission: no weight gives between 1 and rebin_e_perc * maximum_photons photons"; return fail(MCRAT_HIP_EINVAL); }
    const int n_emit = (int)total;
    if (num_emitted) *num_emitted = n_emit;
    if (ph_weight_adjusted) *ph_weight_adjusted = weight;
    if (integrals_not_converged) *integrals_not_converged = (int)flags[0];
    if (n_emit == 0) return fail(MCRAT_HIP_OK);
    // cell -> first pool photon (the counts live in grid_count[0..M)); keep them while the list may be re-allocated
    const size_t scan_ints = (size_t)M + 1 + grid_scan_scratch_ints(M);
    int *d_start = nullptr;
    if (hipMalloc((void **)&d_start, sizeof(int) * scan_ints) != hipSuccess) return fail(MCRAT_HIP_ENOMEM);
    auto fail2 = [&](int code) { (void)hipFree(d_start); return fail(code); };
    if (launch_exclusive_scan(c->grid_count, M, d_start, d_start + M + 1, (long long)total, c->stream) != hipSuccess) return fail2(MCRAT_HIP_EHIP);
    // addToPhotonList (photons.c:108-208): the null slots, the list doubled first when it has none
    unsigned long long n_null = 0;
    auto count_nulls = [&]() -> int {
        if (launch_null_count(c->ph, c->grid_count, c->d_grid_total, c->stream) != hipSuccess ||
            hipMemcpyAsync(&n_null, c->d_grid_total, sizeof n_null, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) return MCRAT_HIP_EHIP;
        return MCRAT_HIP_OK;
    };
    if ((rc = count_nulls())) return fail2(rc);
    if (n_null == 0) {                                                              // num_photons >= list_capacity && num_null_photons <= num (:112)
        const long long cap = c->ph.n;
        const long long new_cap = (cap * 2 > cap + n_emit) ? cap * 2 : cap * (n_emit / cap);
        if (new_cap > 0x7fffffffLL) return fail2(MCRAT_HIP_EINVAL);
        if ((rc = grow_photons(c, (int)new_cap))) return fail2(rc);
        if ((rc = ensure_counts(c, (size_t)((c->ph.n + 255) / 256 + 8)))) return fail2(rc);
        if ((rc = count_nulls())) return fail2(rc);
    }
    if ((unsigned long long)n_emit > n_null) {
        c->last_error = "cyclo-synchrotron emission: fewer null slots than photons to add (the reference exits with \"Adding to the photon list has failed\")";
        return fail2(MCRAT_HIP_EINVAL);
    }
    const long long nblk = (c->ph.n + 255) / 256;
    const size_t null_bytes = sizeof(int) * ((size_t)nblk + 1 + grid_scan_scratch_ints(nblk) + (size_t)n_null);
    if ((rc = ensure_aos(c, null_bytes))) return fail2(rc);
    int *blk_start = static_cast<int *>(c->aos_buf), *scratch = blk_start + nblk + 1, *null_slots = scratch + grid_scan_scratch_ints(nblk);
    if (launch_exclusive_scan(c->grid_count, nblk, blk_start, scratch, (long long)n_null, c->stream) != hipSuccess ||
        launch_null_write(c->ph, blk_start, null_slots, c->stream) != hipSuccess ||
        launch_cs_emit_generate(p, c->hy, c->hcol, weight, key, d_start, n_emit, null_slots, c->ph, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { c->last_error = "cyclo-synchrotron emission: generate pass failed"; return fail2(MCRAT_HIP_EHIP); }
    c->frame_open = false;
    drop_graph(c);
    return fail2(MCRAT_HIP_OK);
}

extern "C" int mcrat_hip_set_photons_soa(mcrat_hip_ctx *c, const mcrat_hip_photon_soa *s)
{
    if (!c || !s || s->n <= 0) return MCRAT_HIP_EINVAL;
    if (!s->type || !s->p0 || !s->p1 || !s->p2 || !s->p3 || !s->r0 || !s->r1 || !s->r2 || !s->num_scatt ||
        !s->recalc_properties || !s->weight || !s->nearest_block_index)
        return MCRAT_HIP_EINVAL;
    const int n = s->n;
    int rc = alloc_photons(c, n);
    if (rc) return rc;
    std::vector<unsigned char> flags(n);
    for (int i = 0; i < n; ++i) flags[i] = make_flags(s->type[i], s->weight[i], s->recalc_properties[i]);
    std::vector<const double *> src = {s->r0, s->r1, s->r2, s->p0, s->p1, s->p2, s->p3,
                                       s->comv_p0, s->comv_p1, s->comv_p2, s->comv_p3,
                                       s->s0, s->s1, s->s2, s->s3, s->num_scatt, s->weight,
                                       s->total_optical_depth, s->time_to_scatter};
    return upload_columns(c, n, src, s->nearest_block_index, flags.data(), s->type);
}

extern "C" int mcrat_hip_snapshot_photons(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    if (c->parent) { c->last_error = "snapshot the pool, not one of its views"; return MCRAT_HIP_ESTATE; }
    int rc = MCRAT_HIP_OK;
    if (c->frame_open && c->n_ranks == 0) { HIPCHK(c, launch_flush(c->ph, c->d_state, c->step_blocks, c->stream)); }
    (void)rc;
    if (c->ph_snap && c->ph_snap_bytes < c->ph_bytes) { HIPCHK(c, hipFree(c->ph_snap)); c->ph_snap = nullptr; }
    if (!c->ph_snap) { HIPCHK(c, hipMalloc(&c->ph_snap, c->ph_bytes)); c->ph_snap_bytes = c->ph_bytes; }
    HIPCHK(c, hipMemcpyAsync(c->ph_snap, c->ph_buf, c->ph_bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->snap_lens.clear();
    for (mcrat_hip_ctx *v : c->views) c->snap_lens.push_back((v && v->have_photons) ? v->ph.n : -1);    // lists change length (cyclo-synchrotron)
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_restore_photons(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->parent || !c->have_photons || !c->ph_snap || c->ph_snap_bytes < c->ph_bytes) return MCRAT_HIP_ESTATE;
    HIPCHK(c, hipMemcpyAsync(c->ph_buf, c->ph_snap, c->ph_bytes, hipMemcpyDeviceToDevice, c->stream));
    c->frame_open = false;        // the loop state no longer matches the photons: begin_frame comes next
    for (size_t r = 0; r < c->views.size(); ++r) {
        mcrat_hip_ctx *v = c->views[r];
        if (!v) continue;
        v->frame_open = false;
        if (r < c->snap_lens.size() && c->snap_lens[r] >= 0) v->ph.n = c->snap_lens[r];
    }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_rebin_cyclosynch(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, int max_photons, int *empty_bins_out,
                                          int *num_cyclosynch_ph_emit, int *scatt_cyclosynch_num_ph)
{
    if (!c || !cs || max_photons <= 0) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    int rc = flush_pending(c);
    if (rc) return rc;
    const int n = c->ph.n, three = c->kc.dimensions == DIM_THREE;
    // collect_photon_statistics :273-322
    const int rblk = rebin_range_blocks(n);
    if ((rc = ensure_aos(c, sizeof(RebinRange) * (size_t)rblk))) return rc;
    HIPCHK(c, launch_rebin_range(c->ph, three, static_cast<RebinRange *>(c->aos_buf), c->stream));
    std::vector<RebinRange> part((size_t)rblk);
    HIPCHK(c, hipMemcpyAsync(part.data(), c->aos_buf, sizeof(RebinRange) * (size_t)rblk, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    RebinRange q = part[0];
    for (int k = 1; k < rblk; ++k) {
        q.p0_min = std::fmin(q.p0_min, part[k].p0_min); q.p0_max = std::fmax(q.p0_max, part[k].p0_max);
        q.theta_min = std::fmin(q.theta_min, part[k].theta_min); q.theta_max = std::fmax(q.theta_max, part[k].theta_max);
        q.phi_min = std::fmin(q.phi_min, part[k].phi_min); q.phi_max = std::fmax(q.phi_max, part[k].phi_max);
        q.valid += part[k].valid; q.synch += part[k].synch;
    }
    if (q.valid == 0) { c->last_error = "rebinning: no valid photons found for rebinning"; return MCRAT_HIP_EREFUSED; }
    const double log_p0_min = (q.p0_min > 0 && q.p0_max > 0) ? std::log10(q.p0_min) : 0.0, log_p0_max = (q.p0_min > 0 && q.p0_max > 0) ? std::log10(q.p0_max) : 1.0;
    // calculate_binning_params :324-347, allocate_histograms :351-391
    RebinAxes ax{};
    ax.three = three;
    ax.num_bins = (int)(cs->rebin_e_perc * max_photons);
    ax.num_bins_theta = (int)std::ceil((q.theta_max - q.theta_min) / (cs->rebin_ang * (M_PI / 180.0)));
    ax.num_bins_phi = three ? (int)std::ceil((q.phi_max - q.phi_min) / cs->rebin_ang_phi) : 1;
    const long long total_ll = (long long)ax.num_bins_theta * ax.num_bins * (three ? ax.num_bins_phi : 1);
    if (total_ll > max_photons) { c->last_error = "rebinning would create more photons than max_photons"; return MCRAT_HIP_EREFUSED; }
    if (ax.num_bins <= 0 || ax.num_bins_theta <= 0 || ax.num_bins_phi <= 0) { c->last_error = "rebinning: invalid histogram dimensions"; return MCRAT_HIP_EREFUSED; }
    ax.total_bins = (int)total_ll;
    ax.e_lo = log_p0_min; ax.e_hi = log_p0_max + (log_p0_max - log_p0_min) * 1e-6;
    ax.t_lo = q.theta_min; ax.t_hi = q.theta_max + (q.theta_max - q.theta_min) * 1e-6;
    ax.p_lo = q.phi_min; ax.p_hi = q.phi_max + (q.phi_max - q.phi_min) * 1e-6;
    // scratch: bin of every slot, per-bin counts / cursors / starts, the member lists, the records, the null-slot list
    const int B = ax.total_bins;
    const long long nblk = (n + 255) / 256;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_bin = take(sizeof(int) * (size_t)n), o_cnt = take(sizeof(unsigned) * (size_t)(B + 2)), o_cur = take(sizeof(unsigned) * (size_t)B);
    const size_t o_start = take(sizeof(int) * ((size_t)B + 1 + grid_scan_scratch_ints(B))), o_mem = take(sizeof(int) * (size_t)n);
    const size_t o_rec = take(sizeof(RebinRec) * (size_t)B);
    const size_t o_nblk = take(sizeof(unsigned) * (size_t)(nblk + 8)), o_nstart = take(sizeof(int) * ((size_t)nblk + 1 + grid_scan_scratch_ints(nblk)));
    const size_t o_null = take(sizeof(int) * (size_t)n);
    if ((rc = ensure_aos(c, off))) return rc;
    char *b = static_cast<char *>(c->aos_buf);
    int *bin_of = reinterpret_cast<int *>(b + o_bin), *bin_start = reinterpret_cast<int *>(b + o_start), *members = reinterpret_cast<int *>(b + o_mem);
    unsigned *bin_count = reinterpret_cast<unsigned *>(b + o_cnt), *cursor = reinterpret_cast<unsigned *>(b + o_cur);
    unsigned *empty = bin_count + B;                              // [B]: empty bins, [B + 1]: photons outside the histograms
    RebinRec *recs = reinterpret_cast<RebinRec *>(b + o_rec);
    unsigned *null_cnt = reinterpret_cast<unsigned *>(b + o_nblk);
    int *null_start = reinterpret_cast<int *>(b + o_nstart), *null_slots = reinterpret_cast<int *>(b + o_null);
    if ((rc = ensure_counts(c, 1))) return rc;
    HIPCHK(c, launch_rebin_assign(c->ph, ax, bin_of, bin_count, c->stream));           // zeroes bin_count[0 .. B + 1] first
    HIPCHK(c, hipMemsetAsync(cursor, 0, sizeof(unsigned) * (size_t)B, c->stream));
    // how many photons take part (= the scan's total): eligible photons are all binned or the call fails
    std::vector<unsigned> h_cnt((size_t)B + 2);
    HIPCHK(c, hipMemcpyAsync(h_cnt.data(), bin_count, sizeof(unsigned) * ((size_t)B + 2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    long long members_total = 0;
    for (int k = 0; k < B; ++k) members_total += h_cnt[(size_t)k];
    if (h_cnt[(size_t)B + 1] != 0) { c->last_error = "rebinning: a photon maps to an invalid bin index (the reference exits)"; return MCRAT_HIP_EREFUSED; }
    HIPCHK(c, launch_exclusive_scan(bin_count, B, bin_start, bin_start + B + 1, members_total, c->stream));
    HIPCHK(c, launch_rebin_fill(c->ph, bin_of, bin_start, cursor, members, c->stream));
    HIPCHK(c, launch_rebin_create(c->ph, ax, bin_start, members, recs, empty, c->stream));
    HIPCHK(c, launch_rebin_nullify(c->ph, c->stream));                                            // :573-581
    // addToPhotonList(rebin_ph, total_bins), photons.c:108-208
    unsigned long long n_null = 0;
    HIPCHK(c, launch_null_count(c->ph, null_cnt, c->d_grid_total, c->stream));
    unsigned h_empty[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(&n_null, c->d_grid_total, sizeof n_null, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_empty, empty, sizeof h_empty, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if ((unsigned long long)B > n_null) {
        c->last_error = "rebinning: fewer null slots than rebinned photons (the reference exits with \"Adding to the photon list has failed\")";
        return MCRAT_HIP_EREFUSED;
    }
    HIPCHK(c, launch_exclusive_scan(null_cnt, nblk, null_start, null_start + nblk + 1, (long long)n_null, c->stream));
    HIPCHK(c, launch_null_write(c->ph, null_start, null_slots, c->stream));
    HIPCHK(c, launch_rebin_place(c->ph, recs, B, null_slots, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int null_count = (int)h_empty[0];
    if ((long long)n - (long long)n_null + (B - null_count) < B) {                                // :676-681
        c->last_error = "rebinning: fewer photons in the list than bins after the rebinning";
        return MCRAT_HIP_EREFUSED;
    }
    if (empty_bins_out) *empty_bins_out = null_count;
    if (scatt_cyclosynch_num_ph) *scatt_cyclosynch_num_ph = B - null_count;                       // :689-690
    if (num_cyclosynch_ph_emit) *num_cyclosynch_ph_emit = B + q.synch - null_count;
    drop_graph(c);
    return MCRAT_HIP_OK;
}

// rebinCyclosynchCompPhotons (mc_cyclosynch.c:610-710) for the lists `ids` of a rank pool at once: two launches (one workgroup per list) and two
// host round trips for all of them, against a dozen launches and four round trips per list through mcrat_hip_rebin_cyclosynch on the views.
// The host's part in between -- the histograms' axes from the ranges (:324-391) -- is the same code, so a list comes out bit for bit the same.
// rc[j]: MCRAT_HIP_OK, or MCRAT_HIP_EREFUSED where the reference would refuse or exit (the view's last_error says why); empty / emit / scatt as
// mcrat_hip_rebin_cyclosynch returns them.  Returns the first hard error.
struct PoolRebinResult { int rc, empty_bins, num_cyclosynch_ph_emit, scatt_cyclosynch_num_ph; };
static int pool_rebin_lists(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, int max_photons, const std::vector<int> &ids, std::vector<PoolRebinResult> &res)
{
    const int L = (int)ids.size();
    res.assign((size_t)L, PoolRebinResult{MCRAT_HIP_OK, 0, 0, 0});
    if (L == 0) return MCRAT_HIP_OK;
    const int three = c->kc.dimensions == DIM_THREE;
    std::vector<RebinPoolList> h((size_t)L);
    for (int j = 0; j < L; ++j) {
        mcrat_hip_ctx *v = c->views[ids[(size_t)j]];
        int rc = flush_pending(v);
        if (rc) return rc;
        memset(&h[(size_t)j], 0, sizeof(RebinPoolList));
        h[(size_t)j].first = ids[(size_t)j] * c->rank_stride;
        h[(size_t)j].n = v->ph.n;
    }
    // launch 1: collect_photon_statistics :273-322 of every list
    const size_t o_lists = 0, o_range = align_up(sizeof(RebinPoolList) * (size_t)L, 256);
    int rc = ensure_aos(c, o_range + sizeof(RebinRange) * (size_t)L);
    if (rc) return rc;
    char *b = static_cast<char *>(c->aos_buf);
    HIPCHK(c, hipMemcpyAsync(b + o_lists, h.data(), sizeof(RebinPoolList) * (size_t)L, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_rebin_pool_range(c->ph, three, reinterpret_cast<const RebinPoolList *>(b + o_lists), L, reinterpret_cast<RebinRange *>(b + o_range), c->stream));
    std::vector<RebinRange> range((size_t)L);
    HIPCHK(c, hipMemcpyAsync(range.data(), b + o_range, sizeof(RebinRange) * (size_t)L, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the host's part, list by list: calculate_binning_params :324-347, allocate_histograms :351-391
    std::vector<RebinPoolList> go;
    std::vector<int> go_j;
    size_t scratch = 0;
    for (int j = 0; j < L; ++j) {
        mcrat_hip_ctx *v = c->views[ids[(size_t)j]];
        const RebinRange &q = range[(size_t)j];
        auto refuse = [&](const char *why) { v->last_error = why; res[(size_t)j].rc = MCRAT_HIP_EREFUSED; };
        if (q.valid == 0) { refuse("rebinning: no valid photons found for rebinning"); continue; }
        const double log_p0_min = (q.p0_min > 0 && q.p0_max > 0) ? std::log10(q.p0_min) : 0.0, log_p0_max = (q.p0_min > 0 && q.p0_max > 0) ? std::log10(q.p0_max) : 1.0;
        RebinAxes ax{};
        ax.three = three;
        ax.num_bins = (int)(cs->rebin_e_perc * max_photons);
        ax.num_bins_theta = (int)std::ceil((q.theta_max - q.theta_min) / (cs->rebin_ang * (M_PI / 180.0)));
        ax.num_bins_phi = three ? (int)std::ceil((q.phi_max - q.phi_min) / cs->rebin_ang_phi) : 1;
        const long long total_ll = (long long)ax.num_bins_theta * ax.num_bins * (three ? ax.num_bins_phi : 1);
        if (total_ll > max_photons) { refuse("rebinning would create more photons than max_photons"); continue; }
        if (ax.num_bins <= 0 || ax.num_bins_theta <= 0 || ax.num_bins_phi <= 0) { refuse("rebinning: invalid histogram dimensions"); continue; }
        ax.total_bins = (int)total_ll;
        ax.e_lo = log_p0_min; ax.e_hi = log_p0_max + (log_p0_max - log_p0_min) * 1e-6;
        ax.t_lo = q.theta_min; ax.t_hi = q.theta_max + (q.theta_max - q.theta_min) * 1e-6;
        ax.p_lo = q.phi_min; ax.p_hi = q.phi_max + (q.phi_max - q.phi_min) * 1e-6;
        RebinPoolList l = h[(size_t)j];
        l.ax = ax;
        l.scratch = scratch;
        l.status = -1;
        scratch += rebin_pool_scratch_bytes(l.n, ax.total_bins);
        go.push_back(l);
        go_j.push_back(j);
    }
    const int G = (int)go.size();
    if (G == 0) return MCRAT_HIP_OK;
    // launch 2: everything else
    const size_t o_scr = align_up(sizeof(RebinPoolList) * (size_t)G, 256);
    if ((rc = ensure_aos(c, o_scr + scratch))) return rc;
    b = static_cast<char *>(c->aos_buf);
    HIPCHK(c, hipMemcpyAsync(b, go.data(), sizeof(RebinPoolList) * (size_t)G, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_rebin_pool(c->ph, reinterpret_cast<RebinPoolList *>(b), G, b + o_scr, c->stream));
    HIPCHK(c, hipMemcpyAsync(go.data(), b, sizeof(RebinPoolList) * (size_t)G, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int g = 0; g < G; ++g) {
        const int j = go_j[(size_t)g];
        mcrat_hip_ctx *v = c->views[ids[(size_t)j]];
        const RebinPoolList &l = go[(size_t)g];
        const int B = l.ax.total_bins;
        drop_graph(v);
        if (l.status == 1) { v->last_error = "rebinning: a photon maps to an invalid bin index (the reference exits)"; res[(size_t)j].rc = MCRAT_HIP_EREFUSED; continue; }
        if (l.status == 2) {
            v->last_error = "rebinning: fewer null slots than rebinned photons (the reference exits with \"Adding to the photon list has failed\")";
            res[(size_t)j].rc = MCRAT_HIP_EREFUSED;
            continue;
        }
        if (l.status != 0) { c->last_error = "rebinning: the pool kernel left a list without a status"; return MCRAT_HIP_EHIP; }
        if ((long long)l.n - (long long)l.n_null + (B - l.empty_bins) < B) {                          // :676-681
            v->last_error = "rebinning: fewer photons in the list than bins after the rebinning";
            res[(size_t)j].rc = MCRAT_HIP_EREFUSED;
            continue;
        }
        res[(size_t)j].empty_bins = l.empty_bins;
        res[(size_t)j].scatt_cyclosynch_num_ph = B - l.empty_bins;                                    // :689-690
        res[(size_t)j].num_cyclosynch_ph_emit = B + range[(size_t)j].synch - l.empty_bins;
    }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_num_photon_slots(const mcrat_hip_ctx *c) { return (c && c->have_photons) ? c->ph.n : 0; }

extern "C" int mcrat_hip_profile_totals(const mcrat_hip_ctx *c, double *loop_kernel_ms, long long *loop_kernel_launches)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->cfg.profile) return MCRAT_HIP_ESTATE;
    if (loop_kernel_ms) *loop_kernel_ms = c->prof_step_ms;
    if (loop_kernel_launches) *loop_kernel_launches = c->prof_launches;
    return MCRAT_HIP_OK;
}

static int flush_pending(mcrat_hip_ctx *c)
{
    if (c->n_ranks > 0 || (c->parent && c->rank_current)) return MCRAT_HIP_OK;      // rank_loop_kernel leaves the photons current
    if (c->pending_applied) return MCRAT_HIP_OK;  // between the two halves of a pass: the step kernel has applied it, the event kernel will replace it
    HIPCHK(c, launch_flush(c->ph, c->d_state, c->step_blocks, c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_get_photons_soa(mcrat_hip_ctx *c, const mcrat_hip_photon_soa *s)
{
    if (!c || !s) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    if (s->n != c->ph.n) return MCRAT_HIP_EINVAL;
    const int n = c->ph.n;
    int rc = flush_pending(c);
    if (rc) return rc;
    PhotonDev &p = c->ph;
    const double *cols[19] = {p.r0, p.r1, p.r2, p.p0, p.p1, p.p2, p.p3, p.c0, p.c1, p.c2, p.c3,
                              p.s0, p.s1, p.s2, p.s3, p.num_scatt, p.weight, p.tau, p.tts};
    double *dst[19] = {s->r0, s->r1, s->r2, s->p0, s->p1, s->p2, s->p3, s->comv_p0, s->comv_p1, s->comv_p2, s->comv_p3,
                       s->s0, s->s1, s->s2, s->s3, s->num_scatt, s->weight, s->total_optical_depth, s->time_to_scatter};
    for (int k = 0; k < 19; ++k)
        if (dst[k]) HIPCHK(c, hipMemcpyAsync(dst[k], cols[k], sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    if (s->nearest_block_index) HIPCHK(c, hipMemcpyAsync(s->nearest_block_index, p.idx, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    if (s->type) HIPCHK(c, hipMemcpyAsync(s->type, p.type, n, hipMemcpyDeviceToHost, c->stream));
    std::vector<unsigned char> flags;
    if (s->recalc_properties) {
        flags.resize(n);
        HIPCHK(c, hipMemcpyAsync(flags.data(), p.flags, n, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (s->recalc_properties)
        for (int i = 0; i < n; ++i) s->recalc_properties[i] = (flags[i] & FLAG_RECALC) ? 1 : 0;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_get_photons(mcrat_hip_ctx *c, mcrat_hip_photon_list *l)
{
    if (!c || !l || !l->photons) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    if (l->list_capacity != c->ph.n) return MCRAT_HIP_EINVAL;
    const int n = c->ph.n;
    int rc = flush_pending(c);
    if (rc) return rc;
    const size_t bytes = sizeof(mcrat_hip_photon) * (size_t)n;
    const bool fresh = !c->aos_buf || c->aos_bytes < bytes || c->parent;     // (a view's staging buffer is its pool's)
    if ((rc = ensure_aos(c, bytes))) return rc;
    // photons that came in as SoA columns have no uploaded records: the bytes between the members are the caller's
    if (fresh) HIPCHK(c, hipMemcpyAsync(c->aos_buf, l->photons, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_soa_to_aos(c->ph, c->aos_buf, 0, n, c->stream));
    HIPCHK(c, hipMemcpyAsync(l->photons, c->aos_buf, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->cfg.cyclosynchrotron_switch) {          // the list changed inside the frame: the counts of photons.c:252-275
        int nulls = 0;
        for (int i = 0; i < n; ++i) nulls += l->photons[i].type == 'N';
        l->num_null_photons = nulls;
        l->num_photons = n - nulls;
    }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_get_photons_range(mcrat_hip_ctx *c, int first, int count, mcrat_hip_photon *records)
{
    if (!c || !records || first < 0 || count <= 0) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    if ((long long)first + count > c->ph.n) return MCRAT_HIP_EINVAL;
    int rc = flush_pending(c);
    if (rc) return rc;
    const size_t bytes = sizeof(mcrat_hip_photon) * (size_t)count;
    if ((rc = ensure_aos(c, bytes))) return rc;
    HIPCHK(c, hipMemsetAsync(c->aos_buf, 0, bytes, c->stream));        // the bytes between the members: zero
    HIPCHK(c, launch_soa_to_aos(c->ph, c->aos_buf, first, count, c->stream));
    HIPCHK(c, hipMemcpyAsync(records, c->aos_buf, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_convert_comptonized(mcrat_hip_ctx *c, int *num_converted)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    int rc = ensure_aos(c, sizeof(unsigned));
    if (rc) return rc;
    unsigned n = 0;
    HIPCHK(c, hipMemsetAsync(c->aos_buf, 0, sizeof(unsigned), c->stream));
    HIPCHK(c, launch_convert_comptonized(c->ph, static_cast<unsigned *>(c->aos_buf), c->stream));
    HIPCHK(c, hipMemcpyAsync(&n, c->aos_buf, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (num_converted) *num_converted = (int)n;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_get_output(mcrat_hip_ctx *c, mcrat_hip_output_columns *out)
{
    if (!c || !out) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    const int n = c->ph.n;
    int rc = flush_pending(c);
    if (rc) return rc;
    const long long nblk = (n + 255) / 256;
    { int rc_ = ensure_counts(c, (size_t)nblk); if (rc_) return rc_; }
    unsigned long long total = 0;
    HIPCHK(c, launch_output_count(c->ph, n, c->grid_count, c->d_grid_total, c->stream));
    HIPCHK(c, hipMemcpyAsync(&total, c->d_grid_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int m = (int)total;
    double *dst[17] = {out->p0, out->p1, out->p2, out->p3, out->comv_p0, out->comv_p1, out->comv_p2, out->comv_p3, out->r0, out->r1, out->r2,
                       out->s0, out->s1, out->s2, out->s3, out->num_scatt, out->weight};
    bool any = out->type != nullptr;
    for (double *d : dst) any = any || d;
    if (!any) { out->count = m; return MCRAT_HIP_OK; }            // sizing call
    if (out->count < m) { out->count = m; return MCRAT_HIP_EINVAL; }
    out->count = m;
    if (m == 0) return MCRAT_HIP_OK;
    const size_t stride = align_up(sizeof(double) * (size_t)m, 256);
    const size_t o_scan = 17 * stride + align_up((size_t)m, 256);
    const size_t bytes = o_scan + sizeof(int) * ((size_t)nblk + 1 + grid_scan_scratch_ints(nblk));
    if ((rc = ensure_aos(c, bytes))) return rc;
    char *base = static_cast<char *>(c->aos_buf);
    OutputCols oc;
    for (int k = 0; k < 17; ++k) oc.col[k] = reinterpret_cast<double *>(base + k * stride);
    oc.type = base + 17 * stride;
    int *start = reinterpret_cast<int *>(base + o_scan), *scratch = start + nblk + 1;
    HIPCHK(c, launch_exclusive_scan(c->grid_count, nblk, start, scratch, (long long)total, c->stream));
    HIPCHK(c, launch_output_write(c->ph, n, start, oc, c->stream));
    for (int k = 0; k < 17; ++k)
        if (dst[k]) HIPCHK(c, hipMemcpyAsync(dst[k], oc.col[k], sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    if (out->type) HIPCHK(c, hipMemcpyAsync(out->type, oc.type, (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCRAT_HIP_OK;
}

// ---------------------------------------------------------------------------------------------- asynchronous output
// What saveCheckpoint (mcrat_io.c:838-1009) and printPhotons (mcrat_io.c:114-836) read, taken off the device WHILE THE NEXT FRAME RUNS: post()
// stages the records and the compacted output columns in device buffers of the outbox by kernels in the context's stream order -- the photons
// may change as soon as post() has returned -- and a copy stream of the outbox's own brings them into pinned host memory; wait() (any thread)
// blocks until they have landed.  (The reference writes both files at the end of every frame with its rank idle, mcrat.c:902,907.)
struct mcrat_hip_outbox {
    int device = 0;
    hipStream_t copy = nullptr;
    hipEvent_t staged = nullptr, landed = nullptr;
    void *d_buf = nullptr, *h_buf = nullptr;
    size_t cap = 0;
    size_t o_rec = 0, o_cols = 0, col_stride = 0;
    int n_records = 0, n_output = 0;
    bool posted = false, have_records = false, have_output = false;
};

extern "C" int mcrat_hip_outbox_create(mcrat_hip_ctx *c, mcrat_hip_outbox **out)
{
    if (!c || !out) return MCRAT_HIP_EINVAL;
    mcrat_hip_outbox *b = new (std::nothrow) mcrat_hip_outbox();
    if (!b) return MCRAT_HIP_ENOMEM;
    b->device = c->cfg.device;
    if (hipStreamCreateWithFlags(&b->copy, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&b->staged, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&b->landed, hipEventDisableTiming) != hipSuccess) {
        mcrat_hip_outbox_destroy(b);
        return MCRAT_HIP_EHIP;
    }
    *out = b;
    return MCRAT_HIP_OK;
}

extern "C" void mcrat_hip_outbox_destroy(mcrat_hip_outbox *b)
{
    if (!b) return;
    if (b->copy) { (void)hipStreamSynchronize(b->copy); (void)hipStreamDestroy(b->copy); }
    if (b->staged) (void)hipEventDestroy(b->staged);
    if (b->landed) (void)hipEventDestroy(b->landed);
    if (b->d_buf) (void)hipFree(b->d_buf);
    if (b->h_buf) (void)hipHostFree(b->h_buf);
    delete b;
}

extern "C" int mcrat_hip_outbox_post(mcrat_hip_ctx *c, mcrat_hip_outbox *b, int want_records, int want_output)
{
    if (!c || !b || (!want_records && !want_output)) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    const int n = c->ph.n;
    int rc = flush_pending(c);
    if (rc) return rc;
    if (b->posted) HIPCHK(c, hipEventSynchronize(b->landed));          // (a post over one nobody waited for)
    b->posted = false;
    const long long nblk = (n + 255) / 256;
    unsigned long long total = 0;
    if (want_output) {
        int rc_ = ensure_counts(c, (size_t)nblk);
        if (rc_) return rc_;
        HIPCHK(c, launch_output_count(c->ph, n, c->grid_count, c->d_grid_total, c->stream));
        HIPCHK(c, hipMemcpyAsync(&total, c->d_grid_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const size_t m = (size_t)total;
    const size_t rec_bytes = want_records ? align_up(sizeof(mcrat_hip_photon) * (size_t)n, 256) : 0;
    const size_t stride = align_up(sizeof(double) * (m ? m : 1), 256);
    const size_t cols_bytes = want_output ? 17 * stride + align_up(m ? m : 1, 256) : 0;
    const size_t scan_bytes = want_output ? align_up(sizeof(int) * ((size_t)nblk + 1 + grid_scan_scratch_ints(nblk)), 256) : 0;
    const size_t need = rec_bytes + cols_bytes + scan_bytes;
    if (need > b->cap) {
        const size_t cap = need + need / 4;
        if (b->d_buf) (void)hipFree(b->d_buf);
        if (b->h_buf) (void)hipHostFree(b->h_buf);
        b->d_buf = b->h_buf = nullptr;
        b->cap = 0;
        HIPCHK(c, hipMalloc(&b->d_buf, cap));
        HIPCHK(c, hipHostMalloc(&b->h_buf, cap, hipHostMallocDefault));
        b->cap = cap;
    }
    char *d = static_cast<char *>(b->d_buf), *h = static_cast<char *>(b->h_buf);
    b->o_rec = 0; b->o_cols = rec_bytes; b->col_stride = stride;
    b->n_records = want_records ? n : 0;
    b->n_output = (int)m;
    b->have_records = want_records != 0;
    b->have_output = want_output != 0;
    if (want_records) {
        HIPCHK(c, hipMemsetAsync(d, 0, rec_bytes, c->stream));            // the bytes between the members: zero
        HIPCHK(c, launch_soa_to_aos(c->ph, d, 0, n, c->stream));
    }
    if (want_output && m > 0) {
        OutputCols oc;
        for (int k = 0; k < 17; ++k) oc.col[k] = reinterpret_cast<double *>(d + b->o_cols + k * stride);
        oc.type = d + b->o_cols + 17 * stride;
        int *start = reinterpret_cast<int *>(d + rec_bytes + cols_bytes), *scratch = start + nblk + 1;
        HIPCHK(c, launch_exclusive_scan(c->grid_count, nblk, start, scratch, (long long)total, c->stream));
        HIPCHK(c, launch_output_write(c->ph, n, start, oc, c->stream));
    }
    HIPCHK(c, hipEventRecord(b->staged, c->stream));
    HIPCHK(c, hipStreamWaitEvent(b->copy, b->staged, 0));
    const size_t bytes = rec_bytes + (want_output && m > 0 ? cols_bytes : 0);
    if (bytes) HIPCHK(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b->copy));
    HIPCHK(c, hipEventRecord(b->landed, b->copy));
    b->posted = true;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_outbox_wait(mcrat_hip_outbox *b, const mcrat_hip_photon **records, int *n_records, mcrat_hip_output_columns *cols)
{
    if (!b) return MCRAT_HIP_EINVAL;
    if (!b->posted) return MCRAT_HIP_ESTATE;
    if (hipSetDevice(b->device) != hipSuccess) return MCRAT_HIP_ENODEV;       // (the caller may be a thread of its own)
    if (hipEventSynchronize(b->landed) != hipSuccess) return MCRAT_HIP_EHIP;
    char *h = static_cast<char *>(b->h_buf);
    if (records) *records = b->have_records ? reinterpret_cast<const mcrat_hip_photon *>(h + b->o_rec) : nullptr;
    if (n_records) *n_records = b->n_records;
    if (cols) {
        memset(cols, 0, sizeof *cols);
        cols->count = b->have_output ? b->n_output : 0;
        if (b->have_output && b->n_output > 0) {
            double **dst[17] = {&cols->p0, &cols->p1, &cols->p2, &cols->p3, &cols->comv_p0, &cols->comv_p1, &cols->comv_p2, &cols->comv_p3, &cols->r0, &cols->r1,
                                &cols->r2, &cols->s0, &cols->s1, &cols->s2, &cols->s3, &cols->num_scatt, &cols->weight};
            for (int k = 0; k < 17; ++k) *dst[k] = reinterpret_cast<double *>(h + b->o_cols + k * b->col_stride);
            cols->type = h + b->o_cols + 17 * b->col_stride;
        }
    }
    return MCRAT_HIP_OK;
}

// ---------------------------------------------------------------------------------------------- the loop
static long long read_table_fallbacks(mcrat_hip_ctx *c)
{
    int m = 0;
    if (c->cfg.tau_calculation == MCRAT_HIP_TAU_TABLE && c->d_table_fallbacks &&
        hipMemcpy(&m, c->d_table_fallbacks, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) m = -1;
    return m;
}

static void fill_stats(mcrat_hip_ctx *c, mcrat_hip_frame_stats *s)
{
    if (!s) return;
    const LoopState &h = *c->h_state;
    s->iterations = h.iterations;
    s->photon_steps = h.iterations * (long long)c->ph.n;
    s->slot_steps = h.slot_steps;
    s->frame_scatt_cnt = h.frame_scatt_cnt;
    s->num_photons_find_new_element = h.n_relocated;
    s->not_found = h.not_found;
    s->kn_rejections = h.kn_rejections;
    s->rescans = h.rescans;
    s->last_scattered_index = h.last_scattered_index;
    s->last_scattered_temp = h.last_scattered_temp;
    s->last_time_step = h.last_time_step;
    s->remaining_time = h.remaining_time;
    s->time_now = h.time_now;
    s->step_kernel_ms = c->prof_step_ms;
    s->step_kernel_launches = c->prof_launches;
    s->event_kernel_ms = c->prof_event_ms;
    s->table_fallbacks = read_table_fallbacks(c);
}

static void state_to_stats(const LoopState &h, long long slots, mcrat_hip_frame_stats *s)
{
    memset(s, 0, sizeof *s);
    s->iterations = h.iterations;
    s->photon_steps = h.iterations * slots;
    s->slot_steps = h.slot_steps;
    s->frame_scatt_cnt = h.frame_scatt_cnt;
    s->num_photons_find_new_element = h.n_relocated;
    s->not_found = h.not_found;
    s->kn_rejections = h.kn_rejections;
    s->rescans = h.rescans;
    s->last_scattered_index = h.last_scattered_index;
    s->last_scattered_temp = h.last_scattered_temp;
    s->last_time_step = h.last_time_step;
    s->remaining_time = h.remaining_time;
    s->time_now = h.time_now;
}

static int rank_slots(const mcrat_hip_ctx *c, int r)
{
    if (c->is_pool) return c->h_desc[r].len;
    const int per = c->cfg.virtual_rank_photons;
    return std::min(per, c->ph.n - r * per);
}

static int longest_rank_list(const mcrat_hip_ctx *c)
{
    if (!c->is_pool) return std::min(c->cfg.virtual_rank_photons, c->ph.n);
    int m = 0;
    for (int r = 0; r < c->n_ranks; ++r) m = std::max(m, c->h_desc[r].len);
    return m;
}

// whole-job view of the virtual ranks: counters add up; the clock shown is that of the rank furthest behind
static void fill_rank_stats(mcrat_hip_ctx *c, mcrat_hip_frame_stats *s)
{
    if (!s) return;
    mcrat_hip_frame_stats t, acc;
    memset(&acc, 0, sizeof acc);
    for (int r = 0; r < c->n_ranks; ++r) {
        state_to_stats(c->h_rstates[r], rank_slots(c, r), &t);
        acc.iterations += t.iterations; acc.photon_steps += t.photon_steps; acc.slot_steps += t.slot_steps; acc.frame_scatt_cnt += t.frame_scatt_cnt;
        acc.num_photons_find_new_element += t.num_photons_find_new_element; acc.not_found += t.not_found;
        acc.kn_rejections += t.kn_rejections; acc.rescans += t.rescans;
        if (r == 0 || t.remaining_time > acc.remaining_time) {
            acc.remaining_time = t.remaining_time; acc.time_now = t.time_now; acc.last_time_step = t.last_time_step;
            acc.last_scattered_index = t.last_scattered_index; acc.last_scattered_temp = t.last_scattered_temp;
        }
    }
    *s = acc;
}

#ifdef MCRAT_DIAG
extern "C" __attribute__((visibility("default"))) int mcrat_hip_diag_rank_stamps(mcrat_hip_ctx *c, int rank, long long out[8])
{
    if (!c || !out || rank < 0 || rank >= c->n_ranks) return -1;
    LoopState h;
    if (hipMemcpy(&h, c->d_rstates + rank, sizeof(LoopState), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out[k] = h.stamps[k];
    return 0;
}

extern "C" __attribute__((visibility("default"))) int mcrat_hip_diag_stamps(mcrat_hip_ctx *c, long long out[8])
{
    if (!c || !out) return -1;
    if (hipMemcpy(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (int k = 0; k < 8; ++k) out[k] = c->h_state->stamps[k];
    return 0;
}
#endif

extern "C" int mcrat_hip_num_virtual_ranks(const mcrat_hip_ctx *c) { return c ? c->n_ranks : 0; }

extern "C" int mcrat_hip_rank_stats(mcrat_hip_ctx *c, int rank, mcrat_hip_frame_stats *stats)
{
    if (!c || !stats) return MCRAT_HIP_EINVAL;
    if (c->n_ranks <= 0 || !c->frame_open) return MCRAT_HIP_ESTATE;
    if (rank < 0 || rank >= c->n_ranks) return MCRAT_HIP_EINVAL;
    HIPCHK(c, hipMemcpyAsync(c->h_rstates, c->d_rstates, sizeof(LoopState) * c->n_ranks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    state_to_stats(c->h_rstates[rank], rank_slots(c, rank), stats);
    return MCRAT_HIP_OK;
}

// ---------------------------------------------------------------------------------------------- rank pool
extern "C" int mcrat_hip_pool_create(mcrat_hip_ctx *c, int n_ranks, int slots_per_rank)
{
    if (!c || n_ranks <= 0 || slots_per_rank <= 0) return MCRAT_HIP_EINVAL;
    if (c->parent) { c->last_error = "a rank view cannot hold a pool"; return MCRAT_HIP_ESTATE; }
    if (c->sc_world > 0) { c->last_error = "shared clock and rank pool exclude each other"; return MCRAT_HIP_ESTATE; }
    if (c->d_tape) { c->last_error = "a context with a tape of uniforms holds one list (mcrat_hip_set_rng_tape)"; return MCRAT_HIP_ESTATE; }
    const size_t stride = align_up((size_t)slots_per_rank, 2 * STEP_BLOCK);
    if (stride * (size_t)n_ranks > 0x7fffffffull - 2 * STEP_BLOCK) { c->last_error = "rank pool: more than 2^31 slots"; return MCRAT_HIP_EINVAL; }
    for (mcrat_hip_ctx *v : c->views)
        if (v) { v->parent = nullptr; destroy_view(v); }
    c->views.clear();
    c->is_pool = false;
    c->cfg.virtual_rank_photons = (int)stride;
    int rc = alloc_photons(c, (int)(stride * (size_t)n_ranks));       // zeroed: no slot belongs to a list yet
    if (rc) return rc;
    c->is_pool = true;
    c->views.assign((size_t)n_ranks, nullptr);
    if (c->d_desc) { HIPCHK(c, hipFree(c->d_desc)); c->d_desc = nullptr; }
    if (c->h_desc) { HIPCHK(c, hipHostFree(c->h_desc)); c->h_desc = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->d_desc, sizeof(RankDesc) * (size_t)n_ranks));
    HIPCHK(c, hipHostMalloc((void **)&c->h_desc, sizeof(RankDesc) * (size_t)n_ranks, hipHostMallocDefault));
    memset(c->h_desc, 0, sizeof(RankDesc) * (size_t)n_ranks);
    HIPCHK(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RankDesc) * (size_t)n_ranks, hipMemcpyHostToDevice, c->stream));
    memset(c->h_rstates, 0, sizeof(LoopState) * (size_t)n_ranks);
    for (int r = 0; r < n_ranks; ++r) { c->h_rstates[r].done = 1; c->h_rstates[r].skip_idx = -1; c->h_rstates[r].last_scattered_index = -1; }
    HIPCHK(c, hipMemcpyAsync(c->d_rstates, c->h_rstates, sizeof(LoopState) * (size_t)n_ranks, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_photons = true;
    c->frame_open = false;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_pool_rank(mcrat_hip_ctx *c, int rank, uint32_t rng_stream, mcrat_hip_ctx **view)
{
    if (!c || !view) return MCRAT_HIP_EINVAL;
    *view = nullptr;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    if (rank < 0 || rank >= c->n_ranks) return MCRAT_HIP_EINVAL;
    mcrat_hip_ctx *v = c->views[rank];
    if (!v) {
        v = new (std::nothrow) mcrat_hip_ctx();
        if (!v) return MCRAT_HIP_ENOMEM;
        v->cfg = c->cfg;
        v->cfg.virtual_rank_photons = 0;
        v->cfg.stream = c->stream;
        v->kc = c->kc;
        v->stream = c->stream;
        v->own_stream = false;
        v->parent = c;
        v->view_rank = rank;
        v->d_state = c->d_rstates + rank;         // windows into the pool's blocks; nothing here is owned by the view
        v->h_state = c->h_rstates + rank;
        v->d_red = c->d_red; v->h_red = c->h_red;
        v->d_table_fallbacks = c->d_table_fallbacks;
        c->views[rank] = v;
        sync_views(c);
    }
    v->cfg.rng_stream = rng_stream;
    v->key.stream = rng_stream & 0xffffffu;
    v->view_stream = v->key.stream;
    *view = v;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_pool_layout(const mcrat_hip_ctx *c, int *n_ranks, int *slots_per_rank)
{
    if (!c || !c->is_pool) return MCRAT_HIP_ESTATE;
    if (n_ranks) *n_ranks = c->n_ranks;
    if (slots_per_rank) *slots_per_rank = c->rank_stride;
    return MCRAT_HIP_OK;
}

// mcrat_hip_begin_frame for many views in one launch: every list with open[r] != 0 gets its own seed and clock
extern "C" int mcrat_hip_pool_begin_frames(mcrat_hip_ctx *c, const int *open, const uint64_t *seeds, const double *time_now, const double *remaining_time)
{
    if (!c || !open || !seeds || !time_now || !remaining_time) return MCRAT_HIP_EINVAL;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    if (!c->have_hydro) return MCRAT_HIP_ESTATE;
    if (c->cfg.tau_calculation == MCRAT_HIP_TAU_TABLE && !c->d_hot_table) {
        c->last_error = "TAU_CALCULATION == TABLE needs mcrat_hip_set_hot_cross_section first";
        return MCRAT_HIP_ESTATE;
    }
    const int R = c->n_ranks;
    std::vector<int> op((size_t)R, 0);
    for (int r = 0; r < R; ++r) {
        mcrat_hip_ctx *v = c->views[r];
        if (!open[r]) continue;
        if (!v || !v->have_photons) { c->last_error = "pool_begin_frames: a list that does not exist was asked to open a frame"; return MCRAT_HIP_ESTATE; }
        op[(size_t)r] = 1;
    }
    const size_t bytes = (sizeof(int) + 2 * sizeof(double)) * (size_t)R;
    int rc = ensure_aos(c, bytes + 64);
    if (rc) return rc;
    double *d_t = static_cast<double *>(c->aos_buf), *d_rem = d_t + R;
    int *d_open = reinterpret_cast<int *>(d_rem + R);
    HIPCHK(c, hipMemcpyAsync(d_t, time_now, sizeof(double) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rem, remaining_time, sizeof(double) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_open, op.data(), sizeof(int) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_table_fallbacks, 0, sizeof(int), c->stream));
    HIPCHK(c, launch_init_states_multi(c->d_rstates, R, d_open, d_t, d_rem, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));        // the arrays above are the caller's and `op`
    for (int r = 0; r < R; ++r) {
        if (!op[(size_t)r]) continue;
        mcrat_hip_ctx *v = c->views[r];
        if (v->graph_exec && v->key.seed != seeds[r]) drop_graph(v);
        v->key.seed = seeds[r];
        v->find_switch = 1;
        v->pending_applied = false;
        v->rank_current = false;
        v->frame_open = true;
        v->prof_step_ms = v->prof_event_ms = 0;
        v->prof_launches = 0;
        LoopState &h = *v->h_state;
        memset(&h, 0, sizeof h);
        h.remaining_time = remaining_time[r]; h.time_now = time_now[r]; h.done = !(remaining_time[r] > 0);
        h.skip_idx = -1; h.last_scattered_index = -1; h.force_relocate = 1;
    }
    c->rank_block_fixed = false;
    return MCRAT_HIP_OK;
}

// the loop statistics of every list (what mcrat_hip_frame_statistics gives for one view), one read-back
extern "C" int mcrat_hip_pool_frame_stats(mcrat_hip_ctx *c, mcrat_hip_frame_stats *out)
{
    if (!c || !out) return MCRAT_HIP_EINVAL;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    HIPCHK(c, hipMemcpyAsync(c->h_rstates, c->d_rstates, sizeof(LoopState) * (size_t)c->n_ranks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < c->n_ranks; ++r) {
        const mcrat_hip_ctx *v = c->views[r];
        state_to_stats(c->h_rstates[r], (v && v->have_photons) ? v->ph.n : 0, &out[r]);
    }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_pool_summaries(mcrat_hip_ctx *c, mcrat_hip_rank_summary *out)
{
    if (!c || !out) return MCRAT_HIP_EINVAL;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    const int R = c->n_ranks;
    for (int r = 0; r < R; ++r) {                  // every list that exists, whether or not it is in a frame
        const mcrat_hip_ctx *v = c->views[r];
        RankDesc d{};
        if (v && v->have_photons) { d.len = v->ph.n; d.stream = v->key.stream; d.seed = v->key.seed; }
        c->h_desc[r] = d;
    }
    const size_t bytes = (sizeof(ReducePartial) + sizeof(int)) * (size_t)R + sizeof(RankDesc) * (size_t)R;
    int rc = ensure_aos(c, bytes);
    if (rc) return rc;
    ReducePartial *d_part = static_cast<ReducePartial *>(c->aos_buf);
    RankDesc *d_desc = reinterpret_cast<RankDesc *>(d_part + R);
    int *d_nout = reinterpret_cast<int *>(d_desc + R);
    HIPCHK(c, hipMemcpyAsync(d_desc, c->h_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, launch_rank_reduce(c->ph, c->rank_stride, R, d_desc, d_part, d_nout, c->stream));
    std::vector<ReducePartial> part((size_t)R);
    std::vector<int> nout((size_t)R);
    HIPCHK(c, hipMemcpyAsync(part.data(), d_part, sizeof(ReducePartial) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(nout.data(), d_nout, sizeof(int) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int r = 0; r < R; ++r) {
        const ReducePartial &t = part[(size_t)r];
        mcrat_hip_rank_summary &o = out[r];
        memset(&o, 0, sizeof o);
        o.list_capacity = c->h_desc[r].len;
        if (o.list_capacity <= 0) continue;
        o.min_r = t.r_min; o.max_r = t.r_max; o.min_theta = t.th_min; o.max_theta = t.th_max;
        o.max_scatt = (int)t.max_scatt; o.min_scatt = (int)t.min_scatt;
        o.avg_scatt = t.sum_scatt / (double)t.count; o.avg_r = t.sum_r / (double)t.count;
        o.avg_energy = (t.e_sum * C_LIGHT) / t.w_sum;
        o.num_output = nout[(size_t)r];
    }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_begin_frame(mcrat_hip_ctx *c, uint64_t seed, double time_now, double remaining_time)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->have_hydro || !c->have_photons) return MCRAT_HIP_ESTATE;
    if (c->cfg.tau_calculation == MCRAT_HIP_TAU_TABLE && !c->d_hot_table) {
        c->last_error = "TAU_CALCULATION == TABLE needs mcrat_hip_set_hot_cross_section first";
        return MCRAT_HIP_ESTATE;
    }
    // everything below is ordered on the context's stream behind whatever the previous frame left there: no wait
    HIPCHK(c, hipMemsetAsync(c->d_table_fallbacks, 0, sizeof(int), c->stream));
    LoopState h;
    memset(&h, 0, sizeof h);
    h.remaining_time = remaining_time;
    h.time_now = time_now;
    h.done = !(remaining_time > 0);
    h.skip_idx = -1;
    h.last_scattered_index = -1;
    if (c->parent) h.force_relocate = 1;           // a view's LoopState is the pool's for this list (rank_loop_kernel looks at the flag)
    HIPCHK(c, launch_init_states(c->d_state, c->d_rstates, c->n_ranks, h, c->stream));     // per list: force_relocate = 1, mcrat.c:756
    if (c->is_pool)                                // the pool's own begin_frame: every list, the same seed and clock
        for (mcrat_hip_ctx *v : c->views)
            if (v && v->have_photons) { v->key.seed = seed; v->frame_open = true; v->find_switch = 1; v->pending_applied = false; v->rank_current = false; }
    *c->h_state = h;                               // the host's view until the next read-back (every read-back is followed by a wait)
    HIPCHK(c, hipMemsetAsync(c->shortlist, 0, sizeof(Shortlist), c->stream));
    if (c->sc_world > 0) HIPCHK(c, hipMemsetAsync(c->d_sc, 0, sizeof(ScState), c->stream));
    if (c->graph_exec && c->key.seed != seed) drop_graph(c);      // the captured launches carry the key by value
    c->key.seed = seed;
    c->find_switch = 1;           // mcrat.c:756
    c->pending_applied = false;
    c->rank_current = false;
    c->rank_block_fixed = false;
    c->frame_open = true;
    c->prof_step_ms = c->prof_event_ms = 0;
    c->prof_launches = 0;
    return MCRAT_HIP_OK;
}

// the event half of a pass from the context's random source: the keyed streams, or the caller's tape (then preceded by the pass's free-path
// draws in slot order, kernels.hip tape_draw_kernel)
static hipError_t launch_event_of(mcrat_hip_ctx *c)
{
    if (c->d_tape) {
        TapeDev t;
        t.u = c->d_tape; t.n = c->tape_n; t.cursor = c->d_tape_cursor; t.error = reinterpret_cast<int *>(c->d_tape_cursor + 1);
        return launch_tape_pass(c->kc, c->ph, c->hy, c->d_state, c->key, t, c->partials, c->step_blocks, c->shortlist, c->stream);
    }
    return launch_event(c->kc, c->ph, c->hy, c->d_state, c->key, c->partials, c->step_blocks, c->shortlist, c->stream);
}

static int launch_iteration(mcrat_hip_ctx *c, bool force)
{
    HIPCHK(c, launch_step(c->kc, force, c->ph, c->hy, c->d_state, c->key, c->partials, c->step_blocks, c->shortlist, c->stream));
    HIPCHK(c, launch_event_of(c));
    return MCRAT_HIP_OK;
}

// The random stream as an INPUT (SURVEY.md section 8c "tape").  MCRaT draws everything from one sequential ranlxs0 stream
// (mcrat.c:99-103,701); a maintainer who records that stream (tools/ref_harness: the doubles the generator returned, in call order) hands it
// over here and the loop consumes it exactly as MCRaT does -- one gsl_rng_uniform_pos per located slot in ascending slot order
// (mclib.c:646-675), then photonEvent's draws (electron.c:81,196,217-233; mcrat_scattering.c:519-574), gsl_rng_uniform_pos skipping zeros and the
// polar Gaussian taking as many pairs as it needs -- so that photons can be compared with MCRaT's own, photon for photon.  A validation mode: one
// list per context (no rank pool, no shared clock, no cyclo-synchrotron hook), the free-path draws of a pass walked by one workgroup.
// n == 0 (or uniforms == NULL) returns the context to its keyed streams.  The seed of begin_frame is ignored while a tape is set.
extern "C" int mcrat_hip_set_rng_tape(mcrat_hip_ctx *c, const double *uniforms, long long n)
{
    if (!c || n < 0) return MCRAT_HIP_EINVAL;
    if (c->parent || c->is_pool || c->n_ranks > 0 || c->cfg.virtual_rank_photons > 0 || c->sc_world > 0 || c->cfg.cyclosynchrotron_switch) {
        c->last_error = "the tape of uniforms is for one list with one clock (no rank pool, virtual ranks, shared clock or cyclo-synchrotron hook)";
        return MCRAT_HIP_ESTATE;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    drop_graph(c);
    if (c->d_tape) { (void)hipFree(c->d_tape); c->d_tape = nullptr; c->tape_n = 0; }
    if (!uniforms || n == 0) return MCRAT_HIP_OK;
    for (long long k = 0; k < n; ++k)
        if (!(uniforms[k] >= 0.0 && uniforms[k] < 1.0)) { c->last_error = "the tape holds a value outside [0, 1)"; return MCRAT_HIP_EINVAL; }
    if (!c->d_tape_cursor) HIPCHK(c, hipMalloc((void **)&c->d_tape_cursor, 2 * sizeof(long long)));
    HIPCHK(c, hipMemset(c->d_tape_cursor, 0, 2 * sizeof(long long)));
    HIPCHK(c, hipMalloc((void **)&c->d_tape, sizeof(double) * (size_t)n));
    HIPCHK(c, hipMemcpy(c->d_tape, uniforms, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    c->tape_n = n;
    return MCRAT_HIP_OK;
}

// how far the tape has been read (entries consumed so far, zeros skipped by uniform_pos included); *ran_out != 0: the loop needed more entries
// than the tape holds (its results are then meaningless).  Synchronises the stream.
extern "C" int mcrat_hip_rng_tape_position(mcrat_hip_ctx *c, long long *position, int *ran_out)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->d_tape) return MCRAT_HIP_ESTATE;
    long long h[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(h, c->d_tape_cursor, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (position) *position = h[0];
    if (ran_out) *ran_out = (int)(h[1] & 0xffffffffll) != 0;
    return MCRAT_HIP_OK;
}

static int ensure_events(mcrat_hip_ctx *c, size_t n)
{
    while (c->ev.size() < n) {
        hipEvent_t e;
        HIPCHK(c, hipEventCreate(&e));
        c->ev.push_back(e);
    }
    return MCRAT_HIP_OK;
}

static int ensure_graph(mcrat_hip_ctx *c, int batch)
{
    if (c->graph_exec && c->graph_batch == batch) return MCRAT_HIP_OK;
    drop_graph(c);
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = MCRAT_HIP_OK;
    for (int b = 0; b < batch && rc == MCRAT_HIP_OK; ++b) rc = launch_iteration(c, false);
    hipError_t e = hipStreamEndCapture(c->stream, &c->graph);
    if (rc != MCRAT_HIP_OK) return rc;
    HIPCHK(c, e);
    HIPCHK(c, hipGraphInstantiate(&c->graph_exec, c->graph, nullptr, nullptr, 0));
    c->graph_batch = batch;
    return MCRAT_HIP_OK;
}

static int ensure_events(mcrat_hip_ctx *c, size_t n);

// virtual-rank mode: every launch gives each unfinished list up to `per_launch` passes of its own loop
// Threads per list (launch.hpp).  Lists per CU is what the virtual-rank kernel's throughput hangs on (kernels.hip), so many
// lists get 128-thread workgroups, four to a CU -- unless there are too few lists to fill the device that way, or the lists
// are too long to keep in LDS, or the frames are optically thin: a thin frame is a dozen passes in which half the photons
// change cell, i.e. slow-path throughput per list, and there 256 threads per list do better.  The engine cannot know the
// optical depth before it has run a frame; it looks at the previous one (passes per list).
static void choose_rank_block(mcrat_hip_ctx *c)
{
    if (const char *e = getenv("MCRAT_HIP_RANK_BLOCK")) {
        c->rank_block = (atoi(e) == 128) ? 128 : (atoi(e) == 512 ? 512 : 256);
        c->rank_fuse = c->rank_passes_per_list < 48.0;
        if (const char *f = getenv("MCRAT_HIP_RANK_FUSE")) c->rank_fuse = atoi(f) != 0;
        return;
    }
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool many = c->n_ranks > 2 * cus && longest_rank_list(c) <= 1024;
    // ... and, whatever the frame looks like, when there are many times more lists than the device holds at once: four lists per CU then
    // overlap one list's walk with the others' passes all the time (10 246 lists, thin frames: cfg2 5.70 -> 4.96 ms, cfg3 8.44 -> 7.08 ms;
    // 4098 lists 2.50 -> 2.32 ms; 2049 lists no difference; 1025 lists 0.83 -> 0.85 ms)
    const bool very_many = c->n_ranks >= 12 * cus && longest_rank_list(c) <= 1024;
    c->rank_block = ((many && c->rank_passes_per_list >= 48.0) || very_many) ? 128 : 256;
    // lists of thousands of photons (sample_mc.par:21-22 allows 5000 per rank) of which there are about as many as CUs, or fewer: a list has its
    // CU to itself whatever the workgroup size, so it gets 512 threads -- a pass takes half the trips (200 lists of 5000 photons, cfg2: 5.2 -> ms)
    if (longest_rank_list(c) > 1088 && c->n_ranks <= cus + cus / 4) c->rank_block = 512;
    // the build with the fused pass (kernels.hip, rank_loop_kernel<.., FUSE>) for frames that looked optically thin last time (or
    // have not been seen yet): there most slots change cell between two events
    // (not in spherical geometry: two slots' acos / atan2 side by side cost the fused build 50 B of scratch per lane, and the spherical
    // benchmark frames run 2 % faster without it -- cfg3 at 1e7 photons 8.62 -> 8.43 ms; the cylindrical Stokes frame 1.07 -> 0.94 ms with it)
    c->rank_fuse = c->rank_passes_per_list < 48.0 && c->kc.geometry != GEOM_SPHERICAL;
    if (const char *e = getenv("MCRAT_HIP_RANK_FUSE")) c->rank_fuse = atoi(e) != 0;
}

// rank pool: what the kernel needs to know about every list, from its view
static int pool_describe(mcrat_hip_ctx *c)
{
    for (int r = 0; r < c->n_ranks; ++r) {
        const mcrat_hip_ctx *v = r < (int)c->views.size() ? c->views[r] : nullptr;
        RankDesc d{};
        if (v && v->have_photons && v->frame_open) { d.len = v->ph.n; d.stream = v->key.stream; d.seed = v->key.seed; }
        c->h_desc[r] = d;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RankDesc) * (size_t)c->n_ranks, hipMemcpyHostToDevice, c->stream));
    return MCRAT_HIP_OK;
}

static int run_ranks(mcrat_hip_ctx *c, long long max_iterations, mcrat_hip_frame_stats *stats)
{
    if (c->is_pool) { int rc = pool_describe(c); if (rc) return rc; }
    const int longest = longest_rank_list(c);
    if (!c->rank_block_fixed) { choose_rank_block(c); c->rank_block_fixed = true; }    // one choice per frame (begin_frame resets)
    long long per_launch_cap = 32768;           // bounds one launch to a second or two even for the densest lists (4096: the first 60 frames of
                                                // tools/lundman_run.py, 9.3e6 scatterings in the first one, 9.3 -> 8.9 s)
    if (const char *e = getenv("MCRAT_HIP_RANK_LAUNCH_CAP")) per_launch_cap = atoll(e) > 0 ? atoll(e) : per_launch_cap;
    long long it = 0;
    while (max_iterations <= 0 || it < max_iterations) {
        long long batch = per_launch_cap;
        if (max_iterations > 0 && batch > max_iterations - it) batch = max_iterations - it;
        if (c->cfg.profile) {
            int rc = ensure_events(c, 2);
            if (rc) return rc;
            HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
        }
        HIPCHK(c, launch_rank_loop(c->kc, c->ph, c->hy, c->d_rstates, c->key, c->n_ranks, c->rank_stride, longest, c->is_pool ? c->d_desc : nullptr, nullptr, nullptr, batch,
                                   c->rank_block + (c->rank_fuse ? 1000 : 0), c->stream));
        if (c->cfg.profile) {
            HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev[1]));
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            c->prof_step_ms += ms;
            c->prof_launches += 1;
        }
        it += batch;
        HIPCHK(c, hipMemcpyAsync(c->h_rstates, c->d_rstates, sizeof(LoopState) * c->n_ranks, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bool all_done = true;
        for (int r = 0; r < c->n_ranks && all_done; ++r) {
            if (c->is_pool && c->h_desc[r].len <= 0) continue;     // a window without a list (or without a frame) has nothing to finish
            all_done = c->h_rstates[r].done != 0;
        }
        if (all_done) {
            if (c->is_pool)
                for (mcrat_hip_ctx *v : c->views)
                    if (v && v->frame_open) v->rank_current = true;
            break;
        }
    }
    fill_rank_stats(c, stats);
    {   // what the next frame's choice of workgroup size looks at
        long long it_sum = 0;
        bool done = true;
        for (int r = 0; r < c->n_ranks; ++r) { it_sum += c->h_rstates[r].iterations; done = done && c->h_rstates[r].done; }
        if (done && c->n_ranks > 0) c->rank_passes_per_list = (double)it_sum / c->n_ranks;
    }
    if (stats) { stats->step_kernel_ms = c->prof_step_ms; stats->step_kernel_launches = c->prof_launches; stats->table_fallbacks = read_table_fallbacks(c); }
    return MCRAT_HIP_OK;
}

// The frame queue (launch.hpp, FrameQueueDev; kernels.hip, rank_loop_kernel): the pool's lists through several hydro frames in one launch.
extern "C" int mcrat_hip_pool_run_frames(mcrat_hip_ctx *c, const mcrat_hip_frame_plan *p, mcrat_hip_frame_stats *stats)
{
    if (!c || !p || !stats || p->n_frames <= 0 || !p->open || !p->seeds || !p->time_now || !p->remaining_time) return MCRAT_HIP_EINVAL;
    if (p->chain_clock && !p->frame_end) return MCRAT_HIP_EINVAL;
    if (c->selected_frame >= 0) { c->last_error = "a captured frame is selected (mcrat_hip_pool_select_frame(pool, -1) first)"; return MCRAT_HIP_ESTATE; }
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    if (!c->have_hydro) return MCRAT_HIP_ESTATE;
    if (c->cfg.cyclosynchrotron_switch) { c->last_error = "CYCLOSYNCHROTRON_SWITCH is on: its hook needs the host between passes, one frame per call (mcrat_hip_pool_scatter_frames_cyclosynch)"; return MCRAT_HIP_ESTATE; }
    if (c->cfg.tau_calculation == MCRAT_HIP_TAU_TABLE && !c->d_hot_table) {
        c->last_error = "TAU_CALCULATION == TABLE needs mcrat_hip_set_hot_cross_section first";
        return MCRAT_HIP_ESTATE;
    }
    // the staged hydro frames of the plan: index 0 the pool's own, then every other context named in plan->hydro (it keeps its frame staged while the
    // call runs; same switches as the pool -- its HydroDev is handed to the pool's kernels as it is)
    std::vector<HydroDev> hyv(1, c->hy);
    std::vector<const mcrat_hip_ctx *> hy_ctx(1, c);
    std::vector<int> hy_of_frame((size_t)p->n_frames, 0);
    if (p->hydro)
        for (int f = 0; f < p->n_frames; ++f) {
            const mcrat_hip_ctx *o = p->hydro[f];
            if (!o || o == c) continue;
            if (!o->have_hydro) { c->last_error = "pool_run_frames: a context named in plan->hydro holds no staged frame"; return MCRAT_HIP_ESTATE; }
            if (o->kc.dimensions != c->kc.dimensions || o->kc.geometry != c->kc.geometry || o->kc.table != c->kc.table || o->cfg.device != c->cfg.device) {
                c->last_error = "pool_run_frames: a frame staged on a context with other switches (DIMENSIONS, GEOMETRY, TAU_CALCULATION) or on another device";
                return MCRAT_HIP_EINVAL;
            }
            if (c->kc.table && !o->hy.hot_table) { c->last_error = "pool_run_frames: TAU_CALCULATION == TABLE and a frame's context has no cross-section table"; return MCRAT_HIP_ESTATE; }
            size_t k = 1;
            while (k < hy_ctx.size() && hy_ctx[k] != o) ++k;
            if (k == hy_ctx.size()) { hy_ctx.push_back(o); hyv.push_back(o->hy); }
            hy_of_frame[(size_t)f] = (int)k;
        }
    const int R = c->n_ranks, F = p->n_frames;
    const size_t N = (size_t)R * (size_t)F;
    if (N > 0x7fffffffull) return MCRAT_HIP_EINVAL;
    // MCRAT_HIP_QUEUE_TIMING=1: where the call's host time goes, on stderr (development aid)
    static const bool timing = getenv("MCRAT_HIP_QUEUE_TIMING") && atoi(getenv("MCRAT_HIP_QUEUE_TIMING")) != 0;
    auto clock_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tq0 = clock_us();
    double tq_filled = 0, tq_launched = 0, tq_synced = 0, tq_a = 0, tq_b = 0, tq_c = 0;
    // the lists that take part, each in one run of consecutive frames
    std::vector<int> first((size_t)R, -1), last((size_t)R, -1);
    for (int f = 0; f < F; ++f) {
        const int *open_f = p->open + (size_t)f * R;
        for (int r = 0; r < R; ++r) {
            if (!open_f[r]) continue;
            if (first[r] < 0) first[r] = f;
            else if (last[r] != f - 1) { c->last_error = "pool_run_frames: a list's open frames must be consecutive"; return MCRAT_HIP_EINVAL; }
            last[r] = f;
        }
    }
    for (int r = 0; r < R; ++r) {
        const mcrat_hip_ctx *v = c->views[(size_t)r];
        if (first[r] >= 0 && (!v || !v->have_photons)) { c->last_error = "pool_run_frames: a list that does not exist was asked to open a frame"; return MCRAT_HIP_ESTATE; }
    }
    if (p->restore_each_frame) {
        if (!c->ph_snap || c->ph_snap_bytes < c->ph_bytes) { c->last_error = "pool_run_frames: restore_each_frame without a snapshot (mcrat_hip_snapshot_photons)"; return MCRAT_HIP_ESTATE; }
        for (int r = 0; r < R; ++r)
            if (first[r] >= 0 && ((size_t)r >= c->snap_lens.size() || c->snap_lens[(size_t)r] != c->views[(size_t)r]->ph.n)) {
                c->last_error = "pool_run_frames: a list has changed length since the snapshot";
                return MCRAT_HIP_ESTATE;
            }
    }
    for (int r = 0; r < R; ++r) {
        const mcrat_hip_ctx *v = c->views[(size_t)r];
        RankDesc d{};
        if (first[r] >= 0) { d.len = v->ph.n; d.stream = v->key.stream; d.seed = p->seeds[(size_t)first[r] * R + r]; }
        c->h_desc[r] = d;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    // the queue's block: [ticket | frames_done R | order N | items N] (uploaded per launch) then the records (read back)
    const size_t off_done = 64 * FRAME_QUEUE_XCDS, off_order = align_up(off_done + sizeof(unsigned) * (size_t)R, 64), off_items = align_up(off_order + sizeof(int) * N, 64);
    const size_t off_hy = align_up(off_items + sizeof(FrameItem) * N, 256);
    const size_t off_rec = align_up(off_hy + sizeof(HydroDev) * hyv.size(), 256), bytes = off_rec + sizeof(LoopState) * N;
    if (c->fq_bytes < bytes) {
        if (c->d_fq) { HIPCHK(c, hipFree(c->d_fq)); c->d_fq = nullptr; }
        if (c->h_fq) { HIPCHK(c, hipHostFree(c->h_fq)); c->h_fq = nullptr; }
        c->fq_bytes = 0;
        HIPCHK(c, hipMalloc(&c->d_fq, bytes));
        HIPCHK(c, hipHostMalloc(&c->h_fq, bytes, hipHostMallocDefault));
        c->fq_bytes = bytes;
    }
    tq_a = clock_us();
    char *hb = static_cast<char *>(c->h_fq), *db = static_cast<char *>(c->d_fq);
    unsigned *h_done = reinterpret_cast<unsigned *>(hb + off_done);
    int *h_order = reinterpret_cast<int *>(hb + off_order);
    FrameItem *h_items = reinterpret_cast<FrameItem *>(hb + off_items);
    LoopState *h_rec = reinterpret_cast<LoopState *>(hb + off_rec);
    memset(hb, 0, off_rec);
    for (size_t t = 0; t < N; ++t) {
        FrameItem &it = h_items[t];
        it.seed = p->seeds[t]; it.time_now = p->time_now[t]; it.remaining_time = p->remaining_time[t];
        it.frame_end = p->frame_end ? p->frame_end[t] : 0.0;
        it.open = p->open[t] ? 1 : 0;
        it.hydro = hy_of_frame[t / (size_t)R];
    }
    memcpy(hb + off_hy, hyv.data(), sizeof(HydroDev) * hyv.size());
    tq_b = clock_us();
    HIPCHK(c, hipMemsetAsync(db + off_rec, 0, sizeof(LoopState) * N, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_table_fallbacks, 0, sizeof(int), c->stream));
    if (!c->rank_block_fixed) { choose_rank_block(c); c->rank_block_fixed = true; }
    tq_c = clock_us();
    FrameQueueDev fq{};
    fq.n_frames = F; fq.restore = p->restore_each_frame ? 1 : 0; fq.chain_clock = p->chain_clock ? 1 : 0;
    fq.ticket = reinterpret_cast<unsigned *>(db); fq.frames_done = reinterpret_cast<unsigned *>(db + off_done);
    fq.order = reinterpret_cast<const int *>(db + off_order); fq.items = reinterpret_cast<const FrameItem *>(db + off_items);
    fq.records = reinterpret_cast<LoopState *>(db + off_rec);
    fq.hydro = reinterpret_cast<const HydroDev *>(db + off_hy);
    fq.snap_delta = p->restore_each_frame ? (long long)(static_cast<char *>(c->ph_snap) - static_cast<char *>(c->ph_buf)) : 0;
    c->cap_frames = 0;
    if (p->capture_frames && F > 1) {                        // the lists as every frame but the last leaves them (the frame's outputs: mcrat.c:881-915)
        const size_t need = c->ph_bytes * (size_t)(F - 1);
        if (c->ph_cap_bytes < need) {
            if (c->ph_cap) { HIPCHK(c, hipFree(c->ph_cap)); c->ph_cap = nullptr; c->ph_cap_bytes = 0; }
            HIPCHK(c, hipMalloc(&c->ph_cap, need));
            c->ph_cap_bytes = need;
        }
        fq.capture_delta = (long long)(static_cast<char *>(c->ph_cap) - static_cast<char *>(c->ph_buf));
        fq.capture_stride = (long long)c->ph_bytes;
        c->cap_frames = F - 1;
        // a capture holds the slots [0, list_capacity) of the lists that were open in its frame; every other slot must read as the empty slot it is in
        // the live lists (printPhotons' compaction keeps weight != 0 over the whole pool): the weight column starts from zero
        for (int f = 0; f < F - 1; ++f)
            HIPCHK(c, hipMemsetAsync(reinterpret_cast<char *>(c->ph.weight) + fq.capture_delta + (long long)f * fq.capture_stride, 0,
                                     sizeof(double) * (size_t)c->ph.n, c->stream));
    }
    long long per_frame_cap = 32768;             // passes one list may take per frame and launch (run_ranks' bound on a launch's duration)
    if (const char *e = getenv("MCRAT_HIP_RANK_LAUNCH_CAP")) per_frame_cap = atoll(e) > 0 ? atoll(e) : per_frame_cap;
    const int longest = longest_rank_list(c);
    c->prof_step_ms = 0; c->prof_launches = 0;
    // Which XCD a list belongs to: list r to XCD r % 8, where one launch per frame puts it too.  (Measured against contiguous eighths of the lists --
    // neighbouring lists hold photons of neighbouring cells, so an XCD's L2 would have an eighth of the cells to hold: 0.567 against 0.52 ms per frame
    // on the benchmark frame, the eighths differ in optical depth and the launch ends with the slowest XCD.)
    auto list_class = [&](int r) { return r % FRAME_QUEUE_XCDS; };
    int xcd_of_class[FRAME_QUEUE_XCDS];                      // which XCD's queue the lists of class k are in (identity unless an XCD turned out to start no workgroups)
    for (int x = 0; x < FRAME_QUEUE_XCDS; ++x) xcd_of_class[x] = x;
    std::vector<unsigned> tickets((size_t)FRAME_QUEUE_XCDS * FRAME_TICKET_STRIDE);
    bool one_by_one = getenv("MCRAT_HIP_NO_FRAME_QUEUE") && atoi(getenv("MCRAT_HIP_NO_FRAME_QUEUE")) != 0;      // (A/B: the plan frame by frame)
    for (int attempt = 0; !one_by_one; ++attempt) {
        // the open items in the order they are taken: per XCD (list r belongs to XCD r % 8: a list never changes L2) frame-major; one workgroup per item,
        // and as the hardware deals workgroups round-robin over the XCDs, eight times the longest XCD's list of them
        int n_open = 0, longest_xcd = 0;
        {
            int count[FRAME_QUEUE_XCDS] = {0}, fill[FRAME_QUEUE_XCDS];
            for (int f = 0; f < F; ++f)
                for (int r = 0; r < R; ++r)
                    if (h_items[(size_t)f * R + r].open) count[xcd_of_class[list_class(r)]] += 1;
            for (int x = 0; x < FRAME_QUEUE_XCDS; ++x) {
                fq.order_off[x] = fill[x] = n_open;
                n_open += count[x];
                longest_xcd = std::max(longest_xcd, count[x]);
            }
            for (int f = 0; f < F; ++f)                      // frame-major within an XCD's queue
                for (int r = 0; r < R; ++r)
                    if (h_items[(size_t)f * R + r].open) h_order[fill[xcd_of_class[list_class(r)]]++] = f * R + r;
        }
        fq.order_off[FRAME_QUEUE_XCDS] = n_open;
        const int n_groups = FRAME_QUEUE_XCDS * longest_xcd;
        if (n_groups == 0) break;                            // (no list opens a frame: the call has only sized the queue's buffers)
        tq_filled = clock_us();
        HIPCHK(c, hipMemcpyAsync(db, hb, off_rec, hipMemcpyHostToDevice, c->stream));
        if (c->cfg.profile) {
            int rc = ensure_events(c, 2);
            if (rc) return rc;
            HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
        }
        {
            const hipError_t le = launch_rank_loop(c->kc, c->ph, c->hy, c->d_rstates, c->key, R, c->rank_stride, longest, c->d_desc, nullptr, nullptr, per_frame_cap,
                                                   c->rank_block + (c->rank_fuse ? 1000 : 0), c->stream, &fq, n_groups);
            if (le == hipErrorNotSupported && attempt == 0) { one_by_one = true; break; }      // no queue build of this launch form (kernels.hip)
            HIPCHK(c, le);
        }
        if (c->cfg.profile) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        HIPCHK(c, hipMemcpyAsync(tickets.data(), db, sizeof(unsigned) * tickets.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_done, db + off_done, sizeof(unsigned) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h_rec, db + off_rec, sizeof(LoopState) * N, hipMemcpyDeviceToHost, c->stream));
        tq_launched = clock_us();
        HIPCHK(c, hipStreamSynchronize(c->stream));
        tq_synced = clock_us();
        if (c->cfg.profile) {
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            c->prof_step_ms += ms;
            c->prof_launches += 1;
        }
        bool all = true;
        for (int r = 0; r < R; ++r) {
            if (h_done[r] & FRAME_STALLED) h_done[r] &= ~FRAME_STALLED;            // (the frame that ran into the pass limit = the frames before it are through)
            all = all && (first[r] < 0 || (int)h_done[r] == last[r] + 1);
        }
        if (all) break;
        {   // a device whose workgroups report fewer XCDs than eight (another partition mode): the queues nobody drew from move to XCDs that exist
            int alive[FRAME_QUEUE_XCDS], n_alive = 0;
            for (int x = 0; x < FRAME_QUEUE_XCDS; ++x)
                if (tickets[(size_t)x * FRAME_TICKET_STRIDE] > 0) alive[n_alive++] = x;
            if (n_alive == 0) { c->last_error = "pool_run_frames: no workgroup drew an item"; return MCRAT_HIP_EHIP; }
            for (int k = 0; k < FRAME_QUEUE_XCDS; ++k)
                if (tickets[(size_t)xcd_of_class[k] * FRAME_TICKET_STRIDE] == 0) xcd_of_class[k] = alive[k % n_alive];
        }
        if (attempt >= 1 << 16) { c->last_error = "pool_run_frames: lists that make no progress"; return MCRAT_HIP_EHIP; }
        // Lists whose frame ran into the launch's pass limit (and their later frames, whose workgroups gave up): their finished frames leave the queue, the
        // frame in progress goes on from its LoopState (open = 2), the rest as planned -- with the clock the host now knows.
        memset(hb, 0, off_done);                                                   // ticket
        for (int r = 0; r < R; ++r) {
            if (first[r] < 0) continue;
            const int nf = std::max((int)h_done[r], first[r]);                     // the first frame that is not complete
            for (int f = first[r]; f <= last[r]; ++f) {
                FrameItem &it = h_items[(size_t)f * R + r];
                if (f < nf) { it.open = 0; continue; }
                if (f > nf) continue;
                const LoopState &rec = h_rec[(size_t)f * R + r];
                if (rec.iterations > 0 && !rec.done) it.open = 2;
                else if (p->chain_clock && f > first[r]) {                         // (its previous frame has left the queue: the clock it would have read there)
                    it.time_now = h_rec[(size_t)(f - 1) * R + r].time_now;
                    it.remaining_time = it.frame_end - it.time_now;
                }
            }
        }
    }
    if (one_by_one) {
        // The plan one launch per frame: launch forms without a queue build (128- and 512-thread lists, lists whose columns stay in HBM/L2) -- the same
        // frames, seeds and clocks, so the same photons; what is lost is only that a list need not wait for the others at a frame's end.
        std::vector<double> t_now((size_t)R, 0.0), t_rem((size_t)R, 0.0);
        std::vector<int> op((size_t)R, 0);
        const size_t sb = (sizeof(int) + 2 * sizeof(double)) * (size_t)R;
        int rc = ensure_aos(c, sb + 64);
        if (rc) return rc;
        double *d_t = static_cast<double *>(c->aos_buf), *d_rem = d_t + R;
        int *d_open = reinterpret_cast<int *>(d_rem + R);
        const long long snap = p->restore_each_frame ? (long long)(static_cast<char *>(c->ph_snap) - static_cast<char *>(c->ph_buf)) : 0;
        for (int f = 0; f < F; ++f) {
            bool any = false, all_lists = true;                               // (every list that exists takes part: the whole pool is restored in one copy)
            for (int r = 0; r < R; ++r)
                if (c->views[(size_t)r] && c->views[(size_t)r]->have_photons && !h_items[(size_t)f * R + r].open) all_lists = false;
            if (p->restore_each_frame && all_lists) HIPCHK(c, hipMemcpyAsync(c->ph_buf, c->ph_snap, c->ph_bytes, hipMemcpyDeviceToDevice, c->stream));
            for (int r = 0; r < R; ++r) {
                const size_t t = (size_t)f * R + r;
                op[(size_t)r] = h_items[t].open ? 1 : 0;
                if (!op[(size_t)r]) { c->h_desc[r].len = 0; continue; }
                any = true;
                const mcrat_hip_ctx *v = c->views[(size_t)r];
                c->h_desc[r].len = v->ph.n; c->h_desc[r].stream = v->key.stream; c->h_desc[r].seed = h_items[t].seed;
                t_now[(size_t)r] = h_items[t].time_now; t_rem[(size_t)r] = h_items[t].remaining_time;
                if (p->chain_clock && f > first[r]) { t_now[(size_t)r] = h_rec[t - R].time_now; t_rem[(size_t)r] = h_items[t].frame_end - t_now[(size_t)r]; }
                if (p->restore_each_frame && !all_lists) {                    // the list's window of every column back from the snapshot
                    const size_t b0 = (size_t)r * (size_t)c->rank_stride, len = (size_t)v->ph.n;
                    char *col0 = reinterpret_cast<char *>(c->ph.r0 + b0);
                    HIPCHK(c, hipMemcpy2DAsync(col0, sizeof(double) * c->ph.col_stride, col0 + snap, sizeof(double) * c->ph.col_stride, sizeof(double) * len,
                                               24, hipMemcpyDeviceToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->ph.idx + b0, reinterpret_cast<char *>(c->ph.idx + b0) + snap, sizeof(int) * len, hipMemcpyDeviceToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->ph.flags + b0, reinterpret_cast<char *>(c->ph.flags + b0) + snap, len, hipMemcpyDeviceToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(c->ph.type + b0, reinterpret_cast<char *>(c->ph.type + b0) + snap, len, hipMemcpyDeviceToDevice, c->stream));
                }
            }
            if (!any) continue;
            HIPCHK(c, hipMemcpyAsync(d_t, t_now.data(), sizeof(double) * (size_t)R, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_rem, t_rem.data(), sizeof(double) * (size_t)R, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_open, op.data(), sizeof(int) * (size_t)R, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, launch_init_states_multi(c->d_rstates, R, d_open, d_t, d_rem, c->stream));
            for (;;) {
                if (c->cfg.profile) { rc = ensure_events(c, 2); if (rc) return rc; HIPCHK(c, hipEventRecord(c->ev[0], c->stream)); }
                HIPCHK(c, launch_rank_loop(c->kc, c->ph, hyv[(size_t)hy_of_frame[(size_t)f]], c->d_rstates, c->key, R, c->rank_stride, longest, c->d_desc, nullptr, nullptr,
                                           per_frame_cap, c->rank_block + (c->rank_fuse ? 1000 : 0), c->stream));
                if (c->cfg.profile) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
                HIPCHK(c, hipMemcpyAsync(c->h_rstates, c->d_rstates, sizeof(LoopState) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (c->cfg.profile) { float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->prof_step_ms += ms; c->prof_launches += 1; }
                bool done = true;
                for (int r = 0; r < R && done; ++r) done = !op[(size_t)r] || c->h_rstates[r].done != 0;
                if (done) break;
            }
            for (int r = 0; r < R; ++r)
                if (op[(size_t)r]) h_rec[(size_t)f * R + r] = c->h_rstates[r];
            if (c->cap_frames > 0 && f < F - 1)               // the pool as frame f leaves it
                HIPCHK(c, hipMemcpyAsync(static_cast<char *>(c->ph_cap) + (size_t)f * c->ph_bytes, c->ph_buf, c->ph_bytes, hipMemcpyDeviceToDevice, c->stream));
        }
    }
    for (size_t t = 0; t < N; ++t) {
        const int r = (int)(t % (size_t)R);
        if (p->open[t]) state_to_stats(h_rec[t], c->views[(size_t)r]->ph.n, &stats[t]);
        else memset(&stats[t], 0, sizeof stats[t]);
    }
    // the pool as after the last frame's mcrat_hip_run
    long long it_sum = 0;
    int lists = 0;
    for (int r = 0; r < R; ++r) {
        if (first[r] < 0) continue;
        mcrat_hip_ctx *v = c->views[(size_t)r];
        const size_t t = (size_t)last[r] * R + r;
        if (v->graph_exec && v->key.seed != p->seeds[t]) drop_graph(v);
        v->key.seed = p->seeds[t];
        v->find_switch = 1;
        v->pending_applied = false;
        v->rank_current = true;
        v->frame_open = true;
        v->prof_step_ms = v->prof_event_ms = 0;
        v->prof_launches = 0;
        c->h_rstates[r] = h_rec[t];
        it_sum += h_rec[t].iterations;
        lists += 1;
    }
    if (lists > 0) { c->rank_passes_per_list = (double)it_sum / lists; c->frame_open = true; }
    c->rank_block_fixed = false;
    stats[0].step_kernel_ms = c->prof_step_ms;               // (profile = 1: the launch's duration, on the first item)
    stats[0].step_kernel_launches = c->prof_launches;
    if (timing)
        fprintf(stderr, "pool_run_frames: %d frames x %d lists: plan -> queue %.0f us (checks %.0f, items %.0f, memsets + block choice %.0f, order %.0f), upload + launch calls %.0f us, waiting for the device %.0f us (kernel %.0f us), records -> stats %.0f us\n",
                F, R, tq_filled - tq0, tq_a - tq0, tq_b - tq_a, tq_c - tq_b, tq_filled - tq_c, tq_launched - tq_filled, tq_synced - tq_launched, 1e3 * c->prof_step_ms, clock_us() - tq_synced);
    return MCRAT_HIP_OK;
}

// The pool's read entry points (mcrat_hip_pool_summaries, mcrat_hip_outbox_post, mcrat_hip_get_photons_range, mcrat_hip_get_output) on the lists as
// frame `frame` of the last plan left them: the captures are laid out like the live lists, so the pool's column pointers are simply moved over.
extern "C" int mcrat_hip_pool_select_frame(mcrat_hip_ctx *c, int frame)
{
    if (!c || !c->is_pool) return MCRAT_HIP_EINVAL;
    if (c->selected_frame >= 0) { c->ph = c->ph_live; c->selected_frame = -1; }
    if (frame < 0) return MCRAT_HIP_OK;
    if (frame >= c->cap_frames || !c->ph_cap) { c->last_error = "pool_select_frame: no capture of that frame (mcrat_hip_frame_plan.capture_frames; the last frame is the live lists)"; return MCRAT_HIP_ESTATE; }
    c->ph_live = c->ph;
    const long long delta = (static_cast<char *>(c->ph_cap) - static_cast<char *>(c->ph_buf)) + (long long)frame * (long long)c->ph_bytes;
    auto move = [&](auto *&ptr) { ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(reinterpret_cast<char *>(ptr) + delta); };
    PhotonDev &q = c->ph;
    move(q.r0); move(q.r1); move(q.r2); move(q.p0); move(q.p1); move(q.p2); move(q.p3); move(q.c0); move(q.c1); move(q.c2); move(q.c3);
    move(q.s0); move(q.s1); move(q.s2); move(q.s3); move(q.num_scatt); move(q.weight); move(q.tau); move(q.tts); move(q.u0); move(q.u1); move(q.u2);
    move(q.ntau); move(q.tau_next); move(q.idx); move(q.flags); move(q.type);
    c->selected_frame = frame;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_run(mcrat_hip_ctx *c, long long max_iterations, mcrat_hip_frame_stats *stats)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->selected_frame >= 0) { c->last_error = "a captured frame is selected (mcrat_hip_pool_select_frame(pool, -1) first)"; return MCRAT_HIP_ESTATE; }
    if (c->is_pool) {                              // every list whose view has opened a frame (or the pool's own begin_frame: all)
        if (!c->have_hydro) return MCRAT_HIP_ESTATE;
        if (c->cfg.cyclosynchrotron_switch) { c->last_error = "CYCLOSYNCHROTRON_SWITCH is on: run the views one by one (mcrat_hip_scatter_frame_cyclosynch)"; return MCRAT_HIP_ESTATE; }
        c->frame_open = true;
        return run_ranks(c, max_iterations, stats);
    }
    if (!c->frame_open) return MCRAT_HIP_ESTATE;
    if (c->sc_world > 0) { c->last_error = "shared clock attached: drive the frame with mcrat_hip_shared_clock_*"; return MCRAT_HIP_ESTATE; }
    if (c->cfg.cyclosynchrotron_switch) {
        c->last_error = "CYCLOSYNCHROTRON_SWITCH is on: drive the frame with mcrat_hip_scatter_frame_cyclosynch (the loop has the hook of mcrat.c:786-808)";
        return MCRAT_HIP_ESTATE;
    }
    if (c->n_ranks > 0) return run_ranks(c, max_iterations, stats);
    const int per_sync = c->cfg.iterations_per_sync;
    long long it = 0;
    int rc;
    while (max_iterations <= 0 || it < max_iterations) {
        int batch = per_sync;
        if (max_iterations > 0 && (long long)batch > max_iterations - it) batch = (int)(max_iterations - it);
        if (c->cfg.profile) {
            if ((rc = ensure_events(c, (size_t)3 * batch))) return rc;
            for (int b = 0; b < batch; ++b) {
                HIPCHK(c, hipEventRecord(c->ev[3 * b], c->stream));
                HIPCHK(c, launch_step(c->kc, c->find_switch != 0, c->ph, c->hy, c->d_state, c->key, c->partials, c->step_blocks, c->shortlist, c->stream));
                c->find_switch = 0;
                HIPCHK(c, hipEventRecord(c->ev[3 * b + 1], c->stream));
                HIPCHK(c, launch_event_of(c));
                HIPCHK(c, hipEventRecord(c->ev[3 * b + 2], c->stream));
            }
        } else {
            int b = 0;
            if (c->find_switch) {             // the forced-relocation pass is never part of the graph
                if ((rc = launch_iteration(c, true))) return rc;
                c->find_switch = 0;
                b = 1;
            }
            if (c->cfg.use_graph && b == 0 && batch == per_sync) {
                if ((rc = ensure_graph(c, batch))) return rc;
                HIPCHK(c, hipGraphLaunch(c->graph_exec, c->stream));
            } else {
                for (; b < batch; ++b)
                    if ((rc = launch_iteration(c, false))) return rc;
            }
        }
        it += batch;
        HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->cfg.profile) {
            // only launches that did work are counted: after `done` the kernels return at once
            const long long did = c->h_state->iterations - c->prof_launches;
            for (int b = 0; b < batch && b < did; ++b) {
                float ms = 0;
                HIPCHK(c, hipEventElapsedTime(&ms, c->ev[3 * b], c->ev[3 * b + 1]));
                c->prof_step_ms += ms;
                HIPCHK(c, hipEventElapsedTime(&ms, c->ev[3 * b + 1], c->ev[3 * b + 2]));
                c->prof_event_ms += ms;
            }
            c->prof_launches = c->h_state->iterations;
        }
        if (c->h_state->done) break;
    }
    if ((rc = flush_pending(c))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fill_stats(c, stats);
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_propagate_frame(mcrat_hip_ctx *c, double *time_now, double remaining_time, uint64_t seed,
                                         mcrat_hip_frame_stats *stats)
{
    if (!c || !time_now) return MCRAT_HIP_EINVAL;
    int rc = mcrat_hip_begin_frame(c, seed, *time_now, remaining_time);
    if (rc) return rc;
    mcrat_hip_frame_stats local;
    rc = mcrat_hip_run(c, 0, &local);
    if (rc) return rc;
    *time_now = local.time_now;
    if (stats) *stats = local;
    return MCRAT_HIP_OK;
}

static int fast_refusals(mcrat_hip_ctx *c)
{
    if (!c->have_hydro) return MCRAT_HIP_ESTATE;
    if (c->cfg.cyclosynchrotron_switch) { c->last_error = "FAST mode has no cyclo-synchrotron hooks: use the exact scatter frame"; return MCRAT_HIP_ESTATE; }
    if (c->sc_world > 0) { c->last_error = "shared clock attached: drive the frame with mcrat_hip_shared_clock_*"; return MCRAT_HIP_ESTATE; }
    if (c->cfg.tau_calculation == MCRAT_HIP_TAU_TABLE && !c->d_hot_table) {
        c->last_error = "TAU_CALCULATION == TABLE needs mcrat_hip_set_hot_cross_section first";
        return MCRAT_HIP_ESTATE;
    }
    return MCRAT_HIP_OK;
}

static void fast_stats(const FastCounts &fc, double time_now, mcrat_hip_frame_stats *stats)
{
    memset(stats, 0, sizeof *stats);
    stats->iterations = (long long)fc.passes;
    stats->photon_steps = (long long)fc.photon_steps;
    stats->slot_steps = (long long)fc.photon_steps;
    stats->frame_scatt_cnt = (long long)fc.scatterings;
    stats->kn_rejections = (long long)fc.kn_rejections;
    stats->num_photons_find_new_element = (long long)fc.relocated;
    stats->not_found = (long long)fc.not_found;
    stats->last_scattered_index = -1;
    stats->remaining_time = 0;
    stats->time_now = time_now;
}

// FAST mode's refresh cadence when the caller does not name one (fast_windows <= 0).  The exact loop re-locates a photon and redraws its free
// path whenever ANY photon of its rank scatters -- N_events(rank, frame) times per frame, for the reference's ranks of about 1000 photons that
// is the frame's scatterings per 1000 photons.  A fixed 8 is right for thin frames and biased by a per cent in dense ones (the Lundman run's first
// frames: 68.0 scatterings per photon against the exact mode's 67.3; 128 windows give 67.3).  So the cadence follows the frame: the scatterings per
// 1000 photons the LIST's last fast frame counted (fast_cadence_learn below), between 8 and 2048, 32 before any frame has run.
static int fast_cadence(const mcrat_hip_ctx *c, int fast_windows)
{
    return fast_windows > 0 ? fast_windows : c->fast_auto_windows;
}
// The cadence is a property of the LIST (kept on its context -- a pool's list: on its view): what a 10^3-photon rank of the exact mode would have
// refreshed in the frame this list has just seen -- its own scatterings per thousand of its own photons, between 8 and 2048 windows.  A list's
// photons therefore do not depend on which other lists share its pool, and mcrat_hip_fast_cadence carries the value over a restart.
static void fast_cadence_learn(mcrat_hip_ctx *c, unsigned long long scatterings, long long photons)
{
    if (photons <= 0) return;
    const double per_thousand = 1000.0 * (double)scatterings / (double)photons;
    // (the upper end: at 7 300 scatterings per thousand photons and frame 128 windows still count 0.14 % too many scatterings, 512 0.05 %, 2048 none --
    // tools/fast_cadence_gate.py, 1.9e7 events -- and 2048 windows are still faster than the exact loop there)
    c->fast_auto_windows = per_thousand < 8.0 ? 8 : (per_thousand > 2048.0 ? 2048 : (int)(per_thousand + 0.5));
}

extern "C" int mcrat_hip_fast_cadence(mcrat_hip_ctx *c, int set_windows)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (set_windows > 0) c->fast_auto_windows = set_windows < 8 ? 8 : (set_windows > 2048 ? 2048 : set_windows);
    return c->fast_auto_windows;
}

// MCRAT_HIP_MODE_FAST for the lists of a rank pool, one launch: list r (open[r] != 0) runs its frame of remaining_time[r] with its own seed
// and stream and list-local slot numbers in its keys -- bit for bit what mcrat_hip_propagate_frame_mode(view r, ..., seeds[r], FAST, ...) gives
extern "C" int mcrat_hip_pool_propagate_frames_fast(mcrat_hip_ctx *c, const int *open, const uint64_t *seeds, const double *time_now,
                                                    const double *remaining_time, int fast_windows, mcrat_hip_frame_stats *stats)
{
    if (!c || !open || !seeds || !time_now || !remaining_time) return MCRAT_HIP_EINVAL;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    int rc = fast_refusals(c);
    if (rc) return rc;
    const int R = c->n_ranks;
    std::vector<double> rem((size_t)R, 0.0);
    for (int r = 0; r < R; ++r) {
        mcrat_hip_ctx *v = c->views[r];
        c->h_desc[r] = RankDesc{};
        if (!open[r]) continue;
        if (!v || !v->have_photons) { c->last_error = "pool_propagate_frames_fast: a list that does not exist was asked to run a frame"; return MCRAT_HIP_ESTATE; }
        if ((rc = flush_pending(v))) return rc;
        c->h_desc[r].len = v->ph.n; c->h_desc[r].stream = v->key.stream; c->h_desc[r].seed = seeds[r];
        rem[(size_t)r] = remaining_time[r] > 0 ? remaining_time[r] : 0.0;
    }
    std::vector<int> win((size_t)R, 8);                      // every list its own cadence (fast_cadence_learn): as its view run alone would have
    for (int r = 0; r < R; ++r)
        if (open[r]) win[(size_t)r] = fast_cadence(c->views[r], fast_windows);
    const size_t bytes = (sizeof(FastCounts) + sizeof(double) + sizeof(int)) * (size_t)R;
    if ((rc = ensure_aos(c, bytes + 64))) return rc;
    FastCounts *d_cnt = static_cast<FastCounts *>(c->aos_buf);
    double *d_rem = reinterpret_cast<double *>(d_cnt + R);
    int *d_win = reinterpret_cast<int *>(d_rem + R);
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, sizeof(FastCounts) * (size_t)R, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_rem, rem.data(), sizeof(double) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_win, win.data(), sizeof(int) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    const FastLists lists{c->rank_stride, c->d_desc, d_rem, d_win};
    HIPCHK(c, launch_fast_frame(c->kc, c->ph, c->hy, c->key, 0.0, 8, 1 << 22, d_cnt, lists, c->stream));
    std::vector<FastCounts> cnt((size_t)R);
    HIPCHK(c, hipMemcpyAsync(cnt.data(), d_cnt, sizeof(FastCounts) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->frame_open = false;
    unsigned long long unfinished = 0, scatterings = 0;
    long long photons = 0;
    for (int r = 0; r < R; ++r) {
        if (!open[r]) continue;
        mcrat_hip_ctx *v = c->views[r];
        v->frame_open = false; v->pending_applied = false; v->rank_current = true;
        unfinished += cnt[(size_t)r].unfinished;
        scatterings += cnt[(size_t)r].scatterings; photons += v->ph.n;
        if (rem[(size_t)r] > 0) fast_cadence_learn(v, cnt[(size_t)r].scatterings, v->ph.n);
        if (stats) fast_stats(cnt[(size_t)r], time_now[r] + rem[(size_t)r], &stats[r]);
    }
    (void)scatterings; (void)photons;
    if (unfinished) { c->last_error = "FAST mode: photons left with frame time after 2^22 passes (an optical depth of infinity?)"; return MCRAT_HIP_ESTATE; }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_propagate_frame_mode(mcrat_hip_ctx *c, double *time_now, double remaining_time, uint64_t seed, int mode, int fast_windows,
                                              mcrat_hip_frame_stats *stats)
{
    if (!c || !time_now) return MCRAT_HIP_EINVAL;
    if (mode == MCRAT_HIP_MODE_EXACT) return mcrat_hip_propagate_frame(c, time_now, remaining_time, seed, stats);
    if (mode != MCRAT_HIP_MODE_FAST) return MCRAT_HIP_EINVAL;
    int rc = fast_refusals(c);
    if (rc) return rc;
    if (c->is_pool) {                                                  // every list that holds photons, the same seed and frame time
        const int R = c->n_ranks;
        std::vector<int> open((size_t)R, 0);
        std::vector<uint64_t> seeds((size_t)R, seed);
        std::vector<double> t((size_t)R, *time_now), rem((size_t)R, remaining_time);
        std::vector<mcrat_hip_frame_stats> per((size_t)R);
        bool any = false;
        for (int r = 0; r < R; ++r) { open[(size_t)r] = c->views[r] && c->views[r]->have_photons; any = any || open[(size_t)r]; }
        if (!any) return MCRAT_HIP_ESTATE;
        if ((rc = mcrat_hip_pool_propagate_frames_fast(c, open.data(), seeds.data(), t.data(), rem.data(), fast_windows, per.data()))) return rc;
        if (remaining_time > 0) *time_now += remaining_time;
        if (stats) {
            FastCounts sum{};
            for (int r = 0; r < R; ++r) {
                if (!open[(size_t)r]) continue;
                const mcrat_hip_frame_stats &q = per[(size_t)r];
                sum.passes = std::max<unsigned long long>(sum.passes, (unsigned long long)q.iterations);
                sum.photon_steps += (unsigned long long)q.photon_steps; sum.scatterings += (unsigned long long)q.frame_scatt_cnt;
                sum.kn_rejections += (unsigned long long)q.kn_rejections; sum.relocated += (unsigned long long)q.num_photons_find_new_element;
                sum.not_found += (unsigned long long)q.not_found;
            }
            fast_stats(sum, *time_now, stats);
        }
        return MCRAT_HIP_OK;
    }
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    if (c->frame_open && c->n_ranks == 0 && (rc = flush_pending(c))) return rc;       // what a list-mode run still owes the photons
    if (!c->d_fast) HIPCHK(c, hipMalloc(&c->d_fast, sizeof(FastCounts)));
    HIPCHK(c, hipMemsetAsync(c->d_fast, 0, sizeof(FastCounts), c->stream));
    RngKey key = c->key;
    key.seed = seed;
    FastCounts fc{};
    if (remaining_time > 0) {
        HIPCHK(c, launch_fast_frame(c->kc, c->ph, c->hy, key, remaining_time, fast_cadence(c, fast_windows), 1 << 22,
                                    static_cast<FastCounts *>(c->d_fast), FastLists{0, nullptr, nullptr, nullptr}, c->stream));
        HIPCHK(c, hipMemcpyAsync(&fc, c->d_fast, sizeof fc, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (remaining_time > 0) fast_cadence_learn(c, fc.scatterings, c->ph.n);
    c->frame_open = false;
    c->pending_applied = false;
    if (c->parent) c->rank_current = true;
    if (fc.unfinished) { c->last_error = "FAST mode: photons left with frame time after 2^22 passes (an optical depth of infinity?)"; return MCRAT_HIP_ESTATE; }
    if (remaining_time > 0) *time_now += remaining_time;
    if (stats) fast_stats(fc, *time_now, stats);
    return MCRAT_HIP_OK;
}

// The scatter-frame body of main() with CYCLOSYNCHROTRON_SWITCH on (mcrat.c:706-878, between getHydroData and saveCheckpoint): pool
// emission, the loop with the replacement of scattered pool photons (:786-795) and the rebinning trigger (:797-808), the rebinning
// and absorption at the end of the frame (:853-878).  The hook needs the list current after every pass, so a pass here is
// step + event + flush + cs_replace; the hook keeps the frame's counters on the device and parks the loop when it needs the host
// (list growth, the rebinning), so the host reads back once per batch of passes (CsFrame, launch.hpp).
extern "C" int mcrat_hip_scatter_frame_cyclosynch(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, double *time_now, double remaining_time, uint64_t seed,
                                                  double r_inj, double ph_weight_suggest, int max_photons, double theta_min, double theta_max, double fps,
                                                  int emit_pool, long long max_iterations, mcrat_hip_frame_stats *stats, mcrat_hip_cyclosynch_counts *cnt)
{
    if (!c || !cs || !time_now || !cnt || cs->b_field_calc < 0 || cs->b_field_calc > 2 || max_photons <= 0) return MCRAT_HIP_EINVAL;
    if (!c->cfg.cyclosynchrotron_switch) { c->last_error = "the context was created with cyclosynchrotron_switch = 0"; return MCRAT_HIP_ESTATE; }
    if (c->n_ranks > 0 || c->sc_world > 0) return MCRAT_HIP_ESTATE;
    if (!c->have_hydro || !c->hcol_buf || !c->have_photons) return MCRAT_HIP_ESTATE;
    const int carried = cnt->scatt_cyclosynch_num_ph;        // main()'s counter lives across the scatter frames of an injection (mcrat.c:873,921)
    memset(cnt, 0, sizeof *cnt);
    int rc;
    if (emit_pool) {                                                                              // :727-744
        int n = 0, bad = 0;
        double w = 0;
        if ((rc = mcrat_hip_emit_cyclosynch_pool(c, cs, r_inj, ph_weight_suggest, max_photons, theta_min, theta_max, fps, seed, &n, &w, &bad))) return rc;
        cnt->num_cyclosynch_ph_emit = n;
        cnt->pool_weight = w;
        cnt->integrals_not_converged = bad;
    }
    if ((rc = mcrat_hip_begin_frame(c, seed, *time_now, remaining_time))) return rc;
    CsEmitParams p{};
    p.dimensions = c->kc.dimensions; p.geometry = c->kc.geometry; p.b_field_calc = cs->b_field_calc; p.epsilon_b = cs->epsilon_b;
    if (!c->d_cs_hook) HIPCHK(c, hipMalloc((void **)&c->d_cs_hook, sizeof(CsFrame)));
    CsFrame *d_cf = static_cast<CsFrame *>(c->d_cs_hook);
    CsFrame cf{};
    cf.max_photons = max_photons;
    cf.last_iteration = ~0ull;
    cf.scatt_num = carried;
    HIPCHK(c, hipMemcpyAsync(d_cf, &cf, sizeof cf, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                                                  // cf is on the stack
    const int emit_pool_count = cnt->num_cyclosynch_ph_emit;
    int emit_base = emit_pool_count;               // num_cyclosynch_ph_emit = emit_base + replacements since (the rebinning overwrites it)
    const long long per_sync = 32;                 // passes queued per read-back; a parked loop makes the rest of a batch no-ops
    while (!c->h_state->done && (max_iterations <= 0 || c->h_state->iterations < max_iterations)) {   // :761-851
        long long batch = per_sync;
        if (max_iterations > 0 && batch > max_iterations - c->h_state->iterations) batch = max_iterations - c->h_state->iterations;
        auto one_pass = [&](bool force) -> int {
            HIPCHK(c, launch_step(c->kc, force, c->ph, c->hy, c->d_state, c->key, c->partials, c->step_blocks, c->shortlist, c->stream));
            HIPCHK(c, launch_event(c->kc, c->ph, c->hy, c->d_state, c->key, c->partials, c->step_blocks, c->shortlist, c->stream));
            HIPCHK(c, launch_flush(c->ph, c->d_state, c->step_blocks, c->stream));
            HIPCHK(c, launch_cs_replace(p, c->hy, c->hcol, c->key, c->d_state, c->ph, d_cf, 0, c->stream));
            return MCRAT_HIP_OK;
        };
        for (long long b = 0; b < batch; ++b) {       // (a batch as one hipGraph launch was measured: 33.7 against 34.6 us per pass, not kept)
            if ((rc = one_pass(c->find_switch != 0))) return rc;
            c->find_switch = 0;
        }
        HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&cf, d_cf, sizeof cf, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        while (cf.halt) {                          // the loop is parked: do what the hook asked for, then let it go on
            const int why = cf.halt;
            cf.halt = 0;
            c->h_state->done = cf.saved_done;
            HIPCHK(c, hipMemcpyAsync(reinterpret_cast<char *>(c->d_state) + offsetof(LoopState, done), &cf.saved_done, sizeof(int), hipMemcpyHostToDevice,
                                     c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));          // cf lives on the stack and changes below: the copy must have read it
            if (why == CS_HALT_GROW) {                                                            // photons.c:112-121: the list doubles
                if (c->ph.n > 0x3fffffff) { c->last_error = "photon list too long to double"; return MCRAT_HIP_ENOMEM; }
                HIPCHK(c, hipMemcpyAsync(d_cf, &cf, sizeof cf, hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if ((rc = grow_photons(c, 2 * c->ph.n))) return rc;
                HIPCHK(c, launch_cs_replace(p, c->hy, c->hcol, c->key, c->d_state, c->ph, d_cf, 1, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(&cf, d_cf, sizeof cf, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (cf.halt == CS_HALT_GROW) { c->last_error = "the replacement of a scattered pool photon failed after the list was doubled"; return MCRAT_HIP_ENOMEM; }
            } else {                                                                              // :797-808
                int empty = 0, emit_total = emit_base + cf.emitted, scatt = cf.scatt_num;
                rc = mcrat_hip_rebin_cyclosynch(c, cs, max_photons, &empty, &emit_total, &scatt);
                if (rc == MCRAT_HIP_OK) {
                    cnt->rebins += 1;
                    emit_base = emit_total;
                    cf.emitted = 0;
                    cf.scatt_num = scatt;
                } else if (rc != MCRAT_HIP_EREFUSED) {
                    return rc;                     // EREFUSED: one of the reference's refusals, the list is as it was
                }
                HIPCHK(c, hipMemcpyAsync(d_cf, &cf, sizeof cf, hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
        }
    }
    cnt->num_cyclosynch_ph_emit = emit_base + cf.emitted;
    cnt->scatt_cyclosynch_num_ph = cf.scatt_num;
    cnt->n_comptonized = cf.n_comptonized;
    c->pending_applied = false;
    if (emit_pool) {                                                                              // :853-878
        if (cnt->scatt_cyclosynch_num_ph > max_photons) {
            int empty = 0;
            rc = mcrat_hip_rebin_cyclosynch(c, cs, max_photons, &empty, &cnt->num_cyclosynch_ph_emit, &cnt->scatt_cyclosynch_num_ph);
            if (rc == MCRAT_HIP_OK) cnt->rebins += 1;
            else if (rc != MCRAT_HIP_EREFUSED) return rc;
        }
        if (cnt->num_cyclosynch_ph_emit > 0) {
            double w = 0;
            if ((rc = mcrat_hip_absorb_cyclosynch(c, cs, &cnt->frame_abs_cnt, &cnt->scatt_cyclosynch_num_ph, &w))) return rc;
            cnt->n_comptonized -= w;
        }
    }
    HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *time_now = c->h_state->time_now;
    fill_stats(c, stats);
    return MCRAT_HIP_OK;
}

// photonEmitCyclosynch (mc_cyclosynch.c:1219-1460, inject_single_switch = 0) for the lists of a rank pool that ask for it: the emission
// shell of each group of lists that share one (same radii and angles) is found and integrated once, then one workgroup per list runs the
// list's weight loop and places its photons (inject.hip).  A list with more photons than that kernel's tables hold takes the one-list
// path (mcrat_hip_emit_cyclosynch_pool on its view); both give the same photons.
static int pool_emit_cyclosynch(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, const std::vector<mcrat_hip_cyclosynch> &csr, int max_photons, double fps,
                                const mcrat_hip_pool_cs_list *lists, mcrat_hip_cyclosynch_counts *counts, std::vector<int> &emit_base)
{
    const int R = c->n_ranks, M = c->hy.M;
    struct Shell { double rmin, rmax, tmin, tmax; };
    std::vector<Shell> shells;
    std::vector<CsPoolEmit> pe((size_t)R);
    bool any = false;
    int rc;
    for (int r = 0; r < R; ++r) {
        pe[(size_t)r] = CsPoolEmit{};
        c->h_desc[r] = RankDesc{};
        if (!lists[r].open || !lists[r].emit_pool) continue;
        if (!(lists[r].ph_weight_suggest > 0)) return MCRAT_HIP_EINVAL;
        mcrat_hip_ctx *v = c->views[r];
        if ((rc = flush_pending(v))) return rc;
        const mcrat_hip_cyclosynch &q = csr[(size_t)r];
        Shell sh;
        sh.rmin = lists[r].r_inj + (C_LIGHT * (q.scatt_frame_number - q.inj_frame_number) / fps - 0.5 * C_LIGHT / fps);    // calcCyclosynchRLimits :225-244
        sh.rmax = lists[r].r_inj + (C_LIGHT * (q.scatt_frame_number - q.inj_frame_number) / fps + 0.5 * C_LIGHT / fps);
        sh.tmin = lists[r].theta_min; sh.tmax = lists[r].theta_max;
        int g = -1;
        for (size_t k = 0; k < shells.size(); ++k)
            if (shells[k].rmin == sh.rmin && shells[k].rmax == sh.rmax && shells[k].tmin == sh.tmin && shells[k].tmax == sh.tmax) { g = (int)k; break; }
        if (g < 0) { g = (int)shells.size(); shells.push_back(sh); }
        CsPoolEmit &e = pe[(size_t)r];
        e.open = 1; e.group = g; e.seed = lists[r].seed; e.weight_in = lists[r].ph_weight_suggest; e.max_photons = cs->rebin_e_perc * max_photons;
        c->h_desc[r].len = v->ph.n; c->h_desc[r].stream = v->key.stream; c->h_desc[r].seed = lists[r].seed;
        any = true;
    }
    if (!any) return MCRAT_HIP_OK;
    if ((rc = ensure_counts(c, (size_t)M + 8))) return rc;
    const size_t scan_ints = (size_t)M + 1 + grid_scan_scratch_ints(M);
    int *d_start = nullptr;
    CsPoolEmit *d_pe = nullptr;
    CsShellCell *d_shell = nullptr;
    unsigned *d_bad = nullptr;
    size_t shell_cap = 0;
    auto done = [&](int code) { (void)hipFree(d_start); (void)hipFree(d_pe); (void)hipFree(d_shell); (void)hipFree(d_bad); return code; };
    if (hipMalloc((void **)&d_start, sizeof(int) * scan_ints) != hipSuccess || hipMalloc((void **)&d_pe, sizeof(CsPoolEmit) * (size_t)R) != hipSuccess ||
        hipMalloc((void **)&d_bad, sizeof(unsigned)) != hipSuccess) return done(MCRAT_HIP_ENOMEM);
    if (hipMemcpyAsync(d_pe, pe.data(), sizeof(CsPoolEmit) * (size_t)R, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyHostToDevice, c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
    CsEmitParams p{};
    p.dimensions = c->kc.dimensions; p.geometry = c->kc.geometry; p.b_field_calc = cs->b_field_calc; p.epsilon_b = cs->epsilon_b;
    std::vector<unsigned> bad(shells.size(), 0);
    for (size_t g = 0; g < shells.size(); ++g) {
        p.rmin = shells[g].rmin; p.rmax = shells[g].rmax; p.theta_min = shells[g].tmin; p.theta_max = shells[g].tmax;
        int n_shell = 0;
        if (launch_cs_shell_flag(p, c->hy, c->grid_count, c->d_grid_total, &n_shell, c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
        if ((size_t)n_shell > shell_cap) {
            (void)hipFree(d_shell); d_shell = nullptr;
            if (hipMalloc((void **)&d_shell, sizeof(CsShellCell) * (size_t)n_shell) != hipSuccess) return done(MCRAT_HIP_ENOMEM);
            shell_cap = (size_t)n_shell;
        }
        if (n_shell > 0 && launch_cs_shell_write(p, c->hy, c->hcol, c->grid_count, n_shell, d_start, d_start + M + 1, d_shell, d_bad, c->stream) != hipSuccess)
            return done(MCRAT_HIP_EHIP);
        if (n_shell == 0 && hipMemsetAsync(d_bad, 0, sizeof(unsigned), c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
        if (launch_cs_emit_pool(p, c->hy, c->hcol, c->ph, c->rank_stride, R, c->d_desc, d_shell, n_shell, d_pe, (int)g, c->stream) != hipSuccess ||
            hipMemcpyAsync(&bad[g], d_bad, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
            hipStreamSynchronize(c->stream) != hipSuccess) { c->last_error = "cyclo-synchrotron emission of the pool's lists failed"; return done(MCRAT_HIP_EHIP); }
    }
    if (hipMemcpyAsync(pe.data(), d_pe, sizeof(CsPoolEmit) * (size_t)R, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(c->h_desc, c->d_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return done(MCRAT_HIP_EHIP);
    (void)done(0);
    for (int r = 0; r < R; ++r) {
        const CsPoolEmit &e = pe[(size_t)r];
        if (!e.open) continue;
        mcrat_hip_ctx *v = c->views[r];
        int n = e.n_emit, nbad = (int)bad[(size_t)e.group];
        double w = e.weight_out;
        if (nbad > 0) {       // (as mcrat_hip_emit_cyclosynch_pool: inject.hip, qags_planck ran out of intervals)
            c->last_error = "cyclo-synchrotron emission: the photon-density integral of " + std::to_string(nbad) + " cell(s) did not converge within the device's interval limit";
            return MCRAT_HIP_EREFUSED;
        }
        switch (e.error) {
        case 0:
            v->ph.n = c->h_desc[r].len;                                                           // the list may have grown inside its window
            if (n > 0) { v->frame_open = false; drop_graph(v); }
            break;
        case 4:                                                                                   // more photons than the kernel's tables hold
            if ((rc = mcrat_hip_emit_cyclosynch_pool(v, &csr[(size_t)r], lists[r].r_inj, lists[r].ph_weight_suggest, max_photons, lists[r].theta_min,
                                                     lists[r].theta_max, fps, lists[r].seed, &n, &w, &nbad))) { c->last_error = v->last_error; return rc; }
            break;
        case 1:
            c->last_error = "cyclo-synchrotron emission: no weight gives between 1 and rebin_e_perc * maximum_photons photons";
            return MCRAT_HIP_EINVAL;
        case 2:
            c->last_error = "cyclo-synchrotron emission: fewer null slots than photons to add (the reference exits with \"Adding to the photon list has failed\")";
            return MCRAT_HIP_EINVAL;
        default:
            c->last_error = "the list is longer than the pool's slots per rank (mcrat_hip_pool_create)";
            return MCRAT_HIP_ENOMEM;
        }
        counts[r].num_cyclosynch_ph_emit = n;
        counts[r].pool_weight = w;
        counts[r].integrals_not_converged = nbad;
        emit_base[(size_t)r] = n;
    }
    return MCRAT_HIP_OK;
}

// the rebinning of the pool's lists `ids`: all at once (pool_rebin_lists), or -- MCRAT_HIP_POOL_REBIN_EACH=1, the round-2 path kept for the A/B and
// the equality test -- list by list through the views
static int pool_rebin(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, int max_photons, const std::vector<int> &ids, std::vector<PoolRebinResult> &res)
{
    const char *e = getenv("MCRAT_HIP_POOL_REBIN_EACH");
    if (!(e && atoi(e) != 0)) return pool_rebin_lists(c, cs, max_photons, ids, res);
    res.assign(ids.size(), PoolRebinResult{MCRAT_HIP_OK, 0, 0, 0});
    for (size_t j = 0; j < ids.size(); ++j) {
        mcrat_hip_ctx *v = c->views[ids[j]];
        const int rc = mcrat_hip_rebin_cyclosynch(v, cs, max_photons, &res[j].empty_bins, &res[j].num_cyclosynch_ph_emit, &res[j].scatt_cyclosynch_num_ph);
        res[j].rc = rc;
        if (rc != MCRAT_HIP_OK && rc != MCRAT_HIP_EREFUSED) { c->last_error = v->last_error; return rc; }
    }
    return MCRAT_HIP_OK;
}

// The scatter frame of mcrat.c:706-878 with CYCLOSYNCHROTRON_SWITCH on for the lists of a rank pool: what mcrat_hip_scatter_frame_cyclosynch
// does for one list, for every open list -- the loop of all of them in the same launches.  rank_loop_kernel (its CSH build) runs every list
// and, after a pass the hook of :786-808 must look at (photonEvent reported a pool photon; a thousand scatterings are full), the hook itself:
// it converts and replaces the pool photon (doubling the list inside its window of the pool when it has no null slot left) and evaluates
// the rebinning trigger; the list goes on in the same launch.  Only the rebinning itself parks a list for the host (the view's
// mcrat_hip_rebin_cyclosynch).  (First form, MCRAT_HIP_CS_HOOK_KERNEL=1: the list parks after every such pass and cs_replace_pool_kernel,
// one workgroup per parked list, runs the hook between two launches of the loop.)
extern "C" int mcrat_hip_pool_scatter_frames_cyclosynch(mcrat_hip_ctx *c, const mcrat_hip_cyclosynch *cs, int max_photons, double fps,
                                                        const mcrat_hip_pool_cs_list *lists, mcrat_hip_frame_stats *stats,
                                                        mcrat_hip_cyclosynch_counts *counts)
{
    if (!c || !cs || !lists || !counts || cs->b_field_calc < 0 || cs->b_field_calc > 2 || max_photons <= 0 || !(fps > 0)) return MCRAT_HIP_EINVAL;
    if (!c->is_pool) return MCRAT_HIP_ESTATE;
    if (!c->cfg.cyclosynchrotron_switch) { c->last_error = "the pool was created with cyclosynchrotron_switch = 0"; return MCRAT_HIP_ESTATE; }
    if (!c->have_hydro || !c->hcol_buf) return MCRAT_HIP_ESTATE;
    const int R = c->n_ranks;
    int rc;
    std::vector<int> open((size_t)R, 0), emit_base((size_t)R, 0);
    std::vector<uint64_t> seeds((size_t)R, 0);
    std::vector<double> t_now((size_t)R, 0.0), t_rem((size_t)R, 0.0);
    std::vector<mcrat_hip_cyclosynch> csr((size_t)R, *cs);
    std::vector<int> carried((size_t)R, 0);
    for (int r = 0; r < R; ++r) {
        carried[(size_t)r] = counts[r].scatt_cyclosynch_num_ph;      // in: the counter main() carries from the previous frame (mcrat.c:873,921)
        memset(&counts[r], 0, sizeof counts[r]);
        if (!lists[r].open) continue;
        mcrat_hip_ctx *v = c->views[r];
        if (!v || !v->have_photons) { c->last_error = "pool_scatter_frames_cyclosynch: an open list does not exist"; return MCRAT_HIP_ESTATE; }
        open[(size_t)r] = 1;
        seeds[(size_t)r] = lists[r].seed; t_now[(size_t)r] = lists[r].time_now; t_rem[(size_t)r] = lists[r].remaining_time;
        csr[(size_t)r].scatt_frame_number = lists[r].scatt_frame_number;
        csr[(size_t)r].inj_frame_number = lists[r].inj_frame_number;
    }
    if ((rc = pool_emit_cyclosynch(c, cs, csr, max_photons, fps, lists, counts, emit_base))) return rc;                // :727-744
    if ((rc = mcrat_hip_pool_begin_frames(c, open.data(), seeds.data(), t_now.data(), t_rem.data()))) return rc;
    // the hooks' state, one CsFrame per list
    if (c->d_cs_hook) { HIPCHK(c, hipFree(c->d_cs_hook)); c->d_cs_hook = nullptr; }
    HIPCHK(c, hipMalloc((void **)&c->d_cs_hook, sizeof(CsFrame) * (size_t)R));
    CsFrame *d_cf = static_cast<CsFrame *>(c->d_cs_hook);
    std::vector<CsFrame> cf((size_t)R);
    for (int r = 0; r < R; ++r) {
        cf[(size_t)r] = CsFrame{};
        cf[(size_t)r].max_photons = max_photons; cf[(size_t)r].last_iteration = ~0ull; cf[(size_t)r].scatt_num = carried[(size_t)r];
    }
    HIPCHK(c, hipMemcpyAsync(d_cf, cf.data(), sizeof(CsFrame) * (size_t)R, hipMemcpyHostToDevice, c->stream));
    if ((rc = pool_describe(c))) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    CsEmitParams p{};
    p.dimensions = c->kc.dimensions; p.geometry = c->kc.geometry; p.b_field_calc = cs->b_field_calc; p.epsilon_b = cs->epsilon_b;
    // lists that change length: columns stay in HBM/L2 (longest = the window), so LDS does not limit the lists per CU; with more than
    // two lists per CU the 128-thread workgroups put four on one (cfg5 at 1e7 photons: 420 -> 380 ms per frame)
    {
        int cus = 256, dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        c->rank_block = R > 2 * cus ? 128 : 256;
        // ... and with eight and more per CU one wavefront per list (eight on a CU): no wave ever waits at a barrier for the one that walks
        // the event (cfg5: 250 -> 230 ms per frame); only with the hook inside the loop (the hook kernel is written for 128 threads and more)
        if (R > 8 * cus && !(getenv("MCRAT_HIP_CS_HOOK_KERNEL") && atoi(getenv("MCRAT_HIP_CS_HOOK_KERNEL")) != 0)) c->rank_block = 64;
        c->rank_fuse = false;
        if (const char *e = getenv("MCRAT_HIP_RANK_BLOCK")) c->rank_block = (atoi(e) == 64) ? 64 : (atoi(e) == 128) ? 128 : 256;
    }
    // The hook runs inside rank_loop_kernel (its CSH build: cs_hook_body right after the pass, the list goes on in the same launch); with
    // MCRAT_HIP_CS_HOOK_KERNEL=1 the lists park after such a pass instead and cs_replace_pool_kernel runs it between two launches (the
    // first form of this driver, kept for the A/B).  Either way a list only stays parked for the host when it has to be rebinned.
    const bool hook_kernel = getenv("MCRAT_HIP_CS_HOOK_KERNEL") && atoi(getenv("MCRAT_HIP_CS_HOOK_KERNEL")) != 0;
    if (!c->d_cs_args) HIPCHK(c, hipMalloc(&c->d_cs_args, sizeof(CsHookArgs)));
    {
        CsHookArgs ha;
        ha.p = p; ha.h = c->hcol;
        HIPCHK(c, hipMemcpyAsync(c->d_cs_args, &ha, sizeof ha, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const CsHookArgs *d_args = hook_kernel ? nullptr : static_cast<const CsHookArgs *>(c->d_cs_args);
    const int pairs_per_sync = hook_kernel ? 8 : 2;
    for (;;) {
        if (c->cfg.profile) {                                    // the loop's launches between events (bench.py: cfg5's loop-only roofline)
            if ((rc = ensure_events(c, 2))) return rc;
            HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
        }
        for (int k = 0; k < pairs_per_sync; ++k) {
            HIPCHK(c, launch_rank_loop(c->kc, c->ph, c->hy, c->d_rstates, c->key, R, c->rank_stride, 1 << 30, c->d_desc, d_cf, d_args, 4096, c->rank_block,
                                       c->stream));
            if (hook_kernel) HIPCHK(c, launch_cs_replace_pool(p, c->hy, c->hcol, c->d_rstates, c->ph, c->rank_stride, R, c->d_desc, d_cf, c->stream));
        }
        if (c->cfg.profile) {
            HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
            HIPCHK(c, hipEventSynchronize(c->ev[1]));
            float ms = 0;
            HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            c->prof_step_ms += ms;
            c->prof_launches += pairs_per_sync;
        }
        HIPCHK(c, hipMemcpyAsync(c->h_rstates, c->d_rstates, sizeof(LoopState) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(cf.data(), d_cf, sizeof(CsFrame) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_desc, c->d_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bool all_done = true;
        std::vector<int> parked;                                                                  // lists waiting for rebinCyclosynchCompPhotons (:797-808)
        for (int r = 0; r < R; ++r) {
            if (!open[(size_t)r]) continue;
            mcrat_hip_ctx *v = c->views[r];
            v->ph.n = c->h_desc[r].len;                                                           // the list may have doubled in the hook
            CsFrame &f = cf[(size_t)r];
            if (f.halt == CS_HALT_GROW) {
                c->last_error = "a cyclo-synchrotron list outgrew the pool's slots per rank (mcrat_hip_pool_create: allow for the doublings)";
                return MCRAT_HIP_ENOMEM;
            }
            if (f.halt == CS_HALT_REBIN) parked.push_back(r);
        }
        if (!parked.empty()) {                                                                    // all of them in the same two launches
            std::vector<PoolRebinResult> res;
            if ((rc = pool_rebin(c, cs, max_photons, parked, res))) return rc;
            for (size_t j = 0; j < parked.size(); ++j) {
                const int r = parked[j];
                CsFrame &f = cf[(size_t)r];
                if (res[j].rc == MCRAT_HIP_OK) {
                    counts[r].rebins += 1;
                    emit_base[(size_t)r] = res[j].num_cyclosynch_ph_emit;
                    f.emitted = 0;
                    f.scatt_num = res[j].scatt_cyclosynch_num_ph;
                }
                f.halt = 0;
                c->h_rstates[r].done = f.saved_done;
                HIPCHK(c, hipMemcpyAsync(d_cf + r, &f, sizeof f, hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(reinterpret_cast<char *>(c->d_rstates + r) + offsetof(LoopState, done), &c->h_rstates[r].done, sizeof(int),
                                         hipMemcpyHostToDevice, c->stream));
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        for (int r = 0; r < R; ++r)
            if (open[(size_t)r] && c->h_rstates[r].done != LOOP_DONE) all_done = false;
        if (all_done) break;
    }
    std::vector<int> absorb((size_t)R, 0);
    bool any_absorb = false;
    for (int r = 0; r < R; ++r) {
        if (!open[(size_t)r]) continue;
        mcrat_hip_ctx *v = c->views[r];
        const CsFrame &f = cf[(size_t)r];
        counts[r].num_cyclosynch_ph_emit = emit_base[(size_t)r] + f.emitted;
        counts[r].scatt_cyclosynch_num_ph = f.scatt_num;
        counts[r].n_comptonized = f.n_comptonized;
        v->pending_applied = false;
        v->rank_current = true;
        if (stats) state_to_stats(c->h_rstates[r], v->ph.n, &stats[r]);
    }
    {                                                                                             // :853-878, the lists over max_photons all at once
        std::vector<int> over;
        for (int r = 0; r < R; ++r)
            if (open[(size_t)r] && lists[r].emit_pool && counts[r].scatt_cyclosynch_num_ph > max_photons) over.push_back(r);
        std::vector<PoolRebinResult> res;
        if ((rc = pool_rebin(c, cs, max_photons, over, res))) return rc;
        for (size_t j = 0; j < over.size(); ++j) {
            if (res[j].rc != MCRAT_HIP_OK) continue;
            const int r = over[j];
            counts[r].rebins += 1;
            counts[r].num_cyclosynch_ph_emit = res[j].num_cyclosynch_ph_emit;
            counts[r].scatt_cyclosynch_num_ph = res[j].scatt_cyclosynch_num_ph;
        }
        for (int r = 0; r < R; ++r)
            if (open[(size_t)r] && lists[r].emit_pool && counts[r].num_cyclosynch_ph_emit > 0) { absorb[(size_t)r] = 1; any_absorb = true; }
    }
    if (any_absorb) {                                                                             // phAbsCyclosynch of every such list, one launch
        for (int r = 0; r < R; ++r)
            if (absorb[(size_t)r] && (rc = flush_pending(c->views[r]))) return rc;
        for (int r = 0; r < R; ++r) { c->h_desc[r].len = open[(size_t)r] ? c->views[r]->ph.n : 0; }
        const size_t bytes = (sizeof(CsAbsPartial) + sizeof(int)) * (size_t)R;
        if ((rc = ensure_aos(c, bytes + 64))) return rc;
        CsAbsPartial *d_part = static_cast<CsAbsPartial *>(c->aos_buf);
        int *d_sel = reinterpret_cast<int *>(d_part + R);
        std::vector<CsAbsPartial> part((size_t)R);
        HIPCHK(c, hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(RankDesc) * (size_t)R, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_sel, absorb.data(), sizeof(int) * (size_t)R, hipMemcpyHostToDevice, c->stream));
        CsParams ap{c->kc.dimensions, cs->b_field_calc, cs->epsilon_b};
        HIPCHK(c, launch_cs_absorb_pool(ap, c->ph, c->rank_stride, R, c->d_desc, d_sel, c->hy.temp, c->hcol, d_part, c->stream));
        HIPCHK(c, hipMemcpyAsync(part.data(), d_part, sizeof(CsAbsPartial) * (size_t)R, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int r = 0; r < R; ++r) {
            if (!absorb[(size_t)r]) continue;
            counts[r].frame_abs_cnt = (int)part[(size_t)r].abs_count;
            counts[r].scatt_cyclosynch_num_ph = (int)part[(size_t)r].scatt_count;
            counts[r].n_comptonized -= part[(size_t)r].abs_weight;
            drop_graph(c->views[r]);
        }
    }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_step_locate_sample(mcrat_hip_ctx *c, int find_nearest_block_switch)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->frame_open || c->n_ranks > 0) return MCRAT_HIP_ESTATE;
    HIPCHK(c, launch_step(c->kc, find_nearest_block_switch != 0, c->ph, c->hy, c->d_state, c->key, c->partials, c->step_blocks, c->shortlist, c->stream));
    c->find_switch = 0;
    c->pending_applied = true;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_frame_statistics(mcrat_hip_ctx *c, mcrat_hip_frame_stats *stats)
{
    if (!c || !stats) return MCRAT_HIP_EINVAL;
    if (!c->frame_open || c->n_ranks > 0) return MCRAT_HIP_ESTATE;
    HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fill_stats(c, stats);
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_update_photon_position(mcrat_hip_ctx *c, double t)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    int rc;
    if (c->frame_open && c->n_ranks == 0 && (rc = flush_pending(c))) return rc;      // what the loop still owes the photons
    c->pending_applied = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int nseg = 1, skip = -1;
    HIPCHK(c, hipMemcpy(reinterpret_cast<char *>(c->d_state) + offsetof(LoopState, seg), &t, sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(reinterpret_cast<char *>(c->d_state) + offsetof(LoopState, nseg), &nseg, sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(reinterpret_cast<char *>(c->d_state) + offsetof(LoopState, skip_idx), &skip, sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(c, launch_flush(c->ph, c->d_state, c->step_blocks, c->stream));          // r += (p/p0) c t, mclib.c:1067-1095
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_step_event(mcrat_hip_ctx *c, mcrat_hip_frame_stats *stats)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->frame_open || c->n_ranks > 0) return MCRAT_HIP_ESTATE;
    HIPCHK(c, launch_event_of(c));
    c->pending_applied = false;
    HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fill_stats(c, stats);
    return MCRAT_HIP_OK;
}

// ---------------------------------------------------------------------------------------------- shared clock
extern "C" size_t mcrat_hip_shared_clock_bytes_per_rank(void) { return sizeof(ScProposal); }

extern "C" int mcrat_hip_shared_clock_attach(mcrat_hip_ctx *c, int world, int rank, long long slot_base, void *send, void *recv)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (world < 1 || world > SC_MAX_WORLD || rank < 0 || rank >= world) return MCRAT_HIP_EINVAL;
    if (slot_base < 0 || (slot_base & 1) || slot_base > 0xffffffffll - 0x7fffffffll) {
        c->last_error = "slot_base must be even, non-negative and leave room for 2^31 slots below 2^32";
        return MCRAT_HIP_EINVAL;
    }
    if (c->n_ranks > 0 || c->cfg.virtual_rank_photons > 0) { c->last_error = "shared clock and virtual ranks exclude each other"; return MCRAT_HIP_EINVAL; }
    if (c->frame_open && c->h_state && !c->h_state->done) return MCRAT_HIP_ESTATE;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->d_sc) HIPCHK(c, hipMalloc((void **)&c->d_sc, sizeof(ScState)));
    if (c->sc_own_send && c->sc_send) { (void)hipFree(c->sc_send); c->sc_send = nullptr; }
    if (c->sc_own_recv && c->sc_recv) { (void)hipFree(c->sc_recv); c->sc_recv = nullptr; }
    c->sc_own_send = c->sc_own_recv = false;
    if (c->sc_flags) { (void)hipFree(c->sc_flags); c->sc_flags = nullptr; }
    if (c->sc_gather) { (void)hipFree(c->sc_gather); c->sc_gather = nullptr; }
    c->sc_device = c->sc_peers_set = false;
    if (send) c->sc_send = (ScProposal *)send;
    else { HIPCHK(c, hipMalloc((void **)&c->sc_send, sizeof(ScProposal))); c->sc_own_send = true; }
    if (recv) c->sc_recv = (ScProposal *)recv;
    else if (world == 1) c->sc_recv = c->sc_send;
    else { HIPCHK(c, hipMalloc((void **)&c->sc_recv, sizeof(ScProposal) * world)); c->sc_own_recv = true; }
    HIPCHK(c, hipMemset(c->sc_send, 0, sizeof(ScProposal)));
    if (c->sc_recv != c->sc_send) HIPCHK(c, hipMemset(c->sc_recv, 0, sizeof(ScProposal) * world));
    c->sc_world = world;
    c->sc_rank = rank;
    c->key.slot_base = (uint32_t)slot_base;
    c->frame_open = false;
    drop_graph(c);
    return MCRAT_HIP_OK;
}

// the same attachment with the exchange done by the GPUs themselves: no host collective between propose and resolve
extern "C" int mcrat_hip_shared_clock_attach_device(mcrat_hip_ctx *c, int world, int rank, long long slot_base)
{
    int rc = mcrat_hip_shared_clock_attach(c, world, rank, slot_base, nullptr, nullptr);
    if (rc) return rc;
    if (c->sc_own_recv && c->sc_recv) (void)hipFree(c->sc_recv);
    c->sc_recv = nullptr; c->sc_own_recv = false;
    const size_t recv_bytes = sizeof(ScProposal) * 2 * (size_t)world, flag_bytes = sizeof(unsigned long long) * SC_FLAG_WORDS;
    HIPCHK(c, hipMalloc((void **)&c->sc_gather, sizeof(ScProposal) * (size_t)world));
    HIPCHK(c, hipMemset(c->sc_gather, 0, sizeof(ScProposal) * (size_t)world));
    HIPCHK(c, hipExtMallocWithFlags((void **)&c->sc_recv, recv_bytes, hipDeviceMallocFinegrained));
    c->sc_own_recv = true;
    HIPCHK(c, hipExtMallocWithFlags((void **)&c->sc_flags, flag_bytes, hipDeviceMallocFinegrained));
    HIPCHK(c, hipMemset(c->sc_recv, 0, recv_bytes));
    HIPCHK(c, hipMemset(c->sc_flags, 0, flag_bytes));
    HIPCHK(c, hipDeviceSynchronize());
    c->sc_device = true;
    c->sc_fold = true;
    if (const char *e = getenv("MCRAT_HIP_SC_FOLD")) c->sc_fold = atoi(e) != 0;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_peer_buffers(mcrat_hip_ctx *c, void **recv, size_t *recv_bytes, void **flags, size_t *flags_bytes)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->sc_device) return MCRAT_HIP_ESTATE;
    if (recv) *recv = c->sc_recv;
    if (recv_bytes) *recv_bytes = sizeof(ScProposal) * 2 * (size_t)c->sc_world;
    if (flags) *flags = c->sc_flags;
    if (flags_bytes) *flags_bytes = sizeof(unsigned long long) * SC_FLAG_WORDS;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_set_peers(mcrat_hip_ctx *c, void *const *peer_recv, void *const *peer_flags)
{
    if (!c || !peer_recv || !peer_flags) return MCRAT_HIP_EINVAL;
    if (!c->sc_device) return MCRAT_HIP_ESTATE;
    for (int r = 0; r < c->sc_world; ++r) {
        void *pr = r == c->sc_rank ? (void *)c->sc_recv : peer_recv[r], *pf = r == c->sc_rank ? (void *)c->sc_flags : peer_flags[r];
        if (!pr || !pf) return MCRAT_HIP_EINVAL;
        c->sc_peers.recv[r] = static_cast<ScProposal *>(pr);
        c->sc_peers.flag[r] = static_cast<unsigned long long *>(pf);
    }
    c->sc_peers_set = true;
    return MCRAT_HIP_OK;
}

static int sc_wait_spins()
{
    int spins = 4000000;                                             // a few seconds: ranks enter a frame together (a barrier on the host)
    if (const char *e = getenv("MCRAT_HIP_SC_WAIT_SPINS")) spins = atoi(e) > 0 ? atoi(e) : spins;
    return spins;
}

// the exchange inside the round's own kernels (launch.hpp, ScFold), when the GPUs do it themselves
static ScFold sc_fold_of(const mcrat_hip_ctx *c)
{
    ScFold f{};
    f.on = (c->sc_device && c->sc_peers_set && c->sc_fold) ? 1 : 0;
    if (f.on) {
        f.world = c->sc_world; f.rank = c->sc_rank; f.max_spins = sc_wait_spins();
        f.peers = c->sc_peers; f.my_flags = c->sc_flags; f.recv = c->sc_recv; f.gathered = c->sc_gather;
    }
    return f;
}

extern "C" int mcrat_hip_shared_clock_exchange_push(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->sc_device || !c->sc_peers_set || !c->frame_open) return MCRAT_HIP_ESTATE;
    if (c->sc_fold) return MCRAT_HIP_OK;                              // (the propose kernel has pushed)
    HIPCHK(c, launch_sc_push(c->sc_send, c->sc_peers, c->sc_flags, c->sc_world, c->sc_rank, c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_exchange_wait(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->sc_device || !c->sc_peers_set || !c->frame_open) return MCRAT_HIP_ESTATE;
    if (c->sc_fold) return MCRAT_HIP_OK;                              // (the resolve kernel will wait)
    HIPCHK(c, launch_sc_wait(c->sc_flags, c->sc_recv, c->sc_gather, c->sc_world, sc_wait_spins(), c->d_state, c->stream));
    return MCRAT_HIP_OK;
}

// After a wait has given up (mcrat_hip_shared_clock_poll answers MCRAT_HIP_EHIP) the ranks' round numbers may be out of step.  Every rank calls this --
// its give-up word, its round number and the stamps its peers left are cleared --, then the ranks synchronise (any barrier: a peer must not push
// its first new round before this rank has cleared its stamps), then begin_frame as usual.
extern "C" int mcrat_hip_shared_clock_reset_exchange(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (!c->sc_device) return MCRAT_HIP_ESTATE;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemset(c->sc_flags, 0, sizeof(unsigned long long) * (size_t)SC_FLAG_WORDS));
    c->frame_open = false;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_exchange(mcrat_hip_ctx *c)
{
    int rc = mcrat_hip_shared_clock_exchange_push(c);
    return rc ? rc : mcrat_hip_shared_clock_exchange_wait(c);
}

extern "C" int mcrat_hip_shared_clock_buffers(mcrat_hip_ctx *c, void **send, void **recv)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->sc_world <= 0) return MCRAT_HIP_ESTATE;
    if (send) *send = c->sc_send;
    if (recv) *recv = c->sc_recv;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_propose(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->sc_world <= 0 || !c->frame_open) return MCRAT_HIP_ESTATE;
    if (c->sc_device && !c->sc_peers_set && c->sc_fold) { c->last_error = "shared clock: the peers' buffers are not set (mcrat_hip_shared_clock_set_peers)"; return MCRAT_HIP_ESTATE; }
    HIPCHK(c, launch_sc_propose(c->kc, c->find_switch != 0, c->ph, c->hy, c->d_state, c->d_sc, c->key, c->partials, c->step_blocks,
                                c->shortlist, c->sc_send, sc_fold_of(c), c->stream));
    c->find_switch = 0;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_resolve(mcrat_hip_ctx *c)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->sc_world <= 0 || !c->frame_open) return MCRAT_HIP_ESTATE;
    const ScProposal *all = c->sc_device ? c->sc_gather : c->sc_recv;                   // device exchange: the wait kernel has copied the round out
    HIPCHK(c, launch_sc_resolve(c->kc, c->ph, c->hy, c->d_state, c->d_sc, c->key, all, c->sc_world, sc_fold_of(c), c->stream));
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_poll(mcrat_hip_ctx *c, int *frame_done, mcrat_hip_frame_stats *stats)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->sc_world <= 0 || !c->frame_open) return MCRAT_HIP_ESTATE;
    HIPCHK(c, hipMemcpyAsync(c->h_state, c->d_state, sizeof(LoopState), hipMemcpyDeviceToHost, c->stream));
    unsigned long long gave_up = 0;
    if (c->sc_device) HIPCHK(c, hipMemcpyAsync(&gave_up, c->sc_flags + SC_GAVE_UP_WORD, sizeof gave_up, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (gave_up) { c->last_error = "shared clock: a peer's proposal did not arrive (device-initiated exchange gave up waiting)"; return MCRAT_HIP_EHIP; }
    if (frame_done) *frame_done = c->h_state->done == LOOP_DONE;
    fill_stats(c, stats);
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_shared_clock_finish(mcrat_hip_ctx *c, mcrat_hip_frame_stats *stats)
{
    if (!c) return MCRAT_HIP_EINVAL;
    if (c->sc_world <= 0 || !c->frame_open) return MCRAT_HIP_ESTATE;
    int rc;
    if ((rc = flush_pending(c))) return rc;
    return mcrat_hip_shared_clock_poll(c, nullptr, stats);
}

// ---------------------------------------------------------------------------------------------- reductions
static int run_reduce(mcrat_hip_ctx *c, ReducePartial &t)
{
    if (!c->have_photons) return MCRAT_HIP_ESTATE;
    int rc = flush_pending(c);
    if (rc) return rc;
    const int blocks = std::min(mcrat_hip_ctx::RED_BLOCKS, std::max(1, (c->ph.n + 255) / 256));
    HIPCHK(c, launch_reduce(c->ph, c->d_red, blocks, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_red, c->d_red, sizeof(ReducePartial) * blocks, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    t = c->h_red[0];
    for (int b = 1; b < blocks; ++b) {
        const ReducePartial &p = c->h_red[b];
        t.r_min = std::min(t.r_min, p.r_min); t.r_max = std::max(t.r_max, p.r_max);
        t.th_min = std::min(t.th_min, p.th_min); t.th_max = std::max(t.th_max, p.th_max);
        t.sum_scatt += p.sum_scatt; t.sum_r += p.sum_r; t.e_sum += p.e_sum; t.w_sum += p.w_sum;
        t.max_scatt = std::max(t.max_scatt, p.max_scatt); t.min_scatt = std::min(t.min_scatt, p.min_scatt);
        t.count += p.count;
    }
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_ph_minmax(mcrat_hip_ctx *c, double *min_r, double *max_r, double *min_theta, double *max_theta)
{
    if (!c || !min_r || !max_r || !min_theta || !max_theta) return MCRAT_HIP_EINVAL;
    ReducePartial t;
    int rc = run_reduce(c, t);
    if (rc) return rc;
    *min_r = t.r_min; *max_r = t.r_max; *min_theta = t.th_min; *max_theta = t.th_max;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_scatt_stats(mcrat_hip_ctx *c, int *max_scatt, int *min_scatt, double *avg_scatt, double *avg_r)
{
    if (!c || !max_scatt || !min_scatt || !avg_scatt || !avg_r) return MCRAT_HIP_EINVAL;
    ReducePartial t;
    int rc = run_reduce(c, t);
    if (rc) return rc;
    *max_scatt = (int)t.max_scatt; *min_scatt = (int)t.min_scatt;
    *avg_scatt = t.sum_scatt / (double)t.count; *avg_r = t.sum_r / (double)t.count;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_avg_energy(mcrat_hip_ctx *c, double *erg)
{
    if (!c || !erg) return MCRAT_HIP_EINVAL;
    ReducePartial t;
    int rc = run_reduce(c, t);
    if (rc) return rc;
    *erg = (t.e_sum * C_LIGHT) / t.w_sum;
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_eval_function(mcrat_hip_ctx *c, int fn, int n, const double *in, double *out, uint64_t seed)
{
    static const int in_w[8] = {0, 1, 7, 7, 13, 5, 5, 9}, out_w[8] = {0, 1, 4, 4, 4, 4, 4, 13};
    if (!c || n <= 0 || !in || !out || fn < 1 || fn > 7) return MCRAT_HIP_EINVAL;
    const size_t bi = sizeof(double) * (size_t)in_w[fn] * (size_t)n, bo = sizeof(double) * (size_t)out_w[fn] * (size_t)n;
    double *d_in = nullptr, *d_out = nullptr;
    HIPCHK(c, hipMalloc((void **)&d_in, bi));
    if (hipMalloc((void **)&d_out, bo) != hipSuccess) { (void)hipFree(d_in); return MCRAT_HIP_ENOMEM; }
    hipError_t e = hipMemcpyAsync(d_in, in, bi, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_eval_function(fn, c->kc.stokes, n, d_in, d_out, seed, c->key.stream, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bo, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    HIPCHK(c, e);
    return MCRAT_HIP_OK;
}

extern "C" int mcrat_hip_lookup_cell(mcrat_hip_ctx *c, int n, const double *a0, const double *a1, const double *a2, int *out)
{
    if (!c || n <= 0 || !a0 || !a1 || !out) return MCRAT_HIP_EINVAL;
    if (!c->have_hydro) return MCRAT_HIP_ESTATE;
    double *d = nullptr;
    int *dout = nullptr;
    HIPCHK(c, hipMalloc((void **)&d, sizeof(double) * 3 * (size_t)n));
    if (hipMalloc((void **)&dout, sizeof(int) * (size_t)n) != hipSuccess) { (void)hipFree(d); return MCRAT_HIP_ENOMEM; }
    int rc = MCRAT_HIP_OK;
    std::vector<double> zeros;
    if (!a2) { zeros.assign(n, 0.0); a2 = zeros.data(); }
    if (hipMemcpy(d, a0, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, a1, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + 2 * (size_t)n, a2, sizeof(double) * n, hipMemcpyHostToDevice) != hipSuccess ||
        launch_lookup(c->kc, c->hy, n, d, d + n, d + 2 * (size_t)n, dout, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(out, dout, sizeof(int) * n, hipMemcpyDeviceToHost) != hipSuccess) {
        c->last_error = "lookup_cell failed";
        rc = MCRAT_HIP_EHIP;
    }
    (void)hipFree(d);
    (void)hipFree(dout);
    return rc;
}
