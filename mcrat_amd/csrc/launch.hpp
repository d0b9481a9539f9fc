// launch.hpp -- host-callable launchers of the kernels in kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace mcrat {

struct KernelConfig {
    int dimensions;
    int geometry;
    int stokes;
    int table;      // TAU_CALCULATION == TABLE: the kernels of kernels_table.hip
};

struct ReducePartial {      // one per workgroup of the reduction kernels
    double r_min, r_max, th_min, th_max;       // phMinMax (weight != 0)
    double sum_scatt, sum_r, e_sum, w_sum;     // phScattStats / averagePhotonEnergy
    double max_scatt, min_scatt;
    long long count;
};

int step_grid_blocks(int n_pad);   // workgroups of the step kernel for this capacity

// findContainingHydroCell + calcMeanFreePath (+ the pending updatePhotonPosition) over all slots
hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream);
// candidate selection + photonEvent + loop bookkeeping
hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
// ... with the caller's tape of uniforms as the random source (mcrat_hip_set_rng_tape): the pass's free-path draws in slot order
// (tape_draw_kernel, after launch_step), then the event reading on from where they stopped
struct TapeDev;
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream);
// virtual ranks: every workgroup (of `block` = 128 or 256 threads) runs the whole loop of one independent photon list: the slots
// [r * rank_stride, ...) -- rank_stride of them, or desc[r].len with the list's own seed and stream (rank pool); longest_list sizes the LDS copy
// hook: (device memory) what the cyclo-synchrotron hook needs -- then lists with `cs` run it inside the loop instead of parking for
// cs_replace_pool_kernel after every pass it has to look at
// The frame queue (round 4): ONE launch takes every list through SEVERAL hydro frames.  The reference's ranks are asynchronous processes, each in
// its own frame loop (mcrat.c:457-479, :566-934); a launch per hydro frame makes them wait for each other at every frame's end.  A queue launch has
// as many persistent workgroups as the device holds; each draws (frame, list) items from the queue of the XCD it runs on until that is empty, the k-th
// draw on an XCD getting the k-th open item of that XCD's lists in frame-major order (`order`, `ticket`): a list that is through frame f starts f + 1 as
// soon as a workgroup is free, while other lists are still in f.  The item of a list whose previous frame is still running waits for it (frames_done;
// that item was drawn earlier, by a workgroup that is running and depends on nothing later, so this cannot deadlock).  Every list sees exactly the frames it would have seen one launch at a time: the same seeds, clocks, passes and photons
// (tests/test_gpu_frame_queue.py).
struct FrameItem {                   // list r in frame f: item f * n_ranks + r
    unsigned long long seed;         // the list's seed of this frame (gsl_rng_get, mcrat.c:701)
    double time_now;                 // its clock at the start of the frame ...
    double remaining_time;           // ... and the time left in it (mcrat.c:757), unless the clock is chained (below)
    double frame_end;                // (scatt_frame + increment_scatt_frame) / fps: a chained list gets frame_end - its own time_now, the host's expression
    int open;                        // the list takes part in this frame (a rank joins at its injection frame, mcrat.c:566-700)
    int hydro;                       // which staged hydro frame of the launch it propagates through
};
struct FrameQueueDev {
    int n_frames;                    // 0: no queue -- one workgroup per list, one frame, as before
    int restore;                     // every frame starts from the context's snapshot (mcrat_hip_snapshot_photons; benchmarks: the same work every frame)
    int chain_clock;                 // a list's clock carries over from its previous frame of this launch (time_now of the LoopState it left)
    int pad;
    unsigned *ticket;                // [FRAME_QUEUE_XCDS * FRAME_TICKET_STRIDE] per XCD: workgroups that have started there
    const int *order;                // [open items] per XCD (order_off[x] .. order_off[x + 1]) its lists' items in the order they are taken: frame-major
    int order_off[9];
    unsigned *frames_done;           // [n_ranks] f + 1 once item (f, r) is complete; FRAME_STALLED | f: frame f ran into the launch's pass limit
    const FrameItem *items;          // [n_frames * n_ranks]
    const HydroDev *hydro;           // [n_hydro] the staged hydro frames of the launch (FrameItem::hydro indexes it); read through the constant address space
    LoopState *records;              // [n_frames * n_ranks] the LoopState every frame ended with
    long long snap_delta;            // bytes from a column of the live lists to its copy in the snapshot (restore)
    long long capture_delta, capture_stride;   // != 0: at the end of frame f < n_frames - 1 the list's columns are copied to live + capture_delta + f * capture_stride
};
constexpr unsigned FRAME_STALLED = 0x80000000u;
constexpr int FRAME_QUEUE_XCDS = 8, FRAME_TICKET_STRIDE = 16;     // (a ticket per XCD, each on a 64-B line of its own)
// `fq` (n_frames > 0, n_open items in fq->order) makes it a queue launch: one workgroup per open item
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, struct CsFrame *cs, const struct CsHookArgs *hook,
                            long long max_passes, int block, hipStream_t stream, const FrameQueueDev *fq = nullptr, int n_open = 0);
// shared clock with a device-initiated exchange (staging.hip): recv[r] = rank r's receive buffer (2 x world proposals, by round parity),
// flag[r] = rank r's stamps (SC_MAX_WORLD words, one per sender, + a word counting waits that gave up + the rank's own round number)
struct ScPeers { ScProposal *recv[SC_MAX_WORLD]; unsigned long long *flag[SC_MAX_WORLD]; };
constexpr int SC_GAVE_UP_WORD = SC_MAX_WORLD, SC_ROUND_WORD = SC_MAX_WORLD + 1, SC_FLAG_WORDS = SC_MAX_WORLD + 2;
// the same exchange folded into the round's own kernels (on != 0): sc_propose_kernel pushes its proposal when it has written it, sc_resolve_kernel
// waits for the round's stamps before it reads `gathered` -- two launches less per round than push and wait as kernels of their own
struct ScFold { int on, world, rank, max_spins; ScPeers peers; unsigned long long *my_flags; const ScProposal *recv; ScProposal *gathered; };
hipError_t launch_sc_push(const ScProposal *send, const ScPeers &peers, unsigned long long *my_flags, int world, int rank, hipStream_t stream);
hipError_t launch_sc_wait(unsigned long long *my_flags, const ScProposal *recv, ScProposal *gathered, int world, int max_spins, LoopState *st, hipStream_t stream);
// FAST mode: every photon through the whole frame on its own clock (kernels.hip, fast_frame_kernel); the counters add up over launches
struct FastCounts { unsigned long long photon_steps, scatterings, kn_rejections, relocated, not_found, unfinished, passes; };
// desc != nullptr: the photons are the lists of a rank pool (list r in slots [r * stride, r * stride + desc[r].len), stride a multiple of 256);
// every list then has its own seed and stream (desc), its own frame time and its own counters (counts[r]); lists with len 0 or no time stand still
struct FastLists { int stride; const RankDesc *desc; const double *remaining_time; const int *windows; };   // windows: per list (nullptr: the launch's)
hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream);
// one list over several GPUs with one clock: {step, midpass re-read, proposal of this GPU's earliest candidates} ...
hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream);
// ... and, after the host has all-gathered the proposals into `all`, the replicated walk of photonEvent
hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream);
// apply the pending advance (end of run / before photons are read back) and clear it
hipError_t launch_flush(const PhotonDev &ph, LoopState *st, int blocks, hipStream_t stream);
hipError_t launch_k2e(const double *temp, double *k2e, int M, hipStream_t stream);
hipError_t launch_reduce(const PhotonDev &ph, ReducePartial *out, int blocks, hipStream_t stream);
// struct photon records <-> SoA columns on the device (staging.hip); `aos` is a device buffer of n 176-B records
hipError_t launch_init_states(LoopState *single, LoopState *ranks, int n_ranks, const LoopState &v, hipStream_t stream);
hipError_t launch_convert_comptonized(const PhotonDev &ph, unsigned *converted, hipStream_t stream);   // 'k' with weight != 0 -> 'c' (mcrat_io.c:896-900)
hipError_t launch_clear_slots(const PhotonDev &ph, int first, int count, hipStream_t stream);   // every column zeroed: no list's slots
// rank pool: the per-frame reductions and printPhotons' count for every list (desc[r].len slots from r * stride), one launch
hipError_t launch_rank_reduce(const PhotonDev &ph, int stride, int n_ranks, const RankDesc *desc, ReducePartial *out, int *n_out, hipStream_t stream);
hipError_t launch_init_states_multi(LoopState *ranks, int n_ranks, const int *open, const double *time_now, const double *remaining, hipStream_t stream);
hipError_t launch_aos_to_soa(const void *aos, const PhotonDev &ph, int n, hipStream_t stream);
// many lists of a rank pool at once: desc = `count` device records {int rank, int n, long long first record in aos}; the rest of each window is cleared
hipError_t launch_pool_aos_to_soa(const void *aos, const PhotonDev &pool, int stride, const void *desc, int count, hipStream_t stream);
hipError_t launch_soa_to_aos(const PhotonDev &ph, void *aos, int first, int n, hipStream_t stream);   // record k = slot first + k
// printPhotons' arrays (mcrat_io.c:137-181): photons with weight != 0, slot order; col: p0-3, comv_p0-3, r0-2, s0-3, num_scatt, weight
struct OutputCols {
    double *col[17];
    char *type;
};
hipError_t launch_output_count(const PhotonDev &ph, int n, unsigned *block_count, unsigned long long *d_total, hipStream_t stream);
hipError_t launch_output_write(const PhotonDev &ph, int n, const int *block_start, const OutputCols &out, hipStream_t stream);

// the cell-lookup grid, built on the device (grid_build.hip)
struct GridPlan {
    double org[3], inv[3];
    int dim[3], logmap[3];
    int naxes;
};
hipError_t grid_count(const GridPlan &p, const CellGeom *geom, const CellGeom2 *geom2, int M, unsigned *count, long long nb,
                      unsigned long long *d_total, hipStream_t stream);
size_t grid_scan_scratch_ints(long long nb);
hipError_t launch_exclusive_scan(const unsigned *count, long long n, int *start, int *scratch, long long total, hipStream_t stream);

// photonInjection on the device (inject.hip; mclib.c:9-300)
struct InjectParams {
    int dimensions, geometry;
    double rmin, rmax, theta_min, theta_max;   // the injection slab, mclib.c:34-35,57
    double num_dens_coeff;                     // 8.44f / 20.29f widened (mclib.c:23-32)
    int wien;                                  // spect == 'w'
};
// Poisson photon count of every cell for one weight (mclib.c:87-136); *total receives the sum
hipError_t launch_inject_count(const InjectParams &p, const HydroDev &hy, double ph_weight_adjusted, unsigned long long attempt, RngKey key,
                               unsigned *count, unsigned long long *d_total, hipStream_t stream);
// the photons themselves (mclib.c:150-296), ordered by cell then draw, into the SoA columns
hipError_t launch_inject_generate(const InjectParams &p, const HydroDev &hy, double ph_weight_adjusted, RngKey key, const int *start,
                                  const PhotonDev &ph, hipStream_t stream);
// photonInjection for the lists of a rank pool (inject.hip): the slab's cells with the two list-independent factors of mclib.c:110, once per group
// of lists that share the slab, then one workgroup per list
struct alignas(8) InjectSlabCell { int cell; int pad; double v43, g; };
constexpr int POOL_INJECT_CAP = 8192;        // photons one list may receive (LDS table of their cells)
struct alignas(8) PoolInject {
    int inject, group;               // in: takes part; which slab
    unsigned long long seed;         // in
    unsigned stream; int pad;        // in: the list's RNG stream
    double weight_in;                // in: ph_weight
    int min_photons, max_photons;    // in
    double weight_out;               // out
    int n, error;                    // out: photons; 0 ok, 1 no weight fits, 2 no photons, 3 more than the list's window (or the kernel's table) holds
};
hipError_t launch_inject_slab_flag(const InjectParams &p, const HydroDev &hy, unsigned *flag, unsigned long long *d_total, int *n_slab, hipStream_t stream);   // waits
hipError_t launch_inject_slab_write(const InjectParams &p, const HydroDev &hy, const unsigned *flag, int n_slab, int *start, int *scratch, InjectSlabCell *out,
                                    hipStream_t stream);
hipError_t launch_inject_pool(const InjectParams &p, const HydroDev &hy, const PhotonDev &pool, int stride, int n_ranks, const InjectSlabCell *slab, int n_slab,
                              PoolInject *lists, int group, hipStream_t stream);
// getHydroData on the device (ingest.hip; mcrat_io.c:1898-1990)
struct HydroCols {          // struct hydro_dataframe's columns (mcrat.h:194-244), device arrays of M doubles
    double *r0, *r1, *r2, *s0, *s1, *s2;
    double *v0, *v1, *v2;
    double *dens, *dens_lab, *pres, *temp, *gamma;
    double *r, *theta;
    double *B0, *B1, *B2;   // magnetic field (B_FIELD_CALC == SIMULATION; mcrat_hip_set_hydro_extras)
};
struct SlabDev {
    int dimensions, geometry, ph_inj_switch;
    double r_inj_095;                          // 0.95 r_inj
    double r_lo, r_hi, th_lo, th_hi;           // the widened slab for the current elem_factor
};
struct FlashDev {           // device copies of a FLASH checkpoint's datasets (mclib_flash.c:143-193)
    const double *coord, *bsize;
    const int *node;
    const double *velx, *vely, *dens, *pres;
    int coord_stride, bsize_stride;
    long long n_blocks;
    double L, D, P;
};
struct PlutoDev {           // device copies of readGridFile's arrays and the .dbl variable blocks
    int nx, ny, nz;
    const double *x1, *dx1, *x2, *dx2, *x3, *dx3;
    const double *rho, *vx1, *vx2, *vx3, *prs;
    double L, D, P;
};
struct ChomboBox {          // one box of a PLUTO-Chombo level (mclib_pluto.c:520-545)
    long long first_cell;   // (start_displacement + box_offset) / num_vars: where its cells sit in the reader's cell numbering
    long long data_off;     // start_displacement + box_offset: its data in the concatenated "data:datatype=0" arrays
    int level;
    int lo[3], n[3];        // lo_i, lo_j, lo_k; cells per axis
    int cb[3];              // where this level's 1-D coordinate arrays start in ChomboDev::x / dx
    int pad[2];
};
struct ChomboDev {
    const ChomboBox *boxes;
    int n_boxes;
    long long cells;
    const double *data;
    const double *x[3], *dx[3];        // per axis: the levels' x*_array / dx*_array (mclib_pluto.c:446-517), concatenated
    const unsigned char *covered;      // good_node_buffer == 0 (:206-345); null when the mask is not consulted
    int kv[5];                         // component index of rho, vx1, vx2, vx3, prs (-1: absent)
    double L, D, P;
};
hipError_t ingest_count_chombo(const ChomboDev &h, const SlabDev &slab, unsigned *block_count, unsigned long long *d_total, hipStream_t stream);
hipError_t ingest_write_chombo(const ChomboDev &h, const SlabDev &slab, const int *block_start, const HydroCols &out, hipStream_t stream);
// marks the cells of boxes [c0, c1) that a box of [f0, f1) (the next finer level) covers
hipError_t launch_chombo_mask(const ChomboBox *boxes, int c0, int c1, int f0, int f1, int ref_ratio, int three, unsigned char *covered, hipStream_t stream);
// phAbsCyclosynch (mc_cyclosynch.c:1571-1623) on the device (staging.hip): photons at or below their cell's cyclotron frequency and
// all pool photons become null slots (setNullPhoton, photons.c:210-250)
struct CsParams {
    int dimensions, b_field_calc;
    double epsilon_b;
};
struct CsAbsPartial { double abs_weight; int abs_count, scatt_count; };      // one per workgroup
int cs_absorb_blocks(int n);
hipError_t launch_cs_absorb(const CsParams &p, const PhotonDev &ph, const double *temp, const HydroCols &h, CsAbsPartial *partials, hipStream_t stream);
struct OutflowDev {
    int simulation_type;
    double gamma_infinity, lumi, r00, t_comov, ddensity, theta_j, p;
};
struct StagePartial {       // one per workgroup of stage_cells_kernel
    double lo[3], hi[3], smin[3], smax[3];
    int any_hot, pad;
};
long long ingest_blocks(long long n_virtual);   // workgroups (and block_count entries) of the two selection passes
hipError_t ingest_count_flash(const FlashDev &f, const SlabDev &slab, unsigned *block_count, unsigned long long *d_total, hipStream_t stream);
hipError_t ingest_write_flash(const FlashDev &f, const SlabDev &slab, const int *block_start, const HydroCols &out, hipStream_t stream);
hipError_t ingest_count_pluto(const PlutoDev &g, const SlabDev &slab, unsigned *block_count, unsigned long long *d_total, hipStream_t stream);
hipError_t ingest_write_pluto(const PlutoDev &g, const SlabDev &slab, const int *block_start, const HydroCols &out, hipStream_t stream);
hipError_t launch_fill_spherical(int dims, int geom, const HydroCols &h, int M, hipStream_t stream);
hipError_t launch_outflow_prep(int dims, int geom, const OutflowDev &o, const HydroCols &h, int M, hipStream_t stream);
int stage_cells_blocks(int M);
hipError_t launch_stage_cells(int dims, int geom, const HydroCols &h, int M, CellGeom *og, CellGeom2 *og2, CellFluid *of, double *ofc, double *otemp,
                              double *ogamma, StagePartial *partials, double *samples, int stride, int nsamp, hipStream_t stream);
// createHotCrossSection on the device (hot_table.hip; hot_x_section.c:82-133,324-400)
constexpr int HOT_TABLE_SUBSTREAMS = 256;   // sample k of an entry belongs to substream k % 256 (part of the table's definition)
struct HotTableParams {
    int n_ph_e, n_t;
    double log_ph_e_min, log_ph_e_max, log_t_min, log_t_max;
    long long calls;
    unsigned long long seed;
};
hipError_t launch_hot_table(const HotTableParams &p, double *table, hipStream_t stream);
// photonEmitCyclosynch, inject_single_switch == 0, on the device (inject.hip; mc_cyclosynch.c:1200-1460)
struct CsEmitParams {
    int dimensions, geometry, b_field_calc;
    double epsilon_b;
    double rmin, rmax, theta_min, theta_max;        // the shell of :1203-1204 and the thread's angle range
};
struct CsHookArgs { CsEmitParams p; HydroCols h; };      // (see launch_rank_loop)
hipError_t launch_cs_emit_count(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, double ph_weight_adjusted, unsigned long long attempt,
                                RngKey key, unsigned *count, unsigned long long *d_total, unsigned *d_flags, hipStream_t stream);
hipError_t launch_cs_emit_generate(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, double ph_weight_adjusted, RngKey key, const int *start,
                                   int n_emit, const int *null_slots, const PhotonDev &ph, hipStream_t stream);
// the hook of mcrat.c:786-808 after every pass of a cyclo-synchrotron frame (one workgroup on the context's stream) and the state it
// keeps on the device, so that the host reads a frame's progress back once per batch of passes: when the hook needs the host (the list
// has no null slot left; the rebinning is due) it parks the loop -- LoopState::done = LOOP_CS_HALT makes the step and event kernels of
// the passes already queued return at once -- and says why in `halt`
constexpr int LOOP_CS_HALT = 3;      // LoopState::done value
constexpr int CS_HALT_GROW = 1;      // nothing was done for this pass: double the list, launch the hook again with resume = 1
constexpr int CS_HALT_REBIN = 2;     // the pass is complete; rebinCyclosynchCompPhotons is due (mcrat.c:797-808)
constexpr int CS_HALT_HOOK = 3;      // rank pool, hook-kernel form: rank_loop_kernel parked the list after a pass the hook has to look at (cs_replace_pool_kernel)
struct CsFrame {
    int halt;
    int saved_done;                  // LoopState::done as the pass left it
    int emitted;                     // replacements since the host last looked (num_cyclosynch_ph_emit)
    int scatt_num;                   // scatt_cyclosynch_num_ph
    int max_photons;
    int pad;
    unsigned long long last_iteration;   // LoopState::iteration of the last pass the hook handled (queued passes after `done` are no-ops)
    double n_comptonized;            // mcrat.c:788, summed in pass order
};
hipError_t launch_cs_replace(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, RngKey key, LoopState *st, const PhotonDev &ph,
                             CsFrame *frame, int resume, hipStream_t stream);
// pool emission for the lists of a rank pool (inject.hip): the emission shell's cells with what photonEmitCyclosynch computes per cell, once per
// group of lists that share the shell, then one workgroup per list
struct alignas(8) CsShellCell { int cell; int pad; double integral; double volume; };
struct alignas(8) CsPoolEmit {
    int open, group;                 // in: takes part; which shell it emits into
    unsigned long long seed;         // in: the list's emission seed
    double weight_in, max_photons;   // in: ph_weight_suggest; rebin_e_perc * maximum_photons
    double weight_out;               // out: the adjusted weight
    int n_emit, error;               // out: photons emitted; 0 ok, 1 no weight fits, 2 fewer null slots than photons, 3 the list would outgrow its
                                     //      window of the pool, 4 more photons than the kernel's tables hold
};
hipError_t launch_cs_shell_flag(const CsEmitParams &p, const HydroDev &hy, unsigned *flag, unsigned long long *d_total, int *n_shell, hipStream_t stream);   // waits
hipError_t launch_cs_shell_write(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, const unsigned *flag, int n_shell, int *start, int *scratch,
                                 CsShellCell *out, unsigned *not_converged, hipStream_t stream);
hipError_t launch_cs_emit_pool(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, const PhotonDev &pool, int stride, int n_ranks, RankDesc *desc,
                               const CsShellCell *shell, int n_shell, CsPoolEmit *lists, int group, hipStream_t stream);
// phAbsCyclosynch for the open lists of a rank pool, one workgroup per list (staging.hip)
hipError_t launch_cs_absorb_pool(const CsParams &p, const PhotonDev &pool, int stride, int n_ranks, const RankDesc *desc, const int *open, const double *temp,
                                 const HydroCols &h, CsAbsPartial *per_list, hipStream_t stream);
// the hook for every parked list of a rank pool (one workgroup per list; a list out of null slots doubles inside its window of the pool)
hipError_t launch_cs_replace_pool(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, LoopState *states, const PhotonDev &pool, int stride,
                                  int n_ranks, RankDesc *desc, CsFrame *frames, hipStream_t stream);
// the list's null slots in ascending order (addToPhotonList's null_ph_indexes, photons.c:181-189): count per 256 slots, then write
hipError_t launch_null_count(const PhotonDev &ph, unsigned *block_count, unsigned long long *d_total, hipStream_t stream);
hipError_t launch_null_write(const PhotonDev &ph, const int *block_start, int *null_slots, hipStream_t stream);
// slots [first, first + count) become null photons (reallocatePhotonListMemory, photons.c:72-78)
hipError_t launch_null_fill(const PhotonDev &ph, int first, int count, hipStream_t stream);
// rebinCyclosynchCompPhotons on the device (inject.hip; mc_cyclosynch.c:246-712)
struct RebinRange {            // collect_photon_statistics :273-322, one per workgroup, finished on the host
    double p0_min, p0_max, theta_min, theta_max, phi_min, phi_max;
    int valid, synch;
};
struct RebinAxes {             // the three uniform histograms' ranges (:360-391) and the bin counts (:324-347)
    double e_lo, e_hi, t_lo, t_hi, p_lo, p_hi;
    int num_bins, num_bins_theta, num_bins_phi, total_bins, three;
};
int rebin_range_blocks(int n);
hipError_t launch_rebin_range(const PhotonDev &ph, int three, RebinRange *partials, hipStream_t stream);
// bin of every slot (-1: not rebinned; -2: outside the histograms, the reference's exit(1)) and the number of slots per bin
hipError_t launch_rebin_assign(const PhotonDev &ph, const RebinAxes &ax, int *bin_of, unsigned *bin_count, hipStream_t stream);
// slots of each bin (bin_start from the scan of bin_count); then one thread per bin: the weighted sums in slot order and the
// rebinned photon as a record (create_rebinned_photons :504-607)
hipError_t launch_rebin_fill(const PhotonDev &ph, const int *bin_of, const int *bin_start, unsigned *cursor, int *members, hipStream_t stream);
struct RebinRec {              // one rebinned photon (or an empty bin: valid == 0)
    double weight, p0, p1, p2, p3, r0, r1, r2, s0, s1, s2, s3, num_scatt;
    int valid, pad;
};
hipError_t launch_rebin_create(const PhotonDev &ph, const RebinAxes &ax, const int *bin_start, int *members, RebinRec *recs,
                               unsigned *empty_bins, hipStream_t stream);
// after the old photons are nullified: record b into the b-th null slot (addToPhotonList, photons.c:190-199)
hipError_t launch_rebin_place(const PhotonDev &ph, const RebinRec *recs, int total_bins, const int *null_slots, hipStream_t stream);
// setNullPhoton for every 'k' and 'c' photon (:573-581)
hipError_t launch_rebin_nullify(const PhotonDev &ph, hipStream_t stream);
// the rebinning of many lists of a rank pool in two launches, one workgroup per list (inject.hip, rebin_pool_kernel)
struct RebinPoolList {
    int first, n;              // the list's window in the pool's slots
    RebinAxes ax;              // (second launch) the histograms, fixed by the host from the first launch's ranges
    unsigned long long scratch;   // byte offset of the list's scratch (rebin_pool_scratch_bytes) in the buffer handed to the second launch
    int status;                // out: 0 rebinned; 1 a photon maps outside the histograms (list untouched); 2 fewer null slots than bins
    int empty_bins, n_null;    // out
    int pad;
};
size_t rebin_pool_scratch_bytes(int n, int total_bins);
hipError_t launch_rebin_pool_range(const PhotonDev &pool, int three, const RebinPoolList *lists, int n_lists, RebinRange *out, hipStream_t stream);
hipError_t launch_rebin_pool(const PhotonDev &pool, RebinPoolList *lists, int n_lists, char *scratch, hipStream_t stream);
hipError_t grid_build(const GridPlan &p, const CellGeom *geom, const CellGeom2 *geom2, const CellFluid *fluid, const double *fluid_c, int M,
                      unsigned *count, int *start, int *scan_scratch, int *entries, FatCell *cells, BucketDir *dir, long long nb,
                      long long total, hipStream_t stream);
hipError_t launch_lookup(const KernelConfig &kc, const HydroDev &hy, int n, const double *a0, const double *a1,
                         const double *a2, int *out, hipStream_t stream);
// physics.hpp's functions on arrays (functions.hip; mcrat_hip_eval_function)
hipError_t launch_eval_function(int fn, int stokes, int n, const double *in, double *out, uint64_t seed, uint32_t stream_id, hipStream_t stream);

}  // namespace mcrat
