// kernels.hip -- HIP kernels of the photon loop for gfx950 (wave64).
//
//   step_kernel    one lane per photon-slot pair, coalesced 16-B SoA loads: pending updatePhotonPosition
//                  (mclib.c:1054) -> domain + in-cell test of findContainingHydroCell (mclib.c:485-507) ->
//                  free-time draw of calcMeanFreePath (mclib.c:675-687), fused in ONE pass over the photons.
//                  Slots that left their cell or need tau recomputed (about 1 % per iteration) are
//                  ballot-compacted into an LDS queue and finished by the same workgroup with dense lanes
//                  (cell search, comoving 4-momentum, optical depth: mclib.c:528-586, optical_depth.c:7-59).
//                  The argsort of mclib.c:702-712 is replaced by min-selection: only the prefix of the sorted
//                  list is ever consumed (mclib.c:1128-1133).  HBM-bound; this is the kernel priced against
//                  the roofline.
//   event_kernel   one workgroup: sorts the iteration's shortlist of early candidates, walks it as
//                  photonEvent does (mclib.c:1107-1356), scatters at most one photon and does the time
//                  bookkeeping of mcrat.c:777-846.
//   flush_kernel   applies the advance still pending when a run stops.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "device_types.hpp"
#include "launch.hpp"
#include "sc_exchange.hpp"
#include "physics.hpp"
#include "photon_cols.hpp"
#include "rng.hpp"
#include "cs_device.hpp"

// This file is a template for six translation units (kernels_d{0,1,2}.hip, kernels_table_d{0,1,2}.hip): one per
// TAU_CALCULATION (DIRECT, optical_depth.c:125-127 / TABLE, :132-149) and DIMENSIONS (TWO, TWO_POINT_FIVE, THREE), each
// compiled into its own namespace, side by side.  The reference makes the same choices at compile time
// (mcrat_input.h); a run-time TAU branch costs the DIRECT kernels registers they do not have (step_kernel sits exactly
// at its 168-VGPR budget), and one TU for all of it takes minutes to compile.  launchers.hip picks the namespace.
#if !defined(MCRAT_TAU_TABLE_TU) || !defined(MCRAT_TU_DIMS)
#error "compile kernels.hip through kernels_d*.hip / kernels_table_d*.hip"
#endif
#define MCRAT_NS_CAT2(a, b) a##b
#define MCRAT_NS_CAT(a, b) MCRAT_NS_CAT2(a, b)
#if MCRAT_TAU_TABLE_TU
#define MCRAT_TU_NS MCRAT_NS_CAT(tau_table_d, MCRAT_TU_DIMS)
#else
#define MCRAT_TU_NS MCRAT_NS_CAT(tau_direct_d, MCRAT_TU_DIMS)
#endif

namespace mcrat {
namespace MCRAT_TU_NS {

constexpr bool TABLE_MODE = MCRAT_TAU_TABLE_TU != 0;

// Diagnostic build only (-DMCRAT_DIAG, tools/diag_build.sh): g_diag bits knock out parts of the step kernel so that
// their cost can be read off the kernel trace.  Results are wrong with any bit set; the product build has no such code.
#ifdef MCRAT_DIAG
__device__ int g_diag = 0;
#define MC_DIAG(bit) (g_diag & (bit))
#else
#define MC_DIAG(bit) 0
#endif
#ifdef MCRAT_DIAG
#define MC_STAMP(st, k)                                         \
    do {                                                        \
        __builtin_amdgcn_sched_barrier(0);                      \
        (st)->stamps[k] = (long long)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                      \
    } while (0)
#else
#define MC_STAMP(st, k) do { } while (0)
#endif
enum { DIAG_SKIP_SLOW = 1, DIAG_SKIP_INCELL = 2, DIAG_SKIP_SAMPLE = 4, DIAG_SKIP_ADVANCE = 8, DIAG_SKIP_PHILOX = 16, DIAG_SKIP_COORDS = 32,
       DIAG_SLOW_NO_SEARCH = 64, DIAG_SLOW_NO_PHYSICS = 128, DIAG_SLOW_EMPTY = 256 };

// ------------------------------------------------------------------ top-K of (time, slot), ascending, ties by slot
__device__ __forceinline__ bool cand_less(double ta, int ia, double tb, int ib)
{
    return (ta < tb) || (ta == tb && ia < ib);
}

struct TopK {
    double t[TOPK];
    int i[TOPK];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < TOPK; ++k) { t[k] = INFINITY; i[k] = INT_MAX; }
    }
    __device__ __forceinline__ void insert(double tt, int ii)
    {
        if (tt != tt) tt = INFINITY;   // a NaN free time (cell at rest, optical_depth.c:46) never wins
        if (!cand_less(tt, ii, t[TOPK - 1], i[TOPK - 1])) return;
        t[TOPK - 1] = tt; i[TOPK - 1] = ii;
#pragma unroll
        for (int k = TOPK - 1; k > 0; --k) {
            if (cand_less(t[k], i[k], t[k - 1], i[k - 1])) {
                const double a = t[k]; t[k] = t[k - 1]; t[k - 1] = a;
                const int b = i[k]; i[k] = i[k - 1]; i[k - 1] = b;
            }
        }
    }
    __device__ __forceinline__ void pop()
    {
#pragma unroll
        for (int k = 0; k < TOPK - 1; ++k) { t[k] = t[k + 1]; i[k] = i[k + 1]; }
        t[TOPK - 1] = INFINITY; i[TOPK - 1] = INT_MAX;
    }
};

__device__ __forceinline__ void wave_min_pair(double &t, int &i)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ot = __shfl_xor(t, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (cand_less(ot, oi, t, i)) { t = ot; i = oi; }
    }
}

// wave64 lexicographic min of (t, i) on the VALU's DPP path (no LDS traffic): prefix-min inside each row of
// 16 lanes (row_shr 1,2,4,8), then across rows (row_bcast 15, 31); lane 63 ends with the wave minimum,
// which is broadcast through readlane.  Lanes without a DPP source keep their own value, harmless for a min.
template <int CTRL>
__device__ __forceinline__ void dpp_min_step(double &t, int &i)
{
    const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
    const int lo = (int)(unsigned)(tb & 0xffffffffu), hi = (int)(unsigned)(tb >> 32);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    const int oi = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false);
    const double ot = __longlong_as_double((long long)(((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo));
    if (cand_less(ot, oi, t, i)) { t = ot; i = oi; }
}

__device__ __forceinline__ void wave_min_pair_dpp(double &t, int &i)
{
    dpp_min_step<0x111>(t, i);   // row_shr:1
    dpp_min_step<0x112>(t, i);   // row_shr:2
    dpp_min_step<0x114>(t, i);   // row_shr:4
    dpp_min_step<0x118>(t, i);   // row_shr:8
    dpp_min_step<0x142>(t, i);   // row_bcast:15
    dpp_min_step<0x143>(t, i);   // row_bcast:31
    const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(tb & 0xffffffffu), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(tb >> 32), 63);
    t = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    i = __builtin_amdgcn_readlane(i, 63);
}

// the K smallest of all lanes' lists -> out[0..K) (ascending).  NW = waves in the workgroup, NW*TOPK <= 64.
template <int NW>
__device__ __forceinline__ void block_topk(TopK &mine, Cand (*s_w)[TOPK], Cand *out)
{
    static_assert(NW * TOPK <= 64, "second stage is one wave");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < TOPK; ++r) {
        double ht = mine.t[0];
        int hi = mine.i[0];
        wave_min_pair(ht, hi);
        if (mine.i[0] == hi && mine.t[0] == ht) mine.pop();
        if (lane == 0) { s_w[w][r].t = ht; s_w[w][r].idx = hi; }
    }
    __syncthreads();
    if (w == 0) {
        TopK m2;
        m2.init();
        if (lane < NW * TOPK) m2.insert(s_w[lane / TOPK][lane % TOPK].t, s_w[lane / TOPK][lane % TOPK].idx);
#pragma unroll
        for (int r = 0; r < TOPK; ++r) {
            double ht = m2.t[0];
            int hi = m2.i[0];
            wave_min_pair(ht, hi);
            if (m2.i[0] == hi && m2.t[0] == ht) m2.pop();
            if (lane == 0) { out[r].t = ht; out[r].idx = hi; out[r].pad = 0; }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ step kernel
// x / C_LIGHT, correctly rounded, without the ~30-instruction IEEE division sequence: q = x*(1/c), one FMA for
// the exact residual, one FMA to correct (Markstein's division by a constant).  Bit-identical to x / C_LIGHT on
// 4e8 random doubles over 2^-40..2^80 (checked on the host); see DESIGN.md "Numerics".
__device__ __forceinline__ double div_by_c(double x)
{
    constexpr double INV_C = 1.0 / C_LIGHT;
    const double q = x * INV_C;
    const double r = fma(-q, C_LIGHT, x);
    return fma(r, INV_C, q);
}

// free time of a photon in a known cell: mclib.c:675-687
__device__ __forceinline__ double sample_free_time(double ntau /* -1.0 / tau */, uint64_t bits)
{
    const double rnd = bits_to_uniform_pos(bits);
    const double mfp = ntau * log(rnd);
    return div_by_c(mfp);
}

// blocks b and b+8 share an XCD (and its L2): give every XCD one contiguous eighth of the photon chunks so
// that the cell records a workgroup gathers are shared with its L2 neighbours (placement is a speed matter only)
__device__ __forceinline__ int xcd_contiguous_chunk(int b, int G)
{
    const int x = b & 7, q = G >> 3, rem = G & 7;
    return x * q + (x < rem ? x : rem) + (b >> 3);
}

struct MinCand {
    double t;
    int i;
    __device__ __forceinline__ void init() { t = INFINITY; i = INT_MAX; }
    __device__ __forceinline__ void offer(double tt, int ii)
    {
        if (tt != tt) tt = INFINITY;   // a NaN free time (cell at rest, optical_depth.c:46) never wins
        if (cand_less(tt, ii, t, i)) { t = tt; i = ii; }
    }
};

__device__ __forceinline__ void shortlist_push(Shortlist *sl, double t, int i)
{
    const int pos = atomicAdd(&sl->count, 1);
    if (pos < SHORTLIST_CAP) { sl->items[pos].t = t; sl->items[pos].idx = i; sl->items[pos].pad = 0; }
}

constexpr int Q_RECALC_ONLY = (int)0x80000000;   // queue entry flag: the slot is still in its cell, only tau is stale
constexpr int Q_DEFERRED = 0x40000000;           // rank_loop_kernel, TABLE: the slot's look-up fell off the table; a wavefront integrates it after the queue

// streaming half of an iteration for one slot.  Returns the free time, or sets `queue` (0: no, 1: re-locate,
// 2: recalc tau only) when the slot must go through the slow path; the time returned then is a placeholder.
template <int DIMS, int GEOM, bool FORCE, class PH>
__device__ __forceinline__ double fast_one(const PH &ph, const HydroDev &hy, int i, unsigned fl, int cell,
                                           double r0, double r1, double r2, double ntau, uint64_t bits, int &queue, int &bucket,
                                           double &a0, double &a1, double &a2)
{
    queue = 0;
    bucket = -1;
    a0 = a1 = a2 = 0;
    if (!(fl & FLAG_VALID)) return INFINITY;
    if (MC_DIAG(DIAG_SKIP_COORDS)) { a0 = r0 + r1; a1 = r2; a2 = 0; }
    else phys::hydro_coords<DIMS, GEOM>(r0, r1, r2, a0, a1, a2);
    if (phys::in_domain<DIMS>(hy, a0, a1, a2) && (cell != -1)) {               // mclib.c:492-505
        if constexpr (FORCE) {
            queue = 1;                                                           // mclib.c:528, find_nearest_block_switch == 1
        } else {
            if (MC_DIAG(DIAG_SKIP_INCELL)) queue = 0;
            else if (!phys::check_in_block<DIMS>(hy, cell, a0, a1, a2)) queue = 1;    // mclib.c:507,528
            else if (fl & FLAG_RECALC) {                                         // mclib.c:668
                if (fl & FLAG_TAU_FRESH) {
                    ph.flags(i) = (unsigned char)(fl & ~(FLAG_RECALC | FLAG_TAU_FRESH));
                    ph.tau(i) = ph.tau_next(i);
                } else {
                    queue = 2;
                }
            }
        }
        if (queue == 1) bucket = phys::grid_bucket_of<DIMS>(hy.grid, a0, a1, a2);        // where the slow path will search
        if (queue) return INFINITY;
        if (MC_DIAG(DIAG_SKIP_SAMPLE)) return 1e-3 + (double)i * 1e-12 + (double)(bits & 1) * 0.0 + ntau * 0.0;
        return sample_free_time(ntau, bits);
    }
    if (cell != -1) ph.idx(i) = -1;                                              // mclib.c:592
    return 1e12 / C_LIGHT;                                                       // mclib.c:620,684
}

// slow path of one queued slot: mclib.c:528-586 (re-location, comoving momentum, optical depth) or the
// recalc_properties branch of calcMeanFreePath (mclib.c:668-673), then its free-time draw.  Two dependent load
// rounds: {photon columns, bucket range} then {bucket entries | the cell's fluid record}.
// `bits` is the slot's free-path draw of this pass: phase 1 has computed the pair's Philox block anyway (one block serves two
// slots) and hands the 64 bits over -- for queued slots through the slot's time_to_scatter entry, which this function overwrites.
// (LOGS: `bits` is not the draw but the bit pattern of log(u) of the draw -- rank_loop_kernel's waves that sit out the event walk compute a pass's logarithms ahead of time)
__device__ __forceinline__ double free_time_from_log(double ntau, double log_u) { return div_by_c(ntau * log_u); }   // = sample_free_time, its log at hand

// TAU_CALCULATION == TABLE: what a look-up that falls off the table needs to integrate the cross section afresh (physics.hpp: table_fallback_*;
// hot_x_section.c:563-599) -- the key of the integral's substreams (the pass the reference would evaluate it in, the slot's global index) and who
// integrates: this lane on its own (mode 0), or the caller, who brings the slot to a whole wavefront (mode 1: slow_one hands (eps, theta) back and
// leaves the slot as it was; mode 2: slow_one is told the integral)
struct TableSite {
    uint64_t seed, pass;
    uint32_t stream, slot_base;
    int first;                   // the index (as slow_one sees it) of the list's slot 0
    int mode;
    bool deferred;
    double norm, eps, theta;
    __device__ __forceinline__ TableSite(const RngKey &key, unsigned long long iter, int first_ = 0, int mode_ = 0)
        : seed(key.seed), pass(iter), stream(key.stream), slot_base(key.slot_base), first(first_), mode(mode_), deferred(false), norm(1), eps(0), theta(0) {}
};

template <int DIMS, int GEOM, bool LOGS = false, class PH>
__device__ __forceinline__ double slow_one(const PH &ph, const HydroDev &hy, int i, bool relocate, int bucket, bool count_it,
                                           uint64_t bits, int &relocated, int &not_found, TableSite &ts)
{
    const double r0 = ph.r0(i), r1 = ph.r1(i), r2 = ph.r2(i);
    const double p0 = ph.p0(i), p1 = ph.p1(i), p2 = ph.p2(i), p3 = ph.p3(i);
    const unsigned fl = ph.flags(i);
    int cell;
    bool need_tau = (fl & FLAG_RECALC) != 0;
    bool new_cell = false;
    double fa = 0, fb = 0, fc = 0, fw = 0, fnsig = 0, fgam = 1, fkf = 0.5;
    double a0 = 0, a1 = 0, a2 = 0;
    if (MC_DIAG(DIAG_SLOW_EMPTY)) { ph.tts(i) = 1e-3 + i * 1e-12; return 1e-3 + i * 1e-12; }
    if (MC_DIAG(DIAG_SLOW_NO_SEARCH)) relocate = false;
    if (relocate) {
        phys::hydro_coords<DIMS, GEOM>(r0, r1, r2, a0, a1, a2);
        FatCell hit;
        cell = phys::find_in_bucket<DIMS>(hy.grid, bucket, a0, a1, a2, hit);     // mclib.c:534
        ph.idx(i) = cell;                                                        // mclib.c:536
        if (cell != -1) {
            fa = hit.a; fb = hit.b; fc = hit.c; fw = hit.w; fnsig = hit.nsig; fgam = hit.gam; fkf = phys::kf_of_gamma(hit.gam);
            new_cell = true;
            need_tau = true;                                                     // mclib.c:570
            if (count_it) relocated += 1;                                        // mclib.c:579,608-611
        } else {
            not_found += 1;                                                      // mclib.c:583
        }
    } else {
        cell = ph.idx(i);
        if (cell != -1) {
            const CellFluid f = hy.fluid[cell];
            fa = f.a; fb = f.b; fc = f.c; fw = f.w; fnsig = f.nsig;
        }
    }
    double t;
    if (MC_DIAG(DIAG_SLOW_NO_PHYSICS)) need_tau = false;
    if (cell != -1) {
        double ntau;
        if (need_tau) {
            double cphi, sphi;
            if (new_cell) phys::relocation_azimuth<DIMS, GEOM>(r0, r1, a0, cphi, sphi);   // photon azimuth, mclib.c:549-552
            else phys::cos_sin_of_atan2(r1, r0, cphi, sphi);                              // optical_depth.c:30-35
            double beta[3];
            phys::beta_from_record<DIMS>(fa, fb, fc, cphi, sphi, beta);
            double comv0 = 0;
            if (new_cell) {
                const double lab[4] = {p0, p1, p2, p3};
                double comv[4];
                phys::boost_with<true>(beta, fgam, fkf, lab, comv);              // mclib.c:558
                ph.c0(i) = comv[0]; ph.c1(i) = comv[1]; ph.c2(i) = comv[2]; ph.c3(i) = comv[3];
                comv0 = comv[0];
            }
            double norm = 1.0;
            if constexpr (TABLE_MODE) {                                          // TAU_CALCULATION == TABLE, optical_depth.c:58
                if (!new_cell) comv0 = ph.c0(i);
                if (ts.mode == 2) {
                    norm = ts.norm;
                } else if (phys::thermal_cross_section_lookup(hy, comv0, hy.temp[cell], norm, ts.eps, ts.theta)) {   // off the table: hot_x_section.c:563-599
                    if (ts.mode == 1) {
                        ts.deferred = true;                                      // (idx and comv_p stored above are what the second visit stores again)
                        if (new_cell && count_it) relocated -= 1;
                        return 0.0;
                    }
                    norm = phys::table_fallback_lane(ts.eps, ts.theta, ts.seed, ts.pass, ts.slot_base + (uint32_t)(i - ts.first), ts.stream, hy.hot_fallback_calls);
                    atomicAdd(hy.table_fallbacks, 1);
                }
            }
            const double tau = phys::optical_depth_staged(beta, fw, fnsig, p1, p2, p3, norm);
            ntau = -phys::rcp_nr(tau);
            ph.tau(i) = tau;
            ph.ntau(i) = ntau;
            if (fl & FLAG_RECALC) ph.flags(i) = (unsigned char)(fl & ~(FLAG_RECALC | FLAG_TAU_FRESH));  // mclib.c:571-576,672
        } else {
            ntau = ph.ntau(i);
        }
        t = LOGS ? free_time_from_log(ntau, __longlong_as_double((long long)bits)) : sample_free_time(ntau, bits);
    } else {
        t = 1e12 / C_LIGHT;
    }
    ph.tts(i) = t;
    return t;
}

// K slots of one thread through the re-location branch of the slow path AT ONCE: the same loads, the same arithmetic in the same
// order as slow_one -- the same bits -- but written as one straight-line block over the K slots (no branch between the stages:
// physics.hpp's helpers select instead of branching), so that the K dependent chains (two gathers, six square roots, sixteen
// divisions, a logarithm each) interleave in the issue slots one chain leaves empty.  Where most photons change cell between two
// events -- optically thin frames, and the forced pass of every frame (mcrat.c:756) -- a pass is this function for every slot, and
// two waves per SIMD hide little of one chain's latency.
// Takes a slot only when the bucket's hint settles it (one 96-B entry, the cell provably the only one that holds the point:
// find_in_bucket); todo[k] stays set for the others -- list walks, points outside every cell -- and the caller falls back to slow_one.
// (DIRECT optical depths only: the TABLE build keeps slow_one.)
// `probe` (rank_loop_kernel's fused passes): the caller has NOT tested the slots against their cached cells -- the entry the hint
// names settles that too.  A point well inside the hinted cell is in no other cell (device_types.hpp, BucketDir), so if that cell is the
// cached one the slot has stayed (checkInBlock, geometry.c:394, would have said so) and nothing is stored for it (probe->same[k] set), and
// if it is another one the slot has left its cell (mclib.c:507,528).  This saves every slot the 32-B gather of its cached cell's geometry.
struct LockstepProbe {
    const int *cached;           // [K] the slots' cached cells
    bool force;                  // forced pass (mcrat.c:756): a slot is re-located even into the cell it is in
    bool *same;                  // [K] out: settled, and still in its cached cell
};

template <int DIMS, int GEOM, int K, bool LOGS = false, class PH>
__device__ __forceinline__ void relocate_lockstep(const PH &ph, const HydroDev &hy, const int (&slot)[K], bool (&todo)[K],
                                                  const double (&r0)[K], const double (&r1)[K], const double (&a0)[K], const double (&a1)[K],
                                                  const double (&a2)[K], const int (&code)[K], const uint64_t (&bits)[K], const unsigned (&fl)[K],
                                                  bool count_it, double (&t)[K], int &relocated, const LockstepProbe *probe = nullptr)
{
    static_assert(!TABLE_MODE, "DIRECT optical depths");
    const GridDev &g = hy.grid;
    BucketDir d[K];
    double p0[K], p1[K], p2[K], p3[K];
    bool take[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        take[k] = todo[k] && code[k] >= 0;
        d[k] = g.dir[take[k] ? (code[k] & GRID_CODE_BUCKET_MASK) : 0];
        const int i = slot[k];
        p0[k] = ph.p0(i); p1[k] = ph.p1(i); p2[k] = ph.p2(i); p3[k] = ph.p3(i);
    }
    FatCell f[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned hint = (d[k].hints >> (4 * ((code[k] >> GRID_CODE_OCT_SHIFT) & 7))) & 15u;
        take[k] = take[k] & (d[k].n > 0) & ((code[k] & GRID_CODE_HINT_OK) != 0) & (hint != GRID_NO_HINT);
        f[k] = g.cells[take[k] ? d[k].e0 + (int)hint : 0];
    }
    double comv[K][4], tau[K], ntau[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        take[k] = take[k] & phys::well_in_fat_cell<DIMS>(f[k], a0[k], a1[k], a2[k]);
        double cphi, sphi;
        phys::relocation_azimuth<DIMS, GEOM>(r0[k], r1[k], a0[k], cphi, sphi);  // photon azimuth, mclib.c:549-552
        double beta[3];
        phys::beta_from_record<DIMS>(f[k].a, f[k].b, f[k].c, cphi, sphi, beta);
        const double lab[4] = {p0[k], p1[k], p2[k], p3[k]};
        phys::boost_with<true>(beta, f[k].gam, phys::kf_of_gamma(f[k].gam), lab, comv[k]);   // mclib.c:558
        tau[k] = phys::optical_depth_staged(beta, f[k].w, f[k].nsig, p1[k], p2[k], p3[k], 1.0);
        ntau[k] = -phys::rcp_nr(tau[k]);
        t[k] = LOGS ? free_time_from_log(ntau[k], __longlong_as_double((long long)bits[k])) : sample_free_time(ntau[k], bits[k]);
    }
    // (the K chains above are meant to run side by side: without this the compiler sinks each slot's arithmetic into that slot's `take` branch
    // below and the chains run one after the other)
#pragma unroll
    for (int k = 0; k < K; ++k)
        asm volatile("" : "+v"(comv[k][0]), "+v"(comv[k][1]), "+v"(comv[k][2]), "+v"(comv[k][3]), "+v"(tau[k]), "+v"(ntau[k]), "+v"(t[k]));
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (!take[k]) continue;
        if (probe) {
            probe->same[k] = !probe->force && f[k].cell == probe->cached[k];
            if (probe->same[k]) { todo[k] = false; continue; }
        }
        const int i = slot[k];
        ph.idx(i) = f[k].cell;                                                   // mclib.c:536
        ph.c0(i) = comv[k][0]; ph.c1(i) = comv[k][1]; ph.c2(i) = comv[k][2]; ph.c3(i) = comv[k][3];
        ph.tau(i) = tau[k];
        ph.ntau(i) = ntau[k];
        if (fl[k] & FLAG_RECALC) ph.flags(i) = (unsigned char)(fl[k] & ~(FLAG_RECALC | FLAG_TAU_FRESH));    // mclib.c:571-576
        ph.tts(i) = t[k];
        if (count_it) relocated += 1;                                            // mclib.c:579,608-611
        todo[k] = false;
    }
}

// the streaming loads of one slot pair
struct PairIn {
    double2 R0, R1, R2, U0, U1, U2, NTAU;
    int2 ID;
    uchar2 FL;
};

__device__ __forceinline__ PairIn load_pair(const PhotonDev &ph, int i0, bool need_u)
{
    PairIn in;
    in.R0 = *reinterpret_cast<const double2 *>(ph.r0 + i0);
    in.R1 = *reinterpret_cast<const double2 *>(ph.r1 + i0);
    in.R2 = *reinterpret_cast<const double2 *>(ph.r2 + i0);
    in.NTAU = *reinterpret_cast<const double2 *>(ph.ntau + i0);
    in.ID = *reinterpret_cast<const int2 *>(ph.idx + i0);
    in.FL = *reinterpret_cast<const uchar2 *>(ph.flags + i0);
    if (need_u) {
        in.U0 = *reinterpret_cast<const double2 *>(ph.u0 + i0);
        in.U1 = *reinterpret_cast<const double2 *>(ph.u1 + i0);
        in.U2 = *reinterpret_cast<const double2 *>(ph.u2 + i0);
    } else {
        in.U0 = in.U1 = in.U2 = make_double2(0, 0);
    }
    return in;
}

// (launch bounds: three waves per SIMD -- 768 resident workgroups -- is what the grid is sized for; the kernel sits
// at that register budget, 168 VGPRs)
// One workgroup streams its chunks of 512 slots (phase 1, two chunks in flight per thread), collecting in an
// LDS queue the few slots that need the slow path, then finishes those with dense lanes (phase 2) and
// publishes its minimum.  On the forced pass of a new frame every slot takes the slow path, in line.
template <int DIMS, int GEOM, bool FORCE>
#ifndef STEP_WAVES_PER_SIMD
#define STEP_WAVES_PER_SIMD 3
#endif
__global__ __launch_bounds__(STEP_BLOCK, STEP_WAVES_PER_SIMD) void step_kernel(PhotonDev ph, HydroDev hy, LoopState *st, RngKey key,
                                                          Cand *__restrict__ block_min, Shortlist *sl)
{
    __shared__ int s_qn;
    __shared__ int s_q[STEP_QCAP];          // slot | Q_RECALC_ONLY
    __shared__ int s_qb[STEP_QCAP];         // bucket to search (re-locating slots)
    __shared__ double s_wt[STEP_BLOCK / 64];
    __shared__ int s_wi[STEP_BLOCK / 64];
    if (st->done) return;
    const PtrCols pc(ph);                        // the accessor the shared device functions take (photon_cols.hpp)
    const int nseg = st->nseg;
    const int skip = st->skip_idx;
    const unsigned long long iter = st->iteration;
    const double t_cut = st->t_cut;
    const int G = gridDim.x;
    const int lane = threadIdx.x & 63;

    MinCand best;
    best.init();
    int relocated = 0, not_found = 0;
    TableSite ts(key, iter);                     // TABLE: a look-up off the table is integrated by the lane that meets it (list mode has no wavefront to spare)
    const int nchunks = ph.n_pad / (2 * STEP_BLOCK);
    if (threadIdx.x == 0) s_qn = 0;
    for (int e = threadIdx.x; e < STEP_QCAP; e += STEP_BLOCK) s_q[e] = -1;   // -1: hole left by a wave that overflowed
    __syncthreads();

    auto process = [&](PairIn &in, int pair) {
        const int i0 = pair << 1;
        // pending updatePhotonPosition of the previous iteration, mclib.c:1067-1095: one segment per
        // candidate that photonEvent walked (mclib.c:1138,1332); u = (p * (1/p0)) * C_LIGHT is stored
        if (nseg > 0 && !MC_DIAG(DIAG_SKIP_ADVANCE)) {
            const bool m0 = (in.FL.x & FLAG_MOVES) && (i0 != skip);
            const bool m1 = (in.FL.y & FLAG_MOVES) && (i0 + 1 != skip);
            for (int s = 0; s < nseg; ++s) {
                const double t = st->seg[s];
                if (m0) { in.R0.x += in.U0.x * t; in.R1.x += in.U1.x * t; in.R2.x += in.U2.x * t; }
                if (m1) { in.R0.y += in.U0.y * t; in.R1.y += in.U1.y * t; in.R2.y += in.U2.y * t; }
            }
            *reinterpret_cast<double2 *>(ph.r0 + i0) = in.R0;
            *reinterpret_cast<double2 *>(ph.r1 + i0) = in.R1;
            *reinterpret_cast<double2 *>(ph.r2 + i0) = in.R2;
        }
        // one Philox block serves both slots of the pair
        Philox4 blk;
        if (MC_DIAG(DIAG_SKIP_PHILOX)) { blk.w[0] = pair * 2654435761u; blk.w[1] = pair ^ 0x9e3779b9u; blk.w[2] = ~blk.w[0]; blk.w[3] = blk.w[1] + 7u; }
        else blk = keyed_block(key.seed, iter, (uint32_t)pair + (key.slot_base >> 1), RNG_FREEPATH, key.stream);
        const uint64_t bits0 = (uint64_t)blk.w[0] | ((uint64_t)blk.w[1] << 32);
        const uint64_t bits1 = (uint64_t)blk.w[2] | ((uint64_t)blk.w[3] << 32);

        int q0, q1, b0, b1;
        double2 T;
        double A0[2], A1[2], A2[2];
        T.x = fast_one<DIMS, GEOM, FORCE>(pc, hy, i0, in.FL.x, in.ID.x, in.R0.x, in.R1.x, in.R2.x, in.NTAU.x, bits0, q0, b0, A0[0], A1[0], A2[0]);
        T.y = fast_one<DIMS, GEOM, FORCE>(pc, hy, i0 + 1, in.FL.y, in.ID.y, in.R0.y, in.R1.y, in.R2.y, in.NTAU.y, bits1, q1, b1, A0[1], A1[1], A2[1]);
        if constexpr (FORCE) {
            if constexpr (!TABLE_MODE) {             // both slots of the pair in lockstep; what the hints do not settle goes on below
                const int slot[2] = {i0, i0 + 1}, code[2] = {b0, b1};
                bool todo[2] = {q0 != 0, q1 != 0};
                const double R0[2] = {in.R0.x, in.R0.y}, R1[2] = {in.R1.x, in.R1.y};
                const uint64_t bits[2] = {bits0, bits1};
                const unsigned fl[2] = {in.FL.x, in.FL.y};
                double tt[2];
                int dummy = 0;
                if (todo[0] || todo[1]) {
                    relocate_lockstep<DIMS, GEOM, 2>(pc, hy, slot, todo, R0, R1, A0, A1, A2, code, bits, fl, false, tt, dummy);
                    if (q0 && !todo[0]) { T.x = tt[0]; q0 = 0; }
                    if (q1 && !todo[1]) { T.y = tt[1]; q1 = 0; }
                }
            }
            if (q0) T.x = slow_one<DIMS, GEOM>(pc, hy, i0, true, b0, false, bits0, relocated, not_found, ts);
            if (q1) T.y = slow_one<DIMS, GEOM>(pc, hy, i0 + 1, true, b1, false, bits1, relocated, not_found, ts);
            q0 = q1 = 0;
        } else {
            // ballot-compact the slots that need the slow path into the workgroup's LDS queue (one LDS atomic
            // per wave); if the queue is full the wave finishes its own slots in line
            const unsigned long long m0 = __ballot(q0 != 0), m1 = __ballot(q1 != 0);
            if (m0 | m1) {
                const int c0 = __popcll(m0), c1 = __popcll(m1);
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_qn, c0 + c1);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base + c0 + c1 <= STEP_QCAP) {
                    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
                    if (q0) { const int e = base + __popcll(m0 & below); s_q[e] = i0 | (q0 == 2 ? Q_RECALC_ONLY : 0); s_qb[e] = b0; }
                    if (q1) { const int e = base + c0 + __popcll(m1 & below); s_q[e] = (i0 + 1) | (q1 == 2 ? Q_RECALC_ONLY : 0); s_qb[e] = b1; }
                } else {
                    if (q0) { T.x = slow_one<DIMS, GEOM>(pc, hy, i0, q0 == 1, b0, true, bits0, relocated, not_found, ts); q0 = 0; }
                    if (q1) { T.y = slow_one<DIMS, GEOM>(pc, hy, i0 + 1, q1 == 1, b1, true, bits1, relocated, not_found, ts); q1 = 0; }
                }
            }
        }
        if (q0) T.x = __longlong_as_double((long long)bits0);   // queued slots: the draw, for phase 2 (which stores the free time)
        if (q1) T.y = __longlong_as_double((long long)bits1);
        *reinterpret_cast<double2 *>(ph.tts + i0) = T;
        if ((in.FL.x & FLAG_VALID) && !q0) { best.offer(T.x, i0); if (T.x < t_cut) shortlist_push(sl, T.x, i0); }
        if ((in.FL.y & FLAG_VALID) && !q1) { best.offer(T.y, i0 + 1); if (T.y < t_cut) shortlist_push(sl, T.y, i0 + 1); }
    };

    for (int chunk = xcd_contiguous_chunk(blockIdx.x, G); chunk < nchunks; chunk += G) {
        const int pa = chunk * STEP_BLOCK + threadIdx.x;
        PairIn A = load_pair(ph, pa << 1, nseg > 0);
        process(A, pa);
    }

    if constexpr (!FORCE) {
        __syncthreads();
        int qn = s_qn;
        if (qn > STEP_QCAP) qn = STEP_QCAP;                  // the surplus was finished in line
        if (MC_DIAG(DIAG_SKIP_SLOW)) qn = 0;
        for (int e = threadIdx.x; e < qn; e += STEP_BLOCK) {
            const int entry = s_q[e];
            if (entry == -1) continue;
            const int i = entry & ~Q_RECALC_ONLY;
            const uint64_t bits = (uint64_t)__double_as_longlong(ph.tts[i]);
            const double t = slow_one<DIMS, GEOM>(pc, hy, i, !(entry & Q_RECALC_ONLY), s_qb[e], true, bits, relocated, not_found, ts);
            best.offer(t, i);
            if (t < t_cut) shortlist_push(sl, t, i);
        }
    }

    // workgroup minimum -> block_min[blockIdx.x]
    wave_min_pair_dpp(best.t, best.i);
    if (lane == 0) { s_wt[threadIdx.x >> 6] = best.t; s_wi[threadIdx.x >> 6] = best.i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinCand m;
        m.init();
#pragma unroll
        for (int w = 0; w < STEP_BLOCK / 64; ++w) m.offer(s_wt[w], s_wi[w]);
        Cand c;
        c.t = m.t; c.idx = m.i; c.pad = 0;
        block_min[blockIdx.x] = c;
    }
    if (relocated) atomicAdd(reinterpret_cast<unsigned long long *>(&st->n_relocated), (unsigned long long)relocated);
    if (not_found) atomicAdd(reinterpret_cast<unsigned long long *>(&st->not_found), (unsigned long long)not_found);
}

// ------------------------------------------------------------------ event kernel
enum { EV_RUNNING = 0, EV_DONE = 1, EV_NEED_MORE = 2 };
#ifdef MCRAT_NO_WAVE_WALK
constexpr bool WAVE_WALK = false;    // photonEvent's walk on one lane
#else
constexpr bool WAVE_WALK = true;     // ... on a whole wavefront with the same values in every lane (event_block)
#endif

// thread 0's state while it walks the sorted candidates, photonEvent mclib.c:1128-1339
struct EventWalk {
    double dt_max, old_scatt_time, dt;
    double *seg;          // [MAX_SEG] in LDS (runtime-indexed: keep it out of scratch)
    int nseg, skip, last_idx;
    long long rej;
    bool first;           // the next candidate is the head of the sorted list
    bool called;          // the head lies inside the frame: main() calls photonEvent (mcrat.c:777)
};

// the physics of one candidate between mclib.c:1144 and :1322: fluid frame at the photon's azimuth, Stokes rotation
// into the comoving frame, electron draw, singleScatter, boost back.  r is the candidate's position at the event.
// In two halves (cf. physics.hpp, single_scatter_begin / _finish): scatter_decide goes as far as the Klein-Nishina acceptance test and
// returns false on a rejection (nothing was stored; s is then unspecified); scatter_finish completes an accepted scattering -- it cannot
// fail -- and returns tau_new, the optical depth of the new momentum in the cached cell (see commit_scatter).
// WAVE: called by all 64 lanes of a wavefront with the same arguments (event_block's walk); see phys::sample_thermal_electron.
// Where an event's random numbers come from (rng.hpp): the engine's keyed streams -- one per (pass, candidate), any position one addition
// away, which is what the wave-parallel samplers use -- or the caller's tape (mcrat_hip_set_rng_tape), one sequential stream whose position
// is a device word: a candidate's draws start where the pass's free-path draws (tape_draw_kernel) or the previous candidate stopped.
struct KeyedSource {
    using Stream = EventStream;
    static constexpr bool wave_parallel = true;
    __device__ __forceinline__ Stream open(const RngKey &key, unsigned long long iter, uint32_t slot) const { return event_stream(key.seed, iter, slot, key.stream); }
    __device__ __forceinline__ void close(const Stream &) const {}
};
struct TapeSource {
    using Stream = TapeStream;
    static constexpr bool wave_parallel = false;     // (uniform_pos and the polar method consume as many entries as they need: no skipping ahead)
    TapeDev t;
    __device__ __forceinline__ Stream open(const RngKey &, unsigned long long, uint32_t) const { Stream s = {t.u, t.n, *t.cursor, t.error}; return s; }
    __device__ __forceinline__ void close(const Stream &s) const { *t.cursor = s.pos; }
};

template <class STREAM>
struct EventMidT {
    phys::ScatterMid sm;
    double beta[3];
    double w, nsig, gam, kf;
    double fluid_temp;
    STREAM rng;
};
using EventMid = EventMidT<EventStream>;

template <int DIMS, int GEOM, bool STOKES, bool WAVE = false, class SRC = KeyedSource>
__device__ __forceinline__ bool scatter_decide(const HydroDev &hy, LoopState *st, const RngKey &key, unsigned long long iter,
                                               uint32_t rng_slot, int cell, const double r[3], const double p[4], const double pc[4], double s[4],
                                               EventMidT<typename SRC::Stream> &m, const SRC &src = SRC())
{
    MC_STAMP(st, 2);
    m.fluid_temp = hy.temp[cell];                          // mclib.c:1148
    const CellFluid f = hy.fluid[cell];
    m.w = f.w; m.nsig = f.nsig; m.gam = f.gam; m.kf = f.kf;
    double cphi, sphi;
    phys::cos_sin_of_atan2(r[1], r[0], cphi, sphi);        // ph_phi, mclib.c:1151
    phys::cell_beta<DIMS>(f, cphi, sphi, m.beta);          // mclib.c:1167-1174
    if constexpr (STOKES) phys::stokes_rotation(m.beta, p + 1, pc + 1, s);   // mclib.c:1227
    m.rng = src.open(key, iter, rng_slot);
    const double k2e = hy.k2e ? hy.k2e[cell] : 0.0;
    double el[4];
    MC_STAMP(st, 3);
    phys::single_thermal_electron<WAVE && SRC::wave_parallel>(el, m.fluid_temp, k2e, pc, m.rng);   // mclib.c:1234
    MC_STAMP(st, 4);
    const bool ok = phys::single_scatter_begin<STOKES>(el, pc, s, m.sm, m.rng);   // mclib.c:1245 as far as kleinNishinaScatter's test
    if (!ok) src.close(m.rng);
    return ok;
}

template <int DIMS, int GEOM, bool STOKES, bool WAVE = false, class SRC = KeyedSource>
__device__ __forceinline__ void scatter_finish(const HydroDev &hy, LoopState *st, const RngKey &key, unsigned long long iter, uint32_t rng_slot,
                                               EventMidT<typename SRC::Stream> &m, double p[4], double pc[4], double s[4], double &tau_new, const SRC &src = SRC())
{
    phys::single_scatter_finish<STOKES>(m.sm, pc, s, m.rng);           // the rest of mclib.c:1245
    src.close(m.rng);
    MC_STAMP(st, 5);
    const double nb[3] = {-m.beta[0], -m.beta[1], -m.beta[2]};
    phys::boost_with<true>(nb, m.gam, m.kf, pc, p);                    // mclib.c:1265
    if constexpr (STOKES) phys::stokes_rotation(nb, pc + 1, p + 1, s); // mclib.c:1280
    // recalc_properties = 1 (mclib.c:1322).  The optical depth the next pass would recompute for this slot in its
    // cached cell (calcMeanFreePath, mclib.c:668-673: same position, same cell record, the new momentum) is computed
    // here while everything is in registers; the next pass still runs its in-cell test and re-locates if it fails.
    double norm = 1.0;
    if constexpr (TABLE_MODE) {                                        // optical_depth.c:58
        double eps, theta;
        if (phys::thermal_cross_section_lookup(hy, pc[0], m.fluid_temp, norm, eps, theta)) {
            // off the table (hot_x_section.c:563-599).  The reference meets this look-up in the NEXT pass's calcMeanFreePath: the integral is keyed with
            // that pass, so that a slot which has left its cell by then -- and is looked up again there -- is the only difference to it
            const int calls = hy.hot_fallback_calls;
            const bool whole_wave = WAVE && __ballot(1) == ~0ull;     // (the walk's wavefront: the same values in all lanes)
            norm = whole_wave ? phys::table_fallback_wave(eps, theta, key.seed, iter + 1, rng_slot, key.stream, calls)
                              : phys::table_fallback_lane(eps, theta, key.seed, iter + 1, rng_slot, key.stream, calls);
            if (!WAVE || (threadIdx.x & 63) == 0) atomicAdd(hy.table_fallbacks, 1);
        }
    }
    tau_new = phys::optical_depth_staged(m.beta, m.w, m.nsig, p[1], p[2], p[3], norm);
}

template <int DIMS, int GEOM, bool STOKES, bool WAVE = false, class SRC = KeyedSource>
__device__ __forceinline__ bool scatter_core(const HydroDev &hy, LoopState *st, const RngKey &key, unsigned long long iter,
                                             uint32_t rng_slot, int cell, const double r[3], double p[4], double pc[4], double s[4],
                                             double &fluid_temp, double &tau_new, const SRC &src = SRC())
{
    EventMidT<typename SRC::Stream> m;
    const bool ok = scatter_decide<DIMS, GEOM, STOKES, WAVE, SRC>(hy, st, key, iter, rng_slot, cell, r, p, pc, s, m, src);
    fluid_temp = m.fluid_temp;
    if (!ok) return false;
    scatter_finish<DIMS, GEOM, STOKES, WAVE, SRC>(hy, st, key, iter, rng_slot, m, p, pc, s, tau_new, src);
    return true;
}

// stores of a successful scatter, mclib.c:1290-1322
template <bool STOKES, class PH>
__device__ __forceinline__ void commit_scatter(const PH &ph, int i, const double p[4], const double pc[4], const double s[4],
                                               const double r[3], double tau_new, unsigned cand_flags)
{
    if constexpr (STOKES) { ph.s0(i) = s[0]; ph.s1(i) = s[1]; ph.s2(i) = s[2]; ph.s3(i) = s[3]; }
    ph.p0(i) = p[0]; ph.p1(i) = p[1]; ph.p2(i) = p[2]; ph.p3(i) = p[3];
    {
        const double d = phys::rcp_nr(p[0]);                           // mclib.c:1074-1080 factors of the new momentum
        ph.u0(i) = p[1] * d * C_LIGHT; ph.u1(i) = p[2] * d * C_LIGHT; ph.u2(i) = p[3] * d * C_LIGHT;
    }
    ph.c0(i) = pc[0]; ph.c1(i) = pc[1]; ph.c2(i) = pc[2]; ph.c3(i) = pc[3];
    ph.r0(i) = r[0]; ph.r1(i) = r[1]; ph.r2(i) = r[2];              // already advanced: the next step kernel skips it
    ph.num_scatt(i) += 1;                                              // mclib.c:1317
    ph.tau_next(i) = tau_new;
    ph.ntau(i) = -phys::rcp_nr(tau_new);
    ph.flags(i) = (unsigned char)(cand_flags | FLAG_RECALC | FLAG_TAU_FRESH);
}

// what the walk needs from a candidate's slot: one round of independent loads (this wavefront's latency is its list's)
struct CandCols {
    int idx;                     // the slot these columns belong to (-1: none loaded)
    int cell;
    unsigned flags;
    double p[4], r[3], pc[4], u[3], s[4];
};
template <bool STOKES, class PH>
__device__ __forceinline__ void load_candidate(const PH &ph, int i, CandCols &c)
{
    c.idx = i;
    c.cell = ph.idx(i);
    c.p[0] = ph.p0(i); c.p[1] = ph.p1(i); c.p[2] = ph.p2(i); c.p[3] = ph.p3(i);
    c.r[0] = ph.r0(i); c.r[1] = ph.r1(i); c.r[2] = ph.r2(i);
    c.pc[0] = ph.c0(i); c.pc[1] = ph.c1(i); c.pc[2] = ph.c2(i); c.pc[3] = ph.c3(i);
    c.flags = ph.flags(i);
    c.u[0] = ph.u0(i); c.u[1] = ph.u1(i); c.u[2] = ph.u2(i);
    c.s[0] = 1; c.s[1] = 0; c.s[2] = 0; c.s[3] = 0;
    if constexpr (STOKES) { c.s[0] = ph.s0(i); c.s[1] = ph.s1(i); c.s[2] = ph.s2(i); c.s[3] = ph.s3(i); }
}

// one candidate (scatt_time, i) of the walk.  Returns EV_DONE when the iteration is decided.
template <int DIMS, int GEOM, bool STOKES, bool WAVE = false, class PH, class SRC = KeyedSource>
__device__ __forceinline__ int try_candidate(const PH &ph, const HydroDev &hy, LoopState *st, const RngKey &key,
                                             unsigned long long iter, EventWalk &w, double scatt_time, int i, int slot_base, const SRC &src = SRC())
{
    // *scattered_ph_index (mclib.c:1341) is the last candidate photonEvent looked at; main() does not call
    // photonEvent at all when even the first free time exceeds the frame (mcrat.c:777,834)
    const bool in_frame = scatt_time < w.dt_max;
    if (!(w.first && !in_frame)) w.last_idx = i;
    if (w.first) w.called = in_frame;
    w.first = false;
    if (!in_frame) {                                       // mclib.c:1327-1335
        const double this_seg = w.dt_max - w.old_scatt_time;
        if (w.nseg < MAX_SEG) w.seg[w.nseg++] = this_seg;
        else w.seg[MAX_SEG - 1] += this_seg;
        w.dt = w.dt_max;
        return EV_DONE;
    }
    const double this_seg = scatt_time - w.old_scatt_time;  // mclib.c:1138
    if (w.nseg < MAX_SEG) w.seg[w.nseg++] = this_seg;
    else w.seg[MAX_SEG - 1] += this_seg;
    w.old_scatt_time = scatt_time;
    // one round of independent loads for everything the candidate needs (this wavefront's latency is its list's)
    CandCols cc;
    load_candidate<STOKES>(ph, i, cc);
    const int cell = cc.cell;
    double p[4] = {cc.p[0], cc.p[1], cc.p[2], cc.p[3]};
    double r[3] = {cc.r[0], cc.r[1], cc.r[2]};
    double pc[4] = {cc.pc[0], cc.pc[1], cc.pc[2], cc.pc[3]};
    const unsigned cand_flags = cc.flags;
    const double u0 = cc.u[0], u1 = cc.u[1], u2 = cc.u[2];
    double s[4] = {cc.s[0], cc.s[1], cc.s[2], cc.s[3]};
    if (cell == -1) return EV_RUNNING;                     // cannot scatter (documented deviation: mclib.c:1146-1148 would index [-1])

    if (cand_flags & FLAG_MOVES) {                         // the candidate's own position after mclib.c:1138
        for (int k = 0; k < w.nseg; ++k) {
            r[0] += u0 * w.seg[k];
            r[1] += u1 * w.seg[k];
            r[2] += u2 * w.seg[k];
        }
    }
    double fluid_temp, tau_new;
    if (!scatter_core<DIMS, GEOM, STOKES, WAVE, SRC>(hy, st, key, iter, (uint32_t)(i - slot_base) + key.slot_base, cell, r, p, pc, s, fluid_temp, tau_new, src)) {
        w.rej += 1;
        return EV_RUNNING;
    }
    commit_scatter<STOKES>(ph, i, p, pc, s, r, tau_new, cand_flags);
    st->frame_scatt_cnt += 1;                                          // mclib.c:1318
    st->last_scattered_temp = fluid_temp;
    w.skip = i;
    w.dt = scatt_time;
    MC_STAMP(st, 6);
    return EV_DONE;
}

// LDS scratch of the event walk (one per workgroup)
// (BLOCK threads, one per shortlist entry: the shortlist of a workgroup of BLOCK threads holds BLOCK candidates)
template <int BLOCK>
struct EventSharedT {
    Cand w[BLOCK / 64][TOPK];
    Cand c[TOPK];
    Cand raw[BLOCK];
    Cand list[BLOCK];
    double wt[BLOCK / 64];
    int wi[BLOCK / 64];
    double seg[MAX_SEG];
    double last_t;
    int last_i;
    int status;
};
using EventShared = EventSharedT<EVENT_BLOCK>;

// The second half of a loop pass for one photon list occupying slots [base, base + n): sort the shortlist
// (sh.raw[0..n_raw), complete below t_cut unless it overflowed), walk it as photonEvent does, refill from
// time_to_scatter if it runs out, then the bookkeeping of mcrat.c:782-784 / 837-845 into *st.
// All BLOCK threads call it; `gmin` is the list's minimum candidate (used when the shortlist is empty).
template <int DIMS, int GEOM, bool STOKES, int BLOCK, class PH, class SRC = KeyedSource>
__device__ __forceinline__ void event_block(const PH &ph, const HydroDev &hy, LoopState *st, const RngKey &key,
                                            EventSharedT<BLOCK> &sh, int n_raw, Cand gmin, int base, int n,
                                            unsigned long long iter, double dt_max, int last_idx, double t_est, const SRC &src = SRC())
{
    const int tid = threadIdx.x;
    int n_list = (n_raw > BLOCK) ? 0 : n_raw;              // overflowed: incomplete, ignore it
    if (tid < n_list) {                                    // rank sort (equal (t, idx) pairs cannot occur)
        const Cand me = sh.raw[tid];
        int rank = 0;
        for (int j = 0; j < n_list; ++j) rank += cand_less(sh.raw[j].t, sh.raw[j].idx, me.t, me.idx) ? 1 : 0;
        sh.list[rank] = me;
    }
    if (n_list == 0 && tid == 0) sh.list[0] = gmin;
    // A shortlist of up to 64 entries is sorted by the first wavefront alone -- the one that walks it: its own LDS writes are in order before
    // its reads, no workgroup barrier needed, and the other wavefronts are free until the barrier behind the walk (rank_loop_kernel draws the
    // next pass's random numbers there).  (n_list is the same in every thread: either all pass the barrier or none.)
    if (n_list > 64) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (n_list == 0 && gmin.idx != INT_MAX) n_list = 1;

    if (tid == 0) MC_STAMP(st, 1);
    EventWalk w;
    w.dt_max = dt_max;
    w.seg = sh.seg;
    w.old_scatt_time = 0; w.dt = 0; w.nseg = 0; w.skip = -1; w.rej = 0; w.first = true; w.called = false;
    w.last_idx = last_idx;
    long long rescans = 0;
    const double t_first = (n_list > 0) ? sh.list[0].t : INFINITY;

    const Cand *list = sh.list;
    const int max_rounds = n / TOPK + 3;
    // The walk runs on the whole first wavefront, all 64 lanes with the same values (a wave instruction costs the same for one lane
    // as for 64, and every store below then writes one value to one address from every lane), so that the one loop of the scatter
    // with a long trip count -- the hot electron's rejection sampling -- can try 64 attempts at a time (physics.hpp).
    // (-DMCRAT_NO_WAVE_WALK=1 builds the one-lane walk for A/B runs, tools/hot_bench.py.)
    for (int round = 0; round < max_rounds; ++round) {
        if (WAVE_WALK ? tid < 64 : tid == 0) {
#ifndef MCRAT_NO_WALK_PRIORITY
            __builtin_amdgcn_s_setprio(3);                 // the walk is its list's critical path: it goes first where its SIMD is shared
#endif
            int status = EV_NEED_MORE;
            if (n_list == 0) {                             // every slot was tried (or there is none): mclib.c:1128 loop ends
                w.dt = (round == 0) ? w.dt_max : w.old_scatt_time;
                status = EV_DONE;
            }
            for (int c = 0; c < n_list && status == EV_NEED_MORE; ++c) {
                if (try_candidate<DIMS, GEOM, STOKES, WAVE_WALK>(ph, hy, st, key, iter, w, list[c].t, list[c].idx, base, src) == EV_DONE)
                    status = EV_DONE;
            }
            if (status == EV_NEED_MORE) {
                sh.last_t = list[n_list - 1].t;
                sh.last_i = list[n_list - 1].idx;
                rescans += 1;
            }
            sh.status = status;
#ifndef MCRAT_NO_WALK_PRIORITY
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        __syncthreads();
        if (sh.status != EV_NEED_MORE) break;
        {
            const double lt = sh.last_t;
            const int li = sh.last_i;
            TopK more;
            more.init();
            for (int i = base + tid; i < base + n; i += BLOCK) {
                double t = ph.tts(i);
                if (t != t) t = INFINITY;
                if (cand_less(lt, li, t, i)) more.insert(t, i);
            }
            block_topk<BLOCK / 64>(more, sh.w, sh.c);
        }
        list = sh.c;
        n_list = 0;
        for (int c = 0; c < TOPK; ++c) n_list += (sh.c[c].idx != INT_MAX) ? 1 : 0;
    }

    if (tid == 0) {                                        // mcrat.c:782-784 / 837-845
        st->time_now += w.dt;
        const double rem = w.dt_max - w.dt;
        st->remaining_time = rem;
        st->last_time_step = w.dt;
        st->iteration = iter + 1;
        st->iterations += 1;
#ifndef MCRAT_DIAG
        st->slot_steps += n;
#endif
        st->done = !(rem > 0);
        st->nseg = w.nseg;
        for (int s = 0; s < MAX_SEG; ++s) st->seg[s] = (s < w.nseg) ? w.seg[s] : 0.0;
        st->skip_idx = w.skip;
        st->last_scattered_index = w.last_idx;
        st->kn_rejections += w.rej;
        st->photon_event_called = w.called ? 1 : 0;
        st->rescans += rescans;
        // shortlist threshold for the next pass: ~8 expected entries (speed only)
        if (t_first < INFINITY) {
            const double est = (t_est > 0) ? 0.875 * t_est + 0.125 * t_first : t_first;
            st->t_est = est;
            st->t_cut = 8.0 * est;
        }
        MC_STAMP(st, 7);
    }
}

template <int DIMS, int GEOM, bool STOKES, class SRC = KeyedSource>
__global__ __launch_bounds__(EVENT_BLOCK) void event_kernel(PhotonDev ph, HydroDev hy, LoopState *st, RngKey key,
                                                            const Cand *__restrict__ block_min, int n_blocks, Shortlist *sl, SRC src)
{
    static_assert(EVENT_BLOCK == SHORTLIST_CAP, "one thread per shortlist entry");
    __shared__ EventShared sh;
    const int tid = threadIdx.x;
    // every load this prologue needs is independent of the others: issue them together, then look at `done`
    const int done = st->done;
    const unsigned long long iter = st->iteration;
    const double dt_max = st->remaining_time, t_est = st->t_est;
    const int last_idx = st->last_scattered_index;
    const int n_raw = sl->count;
    const Cand mine = sl->items[tid];                     // EVENT_BLOCK == SHORTLIST_CAP; validity decided by n_raw
    MinCand m;
    m.init();
    for (int e = tid; e < n_blocks; e += EVENT_BLOCK) m.offer(block_min[e].t, block_min[e].idx);
    if (done) return;
    if (tid == 0) MC_STAMP(st, 0);
    if (tid < n_raw) sh.raw[tid] = mine;
    wave_min_pair_dpp(m.t, m.i);
    if ((tid & 63) == 0) { sh.wt[tid >> 6] = m.t; sh.wi[tid >> 6] = m.i; }
    __syncthreads();
    MinCand g;
    g.init();
#pragma unroll
    for (int wv = 0; wv < EVENT_BLOCK / 64; ++wv) g.offer(sh.wt[wv], sh.wi[wv]);
    Cand gmin;
    gmin.t = g.t; gmin.idx = g.i; gmin.pad = 0;
    event_block<DIMS, GEOM, STOKES, EVENT_BLOCK>(PtrCols(ph), hy, st, key, sh, n_raw, gmin, 0, ph.n, iter, dt_max, last_idx, t_est, src);
    if (tid == 0) sl->count = 0;
}

// ------------------------------------------------------------------ the random stream as an input (mcrat_hip_set_rng_tape)
// calcMeanFreePath with the caller's tape (mclib.c:646-675): the slots are taken in ascending order, every slot with a cell (idx != -1) consumes
// one gsl_rng_uniform_pos -- the next entry of the tape that is not 0 -- and its free time is -ln(u) / (tau' c).  Runs after step_kernel, which has
// re-located the slots and refreshed their optical depths exactly as in the keyed mode (its own keyed draws are overwritten here, with its
// minimum and shortlist): ONE workgroup walks the list in chunks of TAPE_BLOCK slots -- a chunk's located slots are ranked by a block scan, the
// tape window behind the cursor is compacted to its non-zero entries by another, slot k of the chunk takes entry k.  A validation mode (this is a
// serial dependency through the whole list by construction), not a fast one.
constexpr int TAPE_BLOCK = 1024;

__device__ __forceinline__ int block_exclusive_scan_1024(int v, int *s_w, int &total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
    if (lane == 63) s_w[w] = x;
    __syncthreads();
    if (w == 0) {
        int t = lane < TAPE_BLOCK / 64 ? s_w[lane] : 0;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int y = __shfl_up(t, off, 64); if (lane >= off) t += y; }
        if (lane < TAPE_BLOCK / 64) s_w[lane] = t;
    }
    __syncthreads();
    total = s_w[TAPE_BLOCK / 64 - 1];
    const int before = (w > 0 ? s_w[w - 1] : 0) + x - v;
    __syncthreads();
    return before;
}

__global__ __launch_bounds__(TAPE_BLOCK) void tape_draw_kernel(PhotonDev ph, const LoopState *__restrict__ st, TapeDev tape, Cand *__restrict__ block_min,
                                                                int n_blocks, Shortlist *sl)
{
    __shared__ double s_u[TAPE_BLOCK];            // the chunk's uniforms: the tape's next non-zero entries
    __shared__ int s_w[TAPE_BLOCK / 64];
    __shared__ long long s_pos;
    __shared__ double s_wt[TAPE_BLOCK / 64];
    __shared__ int s_wi[TAPE_BLOCK / 64];
    if (st->done) return;
    const int tid = threadIdx.x;
    const double t_cut = st->t_cut;
    long long pos = *tape.cursor;
    if (tid == 0) sl->count = 0;
    __syncthreads();
    MinCand best;
    best.init();
    for (int c0 = 0; c0 < ph.n; c0 += TAPE_BLOCK) {
        const int i = c0 + tid;
        const bool valid = i < ph.n && (ph.flags[i] & FLAG_VALID);
        const bool located = valid && ph.idx[i] != -1;
        int need = 0;
        const int rank = block_exclusive_scan_1024(located ? 1 : 0, s_w, need);
        // the next `need` non-zero entries of the tape, window by window
        int have = 0;
        while (have < need) {
            const long long q = pos + tid;
            const double v = q < tape.n ? tape.u[q] : 0.25 + 0.5 * (double)(tid & 1);      // (beyond the tape: flagged below)
            const bool nz = v != 0.0;
            int found = 0;
            const int k = block_exclusive_scan_1024(nz ? 1 : 0, s_w, found);
            if (nz && have + k < need) s_u[have + k] = v;
            if (nz && have + k == need - 1) s_pos = q + 1;                 // the entry that completes the chunk: the cursor stops behind it
            if (q >= tape.n && have + k < need) *tape.error = 1;             // an entry beyond the tape's end was needed
            __syncthreads();
            if (have + found >= need) { pos = s_pos; have = need; }
            else { have += found; pos += TAPE_BLOCK; }
            __syncthreads();
        }
        if (pos > tape.n) { if (tid == 0) *tape.error = 1; pos = tape.n; }
        if (valid) {
            double t = ph.tts[i];                                          // slots without a cell: 1e12 / c, stored by step_kernel (mclib.c:620,684)
            if (located) {
                t = div_by_c(ph.ntau[i] * log(s_u[rank]));                 // mclib.c:675-687
                ph.tts[i] = t;
            }
            best.offer(t, i);
            if (t < t_cut) shortlist_push(sl, t, i);
        }
        __syncthreads();
    }
    wave_min_pair_dpp(best.t, best.i);
    if ((tid & 63) == 0) { s_wt[tid >> 6] = best.t; s_wi[tid >> 6] = best.i; }
    __syncthreads();
    if (tid == 0) {
        MinCand m;
        m.init();
        for (int w = 0; w < TAPE_BLOCK / 64; ++w) m.offer(s_wt[w], s_wi[w]);
        Cand c;
        c.t = m.t; c.idx = m.i; c.pad = 0;
        block_min[0] = c;
        *tape.cursor = pos;
    }
    for (int b = 1 + tid; b < n_blocks; b += TAPE_BLOCK) { Cand c; c.t = INFINITY; c.idx = INT_MAX; c.pad = 0; block_min[b] = c; }
}

// ------------------------------------------------------------------ virtual ranks
// MCRaT is run as many MPI ranks of 10^3 - 5*10^3 photons, each an independent photon list with its own clock
// (Doc/mcrat_doc.tex:165-166,222).  rank_loop_kernel runs R such lists per GPU, one workgroup per
// list, the WHOLE loop of mcrat.c:761-851 inside one launch: the list's hot columns stay in the workgroup's L1/L2,
// the per-pass state lives in LDS, and no pass costs a kernel launch or an N-wide sweep of HBM.  Every list sees
// exactly the arithmetic and random numbers of a single-list context holding only its photons with
// rng_stream = first_stream + r.
// Two layouts.  desc == nullptr: the list is cut into ranks of `stride` consecutive slots (the last may hold fewer), all with the
// launch's seed and the streams key.stream + rank.  desc != nullptr (rank pool, mcrat_hip_pool_*): rank r owns the slots
// [r * stride, r * stride + desc[r].len) -- the reference's ranks hold Poisson-sized lists (mclib.c:87-136) -- and has its own
// seed and stream, as every MPI rank has its own generator (mcrat.c:99-103,701).
// Every column of one list copied between the live lists and a buffer laid out like them (the snapshot; a frame's capture): slot i of a column at byte
// offset +from is read, +to written.  All 24 double columns of a slot are loaded before the first is stored: a workgroup's copy is a handful of memory
// round trips, not one per column and 256 slots.  NOT inlined into rank_loop_kernel: called once per (frame, list) item, and its registers must not
// count against the loop's.
__device__ __attribute__((noinline)) void copy_list_columns(double *r0, unsigned col_stride, int *idx, unsigned char *flags, char *type, int base, int n,
                                                            long long from, long long to, int tid, int block)
{
    for (int il = tid; il < n; il += block) {
        const size_t i = (size_t)base + il;
        double v[N_DOUBLE_COLS];
#pragma unroll
        for (int k = 0; k < N_DOUBLE_COLS; ++k)
            v[k] = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(r0 + (size_t)k * col_stride + i) + from);
        const int ci = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(idx + i) + from);
        const unsigned char cf = *reinterpret_cast<const unsigned char *>(reinterpret_cast<const char *>(flags + i) + from);
        const char ct = *(reinterpret_cast<const char *>(type + i) + from);
#pragma unroll
        for (int k = 0; k < N_DOUBLE_COLS; ++k)
            *reinterpret_cast<double *>(reinterpret_cast<char *>(r0 + (size_t)k * col_stride + i) + to) = v[k];
        *reinterpret_cast<int *>(reinterpret_cast<char *>(idx + i) + to) = ci;
        *reinterpret_cast<unsigned char *>(reinterpret_cast<char *>(flags + i) + to) = cf;
        *(reinterpret_cast<char *>(type + i) + to) = ct;
    }
}

struct RankLayout {
    int n_ranks;
    int stride;           // slots reserved per rank
    int n_total;
    const RankDesc *desc;
    CsFrame *cs;          // CYCLOSYNCHROTRON_SWITCH on: per list, where a pass the hook of mcrat.c:786-808 must look at parks the list
    const CsHookArgs *hook;   // ... or, when given (and the kernel built with CSH), what the hook needs to run right there, inside the loop
    FrameQueueDev fq;     // n_frames > 0: persistent workgroups take (frame, list) items from a ticket (launch.hpp)
};

// Four lists per CU.  Measured by varying the number of lists on a dense jet: a workgroup alone on its CU needs 25 us per
// pass, two per CU 27 us each -- a pass is a chain of latencies (LDS -> sqrt -> gather -> Philox -> log; two dependent loads and a
// boost; 3 500 dependent f64 instructions on one lane), workgroups overlap almost freely, so LISTS PER CU is what counts.  A
// list therefore gets a workgroup of 128 threads (two waves; eight waves per CU = four lists at 256 VGPRs, no spills) and keeps
// only r and -1/tau in LDS (32 B per slot; idx, flags and u stay in HBM/L2), which with the small static scratch below is under
// 40 KB per list.  1000 lists of 1000 photons are then resident all at once.
// The workgroup size is a template parameter: 256 threads per list when there are few lists (each list then gets the CU
// it sits on) or the frame is optically thin (its cost is slow-path throughput per list), 128 otherwise; engine.hip picks.
constexpr int rank_lds_bytes_per_slot(int block) { return block == 256 ? 7 * (int)sizeof(double) + (int)sizeof(int) + 1 : 4 * (int)sizeof(double); }
#ifndef RANK_WAVES_PER_SIMD
#define RANK_WAVES_PER_SIMD 2
#endif
#ifndef RANK_NS_FUSED
#define RANK_NS_FUSED 2        // slots a thread takes through a fused (lock-step) trip: 2; 4 (one trip per pass for 1000-photon lists) spills, DESIGN.md section 4
#endif
#ifndef RANK_FUSE_DEN
#define RANK_FUSE_DEN 2      // a pass takes the fused form when more than 1/RANK_FUSE_DEN of the slots changed cell in the previous one
#endif

// FUSE: the kernel also holds the fused form of a pass (below).  It pays in optically thin frames and costs dense ones code they never
// run (instruction cache, registers), so it is a build of its own and engine.hip picks per frame, as it picks the workgroup size.
// QUEUE: the build that can take (frame, list) items from the frame queue (launch.hpp, FrameQueueDev).  A build of its own because carrying the
// queue's paths costs a build that does not use them 25 spilled doubles per lane (measured: the 2-D spherical Stokes build 128 -> 328 B of scratch, cfg3's
// frame 6.0 -> 8.2 ms), and instantiated only for the 256-thread lists with their columns in LDS; every other launch form runs a plan frame by frame.
template <int DIMS, int GEOM, bool STOKES, bool RESIDENT, int RANK_BLOCK, bool FUSE, bool CSH = false, bool QUEUE = false>
// (one wavefront per SIMD for the small-list builds -- up to 512 registers, spills to AGPRs, no scratch -- measured slower: cfg5's 64-thread lists
// frac 0.394 -> 0.365, cfg3's 128-thread lists 0.32 -> 0.21: the second wavefront hides more latency than the scratch traffic costs)
__global__ __launch_bounds__(RANK_BLOCK, RANK_WAVES_PER_SIMD) void rank_loop_kernel(PhotonDev gph, HydroDev hy_arg, LoopState *states, RngKey key,
                                                                RankLayout lay, long long max_passes, int lds_slots)
{
    constexpr int EVENT_BLOCK = RANK_BLOCK;                    // (shadows the event kernel's block size inside this kernel)
#ifndef RANK_NS_SPHERICAL
#define RANK_NS_SPHERICAL 4
#endif
    constexpr int RANK_NS_QUEUE = (GEOM == GEOM_SPHERICAL) ? RANK_NS_SPHERICAL : 4;   // slots a thread takes through phase 1 together (queue form)
    constexpr int RANK_QCAP = 4 * RANK_BLOCK;                  // slots per chunk = capacity of the slow-path queue
    extern __shared__ __align__(16) unsigned char s_dyn[];     // the list's r and -1/tau columns when it fits (lds_slots >= n)
    __shared__ LoopState st;
    __shared__ EventSharedT<RANK_BLOCK> sh;
    __shared__ int s_qn, s_sln, s_nrel, s_hook, s_len, s_item, s_ndefer;
    __shared__ int s_qb[RANK_QCAP];
    // the slow-path queue shares memory with the event walk's sorted list: the queue is empty before the list is written
    static_assert(sizeof(sh.list) >= sizeof(int) * RANK_QCAP, "queue fits into the sorted-list storage");
    int *const s_q = reinterpret_cast<int *>(sh.list);
    int tid = threadIdx.x, lane = tid & 63;                   // (not const: the queue builds launder them between two items, see the loop below)
    const bool queued = QUEUE ? lay.fq.n_frames > 0 : false;
    // One list through one frame: the whole loop of mcrat.c:761-851.  Without a queue the workgroup does this once, for list blockIdx.x, from the
    // LoopState begin_frame left in states[]; with one (launch.hpp, FrameQueueDev) for one (frame, list) item after the other, each from the item's seed and clock.
    // (the copies of a list between the live columns and the snapshot / a frame's capture: copy_list_columns, a function of its own -- inlined,
    // its 48 registers of columns in flight sat on top of the loop's hoisted per-thread invariants and cost every build 25 spilled doubles)
    auto list_frame = [&](const int rank, const int item, const HydroDev &hy) __attribute__((always_inline)) {
    const int base = rank * lay.stride;
    int n = min(lay.stride, lay.n_total - base);
    RngKey rk = {key.seed, key.stream + (uint32_t)rank, 0u};
    if (lay.desc) {
        const RankDesc d = lay.desc[rank];
        n = min(n, d.len);
        rk.seed = d.seed;
        rk.stream = d.stream;
    }
    if (item >= 0) rk.seed = lay.fq.items[item].seed;
    // (open == 2: the frame was begun by an earlier launch that ran into its pass limit -- the list goes on from its LoopState, as without a queue)
    const bool fresh = item >= 0 && lay.fq.items[item].open == 1;
    // a pool list's *scattered_ph_index is list-local between launches (its view reads it like a context of its own)
    const int idx_shift = lay.desc ? base : 0;
    if (fresh) {                                             // a new frame (cf. init_states_multi_kernel, staging.hip; mcrat.c:754-758)
        static_assert(sizeof(LoopState) / sizeof(int) <= 64, "one wavefront clears the state");
        if (tid < (int)(sizeof(LoopState) / sizeof(int))) reinterpret_cast<int *>(&st)[tid] = 0;
        __syncthreads();
        if (tid == 0) {
            double t_now = lay.fq.items[item].time_now, rem = lay.fq.items[item].remaining_time;
            if (lay.fq.chain_clock && item >= lay.n_ranks && lay.fq.items[item - lay.n_ranks].open) {
                t_now = states[rank].time_now;               // the clock its previous frame of this launch ended on
                rem = lay.fq.items[item].frame_end - t_now;
            }
            st.remaining_time = rem;
            st.time_now = t_now;
            st.done = !(rem > 0);
            st.skip_idx = -1;
            st.last_scattered_index = -1;
            st.force_relocate = 1;                           // mcrat.c:756
        }
    } else if (tid == 0) {
        st = states[rank];
        if (st.last_scattered_index >= 0) st.last_scattered_index += idx_shift;
    }
    __syncthreads();
    if (st.done || n <= 0) {
        if (item >= 0 && tid == 0) { states[rank] = st; lay.fq.records[item] = st; }
        return;
    }
    if (fresh && lay.fq.restore) {                           // the frame starts from the snapshot of the list (mcrat_hip_restore_photons, for this list)
        copy_list_columns(gph.r0, gph.col_stride, gph.idx, gph.flags, gph.type, base, n, lay.fq.snap_delta, 0, tid, RANK_BLOCK);
        __syncthreads();
    }
    // Cyclo-synchrotron lists double when they run out of null slots (photons.c:112-121), so half of a list can be null slots behind the
    // last photon.  A null slot takes no part in a pass -- no cell, never the earliest candidate, time_to_scatter = 1e12/c every time
    // (mclib.c:620,684) -- so the passes of this launch run over the slots up to the last one that is NOT such a settled null slot (a fresh
    // null slot still has to receive its time_to_scatter once).  Nothing observable changes; a doubled list costs a pass what its photons do.
    int n_pass = n;
    auto settle_pass_limit = [&]() {
        int last = -1;
        for (int il = tid; il < n; il += EVENT_BLOCK) {
            const int i = base + il;
            const bool settled_null = gph.type[i] == 'N' && gph.idx[i] == -1 && gph.tts[i] == 1e12 / C_LIGHT;   // (cyclo-synchrotron lists: columns in HBM/L2)
            if (!settled_null) last = il;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
        if (lane == 0) sh.wi[tid >> 6] = last;
        __syncthreads();
#pragma unroll
        for (int wv = 0; wv < EVENT_BLOCK / 64; ++wv) last = max(last, sh.wi[wv]);
        __syncthreads();
        n_pass = min(n, max(2, (last + 2) & ~1));            // slots go in pairs (one Philox block per pair)
    };
    if (lay.cs) settle_pass_limit();
#ifdef MCRAT_DIAG
    long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // ticks: load, forced step, step, event, store; [5] passes
    long long dg_t = (long long)__builtin_amdgcn_s_memtime();
    const bool dg_clock = (g_diag & 1024) != 0;      // only the list's start and end on the 100 MHz clock (and its length in shader-clock ticks)
    const long long dg_real0 = (long long)__builtin_amdgcn_s_memrealtime(), dg_tick0 = dg_t;
#define RANK_TICK(k) do { if (!dg_clock) { __builtin_amdgcn_sched_barrier(0); const long long n_ = (long long)__builtin_amdgcn_s_memtime(); dg[k] += n_ - dg_t; dg_t = n_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define RANK_TICK(k) do { } while (0)
#endif

    // LDS residency.  128-thread workgroups: r and -1/tau, read and written every pass (32 B per slot; idx, flags and u come
    // from HBM/L2), so that four lists fit a CU.  256-thread workgroups (two per CU by their registers anyway): all the
    // per-pass columns (r, u, -1/tau, idx, flags: 61 B per slot).  The columns are copied in once per launch and written back
    // at the end; `ph` is the same PhotonDev with those column pointers aimed at LDS and the biases set, so every device
    // function below works on it unchanged (col[i - bias]).  (RESIDENT is a template parameter so that the pointers provably
    // address LDS and the accesses compile to ds_read / ds_write instead of flat loads.)
    constexpr bool FULL_HOT = RANK_BLOCK == 256;            // (512 threads: long lists, one per CU -- r and -1/tau at 32 B per slot reach 4096 slots)
    constexpr unsigned LDS_MASK = RESIDENT ? (FULL_HOT ? LIST_MASK_FULL : LIST_MASK_SMALL) : 0u;
    using Cols = ListCols<LDS_MASK>;
    static_assert(Cols::lds_bytes_per_slot == (RESIDENT ? (size_t)rank_lds_bytes_per_slot(RANK_BLOCK) : 0), "launch_rank_loop sizes the dynamic LDS with this");
    const Cols ph(gph, RESIDENT ? s_dyn : nullptr, lds_slots, base);
    if constexpr (RESIDENT) {
        for (int il = tid; il < n; il += EVENT_BLOCK) {
            const int i = base + il;
            ph.r0(i) = ph.template gcol<COL_R0>(i); ph.r1(i) = ph.template gcol<COL_R1>(i); ph.r2(i) = ph.template gcol<COL_R2>(i);
            ph.ntau(i) = ph.template gcol<COL_NTAU>(i);
            if constexpr (FULL_HOT) {
                ph.u0(i) = ph.template gcol<COL_U0>(i); ph.u1(i) = ph.template gcol<COL_U1>(i); ph.u2(i) = ph.template gcol<COL_U2>(i);
                ph.idx(i) = ph.g_idx_at(i); ph.flags(i) = ph.g_flags_at(i);
            }
        }
        __syncthreads();
    }
    // The free-path draws of a pass -- one Philox block per slot pair, log(u+) per slot (mclib.c:675-680; rng.hpp) -- depend on the pass number
    // and the slot alone, not on what the event before it does: the wavefronts that sit out the event walk (three of four; the walk is one
    // wavefront's, event_block) compute the NEXT pass's while they would otherwise wait at the barrier, into a scratch column (ListCols::draw_log).
    // Phase 1 then reads 8 B per slot instead of running ten Philox rounds per pair and a logarithm per slot: the same bits, a shorter pass.
#ifndef MCRAT_NO_SHADOW_DRAWS
    constexpr bool SHADOW = RANK_BLOCK > 64;                 // (a list of one wavefront has nobody in the walk's shadow: it draws in phase 1, as before)
#else
    constexpr bool SHADOW = false;                           // (A/B build)
#endif
    auto draw_logs = [&](unsigned long long it, int first_thread, int n_threads) {
        for (int pair = tid - first_thread; 2 * pair < n_pass; pair += n_threads) {
            const Philox4 blk = keyed_block(rk.seed, it, (uint32_t)pair, RNG_FREEPATH, rk.stream);
            const int i = base + 2 * pair;
            ph.draw_log(i) = log(bits_to_uniform_pos((uint64_t)blk.w[0] | ((uint64_t)blk.w[1] << 32)));
            if (2 * pair + 1 < n) ph.draw_log(i + 1) = log(bits_to_uniform_pos((uint64_t)blk.w[2] | ((uint64_t)blk.w[3] << 32)));
        }
    };
    if constexpr (SHADOW) {
        draw_logs(st.iteration, 0, EVENT_BLOCK);             // the first pass of this launch: nobody has drawn for it
        __syncthreads();
    }
    RANK_TICK(0);

    // the advance the last pass left pending (LoopState::seg), applied to every moving slot but the one the event advanced itself
    auto apply_pending = [&]() {
        const int nseg = st.nseg, skip = st.skip_idx;
        if (nseg > 0) {
            for (int il = tid; il < n; il += EVENT_BLOCK) {
                const int i = base + il;
                if ((ph.flags(i) & FLAG_MOVES) && i != skip) {
                    const double u0 = ph.u0(i), u1 = ph.u1(i), u2 = ph.u2(i);
                    double r0 = ph.r0(i), r1 = ph.r1(i), r2 = ph.r2(i);
                    for (int s = 0; s < nseg; ++s) {
                        const double t = st.seg[s];
                        r0 += u0 * t; r1 += u1 * t; r2 += u2 * t;
                    }
                    ph.r0(i) = r0; ph.r1(i) = r1; ph.r2(i) = r2;
                }
            }
        }
    };
    int prev_rel = 0;                                        // slots that changed cell in the previous pass
    for (long long pass = 0; pass < max_passes; ++pass) {
        const int nseg = st.nseg;
        const int skip = st.skip_idx;
        const unsigned long long iter = st.iteration;
        const double t_cut = st.t_cut;
        const bool force = st.force_relocate != 0;          // first pass of a frame, mcrat.c:756
        // Where most slots change cell from one event to the next (the forced pass; optically thin frames, whose events lie many
        // cells' light-crossing times apart) a thread takes its own slots through the re-location in lockstep (relocate_lockstep)
        // instead of queueing them: no hand-over through LDS, and the chains of its slots overlap.  Where few do (dense frames), the
        // queue keeps the lanes of the slow path dense.  Same arithmetic either way.
        const bool fused = FUSE && !TABLE_MODE && (force || RANK_FUSE_DEN * prev_rel > n_pass);
        if (tid == 0) { s_qn = 0; s_sln = 0; s_nrel = 0; s_ndefer = 0; }
        __syncthreads();
        int n_rel = 0;

        MinCand best;
        best.init();
        int relocated = 0, not_found = 0;
        auto shortlist_lds = [&](double t, int i) {
            const int pos = atomicAdd(&s_sln, 1);
            if (pos < RANK_BLOCK) { sh.raw[pos].t = t; sh.raw[pos].idx = i; sh.raw[pos].pad = 0; }
        };
        // ---- phase 1 + phase 2, in chunks of RANK_QCAP slots so that the slow-path queue always holds a chunk's worth
        // (a list of up to 1024 photons is one chunk).  Phase 1: the step of every slot (cf. step_kernel); a thread owns
        // slot pairs so that one Philox block serves two slots, as the draw order prescribes (rng.hpp).
        for (int c0 = 0; c0 < n_pass; c0 += RANK_QCAP) {
            const int c1 = min(n_pass, c0 + RANK_QCAP);
            if (c0 > 0) {
                __syncthreads();                             // the previous chunk's queue has been worked off
                if (tid == 0) { s_qn = 0; s_ndefer = 0; }
                __syncthreads();
            }
            // Queue form: two slot pairs per thread per trip, every stage written over all four slots before the next stage: with
            // two lists per CU a SIMD holds two waves, and four independent chains (LDS loads -> sqrt -> cell-record gather ->
            // Philox -> log) in flight hide part of each other's latency; slots that need the slow path go to the LDS queue.
            // Fused form: one slot pair per trip, its re-locations in lockstep right here (relocate_lockstep); only what the hints
            // do not settle is queued.  Same arithmetic per slot as fast_one / slow_one in both.
            auto phase1 = [&](auto ns_c, auto fused_c) {
                constexpr int NS = decltype(ns_c)::value;
                constexpr bool FUSED = decltype(fused_c)::value;
                constexpr int PAIRS = NS / 2;
                for (int pair = (c0 >> 1) + tid; 2 * pair < c1; pair += PAIRS * EVENT_BLOCK) {
                    int il[NS];
                    bool live[NS];
#pragma unroll
                    for (int j = 0; j < PAIRS; ++j) { il[2 * j] = 2 * (pair + j * EVENT_BLOCK); il[2 * j + 1] = il[2 * j] + 1; }
#pragma unroll
                    for (int k = 0; k < NS; ++k) { live[k] = il[k] < c1; if (!live[k]) il[k] = c0; }
                    double r0[NS], r1[NS], r2[NS], ntau[NS];
                    int cell[NS];
                    unsigned fl[NS];
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        const int i = base + il[k];
                        r0[k] = ph.r0(i); r1[k] = ph.r1(i); r2[k] = ph.r2(i);
                        ntau[k] = ph.ntau(i); cell[k] = ph.idx(i); fl[k] = ph.flags(i);
                    }
                    if (nseg > 0) {                                  // pending updatePhotonPosition, mclib.c:1067-1095
                        double u0[NS], u1[NS], u2[NS];
                        bool mv[NS];
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            const int i = base + il[k];
                            u0[k] = ph.u0(i); u1[k] = ph.u1(i); u2[k] = ph.u2(i);
                            mv[k] = live[k] && (fl[k] & FLAG_MOVES) && (base + il[k] != skip);
                        }
                        for (int sg = 0; sg < nseg; ++sg) {
                            const double t = st.seg[sg];
#pragma unroll
                            for (int k = 0; k < NS; ++k) {
                                const double n0 = r0[k] + u0[k] * t, n1 = r1[k] + u1[k] * t, n2 = r2[k] + u2[k] * t;
                                r0[k] = mv[k] ? n0 : r0[k]; r1[k] = mv[k] ? n1 : r1[k]; r2[k] = mv[k] ? n2 : r2[k];
                            }
                        }
#pragma unroll
                        for (int k = 0; k < NS; ++k)
                            if (mv[k]) { const int i = base + il[k]; ph.r0(i) = r0[k]; ph.r1(i) = r1[k]; ph.r2(i) = r2[k]; }
                    }
                    double a0[NS], a1[NS], a2[NS], tf[NS];
                    bool dom[NS], inb[NS];
                    CellGeom cg[NS];
                    CellGeom2 cg2[NS];
                    // (the fused form looks every slot up in the grid first and learns from the entry whether it has left its cached cell --
                    // LockstepProbe -- so it does not gather the cached cell's geometry here)
                    if constexpr (!(FUSED && !TABLE_MODE)) {
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            const int cc = (cell[k] < 0 || MC_DIAG(DIAG_SKIP_INCELL)) ? 0 : cell[k];
                            cg[k] = hy.geom[cc];                                     // geometry.c:394-417 operands
                            if constexpr (DIMS == DIM_THREE) cg2[k] = hy.geom2[cc];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        if (MC_DIAG(DIAG_SKIP_COORDS)) { a0[k] = r0[k] + r1[k]; a1[k] = r2[k]; a2[k] = 0; }
                        else phys::hydro_coords<DIMS, GEOM>(r0[k], r1[k], r2[k], a0[k], a1[k], a2[k]);
                        dom[k] = phys::in_domain<DIMS>(hy, a0[k], a1[k], a2[k]);      // mclib.c:492-505
                    }
                    uint64_t bits[NS];                                                // the bit pattern of log(u+) of the slot's draw
                    if constexpr (SHADOW) {                                           // drawn in the previous walk's shadow (draw_logs)
#pragma unroll
                        for (int k = 0; k < NS; ++k) bits[k] = (uint64_t)__double_as_longlong(ph.draw_log(base + il[k]));
                    } else {
#pragma unroll
                        for (int j = 0; j < PAIRS; ++j) {
                            const Philox4 blk = keyed_block(rk.seed, iter, (uint32_t)(pair + j * EVENT_BLOCK), RNG_FREEPATH, rk.stream);
                            bits[2 * j] = (uint64_t)__double_as_longlong(log(bits_to_uniform_pos((uint64_t)blk.w[0] | ((uint64_t)blk.w[1] << 32))));
                            bits[2 * j + 1] = (uint64_t)__double_as_longlong(log(bits_to_uniform_pos((uint64_t)blk.w[2] | ((uint64_t)blk.w[3] << 32))));
                        }
                    }
                    // decisions, slot by slot
                    int qd[NS], code[NS];
                    bool settled[NS];
                    double tl[NS];
#pragma unroll
                    for (int k = 0; k < NS; ++k) { settled[k] = false; tl[k] = 0; qd[k] = 0; code[k] = -1; }
                    if constexpr (FUSED && !TABLE_MODE) {
                        bool cand[NS], same[NS], todo[NS];
                        bool any_todo = false;
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            same[k] = false;
                            cand[k] = live[k] & ((fl[k] & FLAG_VALID) != 0) & dom[k] & (cell[k] != -1);
                            if (cand[k]) code[k] = phys::grid_bucket_of<DIMS>(hy.grid, a0[k], a1[k], a2[k]);
                            todo[k] = cand[k];
                            any_todo = any_todo || cand[k];
                        }
                        if (any_todo) {
                            int slot[NS];
#pragma unroll
                            for (int k = 0; k < NS; ++k) slot[k] = base + il[k];
                            const LockstepProbe probe = {cell, force, same};
                            double tt[NS];
                            relocate_lockstep<DIMS, GEOM, NS, true>(ph, hy, slot, todo, r0, r1, a0, a1, a2, code, bits, fl, !force, tt, relocated, &probe);
#pragma unroll
                            for (int k = 0; k < NS; ++k)
                                if (cand[k] && !todo[k] && !same[k]) { settled[k] = true; tl[k] = tt[k]; n_rel += 1; }
                        }
                        bool any = false;
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            if (!cand[k]) continue;
                            bool stays = same[k];
                            if (todo[k]) {                                            // no hint settles this slot: the test on its cached cell
                                stays = !force && phys::check_in_block<DIMS>(hy, cell[k], a0[k], a1[k], a2[k]);   // mclib.c:507,528
                                if (!stays) { qd[k] = 1; n_rel += 1; }
                            }
                            if (stays && (fl[k] & FLAG_RECALC) && !(fl[k] & FLAG_TAU_FRESH)) qd[k] = 2;          // mclib.c:668
                            any = any || (stays && qd[k] == 0);
                        }
                        if (any) {                                                    // the draws of the slots that stay in their cells
#pragma unroll
                            for (int k = 0; k < NS; ++k) tf[k] = free_time_from_log(ntau[k], __longlong_as_double((long long)bits[k]));
                        }
                    } else {
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        inb[k] = (2 * fabs(a0[k] - cg[k].c0) - cg[k].s0 <= 0) & (2 * fabs(a1[k] - cg[k].c1) - cg[k].s1 <= 0);
                        if constexpr (DIMS == DIM_THREE) inb[k] = inb[k] & (2 * fabs(a2[k] - cg2[k].c2) - cg2[k].s2 <= 0);
                        if (MC_DIAG(DIAG_SKIP_INCELL)) inb[k] = true;
                        qd[k] = 0; code[k] = -1;
                        if (!live[k] || !(fl[k] & FLAG_VALID) || !(dom[k] && cell[k] != -1)) continue;
                        if (force || !inb[k]) qd[k] = 1;                              // mclib.c:507,528
                        else if ((fl[k] & FLAG_RECALC) && !(fl[k] & FLAG_TAU_FRESH)) qd[k] = 2;   // mclib.c:668
                        if (qd[k] == 1) { code[k] = phys::grid_bucket_of<DIMS>(hy.grid, a0[k], a1[k], a2[k]); n_rel += 1; }
                    }
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        tf[k] = free_time_from_log(ntau[k], __longlong_as_double((long long)bits[k]));   // mclib.c:675-687
                    }
                    }
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        if (!live[k]) continue;
                        const int i = base + il[k];
                        double t;
                        if (!(fl[k] & FLAG_VALID)) { ph.tts(i) = INFINITY; continue; }
                        if (settled[k]) {                                             // re-located in lockstep above; everything is stored
                            best.offer(tl[k], i);
                            if (tl[k] < t_cut) shortlist_lds(tl[k], i);
                            continue;
                        }
                        if (dom[k] && cell[k] != -1) {
                            const int q = qd[k];
                            if (!q && (fl[k] & FLAG_RECALC)) {                        // mclib.c:668, tau of the new momentum is at hand
                                ph.flags(i) = (unsigned char)(fl[k] & ~(FLAG_RECALC | FLAG_TAU_FRESH));
                                ph.tau(i) = ph.tau_next(i);
                            }
                            if (q) {
                                const int e = atomicAdd(&s_qn, 1);                    // < RANK_QCAP: a chunk has no more slots than that
                                s_q[e] = il[k] | (q == 2 ? Q_RECALC_ONLY : 0);
                                s_qb[e] = code[k];
                                ph.tts(i) = __longlong_as_double((long long)bits[k]);   // the draw, for phase 2 (which stores the free time)
                                continue;
                            } else {
                                t = tf[k];
                                ph.tts(i) = t;
                            }
                        } else {
                            if (cell[k] != -1) ph.idx(i) = -1;                        // mclib.c:592
                            t = 1e12 / C_LIGHT;                                       // mclib.c:620,684
                            ph.tts(i) = t;
                        }
                        best.offer(t, i);
                        if (t < t_cut) shortlist_lds(t, i);
                    }
                }
            };
            if constexpr (FUSE && !TABLE_MODE) {
                if (fused) phase1(std::integral_constant<int, RANK_NS_FUSED>{}, std::true_type{});
                else phase1(std::integral_constant<int, RANK_NS_QUEUE>{}, std::false_type{});
            } else {
                phase1(std::integral_constant<int, RANK_NS_QUEUE>{}, std::false_type{});
            }
            if (!force) RANK_TICK(6);
            if constexpr (FUSE) {                            // how many slots changed cell this pass: the next pass's form
                if (c0 + RANK_QCAP >= n_pass) {
                    int w = n_rel;
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
                    if (lane == 0 && w) atomicAdd(&s_nrel, w);
                }
            }
            __syncthreads();
            if (!force) RANK_TICK(7);
            // ---- phase 2: the queued slots, dense
            const int qn = s_qn;
            TableSite ts(rk, iter, base, TABLE_MODE ? 1 : 0);
            // TAU_CALCULATION == TABLE has a second round: a slot whose look-up fell off the table (hot_x_section.c:563-599: the reference integrates
            // the cross section afresh, 500 000 samples) left the first round as it came, (eps, theta) parked in its tau / -1/tau entries; now a
            // wavefront per such slot integrates (physics.hpp: table_fallback_wave) and its lane 0 takes the slot through slow_one with the value
#pragma nounroll
            for (int round = 0; round < (TABLE_MODE ? 2 : 1); ++round) {
                if (round == 1) {
                    __syncthreads();
                    if (!s_ndefer) break;
                }
                const int e_step = round == 0 ? EVENT_BLOCK : EVENT_BLOCK / 64;
                for (int e = round == 0 ? tid : (tid >> 6); e < qn; e += e_step) {
                    const int q = s_q[e];
                    const int i = base + (q & ~(Q_RECALC_ONLY | Q_DEFERRED));
                    if constexpr (TABLE_MODE) {
                        if (round == 1) {
                            if (!(q & Q_DEFERRED)) continue;                              // (the same for all lanes of the wavefront)
                            const double norm = phys::table_fallback_wave(ph.tau(i), ph.ntau(i), ts.seed, ts.pass, ts.slot_base + (uint32_t)(i - base), ts.stream,
                                                                          hy.hot_fallback_calls);
                            if (lane != 0) continue;
                            atomicAdd(hy.table_fallbacks, 1);
                            ts.mode = 2;
                            ts.norm = norm;
                        }
                    }
                    const uint64_t qbits = (uint64_t)__double_as_longlong(ph.tts(i));
                    const double t = slow_one<DIMS, GEOM, true>(ph, hy, i, !(q & Q_RECALC_ONLY), s_qb[e], !force, qbits, relocated, not_found, ts);
                    if constexpr (TABLE_MODE) {
                        if (ts.deferred) {
                            ts.deferred = false;
                            ph.tau(i) = ts.eps;
                            ph.ntau(i) = ts.theta;
                            s_q[e] = q | Q_DEFERRED;
                            s_ndefer = 1;
                            continue;
                        }
                    }
                    best.offer(t, i);
                    if (t < t_cut) shortlist_lds(t, i);
                }
            }
        }
        wave_min_pair_dpp(best.t, best.i);
        if (lane == 0) { sh.wt[tid >> 6] = best.t; sh.wi[tid >> 6] = best.i; }
        if (relocated) atomicAdd(reinterpret_cast<unsigned long long *>(&st.n_relocated), (unsigned long long)relocated);
        if (not_found) atomicAdd(reinterpret_cast<unsigned long long *>(&st.not_found), (unsigned long long)not_found);
        __syncthreads();
        prev_rel = s_nrel;                                   // (complete since the barrier after phase 1; reset after the next ones)
        MinCand g;
        g.init();
#pragma unroll
        for (int wv = 0; wv < EVENT_BLOCK / 64; ++wv) g.offer(sh.wt[wv], sh.wi[wv]);
        Cand gmin;
        gmin.t = g.t; gmin.idx = g.i; gmin.pad = 0;
        if (force) RANK_TICK(1); else RANK_TICK(2);
        // ---- the event half and the bookkeeping
        const bool frame_goes_on = gmin.t < st.remaining_time;          // (else this pass ends the frame, mcrat.c:834: nobody needs further draws)
        // the wavefronts that do not walk draw for the next pass now; they join the others at the barrier behind the walk (event_block)
        if constexpr (SHADOW) { if (frame_goes_on && tid >= 64) draw_logs(iter + 1, 64, RANK_BLOCK - 64); }
        (void)frame_goes_on;
        event_block<DIMS, GEOM, STOKES, RANK_BLOCK>(ph, hy, &st, rk, sh, s_sln, gmin, base, n_pass, iter, st.remaining_time, st.last_scattered_index, st.t_est);
        if (tid == 0) {
            st.force_relocate = 0;
            // cyclo-synchrotron lists: if photonEvent reported a pool photon (it becomes a comptonised one and is replaced, mcrat.c:786-795)
            // or the rebinning is to be looked at (every 1000 scatterings, :797), the hook has to look at this pass: right here when the
            // kernel carries it (CSH), else the list leaves the loop -- its photons made current below -- for cs_replace_pool_kernel,
            // which lets it go on in the next launch
            s_hook = 0;
            if (lay.cs && st.photon_event_called) {
                const int sidx = st.last_scattered_index;
                const bool pool_photon = sidx >= 0 && gph.type[sidx] == 'p';
                const bool thousand = (st.frame_scatt_cnt % 1000 == 0) && st.frame_scatt_cnt != 0;
                if (pool_photon || thousand) {
                    if (CSH && lay.hook) {
                        s_hook = 1;
                    } else {
                        lay.cs[rank].halt = CS_HALT_HOOK;
                        lay.cs[rank].saved_done = st.done;
                        st.done = LOOP_CS_HALT;
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (CSH) {
            if (s_hook) {
                apply_pending();                             // the hook moves the scattered photon and places a new one: positions must be current
                __syncthreads();
                if (tid == 0) { st.nseg = 0; st.skip_idx = -1; s_len = n; }
                __syncthreads();
                PhotonDev lp = gph;                          // the list as a list of its own, as cs_replace_pool_kernel sees it
                offset_photons(lp, (size_t)base);
                lp.n = n;
                cs_hook_body<RANK_BLOCK>(lay.hook->p, hy, lay.hook->h, rk, &st, lp, &lay.cs[rank], 1, sh.wi, lay.stride, &s_len, idx_shift);
                __threadfence_block();
                __syncthreads();
                if (s_len != n) {                            // the list doubled inside its window (photons.c:112-121)
                    n = s_len;
                    if (tid == 0) const_cast<RankDesc *>(lay.desc)[rank].len = n;
                }
                settle_pass_limit();                         // (the new photon's slot, fresh null slots)
                if constexpr (SHADOW) {
                    draw_logs(st.iteration, 0, EVENT_BLOCK); // (the pass limit may have grown: the next pass's draws for every pair)
                    __syncthreads();
                }
            }
        }
        RANK_TICK(3);
#ifdef MCRAT_DIAG
        dg[5] += 1;
#endif
        if (st.done) break;
    }

    // leave the photons current: apply the advance still pending (cf. flush_kernel)
    {
        apply_pending();
        __syncthreads();
        if constexpr (RESIDENT) {
            for (int il = tid; il < n; il += EVENT_BLOCK) {
                const int i = base + il;
                ph.template gcol<COL_R0>(i) = ph.r0(i); ph.template gcol<COL_R1>(i) = ph.r1(i); ph.template gcol<COL_R2>(i) = ph.r2(i);
                ph.template gcol<COL_NTAU>(i) = ph.ntau(i);
                if constexpr (FULL_HOT) {
                    ph.template gcol<COL_U0>(i) = ph.u0(i); ph.template gcol<COL_U1>(i) = ph.u1(i); ph.template gcol<COL_U2>(i) = ph.u2(i);
                    ph.g_idx_at(i) = ph.idx(i); ph.g_flags_at(i) = ph.flags(i);
                }
            }
        }
#ifdef MCRAT_DIAG
        __syncthreads();
        RANK_TICK(4);
        if (tid == 0) for (int k = 0; k < 8; ++k) st.stamps[k] = dg[k];
        if (tid == 0 && dg_clock) {
            st.stamps[0] = dg_real0; st.stamps[1] = (long long)__builtin_amdgcn_s_memrealtime();
            st.stamps[2] = (long long)__builtin_amdgcn_s_memtime() - dg_tick0;
        }
#endif
        if (tid == 0) {
            st.nseg = 0; st.skip_idx = -1;
            if (st.last_scattered_index >= 0) st.last_scattered_index -= idx_shift;
            states[rank] = st;
            if (item >= 0) lay.fq.records[item] = st;
        }
        // the list as this frame leaves it, for the frame's outputs (printPhotons, saveCheckpoint: mcrat.c:881-906), before its next frame moves it on
        if (item >= 0 && lay.fq.capture_delta != 0 && item / lay.n_ranks < lay.fq.n_frames - 1 && st.done == LOOP_DONE) {
            __syncthreads();
            copy_list_columns(gph.r0, gph.col_stride, gph.idx, gph.flags, gph.type, base, n, 0, lay.fq.capture_delta + (long long)(item / lay.n_ranks) * lay.fq.capture_stride, tid, RANK_BLOCK);
        }
    }
    };   // list_frame

    // ---- which list, which frame
    // A queue launch's workgroups are persistent: one takes item after item from its XCD's queue until that is empty -- a workgroup per item left 13 %
    // of the wave slots empty between a workgroup's end and its successor's start (SQ_WAVE_CYCLES of the launch against 2048 slots x its duration).
    // The loop must not let the compiler carry per-thread values from one item to the next (hoisted out of the loop they live through the whole
    // body: the first persistent form of this kernel paid 37 spilled registers for that): the thread index is laundered at the top of every turn.
    for (;;) {
    if constexpr (QUEUE) { asm volatile("" : "+v"(tid)); lane = tid & 63; }
    int rank = blockIdx.x, item = -1;
    if (queued) {
        // The k-th workgroup to START takes the k-th open item (frame-major): whatever order the hardware starts workgroups in, the workgroup of a
        // list's previous frame has started before this one, runs without waiting for anything later, and so always gets through -- waiting for it
        // cannot deadlock.  (It is rare: with more lists than the device holds at once a list's previous frame ended long before its next item is drawn.)
        // The lists are dealt out to the XCDs (list r belongs to XCD r % 8) and a workgroup takes items of the XCD it runs on (HW_REG_XCC_ID), so a
        // list never changes XCD: what its previous frame stored is in the L2 this workgroup reads through, and the hand-over costs no write-back
        // of that L2 (an agent-scope release per item cost the queue 9 % on the benchmark frame), only the invalidation of this CU's L1.  Should the
        // hardware start fewer workgroups on an XCD than it has items, those items stay undone and the host launches again for them.
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= (unsigned)(FRAME_QUEUE_XCDS - 1);
            const unsigned k = atomicAdd(lay.fq.ticket + xcc * FRAME_TICKET_STRIDE, 1u);
            int it = k < (unsigned)(lay.fq.order_off[xcc + 1] - lay.fq.order_off[xcc]) ? lay.fq.order[lay.fq.order_off[xcc] + (int)k] : -2;   // -2: the queue is empty
            const int f = it / lay.n_ranks, r = it - f * lay.n_ranks;
            if (it >= 0 && f > 0 && lay.fq.items[it - lay.n_ranks].open) {
                unsigned d = 0;
                long long spins = 0;
                for (;;) {
                    d = __hip_atomic_load(&lay.fq.frames_done[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d >= (unsigned)f || spins >= (1ll << 22)) break;      // (through, or stalled: the flag bit makes it large; bounded: ~4 s)
                    __builtin_amdgcn_s_sleep(32);
                    spins += 1;
                }
                if (d != (unsigned)f) it = -1;                                 // its previous frame ran into the launch's pass limit: the host goes on from there
                // Hand-over of a list between workgroups of one XCD (another CU's L1 sees nothing of a workgroup's stores by itself; MI355X_MICROARCH.md,
                // inter-workgroup visibility): the one that ends a frame drains every wave's stores into the XCD's L2 and passes its barrier before ONE
                // lane's agent-scope store of frames_done; the one that takes the list on makes ONE agent-scope acquire (this CU's L1 invalidated) after
                // it has seen frames_done, in front of the barrier that lets its other waves go.
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            s_item = it;
        }
        __syncthreads();
        item = __builtin_amdgcn_readfirstlane(s_item);
        if (item == -2) return;                                // the queue is empty
        if (item < 0) { __syncthreads(); continue; }           // (its list's previous frame stalled: the host goes on from there; on to the next item)
        rank = item % lay.n_ranks;
    }
    if constexpr (QUEUE) {
        // the hydro frame of THIS item: a queue launch may take its lists through several staged frames (FrameQueueDev::hydro; a real run stages frame
        // f + 1 for the slab the photons can reach from frame f before the launch).  The array is read through the constant address space, so that
        // its members arrive by scalar loads like a kernel argument's.
        typedef const HydroDev __attribute__((address_space(4))) *ConstHydro;
        ConstHydro frames = (ConstHydro)lay.fq.hydro;
        const int h = __builtin_amdgcn_readfirstlane(lay.fq.items[item].hydro);
        list_frame(rank, item, *(const HydroDev *)(frames + h));
    } else {
        list_frame(rank, item, hy_arg);
    }
    if (queued) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned frame = (unsigned)(item / lay.n_ranks);
            // (a frame that ran into the launch's pass limit is not through: FRAME_STALLED | frame tells the list's later items to give up)
            __hip_atomic_store(&lay.fq.frames_done[rank], st.done == LOOP_DONE ? frame + 1u : (FRAME_STALLED | frame), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!(QUEUE && queued)) break;
    __syncthreads();
    }
}


// ------------------------------------------------------------------ FAST mode (SURVEY.md section 7, 8b `mode`)
// Within a frozen hydro frame the photons are mutually independent and exponential free paths are memoryless, so the frame can be run
// photon by photon instead of event by event: every lane takes ONE photon through the whole frame on its own clock -- locate, optical
// depth, free-path draw, advance, scatter, again -- with per-photon keyed random numbers.  Statistically equivalent to the loop of
// mcrat.c:761-851, not sequence-equivalent: which photon scatters when is no longer a property of the list.  What the reference does to
// a photon between its own scatterings is kept in distribution: it re-locates the photon and redraws its free path whenever ANY photon
// of the rank scatters, i.e. the cell and optical depth a flight uses are refreshed a number of times per frame; here they are
// refreshed at every boundary of `windows` equal time windows (and after each own scattering).  The same device functions as the
// exact loop do the work (fast_one, slow_one, scatter_core, commit_scatter), so the physics is the exact mode's line for line.
// A pass of the workgroup: phase A, every lane its own photon (streaming, like step_kernel's phase 1 + slow path); the lanes whose
// photon scatters in this pass queue it in LDS (ballot-free: one LDS atomic each, few per pass); phase B, the queue worked off with dense
// lanes -- lane j scatters queue entry j -- so that the 3 500-instruction event code runs with full lanes where there are many events.
#ifndef FAST_BLOCK_THREADS
#define FAST_BLOCK_THREADS 256
#endif
constexpr int FAST_BLOCK = FAST_BLOCK_THREADS;
constexpr uint32_t RNG_FAST_FREEPATH = 8u;

template <int DIMS, int GEOM, bool STOKES>
__global__ __launch_bounds__(FAST_BLOCK, 2) void fast_frame_kernel(PhotonDev ph, HydroDev hy, RngKey key, double remaining_time, int windows,
                                                                   int max_passes, FastCounts *__restrict__ counts, FastLists lists)
{
    __shared__ int s_q[2 * FAST_BLOCK];
    __shared__ int s_nq;
    __shared__ unsigned long long s_cnt[6];
#ifdef MCRAT_DIAG
    __shared__ LoopState s_stamps;                   // scatter_core's shader-clock stamps of the diagnostic build need somewhere to go
    LoopState *const stamp_state = &s_stamps;
#else
    LoopState *const stamp_state = nullptr;          // (never dereferenced: MC_STAMP is empty)
#endif
    const int tid = threadIdx.x;
    const PtrCols pcols(ph);                                          // the accessor the shared device functions take (photon_cols.hpp)
    const int i0 = 2 * (blockIdx.x * FAST_BLOCK + tid);               // a lane takes a pair of photons through the frame (their re-locations in lockstep)
    int first = 0, len = ph.n;
    if (lists.desc) {
        // the lists of a rank pool (windows of lists.stride slots, a multiple of 512: a workgroup lies within one list): each list with
        // its own seed, stream and frame time and list-local slot numbers in its keys -- the photons a list ends up with do not depend
        // on which other lists share the pool, as in the exact mode
        const int r = (2 * blockIdx.x * FAST_BLOCK) / lists.stride;
        const RankDesc d = lists.desc[r];
        first = r * lists.stride;
        len = d.len;
        remaining_time = lists.remaining_time[r];
        if (lists.windows) windows = lists.windows[r];                // the list's own refresh cadence (what IT looked like last frame, not the pool)
        if (len <= 0 || !(remaining_time > 0)) return;
        key.seed = d.seed; key.stream = d.stream; key.slot_base = 0;
        counts += r;
    }
    const uint32_t rng_first = key.slot_base - (uint32_t)first;       // key slot of photon i: i + rng_first
    const double window = remaining_time / (double)windows;
    double t_left[2], w_left[2];
    bool done[2], moves[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = i0 + k;
        const bool have = i - first < len && i < ph.n;
        const unsigned fl0 = have ? (unsigned)ph.flags[i] : 0u;
        moves[k] = (fl0 & FLAG_MOVES) != 0;
        t_left[k] = remaining_time; w_left[k] = window;
        done[k] = !(have && (fl0 & FLAG_VALID)) || !(remaining_time > 0);
    }
    int relocated = 0, not_found = 0;
    unsigned steps = 0, scatt = 0, rej = 0;
    if (tid < 6) s_cnt[tid] = 0;
    int pass = 0;
    for (; pass < max_passes; ++pass) {
        if (tid == 0) s_nq = 0;
        __syncthreads();
        if (!done[0] || !done[1]) {
            // the photons' state lives in the columns between passes (61 B in, 24 B out per pass, like step_kernel): nothing but the
            // clocks stays in registers across phase B, whose event code needs them all
            const int slot[2] = {i0, i0 + 1};                                  // (both below the columns' padded capacity)
            unsigned fl[2];
            int cell[2], queue[2], code[2];
            double r0[2], r1[2], r2[2], a0[2], a1[2], a2[2], t[2];
            uint64_t bits[2];
            bool act[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = slot[k];
                act[k] = !done[k];
                if (!act[k]) {                                              // finished (or no photon): its half of the lane idles
                    fl[k] = 0; cell[k] = -1; queue[k] = 0; code[k] = -1; bits[k] = 0; t[k] = 0;
                    r0[k] = r1[k] = r2[k] = a0[k] = a1[k] = a2[k] = 0;
                    continue;
                }
                fl[k] = ph.flags[i];
                cell[k] = ph.idx[i];
                r0[k] = ph.r0[i]; r1[k] = ph.r1[i]; r2[k] = ph.r2[i];
                const double ntau = ph.ntau[i];
                const Philox4 blk = keyed_block(key.seed, (uint64_t)pass, (uint32_t)i + rng_first, RNG_FAST_FREEPATH, key.stream);
                bits[k] = (uint64_t)blk.w[0] | ((uint64_t)blk.w[1] << 32);
                t[k] = fast_one<DIMS, GEOM, false>(pcols, hy, i, fl[k], cell[k], r0[k], r1[k], r2[k], ntau, bits[k], queue[k], code[k], a0[k], a1[k], a2[k]);
                const bool inside = (cell[k] != -1) && phys::in_domain<DIMS>(hy, a0[k], a1[k], a2[k]);
                if (pass == 0 && inside && queue[k] != 1) {                 // find_nearest_grid_switch = 1 on a new frame (mcrat.c:756)
                    queue[k] = 1;
                    code[k] = phys::grid_bucket_of<DIMS>(hy.grid, a0[k], a1[k], a2[k]);
                }
                if (!inside) cell[k] = -1;                                  // (fast_one has stored it, mclib.c:592)
            }
            if constexpr (!TABLE_MODE) {                                    // the pair's re-locations in lockstep; what the hints do not settle goes on below
                bool todo[2] = {queue[0] == 1, queue[1] == 1};
                if (todo[0] || todo[1]) {
                    double tt[2];
                    relocate_lockstep<DIMS, GEOM, 2>(pcols, hy, slot, todo, r0, r1, a0, a1, a2, code, bits, fl, true, tt, relocated);
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        if (queue[k] == 1 && !todo[k]) { t[k] = tt[k]; queue[k] = 0; cell[k] = ph.idx[slot[k]]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (!act[k]) continue;
                const int i = slot[k];
                if (queue[k]) {
                    TableSite ts(key, (unsigned long long)pass);
                    ts.slot_base = rng_first;
                    t[k] = slow_one<DIMS, GEOM>(pcols, hy, i, queue[k] == 1, code[k], true, bits[k], relocated, not_found, ts);
                    cell[k] = ph.idx[i];
                }
                steps += 1;
                double adv;
                if (cell[k] == -1) {                                        // outside the frame's cells: streams to the end of the frame
                    adv = t_left[k];
                    t_left[k] = 0;
                    done[k] = true;
                } else {
                    const double limit = fmin(t_left[k], w_left[k]);
                    if (t[k] < limit) {
                        adv = t[k];
                        t_left[k] -= adv; w_left[k] -= adv;
                        s_q[atomicAdd(&s_nq, 1)] = i;                       // scatters at the end of this flight: phase B
                    } else if (limit == t_left[k]) {
                        adv = limit;
                        t_left[k] = 0;
                        done[k] = true;
                    } else {
                        adv = limit;
                        t_left[k] -= adv;
                        w_left[k] = window;                                 // a window boundary: the next pass re-locates and redraws
                    }
                }
                if (moves[k]) {
                    const double u0 = ph.u0[i], u1 = ph.u1[i], u2 = ph.u2[i];
                    ph.r0[i] = r0[k] + u0 * adv; ph.r1[i] = r1[k] + u1 * adv; ph.r2[i] = r2[k] + u2 * adv;
                }
            }
        }
        __syncthreads();
        const int nq = s_nq;
        for (int j = tid; j < nq; j += FAST_BLOCK) {                      // phase B: one queued photon per lane (at most two rounds)
            const int k = s_q[j];
            const int kc = ph.idx[k];
            double p[4] = {ph.p0[k], ph.p1[k], ph.p2[k], ph.p3[k]};
            double pc[4] = {ph.c0[k], ph.c1[k], ph.c2[k], ph.c3[k]};
            const double r[3] = {ph.r0[k], ph.r1[k], ph.r2[k]};
            double s[4] = {1, 0, 0, 0};
            if constexpr (STOKES) { s[0] = ph.s0[k]; s[1] = ph.s1[k]; s[2] = ph.s2[k]; s[3] = ph.s3[k]; }
            const unsigned kf = ph.flags[k];
            double fluid_temp, tau_new;
            if (scatter_core<DIMS, GEOM, STOKES, false>(hy, stamp_state, key, (unsigned long long)pass, (uint32_t)k + rng_first, kc, r, p, pc, s,
                                                        fluid_temp, tau_new)) {
                commit_scatter<STOKES>(pcols, k, p, pc, s, r, tau_new, kf);
                scatt += 1;
            } else {
                rej += 1;
            }
        }
        if (!__syncthreads_or(!done[0] || !done[1])) { pass += 1; break; }
    }
    // the workgroup's counters
    unsigned long long v[6] = {(unsigned long long)steps, (unsigned long long)scatt, (unsigned long long)rej, (unsigned long long)relocated, (unsigned long long)not_found, (unsigned long long)((done[0] ? 0 : 1) + (done[1] ? 0 : 1))};
    for (int c = 0; c < 6; ++c) {
        unsigned long long x = v[c];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
        if ((tid & 63) == 0 && x) atomicAdd(&s_cnt[c], x);
    }
    __syncthreads();
    if (tid == 0) {
        atomicAdd(&counts->photon_steps, s_cnt[0]);
        atomicAdd(&counts->scatterings, s_cnt[1]);
        atomicAdd(&counts->kn_rejections, s_cnt[2]);
        atomicAdd(&counts->relocated, s_cnt[3]);
        atomicAdd(&counts->not_found, s_cnt[4]);
        atomicAdd(&counts->unfinished, s_cnt[5]);
        atomicMax(&counts->passes, (unsigned long long)pass);
    }
}

// ------------------------------------------------------------------ one list over several GPUs, one clock
// (device_types.hpp, "ScProposal").  A round is: step_kernel (returns at once in a midpass round) ->
// sc_midpass_kernel (returns at once unless midpass) -> sc_propose_kernel -> all-gather of the proposals (host:
// RCCL / MPI) -> sc_resolve_kernel.  Global slot = key.slot_base + local slot; ties are broken by global slot, as
// the single-list engine breaks them by slot.
__device__ __forceinline__ bool sc_less(double ta, long long ga, double tb, long long gb)
{
    return (ta < tb) || (ta == tb && ga < gb);
}

// midpass: apply the advance the walk has accumulated so far (mclib.c:1138 for every candidate already tried) and
// collect the candidates beyond the cursor from time_to_scatter -- the free times of this iteration are NOT redrawn
__global__ __launch_bounds__(STEP_BLOCK) void sc_midpass_kernel(PhotonDev ph, const LoopState *__restrict__ st, const ScState *__restrict__ sc,
                                                                RngKey key, Cand *__restrict__ block_min, Shortlist *sl)
{
    __shared__ double s_wt[STEP_BLOCK / 64];
    __shared__ int s_wi[STEP_BLOCK / 64];
    if (st->done != LOOP_MIDPASS) return;
    const int nseg = st->nseg;
    const double cur_t = sc->cursor_t, cut = sc->cut_mid;
    const long long cur_g = sc->cursor_gid;
    const int G = gridDim.x, lane = threadIdx.x & 63;
    const int nchunks = ph.n_pad / (2 * STEP_BLOCK);
    MinCand best;
    best.init();
    for (int chunk = xcd_contiguous_chunk(blockIdx.x, G); chunk < nchunks; chunk += G) {
        const int i0 = (chunk * STEP_BLOCK + threadIdx.x) << 1;
        PairIn in = load_pair(ph, i0, nseg > 0);
        if (nseg > 0) {
            const bool m0 = (in.FL.x & FLAG_MOVES) != 0, m1 = (in.FL.y & FLAG_MOVES) != 0;
            for (int s = 0; s < nseg; ++s) {
                const double t = st->seg[s];
                if (m0) { in.R0.x += in.U0.x * t; in.R1.x += in.U1.x * t; in.R2.x += in.U2.x * t; }
                if (m1) { in.R0.y += in.U0.y * t; in.R1.y += in.U1.y * t; in.R2.y += in.U2.y * t; }
            }
            *reinterpret_cast<double2 *>(ph.r0 + i0) = in.R0;
            *reinterpret_cast<double2 *>(ph.r1 + i0) = in.R1;
            *reinterpret_cast<double2 *>(ph.r2 + i0) = in.R2;
        }
        const double2 T = *reinterpret_cast<const double2 *>(ph.tts + i0);
        double t0 = T.x, t1 = T.y;
        if (t0 != t0) t0 = INFINITY;
        if (t1 != t1) t1 = INFINITY;
        if ((in.FL.x & FLAG_VALID) && sc_less(cur_t, cur_g, t0, (long long)key.slot_base + i0)) {
            best.offer(t0, i0);
            if (t0 < cut) shortlist_push(sl, t0, i0);
        }
        if ((in.FL.y & FLAG_VALID) && sc_less(cur_t, cur_g, t1, (long long)key.slot_base + i0 + 1)) {
            best.offer(t1, i0 + 1);
            if (t1 < cut) shortlist_push(sl, t1, i0 + 1);
        }
    }
    wave_min_pair_dpp(best.t, best.i);
    if (lane == 0) { s_wt[threadIdx.x >> 6] = best.t; s_wi[threadIdx.x >> 6] = best.i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        MinCand m;
        m.init();
#pragma unroll
        for (int w = 0; w < STEP_BLOCK / 64; ++w) m.offer(s_wt[w], s_wi[w]);
        Cand c;
        c.t = m.t; c.idx = m.i; c.pad = 0;
        block_min[blockIdx.x] = c;
    }
}

// this GPU's earliest candidates of the round, with their data, into `out`
template <bool STOKES>
__device__ __forceinline__ void sc_propose_body(const PhotonDev &ph, LoopState *st, const ScState *__restrict__ sc, const RngKey &key,
                                                const Cand *__restrict__ block_min, int n_blocks, Shortlist *sl, ScProposal *__restrict__ out)
{
    static_assert(EVENT_BLOCK == SHORTLIST_CAP, "one thread per shortlist entry");
    __shared__ Cand s_raw[SHORTLIST_CAP];
    __shared__ Cand s_list[SHORTLIST_CAP];
    __shared__ double s_wt[EVENT_BLOCK / 64];
    __shared__ int s_wi[EVENT_BLOCK / 64];
    const int tid = threadIdx.x;
    const int done = st->done;
    if (done == LOOP_DONE || done == LOOP_SC_GAVE_UP) return;
    const bool midpass = done == LOOP_MIDPASS;
    const double cut = midpass ? sc->cut_mid : st->t_cut;          // the threshold the shortlist was filled with
    const int n_raw = sl->count;
    const Cand mine = sl->items[tid];
    MinCand m;
    m.init();
    for (int e = tid; e < n_blocks; e += EVENT_BLOCK) m.offer(block_min[e].t, block_min[e].idx);
    if (tid < n_raw) s_raw[tid] = mine;
    wave_min_pair_dpp(m.t, m.i);
    if ((tid & 63) == 0) { s_wt[tid >> 6] = m.t; s_wi[tid >> 6] = m.i; }
    __syncthreads();
    MinCand g;
    g.init();
#pragma unroll
    for (int wv = 0; wv < EVENT_BLOCK / 64; ++wv) g.offer(s_wt[wv], s_wi[wv]);

    int n_list = (n_raw > SHORTLIST_CAP) ? 0 : n_raw;              // overflowed: incomplete, ignore it
    if (tid < n_list) {
        const Cand me = s_raw[tid];
        int rank = 0;
        for (int j = 0; j < n_list; ++j) rank += cand_less(s_raw[j].t, s_raw[j].idx, me.t, me.idx) ? 1 : 0;
        s_list[rank] = me;
    }
    // what this GPU vouches for: with a usable shortlist, everything below `cut` (or up to its SC_K-th entry);
    // without one, only its minimum
    double hor_t;
    long long hor_g;
    int n_out;
    if (n_list == 0) {
        if (tid == 0) { s_list[0].t = g.t; s_list[0].idx = g.i; }
        if (g.i == INT_MAX) { n_out = 0; hor_t = INFINITY; hor_g = LLONG_MAX; }      // nothing left to try on this GPU
        else { n_out = 1; hor_t = g.t; hor_g = (long long)key.slot_base + g.i; }
    } else if (n_list > SC_K) {
        n_out = SC_K;
        hor_t = 0; hor_g = 0;                                       // filled below from the SC_K-th entry
    } else {
        n_out = n_list;
        hor_t = cut; hor_g = -1;
    }
    __syncthreads();
    if (n_list > SC_K) { hor_t = s_list[SC_K - 1].t; hor_g = (long long)key.slot_base + s_list[SC_K - 1].idx; }
    if (tid < n_out) {
        const int i = s_list[tid].idx;
        ScRecord rec;
        rec.t = s_list[tid].t;
        rec.gid = (long long)key.slot_base + i;
        rec.r[0] = ph.r0[i]; rec.r[1] = ph.r1[i]; rec.r[2] = ph.r2[i];
        rec.u[0] = ph.u0[i]; rec.u[1] = ph.u1[i]; rec.u[2] = ph.u2[i];
        rec.p[0] = ph.p0[i]; rec.p[1] = ph.p1[i]; rec.p[2] = ph.p2[i]; rec.p[3] = ph.p3[i];
        rec.pc[0] = ph.c0[i]; rec.pc[1] = ph.c1[i]; rec.pc[2] = ph.c2[i]; rec.pc[3] = ph.c3[i];
        if constexpr (STOKES) { rec.s[0] = ph.s0[i]; rec.s[1] = ph.s1[i]; rec.s[2] = ph.s2[i]; rec.s[3] = ph.s3[i]; }
        else { rec.s[0] = 1; rec.s[1] = 0; rec.s[2] = 0; rec.s[3] = 0; }
        rec.cell = ph.idx[i];
        rec.flags = ph.flags[i];
        out->rec[tid] = rec;
    }
    if (tid == 0) {
        out->n = n_out; out->pad0 = 0; out->horizon_t = hor_t; out->horizon_gid = hor_g; out->pad1 = 0;
        sl->count = 0;
        if (midpass) st->nseg = 0;                                  // sc_midpass_kernel has applied them
    }
}

template <bool STOKES>
__global__ __launch_bounds__(EVENT_BLOCK) void sc_propose_kernel(PhotonDev ph, LoopState *st, const ScState *__restrict__ sc, RngKey key,
                                                                 const Cand *__restrict__ block_min, int n_blocks, Shortlist *sl,
                                                                 ScProposal *__restrict__ out, ScFold fold)
{
    sc_propose_body<STOKES>(ph, st, sc, key, block_min, n_blocks, sl, out);
    if (fold.on) {                                                  // the exchange's push, in line (a finished or parked frame still stamps its rounds)
        __threadfence();
        __syncthreads();
        sc_push_body(out, fold.peers, fold.my_flags, fold.world, fold.rank);
    }
}

// every GPU: merge the gathered proposals, walk the chain as photonEvent does (mclib.c:1128-1339), store the
// scatter if the photon is ours, and do the bookkeeping of mcrat.c:782-784 / 837-845 -- identically everywhere
template <int DIMS, int GEOM, bool STOKES>
__global__ __launch_bounds__(EVENT_BLOCK) void sc_resolve_kernel(PhotonDev ph, HydroDev hy, LoopState *st, ScState *sc, RngKey key,
                                                                 const ScProposal *all, int world, ScFold fold)
{
    __shared__ ScRecord s_rec[SC_MAX_WORLD * SC_K];
    __shared__ int s_order[SC_MAX_WORLD * SC_K];
    __shared__ int s_off[SC_MAX_WORLD + 1];
    __shared__ double s_seg[MAX_SEG];
    __shared__ int s_failed;
    const int tid = threadIdx.x;
    if (fold.on) {                                                  // the exchange's wait, in line: the round's proposals into `all` (= fold.gathered)
        if (!sc_wait_body(fold.my_flags, fold.recv, fold.gathered, fold.world, fold.max_spins, st, &s_failed)) return;
    }
    const int done = st->done;
    if (done == LOOP_DONE || done == LOOP_SC_GAVE_UP) return;
    if (tid == 0) {
        int o = 0;
        for (int g = 0; g < world; ++g) { s_off[g] = o; int n = all[g].n; n = n < 0 ? 0 : (n > SC_K ? SC_K : n); o += n; }
        s_off[world] = o;
    }
    __syncthreads();
    const int E = s_off[world];
    if (tid < world * SC_K) {
        const int g = tid / SC_K, k = tid % SC_K;
        if (k < s_off[g + 1] - s_off[g]) s_rec[s_off[g] + k] = all[g].rec[k];
    }
    __syncthreads();
    if (tid < E) {                                                   // rank sort by (t, gid); pairs are distinct
        int rank = 0;
        for (int j = 0; j < E; ++j) rank += sc_less(s_rec[j].t, s_rec[j].gid, s_rec[tid].t, s_rec[tid].gid) ? 1 : 0;
        s_order[rank] = tid;
    }
    __syncthreads();
    if (WAVE_WALK ? tid >= 64 : tid != 0) return;         // the walk: one wavefront, every lane the same values (see event_block)

    double hor_t = INFINITY;
    long long hor_g = LLONG_MAX;
    for (int g = 0; g < world; ++g)
        if (sc_less(all[g].horizon_t, all[g].horizon_gid, hor_t, hor_g)) { hor_t = all[g].horizon_t; hor_g = all[g].horizon_gid; }

    const bool midpass = done == LOOP_MIDPASS;
    const unsigned long long iter = st->iteration;
    const double dt_max = st->remaining_time;
    double old_scatt_time = midpass ? sc->old_scatt_time : 0.0;
    bool first = midpass ? (sc->first != 0) : true;
    double t_first = midpass ? sc->t_first : ((E > 0) ? s_rec[s_order[0]].t : INFINITY);
    int nseg = midpass ? 0 : 0;
    int last_idx = st->last_scattered_index;
    long long rej = 0;
    int skip = -1;
    double dt = 0;
    bool decided = false;
    double last_t = midpass ? sc->cursor_t : 0.0;
    long long last_g = midpass ? sc->cursor_gid : -1;
    const long long lo = (long long)key.slot_base, hi = lo + ph.n;

    for (int e = 0; e < E && !decided; ++e) {
        const ScRecord &c = s_rec[s_order[e]];
        if (sc_less(hor_t, hor_g, c.t, c.gid)) break;               // beyond what every GPU vouches for
        if (nseg == MAX_SEG) break;                                 // continue in a midpass round: segments are never merged
        const bool in_frame = c.t < dt_max;
        if (!(first && !in_frame)) last_idx = (int)c.gid;           // *scattered_ph_index, as a GLOBAL slot
        first = false;
        if (!in_frame) {                                            // mclib.c:1327-1335
            s_seg[nseg++] = dt_max - old_scatt_time;
            dt = dt_max;
            decided = true;
            break;
        }
        s_seg[nseg++] = c.t - old_scatt_time;                       // mclib.c:1138
        old_scatt_time = c.t;
        last_t = c.t; last_g = c.gid;
        if (c.cell == -1) continue;
        double r[3] = {c.r[0], c.r[1], c.r[2]};
        if (c.flags & FLAG_MOVES) {
            for (int k = 0; k < nseg; ++k) {
                r[0] += c.u[0] * s_seg[k];
                r[1] += c.u[1] * s_seg[k];
                r[2] += c.u[2] * s_seg[k];
            }
        }
        double p[4] = {c.p[0], c.p[1], c.p[2], c.p[3]};
        double pc[4] = {c.pc[0], c.pc[1], c.pc[2], c.pc[3]};
        double s[4] = {c.s[0], c.s[1], c.s[2], c.s[3]};
        double fluid_temp, tau_new;
        if (!scatter_core<DIMS, GEOM, STOKES, WAVE_WALK>(hy, st, key, iter, (uint32_t)c.gid, c.cell, r, p, pc, s, fluid_temp, tau_new)) {
            rej += 1;
            continue;
        }
        if (c.gid >= lo && c.gid < hi) {                            // ours
            skip = (int)(c.gid - lo);
            commit_scatter<STOKES>(PtrCols(ph), skip, p, pc, s, r, tau_new, c.flags);
        }
        st->frame_scatt_cnt += 1;                                   // mclib.c:1318
        st->last_scattered_temp = fluid_temp;
        dt = c.t;
        decided = true;
    }
    if (!decided && hor_t == INFINITY && hor_g == LLONG_MAX && nseg < MAX_SEG) {
        // every GPU has run out of slots to try: the loop of mclib.c:1128 ends
        dt = first ? dt_max : old_scatt_time;
        decided = true;
    }
    st->kn_rejections += rej;
    st->last_scattered_index = last_idx;
    st->nseg = nseg;
    for (int k = 0; k < MAX_SEG; ++k) st->seg[k] = (k < nseg) ? s_seg[k] : 0.0;
    sc->rounds += 1;
    if (decided) {                                                  // mcrat.c:782-784 / 837-845
        st->time_now += dt;
        const double rem = dt_max - dt;
        st->remaining_time = rem;
        st->last_time_step = dt;
        st->iteration = iter + 1;
        st->iterations += 1;
#ifndef MCRAT_DIAG
        st->slot_steps += ph.n;
#endif
        st->done = !(rem > 0) ? LOOP_DONE : 0;
        st->skip_idx = skip;
        if (t_first < INFINITY) {
            const double t_est = st->t_est;
            const double est = (t_est > 0) ? 0.875 * t_est + 0.125 * t_first : t_first;
            st->t_est = est;
            st->t_cut = 8.0 * est;
        }
    } else {
        st->done = LOOP_MIDPASS;
        st->skip_idx = -1;
        st->rescans += 1;
        sc->midpass_rounds += 1;
        sc->cursor_t = last_t;
        sc->cursor_gid = last_g;
        sc->old_scatt_time = old_scatt_time;
        sc->first = first ? 1 : 0;
        sc->t_first = t_first;
        const double est = (st->t_est > 0) ? st->t_est : last_t;
        sc->cut_mid = last_t + 8.0 * est;
    }
}

// ------------------------------------------------------------------ flush
__global__ __launch_bounds__(STEP_BLOCK) void flush_kernel(PhotonDev ph, const LoopState *__restrict__ st)
{
    const int nseg = st->nseg;
    if (nseg <= 0) return;
    const int skip = st->skip_idx;
    for (int i = blockIdx.x * STEP_BLOCK + threadIdx.x; i < ph.n; i += gridDim.x * STEP_BLOCK) {
        if ((ph.flags[i] & FLAG_MOVES) && i != skip) {
            const double u0 = ph.u0[i], u1 = ph.u1[i], u2 = ph.u2[i];
            double r0 = ph.r0[i], r1 = ph.r1[i], r2 = ph.r2[i];
            for (int s = 0; s < nseg; ++s) {
                const double t = st->seg[s];
                r0 += u0 * t;
                r1 += u1 * t;
                r2 += u2 * t;
            }
            ph.r0[i] = r0; ph.r1[i] = r1; ph.r2[i] = r2;
        }
    }
}

__global__ void clear_pending_kernel(LoopState *st)
{
    st->nseg = 0;
    st->skip_idx = -1;
}

// ------------------------------------------------------------------ per-cell exp(x) K_2(x)
__global__ void k2e_kernel(const double *__restrict__ temp, double *__restrict__ k2e, int M)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    const double T = temp[c];
    double v = 0.0;
    if (T >= 1e7) v = phys::bessel_k2_scaled((M_EL * C_LIGHT * C_LIGHT) / (K_B * T));
    k2e[c] = v;
}

// ------------------------------------------------------------------ per-frame reductions (phMinMax, phScattStats, averagePhotonEnergy)
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(256) void reduce_kernel(PhotonDev ph, ReducePartial *__restrict__ out)
{
    __shared__ double s[4][10];
    __shared__ long long s_cnt[4];
    double r_min = 1.7976931348623157e308, r_max = 0, th_min = 1.7976931348623157e308, th_max = 0;
    double sum_scatt = 0, sum_r = 0, e_sum = 0, w_sum = 0, max_s = 0, min_s = 2147483647.0;
    long long count = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ph.n; i += gridDim.x * 256) {
        if (!(ph.flags[i] & FLAG_VALID)) continue;                      // a rank pool's unused slots belong to no list
        const double x = ph.r0[i], y = ph.r1[i], z = ph.r2[i], w = ph.weight[i];
        const double r = sqrt(x * x + y * y + z * z);
        if (w != 0) {                                                   // mclib.c:1479
            const double th = acos(z / r);
            r_max = fmax(r_max, r); r_min = fmin(r_min, r);
            th_max = fmax(th_max, th); th_min = fmin(th_min, th);
        }
        const double ns = ph.num_scatt[i];                              // mclib.c:1405-1421 (CYCLOSYNCHROTRON off: no filter)
        sum_scatt += ns; sum_r += r;
        max_s = fmax(max_s, ns); min_s = fmin(min_s, ns);
        e_sum += ph.p0[i] * w; w_sum += w;                              // mclib.c:1377-1378
        count += 1;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    r_min = wave_min(r_min); r_max = wave_max(r_max); th_min = wave_min(th_min); th_max = wave_max(th_max);
    sum_scatt = wave_sum(sum_scatt); sum_r = wave_sum(sum_r); e_sum = wave_sum(e_sum); w_sum = wave_sum(w_sum);
    max_s = wave_max(max_s); min_s = wave_min(min_s);
    double cd = wave_sum((double)count);
    if (lane == 0) {
        s[wv][0] = r_min; s[wv][1] = r_max; s[wv][2] = th_min; s[wv][3] = th_max; s[wv][4] = sum_scatt;
        s[wv][5] = sum_r; s[wv][6] = e_sum; s[wv][7] = w_sum; s[wv][8] = max_s; s[wv][9] = min_s;
        s_cnt[wv] = (long long)cd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ReducePartial p;
        p.r_min = fmin(fmin(s[0][0], s[1][0]), fmin(s[2][0], s[3][0]));
        p.r_max = fmax(fmax(s[0][1], s[1][1]), fmax(s[2][1], s[3][1]));
        p.th_min = fmin(fmin(s[0][2], s[1][2]), fmin(s[2][2], s[3][2]));
        p.th_max = fmax(fmax(s[0][3], s[1][3]), fmax(s[2][3], s[3][3]));
        p.sum_scatt = (s[0][4] + s[1][4]) + (s[2][4] + s[3][4]);
        p.sum_r = (s[0][5] + s[1][5]) + (s[2][5] + s[3][5]);
        p.e_sum = (s[0][6] + s[1][6]) + (s[2][6] + s[3][6]);
        p.w_sum = (s[0][7] + s[1][7]) + (s[2][7] + s[3][7]);
        p.max_scatt = fmax(fmax(s[0][8], s[1][8]), fmax(s[2][8], s[3][8]));
        p.min_scatt = fmin(fmin(s[0][9], s[1][9]), fmin(s[2][9], s[3][9]));
        p.count = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        out[blockIdx.x] = p;
    }
}

// ------------------------------------------------------------------ cell lookup (A/B of findContainingBlock)
template <int DIMS>
__global__ void lookup_kernel(HydroDev hy, int n, const double *a0, const double *a1, const double *a2, int *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = phys::find_containing_block<DIMS>(hy, a0[i], a1[i], (DIMS == DIM_THREE) ? a2[i] : 0.0);
}

// ------------------------------------------------------------------ launchers
int step_grid_blocks(int n_pad)
{
    const int pairs = n_pad / 2;
    int blocks = (pairs + STEP_BLOCK - 1) / STEP_BLOCK;
    // all workgroups resident at once: every workgroup streams its chunks back to back and pays the
    // latency-bound slow path once.  MCRAT_HIP_STEP_BLOCKS overrides the cap (tuning).
    int cap = 768;                        // 3 workgroups per CU, all resident at the kernel's register budget
    if (const char *e = getenv("MCRAT_HIP_STEP_BLOCKS")) { const int v = atoi(e); if (v > 0) cap = v; }
    if (blocks > cap) {                   // balance: every workgroup gets the same number of chunks
        const int per_block = (blocks + cap - 1) / cap;
        blocks = (blocks + per_block - 1) / per_block;
    }
    if (blocks < 1) blocks = 1;
    return blocks;
}

// run f(integral_constant<DIMS>, integral_constant<GEOM>) for the (DIMENSIONS, GEOMETRY) pairs the reference
// supports (mcrat.h:196-204): 2-D / 2.5-D cartesian, cylindrical, spherical; 3-D cartesian, spherical, polar
template <int V> using ic = std::integral_constant<int, V>;

template <class F>
static hipError_t dispatch(const KernelConfig &kc, F &&f)
{
    const int d = kc.dimensions, g = kc.geometry;
    if (d != MCRAT_TU_DIMS) return hipErrorInvalidValue;          // launchers.hip routes by DIMENSIONS
    // (-DMCRAT_DEV_GEOM=<g>: a development build with the kernels of one GEOMETRY only -- a third of the compile time; every other geometry is refused)
#ifdef MCRAT_DEV_GEOM
#define MCRAT_GEOM_BUILT(G) ((G) == MCRAT_DEV_GEOM)
#else
#define MCRAT_GEOM_BUILT(G) true
#endif
    auto run = [&](auto D, auto G) { if constexpr (MCRAT_GEOM_BUILT(decltype(G)::value)) f(D, G); };
    const bool built = MCRAT_GEOM_BUILT(g);
    if (!built) return hipErrorInvalidValue;
#if MCRAT_TU_DIMS == 0
    if (g == GEOM_CARTESIAN) run(ic<DIM_TWO>{}, ic<GEOM_CARTESIAN>{});
    else if (g == GEOM_CYLINDRICAL) run(ic<DIM_TWO>{}, ic<GEOM_CYLINDRICAL>{});
    else if (g == GEOM_SPHERICAL) run(ic<DIM_TWO>{}, ic<GEOM_SPHERICAL>{});
#elif MCRAT_TU_DIMS == 1
    if (g == GEOM_CARTESIAN) run(ic<DIM_TWO_POINT_FIVE>{}, ic<GEOM_CARTESIAN>{});
    else if (g == GEOM_CYLINDRICAL) run(ic<DIM_TWO_POINT_FIVE>{}, ic<GEOM_CYLINDRICAL>{});
    else if (g == GEOM_SPHERICAL) run(ic<DIM_TWO_POINT_FIVE>{}, ic<GEOM_SPHERICAL>{});
#else
    if (g == GEOM_CARTESIAN) run(ic<DIM_THREE>{}, ic<GEOM_CARTESIAN>{});
    else if (g == GEOM_SPHERICAL) run(ic<DIM_THREE>{}, ic<GEOM_SPHERICAL>{});
    else if (g == GEOM_POLAR) run(ic<DIM_THREE>{}, ic<GEOM_POLAR>{});
#endif
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *block_min, int blocks, Shortlist *sl, hipStream_t stream)
{
    return dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        if (force_relocate)
            step_kernel<DV, GV, true><<<dim3(blocks), dim3(STEP_BLOCK), 0, stream>>>(ph, hy, st, key, block_min, sl);
        else
            step_kernel<DV, GV, false><<<dim3(blocks), dim3(STEP_BLOCK), 0, stream>>>(ph, hy, st, key, block_min, sl);
    });
}

hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream)
{
    return dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        if (kc.stokes)
            event_kernel<DV, GV, true><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, key, block_min, n_blocks, sl, KeyedSource());
        else
            event_kernel<DV, GV, false><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, key, block_min, n_blocks, sl, KeyedSource());
    });
}

// the second half of a pass with the caller's tape: the free-path draws in slot order, then the event reading on from where they stopped
hipError_t launch_tape_pass(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key, const TapeDev &tape,
                            Cand *block_min, int n_blocks, Shortlist *sl, hipStream_t stream)
{
    tape_draw_kernel<<<dim3(1), dim3(TAPE_BLOCK), 0, stream>>>(ph, st, tape, block_min, n_blocks, sl);
    TapeSource src;
    src.t = tape;
    return dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        if (kc.stokes)
            event_kernel<DV, GV, true, TapeSource><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, key, block_min, n_blocks, sl, src);
        else
            event_kernel<DV, GV, false, TapeSource><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, key, block_min, n_blocks, sl, src);
    });
}

#ifndef RANK_SMALL
#define RANK_SMALL 128                 // threads of the small workgroup (block == 128 below); -DRANK_SMALL=64 for the A/B of one-wave lists
#endif
hipError_t launch_rank_loop(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *states, RngKey key,
                            int n_ranks, int rank_stride, int longest_list, const RankDesc *desc, CsFrame *cs, const CsHookArgs *hook, long long max_passes,
                            int block, hipStream_t stream, const FrameQueueDev *fq, int n_open)
{
    // block: 64, 128, 256 or 512 threads per list, + 1000 for the build with the fused pass
    const bool fuse = block >= 1000;
    if (fuse) block -= 1000;
    RankLayout lay = {n_ranks, rank_stride, ph.n, desc, cs, hook, FrameQueueDev{}};
    const bool queued = fq && fq->n_frames > 0;
    if (queued) {
        if (cs || n_open <= 0) return hipErrorInvalidValue;   // (cyclo-synchrotron lists go to the host between passes: one frame per launch)
        lay.fq = *fq;
    }
    if (cs && hook && desc) {                      // cyclo-synchrotron lists with the hook inside the loop: columns in HBM/L2, no fused pass
        return dispatch(kc, [&](auto D, auto G) {
            constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
            if (block == 64) {                     // one wavefront per list, eight lists per CU: every resident wave is always at work (no barrier waits)
                if (kc.stokes) rank_loop_kernel<DV, GV, true, false, 64, false, true><<<dim3(n_ranks), dim3(64), 0, stream>>>(ph, hy, states, key, lay, max_passes, 0);
                else rank_loop_kernel<DV, GV, false, false, 64, false, true><<<dim3(n_ranks), dim3(64), 0, stream>>>(ph, hy, states, key, lay, max_passes, 0);
            } else if (block == 128) {
                if (kc.stokes) rank_loop_kernel<DV, GV, true, false, RANK_SMALL, false, true><<<dim3(n_ranks), dim3(RANK_SMALL), 0, stream>>>(ph, hy, states, key, lay, max_passes, 0);
                else rank_loop_kernel<DV, GV, false, false, RANK_SMALL, false, true><<<dim3(n_ranks), dim3(RANK_SMALL), 0, stream>>>(ph, hy, states, key, lay, max_passes, 0);
            } else {
                if (kc.stokes) rank_loop_kernel<DV, GV, true, false, 256, false, true><<<dim3(n_ranks), dim3(256), 0, stream>>>(ph, hy, states, key, lay, max_passes, 0);
                else rank_loop_kernel<DV, GV, false, false, 256, false, true><<<dim3(n_ranks), dim3(256), 0, stream>>>(ph, hy, states, key, lay, max_passes, 0);
            }
        });
    }
    // per-pass columns in LDS (32 B per slot with 128 threads, 61 B with 256: rank_loop_kernel) for lists of up to 1024 photons
    int lds_slots = 0;
    // (two 256-thread lists per CU: 13 KB of static LDS and 61 B per slot each within 160 KB -> 1088 slots; four 128-thread ones at 32 B: 1024)
    // (512 threads -- lists of thousands of photons, one list per CU: 27 KB of static LDS and 32 B per slot within 160 KB -> 4096 slots)
    const int lds_limit = (block == 128) ? 1024 : (block == 512 ? 4096 : 1088);
    if (!getenv("MCRAT_HIP_NO_LDS_LISTS") && longest_list <= lds_limit) lds_slots = (longest_list + 15) & ~15;
    size_t dyn = (size_t)lds_slots * rank_lds_bytes_per_slot(block == 128 ? 128 : (block == 512 ? 512 : 256));
    bool queue_launched = false;
    const hipError_t launched = dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        // static + dynamic LDS may exceed the 64 KiB default: the kernel must be told, and if the runtime refuses
        // the list simply stays in global memory (lds_slots = 0)
        const int grid = queued ? n_open : n_ranks;          // a queue launch: one workgroup per open (frame, list) item
        if (queued) {
            // the queue builds: 256-thread lists with their columns in LDS, with or without the fused pass; anything else is refused and the caller
            // runs the plan frame by frame (engine.hip, mcrat_hip_pool_run_frames)
            if (block != 256 || lds_slots <= 0) return;
            auto launch_q = [&](auto kernel) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) { (void)hipGetLastError(); return; }
                // persistent workgroups: as many as the device holds at once (they are dealt to the XCDs round-robin, an eighth each); more would only
                // start, find their queue empty and leave
                int per_cu = 0, cus = 256, dev = 0, grid = n_open;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel), 256, dyn) == hipSuccess && per_cu > 0 &&
                    hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
                    grid = std::min(n_open, per_cu * cus);
                else
                    (void)hipGetLastError();
                kernel<<<dim3(grid), dim3(256), dyn, stream>>>(ph, hy, states, key, lay, max_passes, lds_slots);
                queue_launched = true;
            };
            if constexpr (!TABLE_MODE && GV != GEOM_SPHERICAL) {
                if (fuse) {
                    if (kc.stokes) launch_q(rank_loop_kernel<DV, GV, true, true, 256, true, false, true>);
                    else launch_q(rank_loop_kernel<DV, GV, false, true, 256, true, false, true>);
                    return;
                }
            }
            if (kc.stokes) launch_q(rank_loop_kernel<DV, GV, true, true, 256, false, false, true>);
            else launch_q(rank_loop_kernel<DV, GV, false, true, 256, false, false, true>);
            return;
        }
        auto launch = [&](auto kernel, auto kernel_global, int threads) {
            if (lds_slots > 0 &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) == hipSuccess) {
                kernel<<<dim3(grid), dim3(threads), dyn, stream>>>(ph, hy, states, key, lay, max_passes, lds_slots);
            } else {
                (void)hipGetLastError();
                kernel_global<<<dim3(grid), dim3(threads), 0, stream>>>(ph, hy, states, key, lay, max_passes, 0);
            }
        };
        // The fused pass exists where engine.hip's choose_rank_block can ask for it: DIRECT optical depths, not in spherical geometry (there it
        // measured slower), 256 or 512 threads.  Everything else has the queue form only -- a third of the instantiations less than building all.
        if constexpr (!TABLE_MODE && GV != GEOM_SPHERICAL) {
            if (fuse && block == 512) {
                if (kc.stokes) launch(rank_loop_kernel<DV, GV, true, true, 512, true>, rank_loop_kernel<DV, GV, true, false, 512, true>, 512);
                else launch(rank_loop_kernel<DV, GV, false, true, 512, true>, rank_loop_kernel<DV, GV, false, false, 512, true>, 512);
                return;
            }
            if (fuse && block != 128) {
                if (kc.stokes) launch(rank_loop_kernel<DV, GV, true, true, 256, true>, rank_loop_kernel<DV, GV, true, false, 256, true>, 256);
                else launch(rank_loop_kernel<DV, GV, false, true, 256, true>, rank_loop_kernel<DV, GV, false, false, 256, true>, 256);
                return;
            }
        }
        if (block == 512) {
            if (kc.stokes) launch(rank_loop_kernel<DV, GV, true, true, 512, false>, rank_loop_kernel<DV, GV, true, false, 512, false>, 512);
            else launch(rank_loop_kernel<DV, GV, false, true, 512, false>, rank_loop_kernel<DV, GV, false, false, 512, false>, 512);
        } else if (block == 128) {                 // (no fused build at 128 threads: it measured no gain on thin frames, round 2, and was 36 instantiations)
            if (kc.stokes) launch(rank_loop_kernel<DV, GV, true, true, RANK_SMALL, false>, rank_loop_kernel<DV, GV, true, false, RANK_SMALL, false>, RANK_SMALL);
            else launch(rank_loop_kernel<DV, GV, false, true, RANK_SMALL, false>, rank_loop_kernel<DV, GV, false, false, RANK_SMALL, false>, RANK_SMALL);
        } else {
            if (kc.stokes) launch(rank_loop_kernel<DV, GV, true, true, 256, false>, rank_loop_kernel<DV, GV, true, false, 256, false>, 256);
            else launch(rank_loop_kernel<DV, GV, false, true, 256, false>, rank_loop_kernel<DV, GV, false, false, 256, false>, 256);
        }
    });
    if (queued && !queue_launched && launched == hipSuccess) return hipErrorNotSupported;      // no queue build of this launch form: frame by frame then
    return launched;
}

hipError_t launch_fast_frame(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, RngKey key, double remaining_time, int windows,
                             int max_passes, FastCounts *counts, const FastLists &lists, hipStream_t stream)
{
    if (lists.desc && (lists.stride <= 0 || lists.stride % (2 * FAST_BLOCK) != 0)) return hipErrorInvalidValue;
    const int blocks = (ph.n + 2 * FAST_BLOCK - 1) / (2 * FAST_BLOCK);     // a lane takes two photons
    return dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        if (kc.stokes) fast_frame_kernel<DV, GV, true><<<dim3(blocks), dim3(FAST_BLOCK), 0, stream>>>(ph, hy, key, remaining_time, windows, max_passes, counts, lists);
        else fast_frame_kernel<DV, GV, false><<<dim3(blocks), dim3(FAST_BLOCK), 0, stream>>>(ph, hy, key, remaining_time, windows, max_passes, counts, lists);
    });
}

hipError_t launch_sc_propose(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy, LoopState *st,
                             ScState *sc, RngKey key, Cand *block_min, int blocks, Shortlist *sl, ScProposal *out, const ScFold &fold, hipStream_t stream)
{
    hipError_t e = MCRAT_TU_NS::launch_step(kc, force_relocate, ph, hy, st, key, block_min, blocks, sl, stream);
    if (e != hipSuccess) return e;
    sc_midpass_kernel<<<dim3(blocks), dim3(STEP_BLOCK), 0, stream>>>(ph, st, sc, key, block_min, sl);
    if (kc.stokes) sc_propose_kernel<true><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, st, sc, key, block_min, blocks, sl, out, fold);
    else sc_propose_kernel<false><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, st, sc, key, block_min, blocks, sl, out, fold);
    return hipGetLastError();
}

hipError_t launch_sc_resolve(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, ScState *sc, RngKey key,
                             const ScProposal *all, int world, const ScFold &fold, hipStream_t stream)
{
    return dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        if (kc.stokes) sc_resolve_kernel<DV, GV, true><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, sc, key, all, world, fold);
        else sc_resolve_kernel<DV, GV, false><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, sc, key, all, world, fold);
    });
}

#if defined(MCRAT_DIAG) && !MCRAT_TAU_TABLE_TU && MCRAT_TU_DIMS == 0
extern "C" __attribute__((visibility("default"))) int mcrat_hip_diag_set(int bits)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &bits, sizeof(int)) == hipSuccess ? 0 : -1;
}
#endif

hipError_t launch_flush(const PhotonDev &ph, LoopState *st, int blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(flush_kernel, dim3(blocks), dim3(STEP_BLOCK), 0, stream, ph, st);
    hipLaunchKernelGGL(clear_pending_kernel, dim3(1), dim3(1), 0, stream, st);
    return hipGetLastError();
}

hipError_t launch_k2e(const double *temp, double *k2e, int M, hipStream_t stream)
{
    hipLaunchKernelGGL(k2e_kernel, dim3((M + 255) / 256), dim3(256), 0, stream, temp, k2e, M);
    return hipGetLastError();
}

hipError_t launch_reduce(const PhotonDev &ph, ReducePartial *out, int blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(reduce_kernel, dim3(blocks), dim3(256), 0, stream, ph, out);
    return hipGetLastError();
}

hipError_t launch_lookup(const KernelConfig &kc, const HydroDev &hy, int n, const double *a0, const double *a1,
                         const double *a2, int *out, hipStream_t stream)
{
    const int blocks = (n + 255) / 256;
    if (kc.dimensions == DIM_THREE)
        hipLaunchKernelGGL((lookup_kernel<DIM_THREE>), dim3(blocks), dim3(256), 0, stream, hy, n, a0, a1, a2, out);
    else
        hipLaunchKernelGGL((lookup_kernel<DIM_TWO>), dim3(blocks), dim3(256), 0, stream, hy, n, a0, a1, a2, out);
    return hipGetLastError();
}

}  // namespace MCRAT_TU_NS

}  // namespace mcrat
