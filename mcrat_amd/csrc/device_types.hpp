// device_types.hpp -- HBM layouts shared by the kernels and the host-side engine.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mcrat {

// Src/mclib.c:4-5, verbatim values
constexpr double A_RAD = 7.56e-15;
constexpr double C_LIGHT = 2.99792458e10;
constexpr double PL_CONST = 6.6260755e-27;
constexpr double K_B = 1.380658e-16;
constexpr double M_P = 1.6726231e-24;
constexpr double THOM_X_SECT = 6.65246e-25;
constexpr double M_EL = 9.1093879e-28;

constexpr int GEOM_CARTESIAN = 0, GEOM_SPHERICAL = 1, GEOM_CYLINDRICAL = 2, GEOM_POLAR = 3;
constexpr int DIM_TWO = 0, DIM_TWO_POINT_FIVE = 1, DIM_THREE = 2;

// per-photon flag byte (replaces the reads of type, weight and recalc_properties on the streaming path)
constexpr unsigned FLAG_RECALC = 1u;   // recalc_properties == 1                       (mcrat.h:161)
constexpr unsigned FLAG_MOVES = 2u;    // type != CS_POOL_PHOTON && weight != 0        (mclib.c:1070)
constexpr unsigned FLAG_VALID = 4u;    // slot index < list_capacity (padding slots are not valid)
constexpr unsigned FLAG_TAU_FRESH = 8u;  // set with RECALC by the event kernel: tau of the new momentum is already stored for the
                                       // cached cell, so a slot that is still in that cell needs no slow path next pass

constexpr int TOPK = 4;          // candidates fetched per rescan of time_to_scatter in the event kernel
constexpr int MAX_SEG = 8;       // advance segments remembered per iteration (one per tried candidate)
constexpr int STEP_BLOCK = 256;  // threads per workgroup of the step kernel; each thread owns slot pairs
constexpr int STEP_QCAP = 1024;  // LDS queue of slots awaiting the slow path, per workgroup
#ifndef MCRAT_EVENT_BLOCK
#define MCRAT_EVENT_BLOCK 256
#endif
constexpr int SHORTLIST_CAP = MCRAT_EVENT_BLOCK;   // early candidates (free time below LoopState::t_cut) collected per iteration
constexpr int EVENT_BLOCK = MCRAT_EVENT_BLOCK;    // one workgroup; only lane 0 runs the scattering physics, so leave it the whole register file

// SoA photon columns in HBM.  Capacity is padded to a multiple of 2*STEP_BLOCK; every column is 256-B aligned.
struct PhotonDev {
    double *r0, *r1, *r2;
    double *p0, *p1, *p2, *p3;
    double *c0, *c1, *c2, *c3;   // comv_p
    double *s0, *s1, *s2, *s3;
    double *num_scatt;
    double *weight;
    double *tau;                 // total_optical_depth [1/cm]
    double *tts;                 // time_to_scatter [s]
    // derived columns, maintained wherever p or tau change, so that the streaming pass needs no division:
    double *u0, *u1, *u2;        // (p_k * (1/p0)) * C_LIGHT: the velocity factor of mclib.c:1076-1080, rounded as there
    double *ntau;                // -1.0 / total_optical_depth, the factor of mclib.c:680
    double *tau_next;            // optical depth precomputed at scatter time (FLAG_TAU_FRESH); becomes `tau` when the next
                                 // pass confirms the slot is still in its cell, so a read-back in between shows what the
                                 // reference would show (recalc_properties = 1, old total_optical_depth)
    int *idx;                    // nearest_block_index
    unsigned char *flags;
    char *type;
    int n;                       // list_capacity
    int n_pad;
    // the 24 double columns above are ONE allocation, column k (in member order, photon_cols.hpp) at r0 + k * col_stride doubles:
    // rank_loop_kernel addresses them through the base pointer and this stride instead of 24 pointers
    unsigned col_stride;
};
// the columns from slot `first` on: a list of a rank pool as a list of its own (the biases stay 0)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void offset_photons(PhotonDev &ph, size_t first)
{
    double **cols[24] = {&ph.r0, &ph.r1, &ph.r2, &ph.p0, &ph.p1, &ph.p2, &ph.p3, &ph.c0, &ph.c1, &ph.c2, &ph.c3, &ph.s0, &ph.s1, &ph.s2, &ph.s3,
                         &ph.num_scatt, &ph.weight, &ph.tau, &ph.tts, &ph.u0, &ph.u1, &ph.u2, &ph.ntau, &ph.tau_next};
    for (int k = 0; k < 24; ++k) *cols[k] += first;
    ph.idx += first; ph.flags += first; ph.type += first;
}

struct alignas(32) CellGeom {    // one 32-B sector per in-cell test (geometry.c:394-417)
    double c0, c1, s0, s1;       // centre and full size on axes 0,1
};
struct alignas(16) CellGeom2 {   // third axis, 3-D only
    double c2, s2;
};
// What the loop needs from a cell whenever it re-locates a photon into it or scatters one there, staged ONCE per frame instead of being
// recomputed per photon (round 3; the values depend on the cell alone):
//   a, b, c   the fluid velocity with the per-cell part of hydroVectorToCartesian applied (geometry.c:189-253; physics.hpp, beta_from_record)
//   w         beta_g / |v|,  beta_g = sqrt(1 - 1/gamma^2) from the cell's `gamma` (optical_depth.c:52), |v| from the velocity components
//             (optical_depth.c:44): calculateOpticalDepth's  beta_g * (v.p) / (|v| |p|)  is then  w * (v.p) / |p|.  A cell at rest gives 0/0 = NaN,
//             as the reference's cosine does (optical_depth.c:46)
//   nsig      (dens_lab / M_P) * THOM_X_SECT                                               (optical_depth.c:57,59)
//   gam, kf   the Lorentz factor of |v| and (gam - 1)/|v|^2 = gam^2/(gam + 1): the two scalars of the boost matrix of lorentzBoost
//             (mclib.c:318-340), whose |v| does not depend on the photon's azimuth
struct alignas(32) CellFluid {
    double a, b, c, w;
    double nsig, gam, kf, pad;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void cell_staged_operands(double a, double b, double c, double gamma_cell, double dens_lab, CellFluid &f)
{
    const double beta_g = __builtin_sqrt(1.0 - 1.0 / (gamma_cell * gamma_cell));
    const double v2 = (a * a + b * b) + c * c;
    f.a = a; f.b = b; f.c = c;
    f.w = beta_g / __builtin_sqrt(v2);
    f.nsig = (dens_lab / M_P) * THOM_X_SECT;
    f.gam = 1.0 / __builtin_sqrt(1.0 - v2);
    f.kf = (f.gam * f.gam) / (f.gam + 1.0);
    f.pad = 0.0;
}

// exact accelerator for findContainingBlock (geometry.c:350-391): uniform buckets (optionally in
// log of the coordinate) whose member lists are ascending in cell index, so the first hit of the
// closed-interval test is the lowest-index containing cell, i.e. what the reference's linear scan returns.
// bucket-list entry: the cell index with copies of everything the slow path needs from that cell, so that a
// re-location costs two dependent load rounds (list range, then one entry: exactly one 128-B line) instead of five
// (alignas(32), not 128: a local copy must not ask for an over-aligned stack slot; the array itself starts on a 256-B boundary)
// Member order: a lane reads its entry in 16-B pieces, and what a re-location costs the CU's vector cache is the NUMBER of pieces (about one
// lane-piece per clock when every lane is on a line of its own, tools/gather_bench.hip) -- so what a 2-D re-location needs comes first, in
// five pieces {c0,c1} {s0,s1} {a,b} {w,nsig} {gam,cell}; kf = gam^2/(gam + 1) is recomputed (one reciprocal) instead of read; 2.5-D adds
// {c,.}, 3-D {c,c2} {s2,.}.
struct alignas(32) FatCell {
    double c0, c1, s0, s1;       // CellGeom
    double a, b, w, nsig;        // CellFluid
    double gam;
    int cell;
    int pad;
    double c;                    // CellFluid (2.5-D, 3-D)
    double c2, s2;               // CellGeom2 (3-D)
    double pad2[3];
};
static_assert(sizeof(FatCell) == 128, "one cache line per bucket-list entry");

// one bucket of the lookup grid: its list, and for each octant (half a bucket per axis; quadrant in 2-D) the one
// list entry whose cell covers that octant, if exactly one cell reaches into it.  A point well inside an octant
// and well inside the hinted cell is in no other cell, so a lookup then costs this record and ONE FatCell
// (128 B) instead of the whole list; every other point takes the exact list walk.
constexpr unsigned GRID_NO_HINT = 15u;
constexpr int GRID_CODE_OCT_SHIFT = 27;      // bucket code = bucket | octant << 27 | (position usable for a hint) << 30
constexpr int GRID_CODE_HINT_OK = 1 << 30;
constexpr int GRID_CODE_BUCKET_MASK = (1 << GRID_CODE_OCT_SHIFT) - 1;
struct alignas(16) BucketDir {
    int e0;                      // first entry in GridDev::cells
    int n;                       // entries
    unsigned hints;              // 4 bits per octant: entry offset 0..14, GRID_NO_HINT = none
    unsigned pad;
};

struct GridDev {
    const BucketDir *dir;        // [nb]
    const FatCell *cells;        // bucket lists, ascending in cell index inside each bucket
    double org[3];
    double inv[3];
    int dim[3];
    int logmap[3];
    int naxes;
};

struct HydroDev {
    const CellGeom *geom;
    const CellGeom2 *geom2;
    const CellFluid *fluid;
    const double *temp;
    const double *fluid_c;       // third velocity component (2.5-D: v2; 3-D: Cartesian z), absent in 2-D (= CellFluid::c; injection and emission read it here)
    const double *gamma;         // the cells' Lorentz factor (photonInjection's count, mclib.c:95; the loop reads CellFluid::w)
    const double *k2e;           // exp(x) K_2(x), x = m_e c^2 / k T, for cells with T >= 1e7 K (else 0)
    int M;
    double dom0[2], dom1[2], dom2[2];
    GridDev grid;
    // TAU_CALCULATION == TABLE (optical_depth.c:132-149): thermal_table[i][j] of hot_x_section.c, log10(sigma/sigma_T) on
    // the (n_ph_e+1) x (n_t+1) grid of hot_x_section.h:2-10; null in DIRECT builds
    const double *hot_table;
    int hot_n_ph_e, hot_n_t;
    double hot_e0, hot_de, hot_t0, hot_dt;
    int *table_fallbacks;           // lookups outside the tabulated range that were integrated afresh (physics.hpp: table_fallback_*)
    int hot_fallback_calls;      // samples of that integral (500 000, hot_x_section.c:348)
};

struct alignas(16) Cand {
    double t;
    int idx;
    int pad;
};

// loop state kept in HBM between kernels (the scalars of mcrat.c:754-851)
struct alignas(256) LoopState {
    double remaining_time;
    double time_now;
    unsigned long long iteration;        // k, also the RNG counter
    int done;                            // remaining_time <= 0
    int nseg;                            // pending advance segments (applied by the next step kernel)
    int skip_idx;                        // photon already advanced by the event kernel (-1: none)
    int last_scattered_index;
    double seg[MAX_SEG];
    double last_time_step;
    double last_scattered_temp;
    long long iterations;
    long long frame_scatt_cnt;
    long long n_relocated;               // num_photons_find_new_element (not counted on forced passes)
    long long not_found;
    long long kn_rejections;
    long long rescans;
    double t_cut;                        // free times below this go on the shortlist (speed only, never results)
    double t_est;                        // running estimate of the smallest free time per iteration
    int force_relocate;                  // virtual-rank mode: the next pass is the forced re-location pass of a new frame
    int photon_event_called;             // this pass took the photonEvent branch of mcrat.c:777 (the cyclo-synchrotron hook of :786 looks at it)
    union {
        long long stamps[8];             // diagnostic build only (-DMCRAT_DIAG): s_memtime at points of the event walk
        long long slot_steps;            // product build: slots actually taken through a pass, summed over the passes (a cyclo-synchrotron list's settled
                                         // null slots behind its last photon are not: rank_loop_kernel, n_pass); iterations x list_capacity otherwise
    };
};

// Candidates of one iteration.  Every slot whose free time is below LoopState::t_cut is appended here
// (a handful per iteration), so the list holds the COMPLETE sorted prefix of the reference's argsort
// (mclib.c:702-712) up to t_cut; the per-workgroup minima cover the case of an empty list.  If the walk of
// photonEvent needs candidates beyond the list, the event kernel rescans time_to_scatter.
struct alignas(16) Shortlist {
    int count;                   // may exceed SHORTLIST_CAP: then the list is incomplete and ignored
    int pad[3];
    Cand items[SHORTLIST_CAP];
};

// one list of a rank pool (kernels.hip, RankLayout)
struct alignas(16) RankDesc {
    int len;                     // list_capacity of this rank's list (0: the rank sits this launch out)
    uint32_t stream;             // its random stream (RngKey::stream of a single-list context holding only its photons)
    uint64_t seed;               // its per-frame seed
};

struct RngKey {
    uint64_t seed;
    uint32_t stream;
    uint32_t slot_base;          // global index of this context's slot 0 (even; 0 unless the list is split over GPUs)
};

// ---- one photon list split over several GPUs, ONE clock (SURVEY.md 8e "exact mode") ------------------------------
// Every GPU owns a contiguous range of the list's slots.  Each round it proposes its earliest candidates WITH the
// data photonEvent needs from them; the proposals are all-gathered and every GPU walks the merged, sorted chain with
// identical arithmetic, so all copies of LoopState stay bit-identical and only the owner of the scattered photon
// stores anything.  The chain is trusted up to the smallest `horizon` of the proposals; if the walk needs more
// (rare: Klein-Nishina rejections) the iteration continues in a "midpass" round that re-reads time_to_scatter
// beyond the cursor instead of redrawing.
constexpr int SC_K = 4;              // candidates one GPU proposes per round
constexpr int SC_MAX_WORLD = 16;
constexpr int LOOP_DONE = 1;         // LoopState::done values
constexpr int LOOP_MIDPASS = 2;      // shared clock only: the iteration is not decided yet, the step kernel must not redraw
constexpr int LOOP_SC_GAVE_UP = 4;   // shared clock, device-initiated exchange: a peer's proposal never arrived; the list stays at its last complete pass
                                     // (3 is LOOP_CS_HALT, launch.hpp)

struct alignas(16) ScRecord {        // one candidate: 176 B
    double t;                        // time_to_scatter
    long long gid;                   // global slot
    double r[3], u[3], p[4], pc[4], s[4];
    int cell;
    unsigned flags;
};

struct alignas(16) ScProposal {      // what one GPU sends per round
    int n;                           // records that follow
    int pad0;
    long long horizon_gid;           // this GPU has told everything it holds with (t, gid) <= (horizon_t, horizon_gid)
    double horizon_t;
    double pad1;
    ScRecord rec[SC_K];
};

struct alignas(64) ScState {         // per GPU; content identical on every GPU
    double cursor_t;                 // midpass: last candidate the walk tried
    long long cursor_gid;
    double old_scatt_time;           // photonEvent's old_scatt_time across rounds (mclib.c:1138)
    double t_first;                  // first candidate time of the iteration (shortlist threshold estimate)
    double cut_mid;                  // midpass: time_to_scatter below this goes on the shortlist
    int first;                       // no candidate tried yet in this iteration
    int pad0;
    long long rounds, midpass_rounds;
};

}  // namespace mcrat
