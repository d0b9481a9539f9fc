// kernels.hip -- HIP kernels of the photon loop for gfx950 (wave64).
//
//   step_kernel   one lane per photon-slot pair, coalesced 16-B SoA loads:
//                 pending updatePhotonPosition (mclib.c:1054) -> findContainingHydroCell (mclib.c:436)
//                 -> calcMeanFreePath (mclib.c:617) fused in ONE pass over the photons, with the argsort
//                 of mclib.c:702-712 replaced by a top-K selection (only the prefix of the sorted list
//                 is ever consumed, mclib.c:1128-1133).  HBM-bound; this is the kernel priced against
//                 the roofline.
//   event_kernel  one workgroup: merges the per-workgroup candidates, walks them in sorted order as
//                 photonEvent does (mclib.c:1107-1356), scatters at most one photon and does the time
//                 bookkeeping of mcrat.c:777-846.
//   flush_kernel  applies the advance still pending when a run stops.
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <type_traits>
#include "device_types.hpp"
#include "launch.hpp"
#include "physics.hpp"
#include "rng.hpp"

namespace mcrat {

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
struct StepCounters {
    int relocated;
    int not_found;
};

// one photon slot through the iteration's first half.  Returns time_to_scatter.
template <int DIMS, int GEOM, bool FORCE>
__device__ __forceinline__ double step_one(const PhotonDev &ph, const HydroDev &hy, int i, unsigned fl, int cell,
                                           double r0, double r1, double r2, double p0, double p1, double p2, double p3,
                                           double tau, uint64_t bits, StepCounters &cnt)
{
    if (!(fl & FLAG_VALID)) return INFINITY;

    double a0, a1, a2;
    phys::hydro_coords<DIMS, GEOM>(r0, r1, r2, a0, a1, a2);
    bool recalc = (fl & FLAG_RECALC) != 0;
    bool need_tau = false;

    if (phys::in_domain<DIMS>(hy, a0, a1, a2) && (cell != -1)) {          // mclib.c:492-505
        bool relocate = FORCE;
        if constexpr (!FORCE) relocate = !phys::check_in_block<DIMS>(hy, cell, a0, a1, a2);   // mclib.c:507,528
        if (relocate) {
            const int found = phys::find_containing_block<DIMS>(hy, a0, a1, a2);             // mclib.c:534
            cell = found;
            ph.idx[i] = found;                                                               // mclib.c:536
            if (found != -1) {
                // comoving 4-momentum in the new cell, mclib.c:541-563
                const double ph_phi = atan2(r1, r0);
                double beta[3];
                phys::cell_beta<DIMS, GEOM>(hy, found, ph_phi, beta);
                const double lab[4] = {p0, p1, p2, p3};
                double comv[4];
                phys::lorentz_boost(beta, lab, comv, true);
                ph.c0[i] = comv[0]; ph.c1[i] = comv[1]; ph.c2[i] = comv[2]; ph.c3[i] = comv[3];
                need_tau = true;                                                             // mclib.c:570
                if constexpr (!FORCE) cnt.relocated += 1;                                    // mclib.c:579,608-611
            } else {
                cnt.not_found += 1;                                                          // mclib.c:583
            }
        }
    } else if (cell != -1) {
        cell = -1;                                                                           // mclib.c:592
        ph.idx[i] = -1;
    }

    double tts;
    if (cell != -1) {                                                                        // mclib.c:657
        if (need_tau || recalc) {                                                            // mclib.c:668-673 / :570-576
            const double ph_phi = atan2(r1, r0);
            double beta[3];
            phys::cell_beta<DIMS, GEOM>(hy, cell, ph_phi, beta);
            const CellFluid f = hy.fluid[cell];
            tau = phys::optical_depth_direct(beta, f.gamma, f.dens_lab, p1, p2, p3);
            ph.tau[i] = tau;
            if (recalc) ph.flags[i] = (unsigned char)(fl & ~FLAG_RECALC);
        }
        const double rnd = bits_to_uniform_pos(bits);                                        // mclib.c:675
        const double mfp = (-1.0 / tau) * log(rnd);                                          // mclib.c:680
        tts = mfp / C_LIGHT;                                                                 // mclib.c:687
    } else {
        tts = 1e12 / C_LIGHT;                                                                // mclib.c:620,684
    }
    return tts;
}

template <int DIMS, int GEOM, bool FORCE>
__global__ __launch_bounds__(STEP_BLOCK) void step_kernel(PhotonDev ph, HydroDev hy, LoopState *st, RngKey key,
                                                          Cand *__restrict__ partials)
{
    __shared__ Cand s_w[STEP_BLOCK / 64][TOPK];
    if (st->done) return;
    const int nseg = st->nseg;
    const int skip = st->skip_idx;
    const unsigned long long iter = st->iteration;

    TopK best;
    best.init();
    StepCounters cnt = {0, 0};
    const int npairs = ph.n_pad >> 1;

    for (int pair = blockIdx.x * STEP_BLOCK + threadIdx.x; pair < npairs; pair += gridDim.x * STEP_BLOCK) {
        const int i0 = pair << 1;
        double2 R0 = *reinterpret_cast<const double2 *>(ph.r0 + i0);
        double2 R1 = *reinterpret_cast<const double2 *>(ph.r1 + i0);
        double2 R2 = *reinterpret_cast<const double2 *>(ph.r2 + i0);
        const double2 P0 = *reinterpret_cast<const double2 *>(ph.p0 + i0);
        const double2 P1 = *reinterpret_cast<const double2 *>(ph.p1 + i0);
        const double2 P2 = *reinterpret_cast<const double2 *>(ph.p2 + i0);
        const double2 P3 = *reinterpret_cast<const double2 *>(ph.p3 + i0);
        const double2 TAU = *reinterpret_cast<const double2 *>(ph.tau + i0);
        const int2 ID = *reinterpret_cast<const int2 *>(ph.idx + i0);
        const uchar2 FL = *reinterpret_cast<const uchar2 *>(ph.flags + i0);

        // one Philox block serves both slots of the pair
        const Philox4 blk = keyed_block(key.seed, iter, (uint32_t)pair, RNG_FREEPATH, key.stream);
        const uint64_t bits0 = (uint64_t)blk.w[0] | ((uint64_t)blk.w[1] << 32);
        const uint64_t bits1 = (uint64_t)blk.w[2] | ((uint64_t)blk.w[3] << 32);

        // pending updatePhotonPosition of the previous iteration, mclib.c:1067-1095: one segment per
        // candidate that photonEvent walked (mclib.c:1138,1332)
        if (nseg > 0) {
            const double d0 = 1.0 / P0.x, d1 = 1.0 / P0.y;
            const bool m0 = (FL.x & FLAG_MOVES) && (i0 != skip);
            const bool m1 = (FL.y & FLAG_MOVES) && (i0 + 1 != skip);
#pragma unroll
            for (int s = 0; s < MAX_SEG; ++s) {
                if (s < nseg) {
                    const double t = st->seg[s];
                    if (m0) {
                        R0.x += P1.x * d0 * C_LIGHT * t;
                        R1.x += P2.x * d0 * C_LIGHT * t;
                        R2.x += P3.x * d0 * C_LIGHT * t;
                    }
                    if (m1) {
                        R0.y += P1.y * d1 * C_LIGHT * t;
                        R1.y += P2.y * d1 * C_LIGHT * t;
                        R2.y += P3.y * d1 * C_LIGHT * t;
                    }
                }
            }
            *reinterpret_cast<double2 *>(ph.r0 + i0) = R0;
            *reinterpret_cast<double2 *>(ph.r1 + i0) = R1;
            *reinterpret_cast<double2 *>(ph.r2 + i0) = R2;
        }

        double2 T;
        T.x = step_one<DIMS, GEOM, FORCE>(ph, hy, i0, FL.x, ID.x, R0.x, R1.x, R2.x, P0.x, P1.x, P2.x, P3.x, TAU.x, bits0, cnt);
        T.y = step_one<DIMS, GEOM, FORCE>(ph, hy, i0 + 1, FL.y, ID.y, R0.y, R1.y, R2.y, P0.y, P1.y, P2.y, P3.y, TAU.y, bits1, cnt);
        *reinterpret_cast<double2 *>(ph.tts + i0) = T;
        if (FL.x & FLAG_VALID) best.insert(T.x, i0);
        if (FL.y & FLAG_VALID) best.insert(T.y, i0 + 1);
    }

    block_topk<STEP_BLOCK / 64>(best, s_w, partials + (size_t)blockIdx.x * TOPK);

    if (cnt.relocated) atomicAdd(reinterpret_cast<unsigned long long *>(&st->n_relocated), (unsigned long long)cnt.relocated);
    if (cnt.not_found) atomicAdd(reinterpret_cast<unsigned long long *>(&st->not_found), (unsigned long long)cnt.not_found);
}

// ------------------------------------------------------------------ event kernel
enum { EV_RUNNING = 0, EV_DONE = 1, EV_NEED_MORE = 2 };

template <int DIMS, int GEOM, bool STOKES>
__global__ __launch_bounds__(EVENT_BLOCK) void event_kernel(PhotonDev ph, HydroDev hy, LoopState *st, RngKey key,
                                                            const Cand *__restrict__ partials, int n_partials)
{
    __shared__ Cand s_w[EVENT_BLOCK / 64][TOPK];
    __shared__ Cand s_c[TOPK];
    __shared__ int s_status;
    __shared__ double s_last_t;
    __shared__ int s_last_i;
    if (st->done) return;

    const int tid = threadIdx.x;
    {
        TopK best;
        best.init();
        for (int e = tid; e < n_partials; e += EVENT_BLOCK) {
            const Cand c = partials[e];
            if (c.idx != INT_MAX) best.insert(c.t, c.idx);
        }
        block_topk<EVENT_BLOCK / 64>(best, s_w, s_c);
    }

    // thread 0's walk through the sorted candidates, photonEvent mclib.c:1128-1339
    const double dt_max = st->remaining_time;
    const unsigned long long iter = st->iteration;
    double old_scatt_time = 0, dt = 0;
    double seg[MAX_SEG];
    int nseg = 0, skip = -1;
    long long rej = 0, rescans = 0;
    int last_idx = st->last_scattered_index;
    if (tid == 0) s_status = EV_RUNNING;
    __syncthreads();

    const int max_rounds = ph.n / TOPK + 2;
    for (int round = 0; round < max_rounds; ++round) {
        if (tid == 0) {
            int status = EV_NEED_MORE;
            for (int c = 0; c < TOPK; ++c) {
                const double scatt_time = s_c[c].t;
                const int i = s_c[c].idx;
                if (i == INT_MAX) {                       // every slot was tried: mclib.c:1128 loop ends
                    dt = old_scatt_time;
                    status = EV_DONE;
                    break;
                }
                // *scattered_ph_index (mclib.c:1341) is the last candidate photonEvent looked at; main() does not
                // call photonEvent at all when even the first free time exceeds the frame (mcrat.c:777,834)
                if (!(round == 0 && c == 0 && !(scatt_time < dt_max))) last_idx = i;
                if (scatt_time < dt_max) {                // mclib.c:1136
                    const double this_seg = scatt_time - old_scatt_time;
                    if (nseg < MAX_SEG) seg[nseg++] = this_seg;
                    else seg[MAX_SEG - 1] += this_seg;
                    old_scatt_time = scatt_time;
                    const int cell = ph.idx[i];
                    if (cell != -1) {
                        double p[4] = {ph.p0[i], ph.p1[i], ph.p2[i], ph.p3[i]};
                        double r[3] = {ph.r0[i], ph.r1[i], ph.r2[i]};
                        if (ph.flags[i] & FLAG_MOVES) {   // the candidate's own position after mclib.c:1138
                            const double d = 1.0 / p[0];
                            for (int s = 0; s < nseg; ++s) {
                                r[0] += p[1] * d * C_LIGHT * seg[s];
                                r[1] += p[2] * d * C_LIGHT * seg[s];
                                r[2] += p[3] * d * C_LIGHT * seg[s];
                            }
                        }
                        const double fluid_temp = hy.temp[cell];                       // mclib.c:1148
                        const double ph_phi = atan2(r[1], r[0]);                       // mclib.c:1151
                        double beta[3];
                        phys::cell_beta<DIMS, GEOM>(hy, cell, ph_phi, beta);           // mclib.c:1167-1174
                        double pc[4] = {ph.c0[i], ph.c1[i], ph.c2[i], ph.c3[i]};
                        double s[4] = {1, 0, 0, 0};
                        if constexpr (STOKES) {
                            s[0] = ph.s0[i]; s[1] = ph.s1[i]; s[2] = ph.s2[i]; s[3] = ph.s3[i];
                            phys::stokes_rotation(beta, p + 1, pc + 1, s);             // mclib.c:1227
                        }
                        EventStream rng = event_stream(key.seed, iter, (uint32_t)i, key.stream);
                        const double k2e = hy.k2e ? hy.k2e[cell] : 0.0;
                        double el[4];
                        phys::single_thermal_electron(el, fluid_temp, k2e, pc, rng);   // mclib.c:1234
                        if (phys::single_scatter<STOKES>(el, pc, s, rng)) {            // mclib.c:1245
                            const double nb[3] = {-1 * beta[0], -1 * beta[1], -1 * beta[2]};
                            phys::lorentz_boost(nb, pc, p, true);                      // mclib.c:1265
                            if constexpr (STOKES) {
                                phys::stokes_rotation(nb, pc + 1, p + 1, s);           // mclib.c:1280
                                ph.s0[i] = s[0]; ph.s1[i] = s[1]; ph.s2[i] = s[2]; ph.s3[i] = s[3];
                            }
                            ph.p0[i] = p[0]; ph.p1[i] = p[1]; ph.p2[i] = p[2]; ph.p3[i] = p[3];
                            ph.c0[i] = pc[0]; ph.c1[i] = pc[1]; ph.c2[i] = pc[2]; ph.c3[i] = pc[3];
                            ph.r0[i] = r[0]; ph.r1[i] = r[1]; ph.r2[i] = r[2];      // already advanced: the next step kernel skips it
                            ph.num_scatt[i] += 1;                                      // mclib.c:1317
                            ph.flags[i] |= (unsigned char)FLAG_RECALC;                 // mclib.c:1322
                            st->frame_scatt_cnt += 1;                                  // mclib.c:1318
                            st->last_scattered_temp = fluid_temp;
                            skip = i;
                            dt = scatt_time;
                            status = EV_DONE;
                            break;
                        }
                        rej += 1;
                    }
                } else {                                   // mclib.c:1327-1335
                    const double this_seg = dt_max - old_scatt_time;
                    if (nseg < MAX_SEG) seg[nseg++] = this_seg;
                    else seg[MAX_SEG - 1] += this_seg;
                    dt = dt_max;
                    status = EV_DONE;
                    break;
                }
            }
            if (status == EV_NEED_MORE) {
                s_last_t = s_c[TOPK - 1].t;
                s_last_i = s_c[TOPK - 1].idx;
                rescans += 1;
            }
            s_status = status;
        }
        __syncthreads();
        if (s_status != EV_NEED_MORE) break;

        // all TOPK candidates were Klein-Nishina rejected: fetch the next TOPK in sorted order
        {
            const double lt = s_last_t;
            const int li = s_last_i;
            TopK more;
            more.init();
            for (int i = tid; i < ph.n; i += EVENT_BLOCK) {
                double t = ph.tts[i];
                if (t != t) t = INFINITY;
                if (cand_less(lt, li, t, i)) more.insert(t, i);
            }
            block_topk<EVENT_BLOCK / 64>(more, s_w, s_c);
        }
    }

    if (tid == 0) {                                        // mcrat.c:782-784 / 837-845
        st->time_now += dt;
        const double rem = dt_max - dt;
        st->remaining_time = rem;
        st->last_time_step = dt;
        st->iteration = iter + 1;
        st->iterations += 1;
        st->done = !(rem > 0);
        st->nseg = nseg;
        for (int s = 0; s < MAX_SEG; ++s) st->seg[s] = (s < nseg) ? seg[s] : 0.0;
        st->skip_idx = skip;
        st->last_scattered_index = last_idx;
        st->kn_rejections += rej;
        st->rescans += rescans;
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
            const double d = 1.0 / ph.p0[i];
            const double p1 = ph.p1[i], p2 = ph.p2[i], p3 = ph.p3[i];
            double r0 = ph.r0[i], r1 = ph.r1[i], r2 = ph.r2[i];
            for (int s = 0; s < nseg; ++s) {
                const double t = st->seg[s];
                r0 += p1 * d * C_LIGHT * t;
                r1 += p2 * d * C_LIGHT * t;
                r2 += p3 * d * C_LIGHT * t;
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
    if (blocks > 2048) blocks = 2048;      // 256 CUs x 8 workgroups; grid-stride the rest
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
    if (d == DIM_TWO && g == GEOM_CARTESIAN) f(ic<DIM_TWO>{}, ic<GEOM_CARTESIAN>{});
    else if (d == DIM_TWO && g == GEOM_CYLINDRICAL) f(ic<DIM_TWO>{}, ic<GEOM_CYLINDRICAL>{});
    else if (d == DIM_TWO && g == GEOM_SPHERICAL) f(ic<DIM_TWO>{}, ic<GEOM_SPHERICAL>{});
    else if (d == DIM_TWO_POINT_FIVE && g == GEOM_CARTESIAN) f(ic<DIM_TWO_POINT_FIVE>{}, ic<GEOM_CARTESIAN>{});
    else if (d == DIM_TWO_POINT_FIVE && g == GEOM_CYLINDRICAL) f(ic<DIM_TWO_POINT_FIVE>{}, ic<GEOM_CYLINDRICAL>{});
    else if (d == DIM_TWO_POINT_FIVE && g == GEOM_SPHERICAL) f(ic<DIM_TWO_POINT_FIVE>{}, ic<GEOM_SPHERICAL>{});
    else if (d == DIM_THREE && g == GEOM_CARTESIAN) f(ic<DIM_THREE>{}, ic<GEOM_CARTESIAN>{});
    else if (d == DIM_THREE && g == GEOM_SPHERICAL) f(ic<DIM_THREE>{}, ic<GEOM_SPHERICAL>{});
    else if (d == DIM_THREE && g == GEOM_POLAR) f(ic<DIM_THREE>{}, ic<GEOM_POLAR>{});
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_step(const KernelConfig &kc, bool force_relocate, const PhotonDev &ph, const HydroDev &hy,
                       LoopState *st, RngKey key, Cand *partials, int blocks, hipStream_t stream)
{
    return dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        if (force_relocate)
            step_kernel<DV, GV, true><<<dim3(blocks), dim3(STEP_BLOCK), 0, stream>>>(ph, hy, st, key, partials);
        else
            step_kernel<DV, GV, false><<<dim3(blocks), dim3(STEP_BLOCK), 0, stream>>>(ph, hy, st, key, partials);
    });
}

hipError_t launch_event(const KernelConfig &kc, const PhotonDev &ph, const HydroDev &hy, LoopState *st, RngKey key,
                        const Cand *partials, int n_partials, hipStream_t stream)
{
    return dispatch(kc, [&](auto D, auto G) {
        constexpr int DV = decltype(D)::value, GV = decltype(G)::value;
        if (kc.stokes)
            event_kernel<DV, GV, true><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, key, partials, n_partials);
        else
            event_kernel<DV, GV, false><<<dim3(1), dim3(EVENT_BLOCK), 0, stream>>>(ph, hy, st, key, partials, n_partials);
    });
}

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

}  // namespace mcrat
