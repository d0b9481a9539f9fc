// staging.hip -- struct photon records (Src/mcrat.h:142-171, 176 B, AoS) <-> the engine's SoA columns, on the device.
// The caller's photon list crosses PCIe once as it lies in memory; the transposition, the derived columns
// (device_types.hpp: u = (p_k (1/p0)) c as mclib.c:1074-1080 forms it, -1/tau as mclib.c:680) and the flag byte are
// produced here instead of in a host loop over 10^6 - 10^8 records.
#include <hip/hip_runtime.h>
#include "../../include/mcrat_hip.h"
#include "device_types.hpp"
#include "launch.hpp"
#include "sc_exchange.hpp"

namespace mcrat {

namespace {

__global__ __launch_bounds__(256) void aos_to_soa_kernel(const mcrat_hip_photon *__restrict__ aos, PhotonDev ph, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const mcrat_hip_photon q = aos[i];
    ph.r0[i] = q.r0; ph.r1[i] = q.r1; ph.r2[i] = q.r2;
    ph.p0[i] = q.p0; ph.p1[i] = q.p1; ph.p2[i] = q.p2; ph.p3[i] = q.p3;
    ph.c0[i] = q.comv_p0; ph.c1[i] = q.comv_p1; ph.c2[i] = q.comv_p2; ph.c3[i] = q.comv_p3;
    ph.s0[i] = q.s0; ph.s1[i] = q.s1; ph.s2[i] = q.s2; ph.s3[i] = q.s3;
    ph.num_scatt[i] = q.num_scatt;
    ph.weight[i] = q.weight;
    ph.tau[i] = q.total_optical_depth;
    ph.tts[i] = q.time_to_scatter;
    double u0 = 0, u1 = 0, u2 = 0;
    if (q.p0 != 0) {
        const double d = 1.0 / q.p0;
        u0 = q.p1 * d * C_LIGHT; u1 = q.p2 * d * C_LIGHT; u2 = q.p3 * d * C_LIGHT;
    }
    ph.u0[i] = u0; ph.u1[i] = u1; ph.u2[i] = u2;
    ph.ntau[i] = -1.0 / q.total_optical_depth;
    ph.tau_next[i] = 0.0;
    ph.idx[i] = q.nearest_block_index;
    unsigned f = FLAG_VALID;
    if (q.type != 'p' && q.weight != 0) f |= FLAG_MOVES;      // mclib.c:1070
    if (q.recalc_properties == 1) f |= FLAG_RECALC;
    ph.flags[i] = (unsigned char)f;
    ph.type[i] = q.type;
}

// the same for many lists of a rank pool in ONE launch (mcrat_hip_pool_set_photons): list j's records aos[d.aos_first ..) go to the window of rank
// d.rank, the rest of the window is cleared (no list: flag byte 0).  grid = (windows' 256-slot chunks, lists)
struct alignas(16) PoolSetDesc { int rank, n; long long aos_first; };

__device__ __forceinline__ void store_record(const PhotonDev &ph, int i, const mcrat_hip_photon &q)
{
    ph.r0[i] = q.r0; ph.r1[i] = q.r1; ph.r2[i] = q.r2;
    ph.p0[i] = q.p0; ph.p1[i] = q.p1; ph.p2[i] = q.p2; ph.p3[i] = q.p3;
    ph.c0[i] = q.comv_p0; ph.c1[i] = q.comv_p1; ph.c2[i] = q.comv_p2; ph.c3[i] = q.comv_p3;
    ph.s0[i] = q.s0; ph.s1[i] = q.s1; ph.s2[i] = q.s2; ph.s3[i] = q.s3;
    ph.num_scatt[i] = q.num_scatt;
    ph.weight[i] = q.weight;
    ph.tau[i] = q.total_optical_depth;
    ph.tts[i] = q.time_to_scatter;
    double u0 = 0, u1 = 0, u2 = 0;
    if (q.p0 != 0) {
        const double d = 1.0 / q.p0;
        u0 = q.p1 * d * C_LIGHT; u1 = q.p2 * d * C_LIGHT; u2 = q.p3 * d * C_LIGHT;
    }
    ph.u0[i] = u0; ph.u1[i] = u1; ph.u2[i] = u2;
    ph.ntau[i] = -1.0 / q.total_optical_depth;
    ph.tau_next[i] = 0.0;
    ph.idx[i] = q.nearest_block_index;
    unsigned f = FLAG_VALID;
    if (q.type != 'p' && q.weight != 0) f |= FLAG_MOVES;      // mclib.c:1070
    if (q.recalc_properties == 1) f |= FLAG_RECALC;
    ph.flags[i] = (unsigned char)f;
    ph.type[i] = q.type;
}

__global__ __launch_bounds__(256) void pool_aos_to_soa_kernel(const mcrat_hip_photon *__restrict__ aos, PhotonDev ph, int stride,
                                                              const PoolSetDesc *__restrict__ desc)
{
    const PoolSetDesc d = desc[blockIdx.y];
    const int il = blockIdx.x * 256 + threadIdx.x;
    if (il >= stride) return;
    const int i = d.rank * stride + il;
    if (il < d.n) {
        store_record(ph, i, aos[d.aos_first + il]);
    } else {
        double *cols[24] = {ph.r0, ph.r1, ph.r2, ph.p0, ph.p1, ph.p2, ph.p3, ph.c0, ph.c1, ph.c2, ph.c3, ph.s0, ph.s1, ph.s2, ph.s3,
                            ph.num_scatt, ph.weight, ph.tau, ph.tts, ph.u0, ph.u1, ph.u2, ph.ntau, ph.tau_next};
#pragma unroll
        for (int c = 0; c < 24; ++c) cols[c][i] = 0.0;
        ph.idx[i] = 0; ph.flags[i] = 0; ph.type[i] = 0;
    }
}

// record k of `aos` is photon slot first + k
__global__ __launch_bounds__(256) void soa_to_aos_kernel(PhotonDev ph, mcrat_hip_photon *__restrict__ aos, int first, int n)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int i = first + k;
    mcrat_hip_photon q = aos[k];          // keeps the bytes between the members as they were uploaded
    q.type = ph.type[i];
    q.r0 = ph.r0[i]; q.r1 = ph.r1[i]; q.r2 = ph.r2[i];
    q.p0 = ph.p0[i]; q.p1 = ph.p1[i]; q.p2 = ph.p2[i]; q.p3 = ph.p3[i];
    q.comv_p0 = ph.c0[i]; q.comv_p1 = ph.c1[i]; q.comv_p2 = ph.c2[i]; q.comv_p3 = ph.c3[i];
    q.s0 = ph.s0[i]; q.s1 = ph.s1[i]; q.s2 = ph.s2[i]; q.s3 = ph.s3[i];
    q.num_scatt = ph.num_scatt[i];
    q.weight = ph.weight[i];
    q.total_optical_depth = ph.tau[i];
    q.time_to_scatter = ph.tts[i];
    q.nearest_block_index = ph.idx[i];
    q.recalc_properties = (ph.flags[i] & FLAG_RECALC) ? 1 : 0;
    aos[k] = q;
}

// printPhotons' gathering loop (mcrat_io.c:137-181): the photons with weight != 0, in slot order, as the arrays it hands to
// H5Dwrite.  Pass 1 counts per workgroup, a scan places the workgroups, pass 2 writes.
// (a workgroup counts OUT_CHUNKS chunks of 256 slots -- the unit pass 2 runs over -- and adds to the total once:
// same-address atomics serialise)
constexpr int OUT_CHUNKS = 8;
__global__ __launch_bounds__(256) void output_count_kernel(PhotonDev ph, int n, unsigned *__restrict__ block_count, unsigned long long *__restrict__ total)
{
    __shared__ unsigned s_w[OUT_CHUNKS][4];
    const int chunks = (n + 255) / 256;
    for (int c = 0; c < OUT_CHUNKS; ++c) {
        const int i = (blockIdx.x * OUT_CHUNKS + c) * 256 + threadIdx.x;
        const bool keep = i < n && ph.weight[i] != 0;
        const unsigned long long m = __ballot(keep);
        if ((threadIdx.x & 63) == 0) s_w[c][threadIdx.x >> 6] = (unsigned)__popcll(m);
    }
    __syncthreads();
    unsigned cnt = 0;
    if (threadIdx.x < OUT_CHUNKS) {
        const int chunk = blockIdx.x * OUT_CHUNKS + threadIdx.x;
        cnt = s_w[threadIdx.x][0] + s_w[threadIdx.x][1] + s_w[threadIdx.x][2] + s_w[threadIdx.x][3];
        if (chunk < chunks) block_count[chunk] = cnt;
    }
    for (int off = OUT_CHUNKS / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if (threadIdx.x == 0 && cnt) atomicAdd(total, (unsigned long long)cnt);
}

__global__ __launch_bounds__(256) void output_write_kernel(PhotonDev ph, int n, const int *__restrict__ block_start, OutputCols out)
{
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool keep = i < n && ph.weight[i] != 0;
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (!keep) return;
    unsigned pos = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    const size_t j = (size_t)block_start[blockIdx.x] + pos;
    const double *src[17] = {ph.p0, ph.p1, ph.p2, ph.p3, ph.c0, ph.c1, ph.c2, ph.c3, ph.r0, ph.r1, ph.r2, ph.s0, ph.s1, ph.s2, ph.s3, ph.num_scatt, ph.weight};
#pragma unroll
    for (int k = 0; k < 17; ++k) out.col[k][j] = src[k][i];
    out.type[j] = ph.type[i];
}

constexpr double CHARGE_EL = 4.8032068e-10;      // Src/mclib.c:4-5

// phAbsCyclosynch, mc_cyclosynch.c:1571-1623; calcB :54-76, calcCyclotronFreq :30-33, getMagneticFieldMagnitude :78-92: photon i
__device__ __forceinline__ void cs_absorb_slot(const CsParams &p, const PhotonDev &ph, const double *__restrict__ temp, const HydroCols &h, int i,
                                               double &abs_weight, int &abs_count, int &scatt_count)
{
    const double weight = ph.weight[i];
    const int cell = ph.idx[i];
    if (!((weight != 0) && (cell != -1))) return;
    double b_field;
    if (p.b_field_calc == 0 || p.b_field_calc == 1) {
        const double el_dens = h.dens[cell] / M_P, T = temp[cell];
        if (p.b_field_calc == 0) b_field = sqrt(p.epsilon_b * 8 * M_PI * 3 * el_dens * K_B * T / 2);
        else b_field = sqrt(8 * M_PI * p.epsilon_b * (el_dens * M_P * C_LIGHT * C_LIGHT + 4 * A_RAD * T * T * T * T / 3));
    } else if (p.dimensions == DIM_TWO) {
        b_field = sqrt(h.B0[cell] * h.B0[cell] + h.B1[cell] * h.B1[cell]);
    } else {
        b_field = sqrt(h.B0[cell] * h.B0[cell] + h.B1[cell] * h.B1[cell] + h.B2[cell] * h.B2[cell]);
    }
    const double nu_c = CHARGE_EL * b_field / (2 * M_PI * M_EL * C_LIGHT);
    const char type = ph.type[i];
    if ((ph.c0[i] * C_LIGHT / PL_CONST <= nu_c) || (type == 'p')) {
        abs_count += 1;
        if (type == 'i' || type == 'c') abs_weight += weight;
        // setNullPhoton, photons.c:210-250 (time_to_scatter is left as it is)
        ph.type[i] = 'N';
        ph.weight[i] = 0;
        ph.idx[i] = -1;
        ph.flags[i] = (unsigned char)FLAG_VALID;
        ph.p0[i] = 0; ph.p1[i] = 0; ph.p2[i] = 0; ph.p3[i] = 0;
        ph.c0[i] = 0; ph.c1[i] = 0; ph.c2[i] = 0; ph.c3[i] = 0;
        ph.r0[i] = 0; ph.r1[i] = 0; ph.r2[i] = 0;
        ph.s0[i] = 0; ph.s1[i] = 0; ph.s2[i] = 0; ph.s3[i] = 0;
        ph.num_scatt[i] = 0;
        ph.tau[i] = 0;
        ph.u0[i] = 0; ph.u1[i] = 0; ph.u2[i] = 0;
        ph.ntau[i] = -1.0 / 0.0;
        ph.tau_next[i] = 0;
    } else if (type == 'k' || type == 'c') {
        scatt_count += 1;
    }
}

// the workgroup's sums of what its threads counted
__device__ __forceinline__ CsAbsPartial cs_absorb_reduce(double abs_weight, int abs_count, int scatt_count, double *s_w, int *s_a, int *s_s)
{
    for (int off = 32; off > 0; off >>= 1) {
        abs_weight += __shfl_down(abs_weight, off);
        abs_count += __shfl_down(abs_count, off);
        scatt_count += __shfl_down(scatt_count, off);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { s_w[threadIdx.x >> 6] = abs_weight; s_a[threadIdx.x >> 6] = abs_count; s_s[threadIdx.x >> 6] = scatt_count; }
    __syncthreads();
    CsAbsPartial o;
    o.abs_weight = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    o.abs_count = s_a[0] + s_a[1] + s_a[2] + s_a[3];
    o.scatt_count = s_s[0] + s_s[1] + s_s[2] + s_s[3];
    return o;
}

__global__ __launch_bounds__(256) void cs_absorb_kernel(CsParams p, PhotonDev ph, const double *__restrict__ temp, HydroCols h,
                                                        CsAbsPartial *__restrict__ partials)
{
    __shared__ double s_w[4];
    __shared__ int s_a[4], s_s[4];
    double abs_weight = 0;
    int abs_count = 0, scatt_count = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ph.n; i += gridDim.x * 256) cs_absorb_slot(p, ph, temp, h, i, abs_weight, abs_count, scatt_count);
    const CsAbsPartial o = cs_absorb_reduce(abs_weight, abs_count, scatt_count, s_w, s_a, s_s);
    if (threadIdx.x == 0) partials[blockIdx.x] = o;
}

// the same for the open lists of a rank pool, one workgroup per list.  The workgroup plays the cs_absorb_blocks(len) workgroups the list
// would have had alone one after the other and adds their sums in that order, so the absorbed weight is the one-list path's to the bit.
__global__ __launch_bounds__(256) void cs_absorb_pool_kernel(CsParams p, PhotonDev pool, int stride, const RankDesc *__restrict__ desc,
                                                             const int *__restrict__ open, const double *__restrict__ temp, HydroCols h,
                                                             CsAbsPartial *__restrict__ per_list)
{
    __shared__ double s_w[4];
    __shared__ int s_a[4], s_s[4];
    const int r = blockIdx.x;
    if (!open[r]) return;
    PhotonDev ph = pool;
    offset_photons(ph, (size_t)r * (size_t)stride);
    ph.n = desc[r].len;
    const int b = (ph.n + 255) / 256, nblk = b < 1 ? 1 : (b > 1024 ? 1024 : b);
    CsAbsPartial sum;
    sum.abs_weight = 0; sum.abs_count = 0; sum.scatt_count = 0;
    for (int vb = 0; vb < nblk; ++vb) {
        double abs_weight = 0;
        int abs_count = 0, scatt_count = 0;
        for (int i = vb * 256 + threadIdx.x; i < ph.n; i += nblk * 256) cs_absorb_slot(p, ph, temp, h, i, abs_weight, abs_count, scatt_count);
        const CsAbsPartial o = cs_absorb_reduce(abs_weight, abs_count, scatt_count, s_w, s_a, s_s);
        sum.abs_weight += o.abs_weight; sum.abs_count += o.abs_count; sum.scatt_count += o.scatt_count;
    }
    if (threadIdx.x == 0) per_list[r] = sum;
}

// the loop state of a new frame (mcrat.c:754-758): one record for the single list, one per virtual rank with the forced
// re-location pending.  The record travels as a kernel argument, so opening a frame needs no host buffer and no wait.
__global__ __launch_bounds__(256) void init_states_kernel(LoopState *__restrict__ single, LoopState *__restrict__ ranks, int n_ranks, LoopState v)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) *single = v;
    if (r < n_ranks) {
        LoopState w = v;
        w.force_relocate = 1;
        ranks[r] = w;
    }
}

// saveCheckpoint's in-place conversion with CYCLOSYNCHROTRON_SWITCH on (mcrat_io.c:896-900, :951-955, :991-995): a comptonised photon
// 'k' with weight != 0 becomes an unabsorbed one 'c'
__global__ __launch_bounds__(256) void convert_comptonized_kernel(PhotonDev ph, unsigned *__restrict__ converted)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool hit = i < ph.n && ph.type[i] == 'k' && ph.weight[i] != 0;
    if (hit) ph.type[i] = 'c';
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(converted, (unsigned)__popcll(m));
}

// slots [first, first + count) of every column zeroed: not part of any list (flag byte 0: not FLAG_VALID)
__global__ __launch_bounds__(256) void clear_slots_kernel(PhotonDev ph, int first, int count)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int i = first + k;
    double *cols[24] = {ph.r0, ph.r1, ph.r2, ph.p0, ph.p1, ph.p2, ph.p3, ph.c0, ph.c1, ph.c2, ph.c3, ph.s0, ph.s1, ph.s2, ph.s3,
                        ph.num_scatt, ph.weight, ph.tau, ph.tts, ph.u0, ph.u1, ph.u2, ph.ntau, ph.tau_next};
#pragma unroll
    for (int c = 0; c < 24; ++c) cols[c][i] = 0.0;
    ph.idx[i] = 0;
    ph.flags[i] = 0;
    ph.type[i] = 0;
}

// rank pool: phMinMax (mclib.c:1465), phScattStats (mclib.c:1385), averagePhotonEnergy (mclib.c:1358) and printPhotons' count of
// photons with weight != 0 (mcrat_io.c:137-181) for every list in one launch, one workgroup per list
__global__ __launch_bounds__(256) void rank_reduce_kernel(PhotonDev ph, int stride, const RankDesc *__restrict__ desc, ReducePartial *__restrict__ out,
                                                          int *__restrict__ n_out)
{
    __shared__ double s[4][10];
    __shared__ long long s_cnt[4][2];
    const int rank = blockIdx.x, base = rank * stride, n = desc[rank].len;
    double r_min = 1.7976931348623157e308, r_max = 0, th_min = 1.7976931348623157e308, th_max = 0;
    double sum_scatt = 0, sum_r = 0, e_sum = 0, w_sum = 0, max_s = 0, min_s = 2147483647.0;
    long long count = 0, kept = 0;
    for (int il = threadIdx.x; il < n; il += 256) {
        const int i = base + il;
        const double x = ph.r0[i], y = ph.r1[i], z = ph.r2[i], w = ph.weight[i];
        const double r = sqrt(x * x + y * y + z * z);
        if (w != 0) {
            const double th = acos(z / r);
            r_max = fmax(r_max, r); r_min = fmin(r_min, r);
            th_max = fmax(th_max, th); th_min = fmin(th_min, th);
            kept += 1;
        }
        const double ns = ph.num_scatt[i];
        sum_scatt += ns; sum_r += r;
        max_s = fmax(max_s, ns); min_s = fmin(min_s, ns);
        e_sum += ph.p0[i] * w; w_sum += w;
        count += 1;
    }
    auto wsum = [](double v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64); return v; };
    auto wmin = [](double v) { for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64)); return v; };
    auto wmax = [](double v) { for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64)); return v; };
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    r_min = wmin(r_min); r_max = wmax(r_max); th_min = wmin(th_min); th_max = wmax(th_max);
    sum_scatt = wsum(sum_scatt); sum_r = wsum(sum_r); e_sum = wsum(e_sum); w_sum = wsum(w_sum);
    max_s = wmax(max_s); min_s = wmin(min_s);
    const double cd = wsum((double)count), kd = wsum((double)kept);
    if (lane == 0) {
        s[wv][0] = r_min; s[wv][1] = r_max; s[wv][2] = th_min; s[wv][3] = th_max; s[wv][4] = sum_scatt;
        s[wv][5] = sum_r; s[wv][6] = e_sum; s[wv][7] = w_sum; s[wv][8] = max_s; s[wv][9] = min_s;
        s_cnt[wv][0] = (long long)cd; s_cnt[wv][1] = (long long)kd;
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
        p.count = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
        out[rank] = p;
        n_out[rank] = (int)(s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1]);
    }
}

// rank pool: a new frame for the lists with open[r] != 0 -- their own clocks -- in one launch (cf. init_states_kernel)
__global__ __launch_bounds__(256) void init_states_multi_kernel(LoopState *__restrict__ ranks, int n_ranks, const int *__restrict__ open,
                                                                const double *__restrict__ time_now, const double *__restrict__ remaining)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n_ranks || !open[r]) return;
    LoopState w = {};
    w.remaining_time = remaining[r];
    w.time_now = time_now[r];
    w.done = !(remaining[r] > 0);
    w.skip_idx = -1;
    w.last_scattered_index = -1;
    w.force_relocate = 1;                          // mcrat.c:756
    ranks[r] = w;
}

}  // namespace

hipError_t launch_init_states_multi(LoopState *ranks, int n_ranks, const int *open, const double *time_now, const double *remaining, hipStream_t stream)
{
    init_states_multi_kernel<<<dim3((n_ranks + 255) / 256), dim3(256), 0, stream>>>(ranks, n_ranks, open, time_now, remaining);
    return hipGetLastError();
}

hipError_t launch_convert_comptonized(const PhotonDev &ph, unsigned *converted, hipStream_t stream)
{
    convert_comptonized_kernel<<<dim3((ph.n + 255) / 256), dim3(256), 0, stream>>>(ph, converted);
    return hipGetLastError();
}

hipError_t launch_clear_slots(const PhotonDev &ph, int first, int count, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    clear_slots_kernel<<<dim3((count + 255) / 256), dim3(256), 0, stream>>>(ph, first, count);
    return hipGetLastError();
}

hipError_t launch_rank_reduce(const PhotonDev &ph, int stride, int n_ranks, const RankDesc *desc, ReducePartial *out, int *n_out, hipStream_t stream)
{
    rank_reduce_kernel<<<dim3(n_ranks), dim3(256), 0, stream>>>(ph, stride, desc, out, n_out);
    return hipGetLastError();
}

hipError_t launch_init_states(LoopState *single, LoopState *ranks, int n_ranks, const LoopState &v, hipStream_t stream)
{
    const int n = n_ranks > 0 ? n_ranks : 1;
    init_states_kernel<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(single, ranks, n_ranks, v);
    return hipGetLastError();
}

hipError_t launch_aos_to_soa(const void *aos, const PhotonDev &ph, int n, hipStream_t stream)
{
    aos_to_soa_kernel<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(static_cast<const mcrat_hip_photon *>(aos), ph, n);
    return hipGetLastError();
}

// desc: `count` device records {rank, n, first record} (three 64-bit-aligned fields, see PoolSetDesc); aos: all lists' records, concatenated
hipError_t launch_pool_aos_to_soa(const void *aos, const PhotonDev &pool, int stride, const void *desc, int count, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    pool_aos_to_soa_kernel<<<dim3((stride + 255) / 256, count), dim3(256), 0, stream>>>(static_cast<const mcrat_hip_photon *>(aos), pool, stride,
                                                                                         static_cast<const PoolSetDesc *>(desc));
    return hipGetLastError();
}

hipError_t launch_soa_to_aos(const PhotonDev &ph, void *aos, int first, int n, hipStream_t stream)
{
    soa_to_aos_kernel<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(ph, static_cast<mcrat_hip_photon *>(aos), first, n);
    return hipGetLastError();
}

int cs_absorb_blocks(int n) { const int b = (n + 255) / 256; return b < 1 ? 1 : (b > 1024 ? 1024 : b); }

hipError_t launch_cs_absorb(const CsParams &p, const PhotonDev &ph, const double *temp, const HydroCols &h, CsAbsPartial *partials, hipStream_t stream)
{
    cs_absorb_kernel<<<dim3(cs_absorb_blocks(ph.n)), dim3(256), 0, stream>>>(p, ph, temp, h, partials);
    return hipGetLastError();
}

// ---- shared clock, device-initiated exchange (one-shot peer writes + a local wait; launch.hpp).  sc_push_kernel copies this GPU's
// proposal of the round into slot `rank` of every peer's receive buffer (peer pointers: the same process' other contexts, hipIpc mappings of
// other processes' buffers, or other GPUs' over xGMI; the buffers are fine-grained allocations) and then stamps the round into the peer's
// flag word for this rank; sc_wait_kernel, lane r, waits until rank r's stamp has reached the round.  Two receive buffers alternate by the
// round's parity: a rank can push round k + 2 only after its own resolve of round k + 1, which needed every peer's push of round k + 1,
// which that peer issued after ITS resolve of round k -- so nobody still reads the buffer that is overwritten.  The wait is bounded.
// Once a wait has given up (my_flags[SC_GAVE_UP_WORD] != 0) the exchange is dead for this frame: push and wait return at once -- no further round
// stamps go out, no later wait spins its budget again -- and the loop state is parked (LoopState::done = LOOP_SC_GAVE_UP: step, propose and resolve
// are no-ops), so the photons stay at the last pass every rank completed.  mcrat_hip_shared_clock_reset_exchange clears the state.
__global__ __launch_bounds__(256) void sc_push_kernel(const ScProposal *__restrict__ send, ScPeers peers, unsigned long long *my_flags, int world, int rank)
{
    sc_push_body(send, peers, my_flags, world, rank);
}

// (if a stamp does not come within max_spins the wait gives up: nothing is copied -- a stale half of the buffer must never be resolved -- and the loop is parked)
__global__ __launch_bounds__(256) void sc_wait_kernel(unsigned long long *my_flags, const ScProposal *recv, ScProposal *gathered, int world, int max_spins,
                                                      LoopState *st)
{
    __shared__ int s_failed;
    (void)sc_wait_body(my_flags, recv, gathered, world, max_spins, st, &s_failed);
}

hipError_t launch_sc_push(const ScProposal *send, const ScPeers &peers, unsigned long long *my_flags, int world, int rank, hipStream_t stream)
{
    sc_push_kernel<<<dim3(1), dim3(256), 0, stream>>>(send, peers, my_flags, world, rank);
    return hipGetLastError();
}

hipError_t launch_sc_wait(unsigned long long *my_flags, const ScProposal *recv, ScProposal *gathered, int world, int max_spins, LoopState *st, hipStream_t stream)
{
    sc_wait_kernel<<<dim3(1), dim3(256), 0, stream>>>(my_flags, recv, gathered, world, max_spins, st);
    return hipGetLastError();
}

hipError_t launch_cs_absorb_pool(const CsParams &p, const PhotonDev &pool, int stride, int n_ranks, const RankDesc *desc, const int *open, const double *temp,
                                 const HydroCols &h, CsAbsPartial *per_list, hipStream_t stream)
{
    cs_absorb_pool_kernel<<<dim3(n_ranks), dim3(256), 0, stream>>>(p, pool, stride, desc, open, temp, h, per_list);
    return hipGetLastError();
}

hipError_t launch_output_count(const PhotonDev &ph, int n, unsigned *block_count, unsigned long long *d_total, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    output_count_kernel<<<dim3(((n + 255) / 256 + OUT_CHUNKS - 1) / OUT_CHUNKS), dim3(256), 0, stream>>>(ph, n, block_count, d_total);
    return hipGetLastError();
}

hipError_t launch_output_write(const PhotonDev &ph, int n, const int *block_start, const OutputCols &out, hipStream_t stream)
{
    output_write_kernel<<<dim3((n + 255) / 256), dim3(256), 0, stream>>>(ph, n, block_start, out);
    return hipGetLastError();
}

}  // namespace mcrat
