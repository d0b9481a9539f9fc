// inject.hip -- photonInjection on the device (Src/mclib.c:9-300): the producer of the loop's input.
// Cells of the staged hydro frame that touch the injection slab r_inj -+ c/(2 fps), theta_min..theta_max get a Poisson
// number of photons with mean (4/3) V gamma a_n T^3 / weight; the weight is adjusted (x10 / x0.5) until the total lies
// in [min_photons, max_photons]; each photon draws a comoving black-body (Bjorkman & Wood 2001) or Wien frequency, an
// isotropic comoving direction, is boosted to the lab frame with its cell's velocity and placed uniformly in the cell.
// Random numbers: rng.hpp's keyed source -- the count of cell i in attempt a from the stream {a, i, INJECT_COUNT}, the
// draws of photon k (numbered over the whole injection, cells ascending) from the stream {0, k, INJECT_PHOTON}, in the
// reference's draw order.  gsl_ran_poisson's algorithm lives in GSL; the count sampler here (Knuth below a mean of 30,
// Hormann's PTRS above) is the one the oracle restates.
#include <hip/hip_runtime.h>
#include <math.h>
#include "device_types.hpp"
#include "launch.hpp"
#include "physics.hpp"
#include "rng.hpp"
#include "cs_device.hpp"

namespace mcrat {

namespace {

constexpr uint32_t RNG_INJECT_COUNT = 2u;
constexpr uint32_t RNG_INJECT_PHOTON = 3u;


// geometry.c:66-106
__device__ __forceinline__ void hydro_to_spherical(int dims, int geom, double r0, double r1, double r2, double &r, double &theta)
{
    r = 0; theta = 0;
    if (dims == DIM_TWO || dims == DIM_TWO_POINT_FIVE) {
        if (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL) { r = sqrt(r0 * r0 + r1 * r1); theta = atan2(r0, r1); }
        if (geom == GEOM_SPHERICAL) { r = r0; theta = r1; }
    } else {
        if (geom == GEOM_CARTESIAN) { r = sqrt(r0 * r0 + r1 * r1 + r2 * r2); theta = acos(r2 / r); }
        if (geom == GEOM_SPHERICAL) { r = r0; theta = r1; }
        if (geom == GEOM_POLAR) { r = sqrt(r0 * r0 + r2 * r2); theta = acos(r2 / r); }
    }
}




// hydroElementVolume, geometry.c:255-296
__device__ __forceinline__ double element_volume(int dims, int geom, const CellRec &c)
{
    const double r0_max = c.c0 + 0.5 * c.s0, r0_min = c.c0 - 0.5 * c.s0;
    const double r1_max = c.c1 + 0.5 * c.s1, r1_min = c.c1 - 0.5 * c.s1;
    double V = 0;
    if (dims == DIM_TWO || dims == DIM_TWO_POINT_FIVE) {
        if (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL) V = M_PI * (r0_max * r0_max - r0_min * r0_min) * c.s1;
        if (geom == GEOM_SPHERICAL) V = (2.0 * M_PI / 3.0) * (r0_max * r0_max * r0_max - r0_min * r0_min * r0_min) * (cos(r1_min) - cos(r1_max));
    } else {
        const double r2_max = c.c2 + 0.5 * c.s2, r2_min = c.c2 - 0.5 * c.s2;
        if (geom == GEOM_CARTESIAN) V = c.s0 * c.s1 * c.s2;
        if (geom == GEOM_SPHERICAL) V = (1.0 / 3.0) * (r0_max * r0_max * r0_max - r0_min * r0_min * r0_min) * (cos(r1_min) - cos(r1_max)) * (r2_max - r2_min);
        if (geom == GEOM_POLAR) V = 0.5 * (r0_max * r0_max - r0_min * r0_min) * c.s1 * c.s2;
    }
    return V;
}

// mclib.c:40-57
__device__ __forceinline__ bool in_injection_slab(const InjectParams &p, const CellRec &c)
{
    double r_in, th_in, r_out, th_out;
    if (p.dimensions == DIM_THREE) {
        hydro_to_spherical(p.dimensions, p.geometry, fabs(c.c0) - 0.5 * c.s0, fabs(c.c1) - 0.5 * c.s1, fabs(c.c2) - 0.5 * c.s2, r_in, th_in);
        hydro_to_spherical(p.dimensions, p.geometry, fabs(c.c0) + 0.5 * c.s0, fabs(c.c1) + 0.5 * c.s1, fabs(c.c2) + 0.5 * c.s2, r_out, th_out);
    } else {
        hydro_to_spherical(p.dimensions, p.geometry, c.c0 - 0.5 * c.s0, c.c1 - 0.5 * c.s1, 0, r_in, th_in);
        hydro_to_spherical(p.dimensions, p.geometry, c.c0 + 0.5 * c.s0, c.c1 + 0.5 * c.s1, 0, r_out, th_out);
    }
    return (p.rmin <= r_out) && (r_in <= p.rmax) && (th_out >= p.theta_min) && (th_in <= p.theta_max);
}

// stands for gsl_ran_poisson (mclib.c:114): see the head of this file
__device__ __forceinline__ long long poisson(EventStream &rng, double mean)
{
    if (!(mean > 0)) return 0;
    if (mean < 30.0) {
        const double L = exp(-mean);
        long long k = 0;
        double prod = 1.0;
        do {
            k += 1;
            prod *= rng.uniform_pos();
        } while (prod > L);
        return k - 1;
    }
    const double smu = sqrt(mean);
    const double b = 0.931 + 2.53 * smu;
    const double a = -0.059 + 0.02483 * b;
    const double inv_alpha = 1.1239 + 1.1328 / (b - 3.4);
    const double v_r = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < phys::REJECTION_CAP; ++it) {
        const double U = rng.uniform() - 0.5;
        const double V = rng.uniform_pos();
        const double us = 0.5 - fabs(U);
        const double kf = floor((2.0 * a / us + b) * U + mean + 0.43);
        if (us >= 0.07 && V <= v_r) return (long long)kf;
        if (kf < 0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(inv_alpha) - log(a / (us * us) + b) <= -mean + kf * log(mean) - lgamma(kf + 1.0)) return (long long)kf;
    }
    return (long long)mean;
}

__global__ __launch_bounds__(256) void inject_count_kernel(InjectParams p, HydroDev hy, double weight, unsigned long long attempt, RngKey key,
                                                           unsigned *__restrict__ count, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long s_sum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (i < hy.M) {
        const CellRec c = load_cell(hy, p.dimensions, i);
        unsigned n = 0;
        if (in_injection_slab(p, c)) {
            const double T = hy.temp[i];
            const double gamma = hy.gamma[i];
            const double ph_dens_calc = (4.0 / 3.0) * element_volume(p.dimensions, p.geometry, c) * ((gamma * p.num_dens_coeff * T * T * T) / weight);   // mclib.c:110
            EventStream rng = keyed_stream(key, attempt, (uint32_t)i, RNG_INJECT_COUNT);
            const long long k = poisson(rng, ph_dens_calc);
            n = (unsigned)(k < 0 ? 0 : (k > 0x7fffffffll ? 0x7fffffffll : k));
        }
        count[i] = n;
        mine = n;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// exclusive scan of one value per thread over the workgroup (256 threads); returns the thread's offset, *total the sum
__device__ __forceinline__ int block_exclusive_scan_256(int v, int *s_w, int *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
    for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off); if (lane >= off) x += y; }
    __syncthreads();
    if (lane == 63) s_w[w] = x;
    __syncthreads();
    int before = 0;
    for (int k = 0; k < w; ++k) before += s_w[k];
    *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return before + x - v;
}

// photon k of an injection, born in cell i (mclib.c:150-296), into slot k of `ph`
__device__ void inject_one(const InjectParams &p, const HydroDev &hy, double weight, const RngKey &key, int k, int i, const PhotonDev &ph)
{
    const CellRec c = load_cell(hy, p.dimensions, i);
    const double T = hy.temp[i];
    EventStream rng = keyed_stream(key, 0ull, (uint32_t)k, RNG_INJECT_PHOTON);
    double fr_dum = 0;
    if (p.wien) {                                                            // mclib.c:175-190
        double y_dum = 1, yfr_dum = 0;
        for (int it = 0; it < phys::REJECTION_CAP && (y_dum > yfr_dum); ++it) {
            fr_dum = rng.uniform_pos() * 6.3e11 * T;
            y_dum = rng.uniform_pos();
            yfr_dum = (1.0 / (1.29e31)) * pow((fr_dum / T), 3.0) / (exp((PL_CONST * fr_dum) / (K_B * T)) - 1);
        }
    } else {                                                                 // mclib.c:199-214
        double test = 0, test_cnt = 0;
        const double r1 = rng.uniform_pos(), r2 = rng.uniform_pos(), r3 = rng.uniform_pos(), r4 = rng.uniform_pos(), r5 = rng.uniform_pos();
        while (test < M_PI * M_PI * M_PI * M_PI * r1 / 90.0) {
            test_cnt += 1;
            test += 1 / (test_cnt * test_cnt * test_cnt * test_cnt);
        }
        fr_dum = -log(r2 * r3 * r4 * r5) / test_cnt;
        fr_dum *= K_B * T / PL_CONST;
    }
    double position_phi = 0;
    if (p.dimensions != DIM_THREE) position_phi = rng.uniform() * 2 * M_PI;   // mclib.c:223-227
    const double com_v_phi = rng.uniform() * 2 * M_PI;
    const double com_v_theta = acos((rng.uniform() * 2) - 1);
    double p_comv[4];
    p_comv[0] = PL_CONST * fr_dum / C_LIGHT;                                 // mclib.c:232-235
    p_comv[1] = (PL_CONST * fr_dum / C_LIGHT) * sin(com_v_theta) * cos(com_v_phi);
    p_comv[2] = (PL_CONST * fr_dum / C_LIGHT) * sin(com_v_theta) * sin(com_v_phi);
    p_comv[3] = (PL_CONST * fr_dum / C_LIGHT) * cos(com_v_theta);
    // fluid velocity of the cell at the photon's azimuth (mclib.c:239-246), from the staged record (physics.hpp, cell_beta)
    const CellFluid f = hy.fluid[i];
    const double fcv = hy.fluid_c ? hy.fluid_c[i] : 0.0;
    const double cphi = cos(position_phi), sphi = sin(position_phi);
    double boost[3];
    if (p.dimensions == DIM_TWO) phys::beta_from_record<DIM_TWO>(f.a, f.b, fcv, cphi, sphi, boost);
    else if (p.dimensions == DIM_TWO_POINT_FIVE) phys::beta_from_record<DIM_TWO_POINT_FIVE>(f.a, f.b, fcv, cphi, sphi, boost);
    else phys::beta_from_record<DIM_THREE>(f.a, f.b, fcv, cphi, sphi, boost);
    boost[0] *= -1; boost[1] *= -1; boost[2] *= -1;
    double l_boost[4];
    phys::lorentz_boost(boost, p_comv, l_boost, true);                       // mclib.c:252
    const double position_rand = rng.uniform_pos() * c.s0 - 0.5 * c.s0;      // mclib.c:265-266
    const double position2_rand = rng.uniform_pos() * c.s1 - 0.5 * c.s1;
    double xyz[3];
    if (p.dimensions == DIM_THREE) {
        const double position3_rand = rng.uniform_pos() * c.s2 - 0.5 * c.s2;
        hydro_to_mcrat(p.dimensions, p.geometry, c.c0 + position_rand, c.c1 + position2_rand, c.c2 + position3_rand, xyz);
    } else {
        hydro_to_mcrat(p.dimensions, p.geometry, c.c0 + position_rand, c.c1 + position2_rand, position_phi, xyz);
    }
    ph.r0[k] = xyz[0]; ph.r1[k] = xyz[1]; ph.r2[k] = xyz[2];
    ph.p0[k] = l_boost[0]; ph.p1[k] = l_boost[1]; ph.p2[k] = l_boost[2]; ph.p3[k] = l_boost[3];
    ph.c0[k] = p_comv[0]; ph.c1[k] = p_comv[1]; ph.c2[k] = p_comv[2]; ph.c3[k] = p_comv[3];
    ph.s0[k] = 1; ph.s1[k] = 0; ph.s2[k] = 0; ph.s3[k] = 0;                  // mclib.c:281-291
    ph.num_scatt[k] = 0;
    ph.weight[k] = weight;
    ph.tau[k] = 0; ph.tts[k] = 0; ph.tau_next[k] = 0;
    double u0 = 0, u1 = 0, u2 = 0;
    if (l_boost[0] != 0) {
        const double d = 1.0 / l_boost[0];
        u0 = l_boost[1] * d * C_LIGHT; u1 = l_boost[2] * d * C_LIGHT; u2 = l_boost[3] * d * C_LIGHT;
    }
    ph.u0[k] = u0; ph.u1[k] = u1; ph.u2[k] = u2;
    ph.ntau[k] = -INFINITY;                                                  // -1 / total_optical_depth with the 0 of a fresh photon
    ph.idx[k] = 0;
    unsigned fl = FLAG_VALID | FLAG_RECALC;                                  // recalc_properties = 1
    if (weight != 0) fl |= FLAG_MOVES;                                       // type 'i' is not a CS-pool photon
    ph.flags[k] = (unsigned char)fl;
    ph.type[k] = 'i';                                                        // INJECTED_PHOTON, mcrat.h
}

__global__ __launch_bounds__(256) void inject_generate_kernel(InjectParams p, HydroDev hy, double weight, RngKey key, const int *__restrict__ start,
                                                              PhotonDev ph)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ph.n) return;
    // the cell of photon k: start[i] <= k < start[i+1]
    int lo = 0, hi = hy.M;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= k) lo = mid; else hi = mid;
    }
    inject_one(p, hy, weight, key, k, lo, ph);
}

// ---- photonInjection for many lists of a rank pool at once.  Whether a cell touches the injection slab, its volume and gamma T^3 do not depend
// on the list: inject_slab_* find the slab's cells once per group of lists with the same slab (an angle bin's ranks), in cell order, with the
// two factors of mclib.c:110; inject_pool_kernel, one workgroup per list, runs the list's weight loop on its own Poisson streams (the keys of
// the one-list path: {seed, attempt, cell, INJECT_COUNT}), and writes its photons ({seed, 0, k, INJECT_PHOTON}) into the list's window.
__global__ __launch_bounds__(256) void inject_slab_flag_kernel(InjectParams p, HydroDev hy, unsigned *__restrict__ flag, unsigned long long *__restrict__ total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned in = 0;
    if (i < hy.M) {
        in = in_injection_slab(p, load_cell(hy, p.dimensions, i)) ? 1u : 0u;
        flag[i] = in;
    }
    const unsigned long long m = __ballot(in != 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(total, (unsigned long long)__popcll(m));
}

__global__ __launch_bounds__(256) void inject_slab_write_kernel(InjectParams p, HydroDev hy, const unsigned *__restrict__ flag, const int *__restrict__ start,
                                                                InjectSlabCell *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hy.M || !flag[i]) return;
    const CellRec c = load_cell(hy, p.dimensions, i);
    const double T = hy.temp[i];
    InjectSlabCell s;
    s.cell = i; s.pad = 0;
    s.v43 = (4.0 / 3.0) * element_volume(p.dimensions, p.geometry, c);       // mclib.c:110: (4/3) V * ((gamma a_n T^3) / weight), in that order
    s.g = hy.gamma[i] * p.num_dens_coeff * T * T * T;
    out[start[i]] = s;
}

__global__ __launch_bounds__(256) void inject_pool_kernel(InjectParams p, HydroDev hy, PhotonDev pool, int stride, const InjectSlabCell *__restrict__ slab,
                                                          int n_slab, PoolInject *lists, int group)
{
    __shared__ unsigned long long s_sum[4];
    __shared__ int s_w[4];
    __shared__ int s_cell[POOL_INJECT_CAP];
    __shared__ double s_weight;
    __shared__ int s_ok, s_attempt;
    const int r = blockIdx.x, tid = threadIdx.x;
    PoolInject &L = lists[r];
    if (!L.inject || L.group != group) return;
    PhotonDev ph = pool;
    offset_photons(ph, (size_t)r * (size_t)stride);
    const RngKey key = {L.seed, L.stream, 0u};
    if (tid == 0) { s_weight = L.weight_in; s_ok = 0; s_attempt = 0; }
    __syncthreads();
    unsigned long long total = 0;
    for (int attempt = 0; attempt <= 200; ++attempt) {                       // mclib.c:87-136
        const double weight = s_weight;
        unsigned long long mine = 0;
        for (int s = tid; s < n_slab; s += 256) {
            const double ph_dens_calc = slab[s].v43 * (slab[s].g / weight);
            EventStream rng = keyed_stream(key, (unsigned long long)attempt, (uint32_t)slab[s].cell, RNG_INJECT_COUNT);
            const long long k = poisson(rng, ph_dens_calc);
            mine += (unsigned long long)(k < 0 ? 0 : (k > 0x7fffffffll ? 0x7fffffffll : k));
        }
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_sum[tid >> 6] = mine;
        __syncthreads();
        total = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        if (tid == 0) {
            if (total > (unsigned long long)L.max_photons) s_weight = weight * 10;
            else if (total < (unsigned long long)L.min_photons) s_weight = weight * 0.5;
            else { s_ok = 1; s_attempt = attempt; }
        }
        __syncthreads();
        if (s_ok) break;
    }
    if (!s_ok) { if (tid == 0) { L.error = 1; L.n = 0; } return; }
    const double weight = s_weight;
    const int attempt = s_attempt, n = (int)total;
    if (tid == 0) { L.n = n; L.weight_out = weight; L.error = n == 0 ? 2 : (n > stride || n > POOL_INJECT_CAP ? 3 : 0); }
    if (n == 0 || n > stride || n > POOL_INJECT_CAP) return;
    // photon k -> its cell: the accepted attempt's counts again, in cell order
    int base = 0;
    for (int c0 = 0; c0 < n_slab; c0 += 256) {
        const int s = c0 + tid;
        int cnt = 0, cell = -1;
        if (s < n_slab) {
            const double ph_dens_calc = slab[s].v43 * (slab[s].g / weight);
            EventStream rng = keyed_stream(key, (unsigned long long)attempt, (uint32_t)slab[s].cell, RNG_INJECT_COUNT);
            const long long k = poisson(rng, ph_dens_calc);
            cnt = (int)(k < 0 ? 0 : (k > 0x7fffffffll ? 0x7fffffffll : k));
            cell = slab[s].cell;
        }
        int chunk_total;
        const int off = block_exclusive_scan_256(cnt, s_w, &chunk_total);
        for (int j = 0; j < cnt; ++j) s_cell[base + off + j] = cell;
        base += chunk_total;
        __syncthreads();
    }
    ph.n = n;
    for (int k = tid; k < n; k += 256) inject_one(p, hy, weight, key, k, s_cell[k], ph);
}

// ---------------------------------------------------------------------------------------------- cyclo-synchrotron pool emission


// mc_cyclosynch.c:1215-1226 (note the strict upper bounds, unlike the injection's slab)
__device__ __forceinline__ bool in_emission_slab(const CsEmitParams &p, const CellRec &c)
{
    double r_in, th_in, r_out, th_out;
    if (p.dimensions == DIM_THREE) {
        hydro_to_spherical(p.dimensions, p.geometry, fabs(c.c0) - 0.5 * c.s0, fabs(c.c1) - 0.5 * c.s1, fabs(c.c2) - 0.5 * c.s2, r_in, th_in);
        hydro_to_spherical(p.dimensions, p.geometry, fabs(c.c0) + 0.5 * c.s0, fabs(c.c1) + 0.5 * c.s1, fabs(c.c2) + 0.5 * c.s2, r_out, th_out);
    } else {
        hydro_to_spherical(p.dimensions, p.geometry, c.c0 - 0.5 * c.s0, c.c1 - 0.5 * c.s1, 0, r_in, th_in);
        hydro_to_spherical(p.dimensions, p.geometry, c.c0 + 0.5 * c.s0, c.c1 + 0.5 * c.s1, 0, r_out, th_out);
    }
    return (p.rmin <= r_out) && (r_in < p.rmax) && (th_out >= p.theta_min) && (th_in < p.theta_max);
}

// blackbody_ph_spect, mc_cyclosynch.c:185-196
__device__ __forceinline__ double planck_tail(double nu, double temp)
{
    return (8 * M_PI * nu * nu) / (exp(PL_CONST * nu / (K_B * temp)) - 1) / (C_LIGHT * C_LIGHT * C_LIGHT);
}

// QUADPACK's 21-point Gauss-Kronrod rule (dqk21) on blackbody_ph_spect over [a, b], as oracle/oracle_cyclosynch.c (orc_qk21) restates it: the
// integral, its error estimate, and the two sums QAGS' tests use
__device__ double qk21_planck(double a, double b, double temp, double &abserr, double &resabs, double &resasc)
{
    const double XGK[11] = {0.995657163025808080735527280689003, 0.973906528517171720077964012084452, 0.930157491355708226001207180059508,
                            0.865063366688984510732096688423493, 0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
                            0.562757134668604683339000099272694, 0.433395394129247190799265943165784, 0.294392862701460198131126603103866,
                            0.148874338981631210884826001129720, 0.0};
    const double WGK[11] = {0.011694638867371874278064396062192, 0.032558162307964727478818972459390, 0.054755896574351996031381300244580,
                            0.075039674810919952767043140916190, 0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
                            0.123491976262065851077958109585166, 0.134709217311473325928054001771707, 0.142775938577060080797094273138717,
                            0.147739104901338491374841515972068, 0.149445554002916905664936468389821};
    const double WG[5] = {0.066671344308688137593568809893332, 0.149451349150580593145776339657697, 0.219086362515982043995534934228163,
                          0.269266719309996355091226921569469, 0.295524224714752870173815619188769};
    const double centr = 0.5 * (a + b), hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    const double fc = planck_tail(centr, temp);
    double fv1[10], fv2[10], resg = 0, resk = WGK[10] * fc, rabs = fabs(resk);
    for (int j = 0; j < 5; j++) {
        const int jtw = 2 * j + 1;
        const double absc = hlgth * XGK[jtw], f1 = planck_tail(centr - absc, temp), f2 = planck_tail(centr + absc, temp);
        fv1[jtw] = f1; fv2[jtw] = f2;
        resg += WG[j] * (f1 + f2);
        resk += WGK[jtw] * (f1 + f2);
        rabs += WGK[jtw] * (fabs(f1) + fabs(f2));
    }
    for (int j = 0; j < 5; j++) {
        const int jtwm1 = 2 * j;
        const double absc = hlgth * XGK[jtwm1], f1 = planck_tail(centr - absc, temp), f2 = planck_tail(centr + absc, temp);
        fv1[jtwm1] = f1; fv2[jtwm1] = f2;
        resk += WGK[jtwm1] * (f1 + f2);
        rabs += WGK[jtwm1] * (fabs(f1) + fabs(f2));
    }
    const double reskh = resk * 0.5;
    double rasc = WGK[10] * fabs(fc - reskh);
    for (int j = 0; j < 10; j++) rasc += WGK[j] * (fabs(fv1[j] - reskh) + fabs(fv2[j] - reskh));
    double err = fabs((resk - resg) * hlgth);
    const double result = resk * hlgth;
    rabs *= dhlgth; rasc *= dhlgth;
    if (rasc != 0 && err != 0) { const double s = pow(200 * err / rasc, 1.5); err = (s < 1) ? rasc * s : rasc; }
    if (rabs > 2.2250738585072014e-308 / (50 * 2.220446049250313e-16)) { const double m = 50 * 2.220446049250313e-16 * rabs; if (m > err) err = m; }
    abserr = err; resabs = rabs; resasc = rasc;
    return result;
}

// gsl_integration_qags(blackbody_ph_spect, 10, nu_c, 0, 1e-2, limit 10000, ...) (:1276) as orc_qags restates it: the rule on the whole interval
// and QAGS' first-step test -- where the integrand is the Rayleigh-Jeans tail (every cell of cfg5) that is all of it -- and otherwise bisection of the
// interval with the largest error estimate until the sum of the estimates meets the tolerance (no epsilon-algorithm extrapolation: the oracle's
// documented stand-in for the rest of QAGS, the same intervals in the same order and the same sums here).  The intervals live in the lane's scratch
// memory: at most QAGS_DEV_INTERVALS of them.  status: 0 met the tolerance, 1 roundoff (GSL_EROUND; the first rule's value stands), 2 ran out of
// intervals -- the caller counts those cells and the host refuses the emission.
constexpr int QAGS_DEV_INTERVALS = 64;
__device__ double qags_planck(double a, double b, double temp, int &status, bool &bisected)
{
    const double epsabs = 0.0, epsrel = 1e-2;
    double err, rabs, rasc;
    const double res = qk21_planck(a, b, temp, err, rabs, rasc);
    double tol = fmax(epsabs, epsrel * fabs(res));
    bisected = false;
    if (err <= 100 * 2.220446049250313e-16 * rabs && err > tol) { status = 1; return res; }
    if ((err <= tol && err != rasc) || err == 0.0) { status = 0; return res; }
    bisected = true;
    double al[QAGS_DEV_INTERVALS], bl[QAGS_DEV_INTERVALS], rl[QAGS_DEV_INTERVALS], el[QAGS_DEV_INTERVALS];
    int n = 1;
    al[0] = a; bl[0] = b; rl[0] = res; el[0] = err;
    double result = res;
    status = 2;
    while (n < QAGS_DEV_INTERVALS) {
        int worst = 0;
        for (int k = 1; k < n; k++) if (el[k] > el[worst]) worst = k;
        const double mid = 0.5 * (al[worst] + bl[worst]);
        double e1, e2, t1, t2;
        const double r1 = qk21_planck(al[worst], mid, temp, e1, t1, t2);
        const double r2 = qk21_planck(mid, bl[worst], temp, e2, t1, t2);
        al[n] = mid; bl[n] = bl[worst]; rl[n] = r2; el[n] = e2;
        bl[worst] = mid; rl[worst] = r1; el[worst] = e1;
        n++;
        double sr = 0, se = 0;
        for (int k = 0; k < n; k++) { sr += rl[k]; se += el[k]; }
        result = sr;
        tol = fmax(epsabs, epsrel * fabs(sr));
        if (se <= tol) { status = 0; break; }
    }
    return result;
}

// mc_cyclosynch.c:1244-1296, one weight: the Poisson count of every cell of the shell; flags[0] counts cells whose integral ran out of
// intervals (qags_planck), flags[1] the cells of the shell, flags[2] the cells whose integral needed more than QAGS' first rule
__global__ __launch_bounds__(256) void cs_emit_count_kernel(CsEmitParams p, HydroDev hy, HydroCols h, double weight, unsigned long long attempt, RngKey key,
                                                            unsigned *__restrict__ count, unsigned long long *__restrict__ total, unsigned *__restrict__ flags)
{
    __shared__ unsigned long long s_sum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (i < hy.M) {
        const CellRec c = load_cell(hy, p.dimensions, i);
        unsigned n = 0;
        if (in_emission_slab(p, c)) {
            if (attempt == 0) atomicAdd(flags + 1, 1u);
            const double nu_c = cs_nu_c(p, hy, h, i);
            int status;
            bool bisected;
            double ph_dens_calc = qags_planck(10, nu_c, hy.temp[i], status, bisected);
            if (status == 2) atomicAdd(flags, 1u);                                                   // out of intervals: the host refuses
            if (bisected && attempt == 0) atomicAdd(flags + 2, 1u);                                  // (statistics: cells past QAGS' first rule)
            ph_dens_calc *= element_volume(p.dimensions, p.geometry, c) / weight;                    // :1277
            EventStream rng = keyed_stream(key, attempt, (uint32_t)i, RNG_CS_COUNT);
            const long long k = poisson(rng, ph_dens_calc);
            n = (unsigned)(k < 0 ? 0 : (k > 0x7fffffffll ? 0x7fffffffll : k));
        }
        count[i] = n;
        mine = n;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}


// mc_cyclosynch.c:1340-1455: pool photon k into null slot null_slots[k], nearest_block_index = 0 (:1436)
__global__ __launch_bounds__(256) void cs_emit_generate_kernel(CsEmitParams p, HydroDev hy, HydroCols h, double weight, RngKey key,
                                                               const int *__restrict__ start, int n_emit, const int *__restrict__ null_slots, PhotonDev ph)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_emit) return;
    int lo = 0, hi = hy.M;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= k) lo = mid; else hi = mid;
    }
    EventStream rng = keyed_stream(key, 0ull, (uint32_t)k, RNG_CS_PHOTON);
    (void)cs_emit_one(p, hy, h, lo, weight, 0, rng, ph, null_slots[k]);
}

// The hook of mcrat.c:786-808 after a pass.  If the pass called photonEvent and the photon it reports is a pool photon, that photon
// becomes a comptonised one and is replaced by a fresh pool photon of its weight in its cell (photonEmitCyclosynch with
// inject_single_switch = 1, mc_cyclosynch.c:1467-1558, into the list's first null slot, photons.c:139-160), and it is itself moved to
// a random place in that cell (:1540-1556); then the rebinning trigger of :797-808.  One workgroup; the pending advance must have
// been applied (flush_kernel) so that the positions are current.  See CsFrame (launch.hpp) for how it parks the loop.
// ---- pool emission for many lists at once (rank pool).  What photonEmitCyclosynch computes per cell -- whether the cell lies in the
// emission shell, the Planck integral below its cyclotron frequency, its volume -- does not depend on the list; lists that emit into the
// same shell (same injection radius, frame numbers and angle range: the ranks of an angle bin) share it.  cs_shell_*: the shell's cells
// in ascending order with those per-cell values; cs_emit_pool_kernel: one workgroup per list runs that list's weight loop on its own
// Poisson streams (the streams of the one-list path: {seed, attempt, cell, RNG_CS_COUNT}), places the photons in its null slots in
// slot order (the list doubling inside its window of the pool when it has none), and generates them ({seed, 0, k, RNG_CS_PHOTON}).
__global__ __launch_bounds__(256) void cs_shell_flag_kernel(CsEmitParams p, HydroDev hy, unsigned *__restrict__ flag, unsigned long long *__restrict__ total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned in = 0;
    if (i < hy.M) {
        in = in_emission_slab(p, load_cell(hy, p.dimensions, i)) ? 1u : 0u;
        flag[i] = in;
    }
    const unsigned long long m = __ballot(in != 0);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(total, (unsigned long long)__popcll(m));
}

__global__ __launch_bounds__(256) void cs_shell_write_kernel(CsEmitParams p, HydroDev hy, HydroCols h, const unsigned *__restrict__ flag,
                                                             const int *__restrict__ start, CsShellCell *__restrict__ out, unsigned *__restrict__ not_converged)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hy.M || !flag[i]) return;
    const CellRec c = load_cell(hy, p.dimensions, i);
    int status;
    bool bisected;
    CsShellCell s;
    s.cell = i; s.pad = 0;
    s.integral = qags_planck(10, cs_nu_c(p, hy, h, i), hy.temp[i], status, bisected);                  // :1276
    s.volume = element_volume(p.dimensions, p.geometry, c);
    if (status == 2) atomicAdd(not_converged, 1u);                                                     // out of intervals: the host refuses
    out[start[i]] = s;
}

constexpr int CS_POOL_EMIT_CAP = 4096;       // pool photons one list may receive in one emission (LDS tables of cells and slots)

__global__ __launch_bounds__(256) void cs_emit_pool_kernel(CsEmitParams p, HydroDev hy, HydroCols h, PhotonDev pool, int stride, RankDesc *desc,
                                                           const CsShellCell *__restrict__ shell, int n_shell, CsPoolEmit *lists, int group)
{
    __shared__ unsigned long long s_sum[4];
    __shared__ int s_w[4];
    __shared__ int s_cell[CS_POOL_EMIT_CAP], s_slot[CS_POOL_EMIT_CAP];
    __shared__ double s_weight;
    __shared__ int s_ok, s_attempt;
    const int r = blockIdx.x, tid = threadIdx.x;
    CsPoolEmit &L = lists[r];
    if (!L.open || L.group != group) return;
    const RankDesc d = desc[r];
    PhotonDev ph = pool;
    offset_photons(ph, (size_t)r * (size_t)stride);
    ph.n = d.len;
    const RngKey key = {L.seed, d.stream, 0u};
    // ---- :1244-1296: the weight loop
    if (tid == 0) { s_weight = L.weight_in; s_ok = 0; s_attempt = 0; }
    __syncthreads();
    const int min_photons = n_shell > 0 ? 1 : 0;                                          // no cell in the shell: nothing to emit (:1236-1239)
    unsigned long long total = 0;
    for (int attempt = 0; attempt <= 400; ++attempt) {
        const double weight = s_weight;
        unsigned long long mine = 0;
        for (int s = tid; s < n_shell; s += 256) {
            double ph_dens_calc = shell[s].integral;
            ph_dens_calc *= shell[s].volume / weight;                                      // :1277
            EventStream rng = keyed_stream(key, (unsigned long long)attempt, (uint32_t)shell[s].cell, RNG_CS_COUNT);
            const long long k = poisson(rng, ph_dens_calc);
            mine += (unsigned long long)(k < 0 ? 0 : (k > 0x7fffffffll ? 0x7fffffffll : k));
        }
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) s_sum[tid >> 6] = mine;
        __syncthreads();
        total = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        if (tid == 0) {
            if ((double)total > L.max_photons) s_weight = weight * 10;
            else if ((long long)total < min_photons) s_weight = weight * 0.5;
            else { s_ok = 1; s_attempt = attempt; }
        }
        __syncthreads();
        if (s_ok) break;
    }
    if (!s_ok) { if (tid == 0) { L.error = 1; L.n_emit = 0; } return; }
    const double weight = s_weight;
    const int attempt = s_attempt, n_emit = (int)total;
    if (tid == 0) { L.n_emit = n_emit; L.weight_out = weight; L.error = 0; }
    if (n_emit == 0) return;
    if (n_emit > CS_POOL_EMIT_CAP) { if (tid == 0) L.error = 4; return; }
    // ---- photon k -> its cell: the counts of the accepted attempt again, in cell order (:1340-1455)
    int base = 0;
    for (int c0 = 0; c0 < n_shell; c0 += 256) {
        const int s = c0 + tid;
        int cnt = 0, cell = -1;
        if (s < n_shell) {
            double ph_dens_calc = shell[s].integral;
            ph_dens_calc *= shell[s].volume / weight;
            EventStream rng = keyed_stream(key, (unsigned long long)attempt, (uint32_t)shell[s].cell, RNG_CS_COUNT);
            const long long k = poisson(rng, ph_dens_calc);
            cnt = (int)(k < 0 ? 0 : (k > 0x7fffffffll ? 0x7fffffffll : k));
            cell = shell[s].cell;
        }
        int chunk_total;
        const int off = block_exclusive_scan_256(cnt, s_w, &chunk_total);
        for (int j = 0; j < cnt; ++j) s_cell[base + off + j] = cell;
        base += chunk_total;
        __syncthreads();
    }
    // ---- addToPhotonList (photons.c:108-208): the null slots in slot order, the list doubled first when it has none
    int n = ph.n, n_null = 0;
    for (int pass = 0; pass < 2; ++pass) {
        n_null = 0;
        for (int c0 = 0; c0 < n; c0 += 256) {
            const int i = c0 + tid;
            const int isn = (i < n && ph.type[i] == 'N') ? 1 : 0;
            int chunk_total;
            const int off = block_exclusive_scan_256(isn, s_w, &chunk_total);
            if (isn && n_null + off < n_emit) s_slot[n_null + off] = i;
            n_null += chunk_total;
            __syncthreads();
        }
        if (n_null != 0 || pass == 1) break;
        const long long cap = n;                                                           // photons.c:112-121
        const long long new_cap = (cap * 2 > cap + n_emit) ? cap * 2 : cap * (n_emit / cap);
        if (new_cap > stride) { if (tid == 0) L.error = 3; return; }
        for (int i = n + tid; i < (int)new_cap; i += 256) { ph.type[i] = 'N'; ph.idx[i] = -1; ph.flags[i] = (unsigned char)FLAG_VALID; ph.ntau[i] = -INFINITY; }
        n = (int)new_cap;
        ph.n = n;
        if (tid == 0) desc[r].len = n;
        __threadfence_block();
        __syncthreads();
    }
    if (n_emit > n_null) { if (tid == 0) L.error = 2; return; }                            // "Adding to the photon list has failed"
    __syncthreads();
    for (int k = tid; k < n_emit; k += 256) {
        EventStream rng = keyed_stream(key, 0ull, (uint32_t)k, RNG_CS_PHOTON);
        (void)cs_emit_one(p, hy, h, s_cell[k], weight, 0, rng, ph, s_slot[k]);
    }
}


__global__ __launch_bounds__(256) void cs_replace_kernel(CsEmitParams p, HydroDev hy, HydroCols h, RngKey key, LoopState *st, PhotonDev ph, CsFrame *cf,
                                                         int resume)
{
    __shared__ int s_min[4];
    cs_hook_body(p, hy, h, key, st, ph, cf, resume, s_min, 0, nullptr);
}

// the same hook for the lists of a rank pool, one workgroup per list: the lists rank_loop_kernel has parked because the pass they just
// finished has something for the hook (LOOP_CS_HALT with CsFrame::halt == CS_HALT_HOOK; the kernel leaves the photons current)
__global__ __launch_bounds__(256) void cs_replace_pool_kernel(CsEmitParams p, HydroDev hy, HydroCols h, LoopState *states, PhotonDev pool, int stride,
                                                              RankDesc *desc, CsFrame *frames)
{
    __shared__ int s_min[4];
    const int r = blockIdx.x;
    LoopState *st = states + r;
    CsFrame *cf = frames + r;
    if (st->done != LOOP_CS_HALT || cf->halt != CS_HALT_HOOK) return;
    const RankDesc d = desc[r];
    PhotonDev ph = pool;
    offset_photons(ph, (size_t)r * (size_t)stride);
    ph.n = d.len;
    const RngKey key = {d.seed, d.stream, 0u};
    cs_hook_body(p, hy, h, key, st, ph, cf, 0, s_min, stride, &desc[r].len);
}

// the null slots of the list, ascending (photons.c:181-189)
constexpr int NULL_CHUNKS = 8;
__global__ __launch_bounds__(256) void null_count_kernel(PhotonDev ph, unsigned *__restrict__ block_count, unsigned long long *__restrict__ total)
{
    __shared__ unsigned s_w[NULL_CHUNKS][4];
    const int chunks = (ph.n + 255) / 256;
    for (int c = 0; c < NULL_CHUNKS; ++c) {
        const int i = (blockIdx.x * NULL_CHUNKS + c) * 256 + threadIdx.x;
        const bool is_null = i < ph.n && ph.type[i] == 'N';
        const unsigned long long m = __ballot(is_null);
        if ((threadIdx.x & 63) == 0) s_w[c][threadIdx.x >> 6] = (unsigned)__popcll(m);
    }
    __syncthreads();
    unsigned cnt = 0;
    if (threadIdx.x < NULL_CHUNKS) {
        const int chunk = blockIdx.x * NULL_CHUNKS + threadIdx.x;
        cnt = s_w[threadIdx.x][0] + s_w[threadIdx.x][1] + s_w[threadIdx.x][2] + s_w[threadIdx.x][3];
        if (chunk < chunks) block_count[chunk] = cnt;
    }
    for (int off = NULL_CHUNKS / 2; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    if (threadIdx.x == 0 && cnt) atomicAdd(total, (unsigned long long)cnt);
}

__global__ __launch_bounds__(256) void null_write_kernel(PhotonDev ph, const int *__restrict__ block_start, int *__restrict__ null_slots)
{
    __shared__ unsigned s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool is_null = i < ph.n && ph.type[i] == 'N';
    const unsigned long long m = __ballot(is_null);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (!is_null) return;
    unsigned pos = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    null_slots[block_start[blockIdx.x] + pos] = i;
}

// setNullPhoton (photons.c:210-250) on fresh slots
__global__ __launch_bounds__(256) void null_fill_kernel(PhotonDev ph, int first, int count)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const int i = first + k;
    ph.type[i] = 'N';
    ph.idx[i] = -1;
    ph.flags[i] = (unsigned char)FLAG_VALID;
    ph.ntau[i] = -INFINITY;            // -1 / total_optical_depth with 0; every other column is zero already
}

// ---------------------------------------------------------------------------------------------- rebinCyclosynchCompPhotons
__device__ __forceinline__ bool rebin_eligible(char type) { return type != 'N' && type != 'p' && type != 'i'; }   // :284,:453

// calculate_photon_position :246-270
__device__ __forceinline__ void rebin_position(const PhotonDev &ph, int i, int three, double &r, double &theta, double &phi)
{
    const double x = ph.r0[i], y = ph.r1[i], z = ph.r2[i];
    r = sqrt(x * x + y * y + z * z);
    phi = 0;
    if (r < 2.2250738585072014e-308) { theta = 0.0; return; }
    theta = acos(z / r);
    if (three) phi = fmod(atan2(y, x) * (180.0 / M_PI) + 360.0, 360.0);
}

// one thread's share of collect_photon_statistics (:273-322): the slots start, start + step, ... of the list
__device__ __forceinline__ RebinRange rebin_range_thread(const PhotonDev &ph, int three, int start, int step)
{
    RebinRange q;
    q.p0_min = 1.7976931348623157e308; q.p0_max = 0; q.theta_min = 1.7976931348623157e308; q.theta_max = 0;
    q.phi_min = 1.7976931348623157e308; q.phi_max = 0; q.valid = 0; q.synch = 0;
    for (int i = start; i < ph.n; i += step) {
        const char type = ph.type[i];
        if (rebin_eligible(type)) {
            const double p0 = ph.p0[i];
            if (p0 > 0) { q.p0_min = fmin(q.p0_min, p0); q.p0_max = fmax(q.p0_max, p0); q.valid += 1; }
            double r, theta, phi;
            rebin_position(ph, i, three, r, theta, phi);
            q.theta_min = fmin(q.theta_min, theta); q.theta_max = fmax(q.theta_max, theta);
            if (three) { q.phi_min = fmin(q.phi_min, phi); q.phi_max = fmax(q.phi_max, phi); }
        }
        if (type == 'p') q.synch += 1;
    }
    return q;
}
// ... combined over the workgroup's 256 threads (minima, maxima and integer sums: the order does not matter); thread 0 holds the result
__device__ __forceinline__ RebinRange rebin_range_block(RebinRange q, RebinRange (&s_p)[4])
{
    for (int off = 32; off > 0; off >>= 1) {
        q.p0_min = fmin(q.p0_min, __shfl_xor(q.p0_min, off)); q.p0_max = fmax(q.p0_max, __shfl_xor(q.p0_max, off));
        q.theta_min = fmin(q.theta_min, __shfl_xor(q.theta_min, off)); q.theta_max = fmax(q.theta_max, __shfl_xor(q.theta_max, off));
        q.phi_min = fmin(q.phi_min, __shfl_xor(q.phi_min, off)); q.phi_max = fmax(q.phi_max, __shfl_xor(q.phi_max, off));
        q.valid += __shfl_xor(q.valid, off); q.synch += __shfl_xor(q.synch, off);
    }
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            const RebinRange &o = s_p[w];
            q.p0_min = fmin(q.p0_min, o.p0_min); q.p0_max = fmax(q.p0_max, o.p0_max);
            q.theta_min = fmin(q.theta_min, o.theta_min); q.theta_max = fmax(q.theta_max, o.theta_max);
            q.phi_min = fmin(q.phi_min, o.phi_min); q.phi_max = fmax(q.phi_max, o.phi_max);
            q.valid += o.valid; q.synch += o.synch;
        }
    }
    return q;
}

__global__ __launch_bounds__(256) void rebin_range_kernel(PhotonDev ph, int three, RebinRange *__restrict__ partials)
{
    __shared__ RebinRange s_p[4];
    const RebinRange q = rebin_range_block(rebin_range_thread(ph, three, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256), s_p);
    if (threadIdx.x == 0) partials[blockIdx.x] = q;
}

// gsl_histogram2d_set_ranges_uniform's edges and gsl_histogram2d_find (see oracle/oracle_cyclosynch.c)
__device__ __forceinline__ double axis_edge(double lo, double hi, int n, int i)
{
    const double f1 = ((double)(n - i)) / (double)n, f2 = ((double)i) / (double)n;
    return f1 * lo + f2 * hi;
}
__device__ __forceinline__ int axis_find(double lo, double hi, int n, double x)
{
    if (!(x >= axis_edge(lo, hi, n, 0)) || !(x < axis_edge(lo, hi, n, n))) return -1;
    int a = 0, b = n;
    while (b - a > 1) {
        const int mid = (a + b) / 2;
        if (x >= axis_edge(lo, hi, n, mid)) a = mid; else b = mid;
    }
    return a;
}

// the bin of slot i (accumulate_bin_statistics :450-466): -1 not rebinned, -2 outside the histograms (the reference's exit(1))
__device__ __forceinline__ int rebin_assign_one(const PhotonDev &ph, const RebinAxes &ax, int i)
{
    if (!rebin_eligible(ph.type[i])) return -1;
    int bin;
    double r, theta, phi;
    rebin_position(ph, i, ax.three, r, theta, phi);
    const double le = log10(ph.p0[i]);
    int idx_x = axis_find(ax.e_lo, ax.e_hi, ax.num_bins, le), idx_y = axis_find(ax.t_lo, ax.t_hi, ax.num_bins_theta, theta), idx_z = 0;
    if (idx_x < 0 || idx_y < 0) { idx_x = 0; idx_y = 0; }                       // gsl_histogram2d_find: both untouched on a domain error
    if (ax.three) {
        const int pz = axis_find(ax.p_lo, ax.p_hi, ax.num_bins_phi, phi);
        const int ex = axis_find(ax.e_lo, ax.e_hi, ax.num_bins, le), ty = axis_find(ax.t_lo, ax.t_hi, ax.num_bins_theta, theta);
        if (ex >= 0 && pz >= 0) { idx_x = ex; idx_z = pz; }
        if (ty >= 0 && pz >= 0) { idx_y = ty; idx_z = pz; }
    }
    if (idx_x < 0 || idx_x >= ax.num_bins || idx_y < 0 || idx_y >= ax.num_bins_theta || (ax.three && (idx_z < 0 || idx_z >= ax.num_bins_phi))) bin = -2;
    else bin = ax.three ? idx_z * ax.num_bins * ax.num_bins_theta + idx_x * ax.num_bins_theta + idx_y : idx_x * ax.num_bins_theta + idx_y;
    if (bin >= ax.total_bins) bin = -2;
    return bin;
}

__global__ __launch_bounds__(256) void rebin_assign_kernel(PhotonDev ph, RebinAxes ax, int *__restrict__ bin_of, unsigned *__restrict__ bin_count)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ph.n) return;
    const int bin = rebin_assign_one(ph, ax, i);
    if (bin != -1) atomicAdd(bin_count + (bin >= 0 ? bin : ax.total_bins + 1), 1u);   // [total_bins + 1]: photons outside the histograms
    bin_of[i] = bin;
}

__global__ __launch_bounds__(256) void rebin_fill_kernel(PhotonDev ph, const int *__restrict__ bin_of, const int *__restrict__ bin_start,
                                                         unsigned *__restrict__ cursor, int *__restrict__ members)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ph.n) return;
    const int b = bin_of[i];
    if (b >= 0) members[bin_start[b] + (int)atomicAdd(cursor + b, 1u)] = i;
}

// create_rebinned_photons :504-607, with accumulate_bin_statistics' sums (:467-497) formed in slot order as the reference forms them: bin b's
// rebinned photon (valid == 0: the bin is empty)
__device__ __forceinline__ RebinRec rebin_create_one(const PhotonDev &ph, const RebinAxes &ax, const int *__restrict__ bin_start, int *__restrict__ members, int b)
{
    const int m0 = bin_start[b], m = bin_start[b + 1] - m0;
    for (int a = 1; a < m; ++a) {                       // the fill is unordered: sort the few members by slot
        const int key = members[m0 + a];
        int j = a - 1;
        while (j >= 0 && members[m0 + j] > key) { members[m0 + j + 1] = members[m0 + j]; --j; }
        members[m0 + j + 1] = key;
    }
    double w_r = 0, w_theta = 0, w_phi_offset = 0, w_s0 = 0, w_s1 = 0, w_s2 = 0, w_s3 = 0, w_scatt = 0, total_weight = 0;
    double w_phi_dir = 0, w_theta_dir = 0, w_energy = 0, w_phi_pos = 0;
    const double RAD_TO_DEG = 180.0 / M_PI, DEG_TO_RAD = M_PI / 180.0;
    for (int k = 0; k < m; ++k) {
        const int i = members[m0 + k];
        double r, theta, phi;
        rebin_position(ph, i, ax.three, r, theta, phi);
        const double w = ph.weight[i], p0 = ph.p0[i], p1 = ph.p1[i], p2 = ph.p2[i], p3 = ph.p3[i];
        w_r += r * w;
        w_theta += theta * w;
        w_phi_offset += (atan2(p2, p1) - atan2(ph.r1[i], ph.r0[i])) * RAD_TO_DEG * w;
        w_s0 += ph.s0[i] * w; w_s1 += ph.s1[i] * w; w_s2 += ph.s2[i] * w; w_s3 += ph.s3[i] * w;
        w_scatt += ph.num_scatt[i] * w;
        total_weight += w;
        const double phi_dir = fmod(atan2(p2, p1) * RAD_TO_DEG + 360.0, 360.0);
        const double theta_dir = acos(p3 / p0) * RAD_TO_DEG;
        w_phi_dir += phi_dir * w;
        w_theta_dir += theta_dir * w;
        w_energy += p0 * w;
        if (ax.three) w_phi_pos += phi * w;
    }
    RebinRec o;
    o.valid = 0; o.pad = 0;
    o.weight = o.p0 = o.p1 = o.p2 = o.p3 = o.r0 = o.r1 = o.r2 = o.s0 = o.s1 = o.s2 = o.s3 = o.num_scatt = 0;
    if (!(total_weight > 0)) return o;
    const double avg_energy = w_energy / total_weight, avg_phi_dir = w_phi_dir / total_weight, avg_theta_dir = w_theta_dir / total_weight;
    const double avg_r = w_r / total_weight, avg_theta_pos = w_theta / total_weight;
    o.valid = 1;
    o.weight = total_weight;
    o.p0 = avg_energy;
    o.p1 = avg_energy * sin(avg_theta_dir * DEG_TO_RAD) * cos(avg_phi_dir * DEG_TO_RAD);
    o.p2 = avg_energy * sin(avg_theta_dir * DEG_TO_RAD) * sin(avg_phi_dir * DEG_TO_RAD);
    o.p3 = avg_energy * cos(avg_theta_dir * DEG_TO_RAD);
    double pos_phi;
    if (ax.three) pos_phi = (w_phi_pos / total_weight) * DEG_TO_RAD;
    else pos_phi = (avg_phi_dir - w_phi_offset / total_weight) * DEG_TO_RAD;
    o.r0 = avg_r * sin(avg_theta_pos) * cos(pos_phi);
    o.r1 = avg_r * sin(avg_theta_pos) * sin(pos_phi);
    o.r2 = avg_r * cos(avg_theta_pos);
    o.s0 = w_s0 / total_weight; o.s1 = w_s1 / total_weight; o.s2 = w_s2 / total_weight; o.s3 = w_s3 / total_weight;
    o.num_scatt = (double)(int)(w_scatt / total_weight + 0.5);
    return o;
}

__global__ __launch_bounds__(256) void rebin_create_kernel(PhotonDev ph, RebinAxes ax, const int *__restrict__ bin_start, int *__restrict__ members,
                                                           RebinRec *__restrict__ recs, unsigned *__restrict__ empty_bins)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= ax.total_bins) return;
    const RebinRec o = rebin_create_one(ph, ax, bin_start, members, b);
    if (!o.valid) atomicAdd(empty_bins, 1u);
    recs[b] = o;
}

// record o into slot s (addToPhotonList, photons.c:190-199)
__device__ __forceinline__ void rebin_place_one(const PhotonDev &ph, const RebinRec &o, int s)
{
    ph.type[s] = 'k';
    ph.weight[s] = o.weight;
    ph.p0[s] = o.p0; ph.p1[s] = o.p1; ph.p2[s] = o.p2; ph.p3[s] = o.p3;
    ph.c0[s] = 0; ph.c1[s] = 0; ph.c2[s] = 0; ph.c3[s] = 0;
    ph.r0[s] = o.r0; ph.r1[s] = o.r1; ph.r2[s] = o.r2;
    ph.s0[s] = o.s0; ph.s1[s] = o.s1; ph.s2[s] = o.s2; ph.s3[s] = o.s3;
    ph.num_scatt[s] = o.num_scatt;
    ph.idx[s] = 0;
    ph.tau[s] = 0; ph.tau_next[s] = 0; ph.tts[s] = 0;    // calloc'ed in the reference
    double u0 = 0, u1 = 0, u2 = 0;
    if (o.p0 != 0) { const double d = 1.0 / o.p0; u0 = o.p1 * d * C_LIGHT; u1 = o.p2 * d * C_LIGHT; u2 = o.p3 * d * C_LIGHT; }
    ph.u0[s] = u0; ph.u1[s] = u1; ph.u2[s] = u2;
    ph.ntau[s] = -INFINITY;
    ph.flags[s] = (unsigned char)(FLAG_VALID | FLAG_RECALC | (o.weight != 0 ? FLAG_MOVES : 0u));
}

__global__ __launch_bounds__(256) void rebin_place_kernel(PhotonDev ph, const RebinRec *__restrict__ recs, int total_bins, const int *__restrict__ null_slots)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= total_bins) return;
    const RebinRec o = recs[b];
    if (!o.valid) return;                                // a null rebinned photon is not copied (photons.c:192)
    rebin_place_one(ph, o, null_slots[b]);
}

// setNullPhoton for a 'k' or 'c' photon (:573-581)
__device__ __forceinline__ void rebin_nullify_one(const PhotonDev &ph, int i)
{
    const char type = ph.type[i];
    if (type != 'c' && type != 'k') return;
    ph.type[i] = 'N';
    ph.weight[i] = 0;
    ph.idx[i] = -1;
    ph.flags[i] = (unsigned char)FLAG_VALID;
    ph.p0[i] = 0; ph.p1[i] = 0; ph.p2[i] = 0; ph.p3[i] = 0;
    ph.c0[i] = 0; ph.c1[i] = 0; ph.c2[i] = 0; ph.c3[i] = 0;
    ph.r0[i] = 0; ph.r1[i] = 0; ph.r2[i] = 0;
    ph.s0[i] = 0; ph.s1[i] = 0; ph.s2[i] = 0; ph.s3[i] = 0;
    ph.num_scatt[i] = 0;
    ph.tau[i] = 0; ph.tau_next[i] = 0;
    ph.u0[i] = 0; ph.u1[i] = 0; ph.u2[i] = 0;
    ph.ntau[i] = -INFINITY;
}

__global__ __launch_bounds__(256) void rebin_nullify_kernel(PhotonDev ph)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ph.n) return;
    rebin_nullify_one(ph, i);
}

// ---- rebinCyclosynchCompPhotons for MANY lists of a rank pool at once (round 3): one workgroup per list, every stage of the per-list kernels
// above inside one launch -- a parked list costs the frame no launches and no host round trips of its own.  Same per-slot and per-bin
// functions, so the lists come out bit for bit as from mcrat_hip_rebin_cyclosynch.  Two launches: the ranges (the host then fixes the
// histograms' axes exactly as it does for one list, mc_cyclosynch.c:324-391), then everything else.
__global__ __launch_bounds__(256) void rebin_pool_range_kernel(PhotonDev pool, int three, const RebinPoolList *__restrict__ lists, RebinRange *__restrict__ out)
{
    __shared__ RebinRange s_p[4];
    PhotonDev ph = pool;
    offset_photons(ph, (size_t)lists[blockIdx.x].first);
    ph.n = lists[blockIdx.x].n;
    const RebinRange q = rebin_range_block(rebin_range_thread(ph, three, threadIdx.x, 256), s_p);
    if (threadIdx.x == 0) out[blockIdx.x] = q;
}

// exclusive prefix sums over the workgroup's 256 threads; returns the thread's offset, *total the sum
__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned (&s_w)[4], unsigned *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
    for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(inc, off); if (lane >= off) inc += o; }
    __syncthreads();                                     // (s_w may still be read from the previous call)
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += s_w[w];
    *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + inc - v;
}

__global__ __launch_bounds__(256) void rebin_pool_kernel(PhotonDev pool, RebinPoolList *__restrict__ lists, char *__restrict__ scratch)
{
    __shared__ unsigned s_w[4];
    __shared__ unsigned s_empty;
    RebinPoolList &L = lists[blockIdx.x];
    PhotonDev ph = pool;
    offset_photons(ph, (size_t)L.first);
    ph.n = L.n;
    const RebinAxes ax = L.ax;
    const int n = L.n, B = ax.total_bins, tid = threadIdx.x;
    const int n_al = (n + 63) & ~63, b_al = (B + 2 + 63) & ~63;
    int *bin_of = reinterpret_cast<int *>(scratch + L.scratch), *members = bin_of + n_al, *null_slots = members + n_al;
    unsigned *bin_count = reinterpret_cast<unsigned *>(null_slots + n_al), *cursor = bin_count + b_al;      // bin_count[B + 1]: photons outside the histograms
    int *bin_start = reinterpret_cast<int *>(cursor + b_al);
    RebinRec *recs = reinterpret_cast<RebinRec *>(bin_start + b_al);
    if (tid == 0) s_empty = 0;
    for (int k = tid; k < B + 2; k += 256) { bin_count[k] = 0; cursor[k] = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {                                       // accumulate_bin_statistics :450-466
        const int bin = rebin_assign_one(ph, ax, i);
        if (bin != -1) atomicAdd(bin_count + (bin >= 0 ? bin : B + 1), 1u);
        bin_of[i] = bin;
    }
    __threadfence_block();
    __syncthreads();
    if (bin_count[B + 1] != 0) {                                               // the reference exits; the list is left as it was
        if (tid == 0) L.status = 1;
        return;
    }
    unsigned run = 0;                                                          // bin_start = exclusive prefix sums of bin_count
    for (int k0 = 0; k0 < B; k0 += 256) {
        const int k = k0 + tid;
        unsigned total;
        const unsigned off = block_exclusive_scan_256(k < B ? bin_count[k] : 0u, s_w, &total);
        if (k < B) bin_start[k] = (int)(run + off);
        run += total;
    }
    if (tid == 0) bin_start[B] = (int)run;
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int b = bin_of[i];
        if (b >= 0) members[bin_start[b] + (int)atomicAdd(cursor + b, 1u)] = i;
    }
    __threadfence_block();
    __syncthreads();
    for (int b = tid; b < B; b += 256) {                                       // create_rebinned_photons :504-607
        const RebinRec o = rebin_create_one(ph, ax, bin_start, members, b);
        if (!o.valid) atomicAdd(&s_empty, 1u);
        recs[b] = o;
    }
    __threadfence_block();
    __syncthreads();
    for (int i = tid; i < n; i += 256) rebin_nullify_one(ph, i);               // :573-581
    __threadfence_block();
    __syncthreads();
    run = 0;                                                                   // the null slots in ascending order (photons.c:181-189)
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const bool is_null = i < n && ph.type[i] == 'N';
        unsigned total;
        const unsigned off = block_exclusive_scan_256(is_null ? 1u : 0u, s_w, &total);
        if (is_null) null_slots[run + off] = i;
        run += total;
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) { L.n_null = (int)run; L.empty_bins = (int)s_empty; L.status = ((unsigned)B > run) ? 2 : 0; }
    if ((unsigned)B > run) return;                                             // "Adding to the photon list has failed": the reference exits
    for (int b = tid; b < B; b += 256) {                                       // addToPhotonList, photons.c:190-199
        const RebinRec o = recs[b];
        if (o.valid) rebin_place_one(ph, o, null_slots[b]);
    }
}

}  // namespace

int rebin_range_blocks(int n) { const int b = (n + 255) / 256; return b < 1 ? 1 : (b > 512 ? 512 : b); }

hipError_t launch_rebin_range(const PhotonDev &ph, int three, RebinRange *partials, hipStream_t stream)
{
    rebin_range_kernel<<<dim3(rebin_range_blocks(ph.n)), dim3(256), 0, stream>>>(ph, three, partials);
    return hipGetLastError();
}

hipError_t launch_rebin_assign(const PhotonDev &ph, const RebinAxes &ax, int *bin_of, unsigned *bin_count, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(bin_count, 0, sizeof(unsigned) * ((size_t)ax.total_bins + 2), stream);     // + empty bins, photons outside
    if (e != hipSuccess) return e;
    rebin_assign_kernel<<<dim3((ph.n + 255) / 256), dim3(256), 0, stream>>>(ph, ax, bin_of, bin_count);
    return hipGetLastError();
}

hipError_t launch_rebin_fill(const PhotonDev &ph, const int *bin_of, const int *bin_start, unsigned *cursor, int *members, hipStream_t stream)
{
    rebin_fill_kernel<<<dim3((ph.n + 255) / 256), dim3(256), 0, stream>>>(ph, bin_of, bin_start, cursor, members);
    return hipGetLastError();
}

hipError_t launch_rebin_create(const PhotonDev &ph, const RebinAxes &ax, const int *bin_start, int *members, RebinRec *recs,
                               unsigned *empty_bins, hipStream_t stream)
{
    rebin_create_kernel<<<dim3((ax.total_bins + 255) / 256), dim3(256), 0, stream>>>(ph, ax, bin_start, members, recs, empty_bins);
    return hipGetLastError();
}

hipError_t launch_rebin_place(const PhotonDev &ph, const RebinRec *recs, int total_bins, const int *null_slots, hipStream_t stream)
{
    rebin_place_kernel<<<dim3((total_bins + 255) / 256), dim3(256), 0, stream>>>(ph, recs, total_bins, null_slots);
    return hipGetLastError();
}

size_t rebin_pool_scratch_bytes(int n, int total_bins)
{
    const size_t n_al = ((size_t)n + 63) & ~(size_t)63, b_al = ((size_t)total_bins + 2 + 63) & ~(size_t)63;
    return (sizeof(int) * (3 * n_al + 3 * b_al) + sizeof(RebinRec) * (size_t)total_bins + 255) & ~(size_t)255;
}

hipError_t launch_rebin_pool_range(const PhotonDev &pool, int three, const RebinPoolList *lists, int n_lists, RebinRange *out, hipStream_t stream)
{
    if (n_lists <= 0) return hipSuccess;
    rebin_pool_range_kernel<<<dim3(n_lists), dim3(256), 0, stream>>>(pool, three, lists, out);
    return hipGetLastError();
}

hipError_t launch_rebin_pool(const PhotonDev &pool, RebinPoolList *lists, int n_lists, char *scratch, hipStream_t stream)
{
    if (n_lists <= 0) return hipSuccess;
    rebin_pool_kernel<<<dim3(n_lists), dim3(256), 0, stream>>>(pool, lists, scratch);
    return hipGetLastError();
}

hipError_t launch_rebin_nullify(const PhotonDev &ph, hipStream_t stream)
{
    rebin_nullify_kernel<<<dim3((ph.n + 255) / 256), dim3(256), 0, stream>>>(ph);
    return hipGetLastError();
}

hipError_t launch_cs_replace(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, RngKey key, LoopState *st, const PhotonDev &ph,
                             CsFrame *frame, int resume, hipStream_t stream)
{
    cs_replace_kernel<<<dim3(1), dim3(256), 0, stream>>>(p, hy, h, key, st, ph, frame, resume);
    return hipGetLastError();
}

hipError_t launch_inject_slab_flag(const InjectParams &p, const HydroDev &hy, unsigned *flag, unsigned long long *d_total, int *n_slab, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    inject_slab_flag_kernel<<<dim3((hy.M + 255) / 256), dim3(256), 0, stream>>>(p, hy, flag, d_total);
    unsigned long long total = 0;
    if ((e = hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    *n_slab = (int)total;
    return hipGetLastError();
}

hipError_t launch_inject_slab_write(const InjectParams &p, const HydroDev &hy, const unsigned *flag, int n_slab, int *start, int *scratch, InjectSlabCell *out,
                                    hipStream_t stream)
{
    hipError_t e = launch_exclusive_scan(flag, hy.M, start, scratch, (long long)n_slab, stream);
    if (e != hipSuccess) return e;
    inject_slab_write_kernel<<<dim3((hy.M + 255) / 256), dim3(256), 0, stream>>>(p, hy, flag, start, out);
    return hipGetLastError();
}

hipError_t launch_inject_pool(const InjectParams &p, const HydroDev &hy, const PhotonDev &pool, int stride, int n_ranks, const InjectSlabCell *slab, int n_slab,
                              PoolInject *lists, int group, hipStream_t stream)
{
    inject_pool_kernel<<<dim3(n_ranks), dim3(256), 0, stream>>>(p, hy, pool, stride, slab, n_slab, lists, group);
    return hipGetLastError();
}

hipError_t launch_cs_shell_flag(const CsEmitParams &p, const HydroDev &hy, unsigned *flag, unsigned long long *d_total, int *n_shell, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    cs_shell_flag_kernel<<<dim3((hy.M + 255) / 256), dim3(256), 0, stream>>>(p, hy, flag, d_total);
    unsigned long long total = 0;
    if ((e = hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    *n_shell = (int)total;
    return hipGetLastError();
}

hipError_t launch_cs_shell_write(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, const unsigned *flag, int n_shell, int *start, int *scratch,
                                 CsShellCell *out, unsigned *not_converged, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(not_converged, 0, sizeof(unsigned), stream);
    if (e != hipSuccess) return e;
    if ((e = launch_exclusive_scan(flag, hy.M, start, scratch, (long long)n_shell, stream)) != hipSuccess) return e;
    cs_shell_write_kernel<<<dim3((hy.M + 255) / 256), dim3(256), 0, stream>>>(p, hy, h, flag, start, out, not_converged);
    return hipGetLastError();
}

hipError_t launch_cs_emit_pool(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, const PhotonDev &pool, int stride, int n_ranks, RankDesc *desc,
                               const CsShellCell *shell, int n_shell, CsPoolEmit *lists, int group, hipStream_t stream)
{
    cs_emit_pool_kernel<<<dim3(n_ranks), dim3(256), 0, stream>>>(p, hy, h, pool, stride, desc, shell, n_shell, lists, group);
    return hipGetLastError();
}

hipError_t launch_cs_replace_pool(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, LoopState *states, const PhotonDev &pool, int stride,
                                  int n_ranks, RankDesc *desc, CsFrame *frames, hipStream_t stream)
{
    cs_replace_pool_kernel<<<dim3(n_ranks), dim3(256), 0, stream>>>(p, hy, h, states, pool, stride, desc, frames);
    return hipGetLastError();
}

hipError_t launch_cs_emit_count(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, double ph_weight_adjusted, unsigned long long attempt,
                                RngKey key, unsigned *count, unsigned long long *d_total, unsigned *d_flags, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    cs_emit_count_kernel<<<dim3((hy.M + 255) / 256), dim3(256), 0, stream>>>(p, hy, h, ph_weight_adjusted, attempt, key, count, d_total, d_flags);
    return hipGetLastError();
}

hipError_t launch_cs_emit_generate(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, double ph_weight_adjusted, RngKey key, const int *start,
                                   int n_emit, const int *null_slots, const PhotonDev &ph, hipStream_t stream)
{
    if (n_emit <= 0) return hipSuccess;
    cs_emit_generate_kernel<<<dim3((n_emit + 255) / 256), dim3(256), 0, stream>>>(p, hy, h, ph_weight_adjusted, key, start, n_emit, null_slots, ph);
    return hipGetLastError();
}

hipError_t launch_null_count(const PhotonDev &ph, unsigned *block_count, unsigned long long *d_total, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const int chunks = (ph.n + 255) / 256;
    null_count_kernel<<<dim3((chunks + NULL_CHUNKS - 1) / NULL_CHUNKS), dim3(256), 0, stream>>>(ph, block_count, d_total);
    return hipGetLastError();
}

hipError_t launch_null_write(const PhotonDev &ph, const int *block_start, int *null_slots, hipStream_t stream)
{
    null_write_kernel<<<dim3((ph.n + 255) / 256), dim3(256), 0, stream>>>(ph, block_start, null_slots);
    return hipGetLastError();
}

hipError_t launch_null_fill(const PhotonDev &ph, int first, int count, hipStream_t stream)
{
    if (count <= 0) return hipSuccess;
    null_fill_kernel<<<dim3((count + 255) / 256), dim3(256), 0, stream>>>(ph, first, count);
    return hipGetLastError();
}

hipError_t launch_inject_count(const InjectParams &p, const HydroDev &hy, double ph_weight_adjusted, unsigned long long attempt, RngKey key,
                               unsigned *count, unsigned long long *d_total, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    inject_count_kernel<<<dim3((hy.M + 255) / 256), dim3(256), 0, stream>>>(p, hy, ph_weight_adjusted, attempt, key, count, d_total);
    return hipGetLastError();
}

hipError_t launch_inject_generate(const InjectParams &p, const HydroDev &hy, double ph_weight_adjusted, RngKey key, const int *start,
                                  const PhotonDev &ph, hipStream_t stream)
{
    inject_generate_kernel<<<dim3((ph.n + 255) / 256), dim3(256), 0, stream>>>(p, hy, ph_weight_adjusted, key, start, ph);
    return hipGetLastError();
}

}  // namespace mcrat
