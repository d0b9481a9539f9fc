// hot_table.hip -- createHotCrossSection on the device (Src/hot_x_section.c:82-133, 324-400): the table
// TAU_CALCULATION == TABLE interpolates.  Entry (i, j) is log10 of
//     0.5 * int_1^{1+12 theta} dgamma int_-1^1 dmu  f_MJ(gamma; theta) * sigma_KN(eps gamma (1 - mu beta)) / sigma_T * (1 - mu beta)
// for the photon energy eps = 10^(LOG_PH_E_MIN + i d_e) and the temperature theta = 10^(LOG_T_MIN + j d_t), both in units of
// m_e c^2 -- a plain Monte-Carlo integral of 500 000 samples per entry in the reference (gsl_monte_plain_integrate: volume
// times the sample mean of the integrand at uniformly drawn points): 9e9 integrand evaluations for the 221 x 81 table, a
// quarter of an hour on one host core, under a second here.  One workgroup per entry; thread s draws the samples
// s, s + 256, ... from its own keyed stream {iteration = entry, word2 = s, purpose = HOT_TABLE} (x0 then x1 per sample, in
// the reference's draw order), so the table does not depend on the launch shape and the oracle can reproduce it.
#include <hip/hip_runtime.h>
#include <math.h>
#include "device_types.hpp"
#include "launch.hpp"
#include "physics.hpp"
#include "rng.hpp"

namespace mcrat {

namespace {

constexpr uint32_t RNG_HOT_TABLE = 4u;

__global__ __launch_bounds__(HOT_TABLE_SUBSTREAMS) void hot_table_kernel(HotTableParams p, double *__restrict__ table)
{
    __shared__ double s_w[HOT_TABLE_SUBSTREAMS / 64];
    const int entry = blockIdx.x;
    const int i = entry / (p.n_t + 1), j = entry % (p.n_t + 1);
    const double dt = (p.log_t_max - p.log_t_min) / p.n_t, dph_e = (p.log_ph_e_max - p.log_ph_e_min) / p.n_ph_e;   // hot_x_section.c:85
    const double ph_comv = pow(10., p.log_ph_e_min + i * dph_e);
    const double theta = pow(10., p.log_t_min + j * dt);
    const double norm = phys::mj_normalisation(theta);

    const Philox4 b = keyed_block(p.seed, (unsigned long long)entry, (uint32_t)threadIdx.x, RNG_HOT_TABLE, 0u);
    EventStream rng;
    rng.state = (uint64_t)b.w[0] | ((uint64_t)b.w[1] << 32);
    // the integrand and the sample loop: physics.hpp (shared with the loop's look-ups off the table)
    double sum = phys::hot_substream_sum(ph_comv, theta, norm, rng, threadIdx.x, p.calls, HOT_TABLE_SUBSTREAMS);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        double total = 0;
        for (int w = 0; w < HOT_TABLE_SUBSTREAMS / 64; ++w) total += s_w[w];
        table[entry] = log10(phys::hot_integral_of_total(total, theta, p.calls));     // volume x mean, :355, :96
    }
}

}  // namespace

hipError_t launch_hot_table(const HotTableParams &p, double *table, hipStream_t stream)
{
    const int entries = (p.n_ph_e + 1) * (p.n_t + 1);
    hipLaunchKernelGGL(hot_table_kernel, dim3(entries), dim3(HOT_TABLE_SUBSTREAMS), 0, stream, p, table);
    return hipGetLastError();
}

}  // namespace mcrat
