// functions.hip -- the device functions of physics.hpp evaluated one by one on arrays (mcrat_hip_eval_function): the entry the
// function-level parity tests use (tests/test_gpu_functions.py).  The loop kernels reach these functions only
// along trajectories; here every one of them meets its edge cases directly -- both sides of the Klein-Nishina seam at 1e-3, a fluid at
// rest, gamma = 100, photon directions along the flow or along z, both sides of the 1e7 K switch of the electron sampler,
// unpolarised light -- against the oracle's restatement of the reference function.  Not used by the loop.
#include <hip/hip_runtime.h>
#include "../../include/mcrat_hip.h"
#include "device_types.hpp"
#include "launch.hpp"
#include "physics.hpp"
#include "rng.hpp"

namespace mcrat {

namespace {

// the event stream of item i: what the oracle gets from orc_rng_init(seed, stream); orc_rng_set_iteration(i); orc_rng_event_begin(0)
__device__ __forceinline__ EventStream item_stream(uint64_t seed, uint32_t stream, int i) { return event_stream(seed, (uint64_t)i, 0u, stream); }

__device__ __forceinline__ double k2e_of(double temp)
{
    return temp >= 1e7 ? phys::bessel_k2_scaled((M_EL * C_LIGHT * C_LIGHT) / (K_B * temp)) : 0.0;     // k2e_kernel, kernels.hip
}

template <bool STOKES>
__global__ __launch_bounds__(64) void eval_kernel(int fn, int n, const double *__restrict__ in, double *__restrict__ out, uint64_t seed, uint32_t stream)
{
    // one wavefront per item for the wave-wide sampler (all 64 lanes the same item), one lane per item otherwise
    const bool wave_items = fn == MCRAT_HIP_FN_THERMAL_ELECTRON_WAVE;
    const int i = wave_items ? (int)blockIdx.x : (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= n) return;
    switch (fn) {
    case MCRAT_HIP_FN_KN_CROSS_SECTION:
        out[i] = phys::kn_cross_section(in[i]);
        break;
    case MCRAT_HIP_FN_LORENTZ_BOOST_PHOTON:
    case MCRAT_HIP_FN_LORENTZ_BOOST_ELECTRON: {
        const double *r = in + (size_t)7 * i;
        const double b[3] = {r[0], r[1], r[2]}, p[4] = {r[3], r[4], r[5], r[6]};
        double res[4];
        phys::lorentz_boost(b, p, res, fn == MCRAT_HIP_FN_LORENTZ_BOOST_PHOTON);
        for (int k = 0; k < 4; ++k) out[(size_t)4 * i + k] = res[k];
        break;
    }
    case MCRAT_HIP_FN_STOKES_ROTATION: {
        const double *r = in + (size_t)13 * i;
        const double v[3] = {r[0], r[1], r[2]}, k1[3] = {r[3], r[4], r[5]}, k2[3] = {r[6], r[7], r[8]};
        double s[4] = {r[9], r[10], r[11], r[12]};
        phys::stokes_rotation(v, k1, k2, s);
        for (int k = 0; k < 4; ++k) out[(size_t)4 * i + k] = s[k];
        break;
    }
    case MCRAT_HIP_FN_THERMAL_ELECTRON:
    case MCRAT_HIP_FN_THERMAL_ELECTRON_WAVE: {
        const double *r = in + (size_t)5 * i;
        const double ph[4] = {r[1], r[2], r[3], r[4]};
        EventStream rng = item_stream(seed, stream, i);
        double el[4];
        if (wave_items) phys::single_thermal_electron<true>(el, r[0], k2e_of(r[0]), ph, rng);
        else phys::single_thermal_electron<false>(el, r[0], k2e_of(r[0]), ph, rng);
        if (!wave_items || threadIdx.x == 0)
            for (int k = 0; k < 4; ++k) out[(size_t)4 * i + k] = el[k];
        break;
    }
    case MCRAT_HIP_FN_ELECTRON_AND_SCATTER: {
        const double *r = in + (size_t)9 * i;
        double ph[4] = {r[1], r[2], r[3], r[4]}, s[4] = {r[5], r[6], r[7], r[8]};
        EventStream rng = item_stream(seed, stream, i);
        double el[4];
        phys::single_thermal_electron<false>(el, r[0], k2e_of(r[0]), ph, rng);
        const bool ok = phys::single_scatter<STOKES>(el, ph, s, rng);
        double *o = out + (size_t)13 * i;
        for (int k = 0; k < 4; ++k) { o[k] = el[k]; o[4 + k] = ph[k]; o[8 + k] = s[k]; }
        o[12] = ok ? 1.0 : 0.0;
        break;
    }
    default:
        break;
    }
}

}  // namespace

hipError_t launch_eval_function(int fn, int stokes, int n, const double *in, double *out, uint64_t seed, uint32_t stream_id, hipStream_t stream)
{
    const int blocks = (fn == MCRAT_HIP_FN_THERMAL_ELECTRON_WAVE) ? n : (n + 63) / 64;
    if (stokes) eval_kernel<true><<<dim3(blocks), dim3(64), 0, stream>>>(fn, n, in, out, seed, stream_id);
    else eval_kernel<false><<<dim3(blocks), dim3(64), 0, stream>>>(fn, n, in, out, seed, stream_id);
    return hipGetLastError();
}

}  // namespace mcrat
