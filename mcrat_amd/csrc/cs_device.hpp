// cs_device.hpp -- the device functions of the cyclo-synchrotron hook of mcrat.c:786-808 (conversion of a scattered pool photon, its replacement
// by photonEmitCyclosynch with inject_single_switch = 1, the rebinning trigger), shared by inject.hip's hook kernels and -- for the lists of a
// rank pool, which run the hook without leaving their loop -- by rank_loop_kernel (kernels.hip).
#pragma once
#include <limits.h>
#include "device_types.hpp"
#include "launch.hpp"
#include "physics.hpp"
#include "rng.hpp"

namespace mcrat {
namespace {

constexpr uint32_t RNG_CS_COUNT = 5u;
constexpr uint32_t RNG_CS_PHOTON = 6u;
constexpr uint32_t RNG_CS_SINGLE = 7u;
constexpr double CHARGE_EL = 4.8032068e-10;      // Src/mclib.c:4-5

__device__ __forceinline__ EventStream keyed_stream(const RngKey &key, unsigned long long iteration, uint32_t word2, uint32_t purpose)
{
    const Philox4 b = keyed_block(key.seed, iteration, word2, purpose, key.stream);
    EventStream s;
    s.state = (uint64_t)b.w[0] | ((uint64_t)b.w[1] << 32);
    return s;
}

// geometry.c:108-156
__device__ __forceinline__ void hydro_to_mcrat(int dims, int geom, double r0, double r1, double r2, double out[3])
{
    double x = 0, y = 0, z = 0;
    if (dims == DIM_TWO || dims == DIM_TWO_POINT_FIVE) {
        if (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL) { x = r0 * cos(r2); y = r0 * sin(r2); z = r1; }
        if (geom == GEOM_SPHERICAL) { x = r0 * sin(r1) * cos(r2); y = r0 * sin(r1) * sin(r2); z = r0 * cos(r1); }
    } else {
        if (geom == GEOM_CARTESIAN) { x = r0; y = r1; z = r2; }
        if (geom == GEOM_SPHERICAL) { x = r0 * sin(r1) * cos(r2); y = r0 * sin(r1) * sin(r2); z = r0 * cos(r1); }
        if (geom == GEOM_POLAR) { x = r0 * cos(r1); y = r0 * sin(r1); z = r2; }
    }
    out[0] = x; out[1] = y; out[2] = z;
}

struct CellRec {
    double c0, c1, c2, s0, s1, s2;
};

__device__ __forceinline__ CellRec load_cell(const HydroDev &hy, int dims, int i)
{
    const CellGeom g = hy.geom[i];
    CellRec c;
    c.c0 = g.c0; c.c1 = g.c1; c.s0 = g.s0; c.s1 = g.s1; c.c2 = 0; c.s2 = 0;
    if (dims == DIM_THREE) { const CellGeom2 g2 = hy.geom2[i]; c.c2 = g2.c2; c.s2 = g2.s2; }
    return c;
}

// getMagneticFieldMagnitude + calcCyclotronFreq, mc_cyclosynch.c:30-33,54-92
__device__ __forceinline__ double cs_nu_c(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, int i)
{
    double b_field;
    if (p.b_field_calc == 0 || p.b_field_calc == 1) {
        const double el_dens = h.dens[i] / M_P, T = hy.temp[i];
        if (p.b_field_calc == 0) b_field = sqrt(p.epsilon_b * 8 * M_PI * 3 * el_dens * K_B * T / 2);
        else b_field = sqrt(8 * M_PI * p.epsilon_b * (el_dens * M_P * C_LIGHT * C_LIGHT + 4 * A_RAD * T * T * T * T / 3));
    } else if (p.dimensions == DIM_TWO) {
        b_field = sqrt(h.B0[i] * h.B0[i] + h.B1[i] * h.B1[i]);
    } else {
        b_field = sqrt(h.B0[i] * h.B0[i] + h.B1[i] * h.B1[i] + h.B2[i] * h.B2[i]);
    }
    return CHARGE_EL * b_field / (2 * M_PI * M_EL * C_LIGHT);
}

// one cyclo-synchrotron photon at the centre of cell i with the cell's cyclotron frequency into slot s (:1380-1440); the direction
// takes three uniform draws (two in 3-D).  Returns the azimuth drawn for the position.
__device__ double cs_emit_one(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, int i, double weight, int block_index, EventStream &rng,
                              const PhotonDev &ph, int s)
{
    const CellRec c = load_cell(hy, p.dimensions, i);
    const double fr_dum = cs_nu_c(p, hy, h, i);
    double position_phi = 0;
    if (p.dimensions != DIM_THREE) position_phi = rng.uniform() * 2 * M_PI;
    const double com_v_phi = rng.uniform() * 2 * M_PI;
    const double com_v_theta = rng.uniform() * M_PI;                           // uniform in the angle, as the reference has it (:1388)
    double p_comv[4];
    p_comv[0] = PL_CONST * fr_dum / C_LIGHT;
    p_comv[1] = (PL_CONST * fr_dum / C_LIGHT) * sin(com_v_theta) * cos(com_v_phi);
    p_comv[2] = (PL_CONST * fr_dum / C_LIGHT) * sin(com_v_theta) * sin(com_v_phi);
    p_comv[3] = (PL_CONST * fr_dum / C_LIGHT) * cos(com_v_theta);
    const CellFluid f = hy.fluid[i];
    const double fcv = hy.fluid_c ? hy.fluid_c[i] : 0.0;
    const double cphi = cos(position_phi), sphi = sin(position_phi);
    double boost[3];
    if (p.dimensions == DIM_TWO) phys::beta_from_record<DIM_TWO>(f.a, f.b, fcv, cphi, sphi, boost);
    else if (p.dimensions == DIM_TWO_POINT_FIVE) phys::beta_from_record<DIM_TWO_POINT_FIVE>(f.a, f.b, fcv, cphi, sphi, boost);
    else phys::beta_from_record<DIM_THREE>(f.a, f.b, fcv, cphi, sphi, boost);
    boost[0] *= -1; boost[1] *= -1; boost[2] *= -1;
    double l_boost[4];
    phys::lorentz_boost(boost, p_comv, l_boost, true);
    double xyz[3];
    if (p.dimensions == DIM_THREE) hydro_to_mcrat(p.dimensions, p.geometry, c.c0, c.c1, c.c2, xyz);
    else hydro_to_mcrat(p.dimensions, p.geometry, c.c0, c.c1, position_phi, xyz);
    ph.r0[s] = xyz[0]; ph.r1[s] = xyz[1]; ph.r2[s] = xyz[2];
    ph.p0[s] = l_boost[0]; ph.p1[s] = l_boost[1]; ph.p2[s] = l_boost[2]; ph.p3[s] = l_boost[3];
    ph.c0[s] = p_comv[0]; ph.c1[s] = p_comv[1]; ph.c2[s] = p_comv[2]; ph.c3[s] = p_comv[3];
    ph.s0[s] = 1; ph.s1[s] = 0; ph.s2[s] = 0; ph.s3[s] = 0;
    ph.num_scatt[s] = 0;
    ph.weight[s] = weight;
    ph.tau[s] = 0; ph.tts[s] = 0; ph.tau_next[s] = 0;
    double u0 = 0, u1 = 0, u2 = 0;
    if (l_boost[0] != 0) {
        const double d = 1.0 / l_boost[0];
        u0 = l_boost[1] * d * C_LIGHT; u1 = l_boost[2] * d * C_LIGHT; u2 = l_boost[3] * d * C_LIGHT;
    }
    ph.u0[s] = u0; ph.u1[s] = u1; ph.u2[s] = u2;
    ph.ntau[s] = -INFINITY;
    ph.idx[s] = block_index;
    ph.flags[s] = (unsigned char)(FLAG_VALID | FLAG_RECALC);                    // a pool photon does not move (mclib.c:1070)
    ph.type[s] = 'p';
    return position_phi;
}

// (body shared by the single-list hook and the rank pool's; all 256 threads call it.  grow_cap: the list may double in place up to
// this many slots -- the pool's slots per rank -- instead of parking for the host; *len_out receives the new length.)
// BLOCK: the threads of the calling workgroup (s_min: BLOCK / 64 ints of LDS); idx_shift: what *st's last_scattered_index is ahead of the
// list's own slot numbers (rank_loop_kernel keeps it in the pool's numbering while a list runs).
template <int BLOCK = 256>
__device__ __forceinline__ void cs_hook_body(const CsEmitParams &p, const HydroDev &hy, const HydroCols &h, const RngKey &key, LoopState *st, PhotonDev ph,
                                             CsFrame *cf, int resume, int *s_min, int grow_cap, int *len_out, int idx_shift = 0)
{
    const int tid = threadIdx.x;
    const unsigned long long it = st->iteration;
    if (!resume && ((cf->halt != 0 && cf->halt != CS_HALT_HOOK) || it == cf->last_iteration)) return;   // parked for the host, or a queued pass that did nothing
    const int sidx = st->last_scattered_index - idx_shift;
    const int called = st->photon_event_called;
    const bool fire = called && sidx >= 0 && sidx < ph.n && ph.type[sidx] == 'p';
    int slot = INT_MAX;
    if (fire) {
        int mine = INT_MAX;
        for (int i = tid; i < ph.n; i += BLOCK)
            if (ph.type[i] == 'N') { mine = i; break; }
        for (int off = 32; off > 0; off >>= 1) mine = min(mine, __shfl_xor(mine, off));
        if ((tid & 63) == 0) s_min[tid >> 6] = mine;
        __syncthreads();
        slot = s_min[0];
#pragma unroll
        for (int wv = 1; wv < BLOCK / 64; ++wv) slot = min(slot, s_min[wv]);
        if (slot == INT_MAX && 2 * ph.n <= grow_cap) {                          // photons.c:112-121 inside the pool's window: the list doubles
            const int old_n = ph.n;
            for (int i = old_n + tid; i < 2 * old_n; i += BLOCK) {                // (reallocatePhotonListMemory, photons.c:72-78; the columns are zero)
                ph.type[i] = 'N'; ph.idx[i] = -1; ph.flags[i] = (unsigned char)FLAG_VALID; ph.ntau[i] = -INFINITY;
            }
            ph.n = 2 * old_n;
            slot = old_n;
            if (tid == 0 && len_out) *len_out = ph.n;
            __threadfence_block();
            __syncthreads();
        }
    }
    if (tid != 0) return;
    if (fire) {
        if (slot == INT_MAX) {                                                  // photons.c:112-121: the host doubles the list
            cf->halt = CS_HALT_GROW;
            if (st->done != LOOP_CS_HALT) { cf->saved_done = st->done; st->done = LOOP_CS_HALT; }
            return;
        }
        const int i = ph.idx[sidx];
        const double weight = ph.weight[sidx];
        ph.type[sidx] = 'k';                                                    // mcrat.c:789
        EventStream rng = keyed_stream(key, it - 1ull, (uint32_t)sidx, RNG_CS_SINGLE);
        const double position_phi = cs_emit_one(p, hy, h, i, weight, i, rng, ph, slot);
        const CellRec c = load_cell(hy, p.dimensions, i);
        const double position_rand = rng.uniform_pos() * c.s0 - c.s0 / 2.0;
        const double position2_rand = rng.uniform_pos() * c.s1 - c.s1 / 2.0;
        double xyz[3];
        if (p.dimensions == DIM_THREE) {
            const double position3_rand = rng.uniform_pos() * c.s2 - c.s2 / 2.0;
            hydro_to_mcrat(p.dimensions, p.geometry, c.c0 + position_rand, c.c1 + position2_rand, c.c2 + position3_rand, xyz);
        } else {
            hydro_to_mcrat(p.dimensions, p.geometry, c.c0 + position_rand, c.c1 + position2_rand, position_phi, xyz);
        }
        ph.r0[sidx] = xyz[0]; ph.r1[sidx] = xyz[1]; ph.r2[sidx] = xyz[2];
        // it moves from now on (mclib.c:1070); a tau stored at scatter time belonged to the old azimuth
        unsigned f = ph.flags[sidx] & ~FLAG_TAU_FRESH;
        if (weight != 0) f |= FLAG_MOVES;
        ph.flags[sidx] = (unsigned char)f;
        cf->n_comptonized += weight;                                            // mcrat.c:788-794
        cf->emitted += 1;
        cf->scatt_num += 1;
    }
    cf->last_iteration = it;
    const long long fsc = st->frame_scatt_cnt;
    if (called && (fsc % 1000 == 0) && fsc != 0 && cf->scatt_num > cf->max_photons) {       // mcrat.c:797-808
        cf->halt = CS_HALT_REBIN;
        if (st->done != LOOP_CS_HALT) { cf->saved_done = st->done; st->done = LOOP_CS_HALT; }
    } else if (cf->halt == CS_HALT_HOOK) {                                      // the pool's loop parked for this hook: let it go on
        cf->halt = 0;
        st->done = cf->saved_done;
    }
}

}  // namespace
}  // namespace mcrat
