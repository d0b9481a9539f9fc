// rng.hpp -- counter-based random source of the engine (device + host).
//
// The reference draws from one sequential gsl_rng_ranlxs0 stream
// (Src/mcrat.c:99-103); a sequential stream cannot be evaluated one lane per
// photon, so the engine defines its own keyed source with the same draw ORDER
// and the same open/closed interval conventions:
//
//   free path of photon slot i in loop iteration k   (mclib.c:675, uniform_pos)
//       Philox4x32-10( ctr = {k_lo, k_hi, i>>1, FREEPATH | stream<<8}, key = seed )
//       slot i takes words {2(i&1), 2(i&1)+1}: one Philox block serves two slots.
//   event draws of candidate slot i in iteration k   (electron.c:81,196,217,219,233;
//       mcrat_scattering.c:519,535-536,561,573-574)
//       SplitMix64 stream seeded with the first 64 bits of
//       Philox4x32-10( ctr = {k_lo, k_hi, i, EVENT | stream<<8}, key = seed ).
//
// Philox4x32-10: Salmon, Moraes, Dror, Shaw, SC'11.  SplitMix64: Steele, Lea, Flood 2014.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MC_HD __host__ __device__ __forceinline__
#else
#define MC_HD inline
#endif

namespace mcrat {

constexpr uint32_t RNG_FREEPATH = 0u;
constexpr uint32_t RNG_EVENT = 1u;

struct Philox4 {
    uint32_t w[4];
};

MC_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int round = 0; round < 10; ++round) {
        const uint64_t prod0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t prod1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(prod1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)prod1;
        const uint32_t n2 = (uint32_t)(prod0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)prod0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 r;
    r.w[0] = c0; r.w[1] = c1; r.w[2] = c2; r.w[3] = c3;
    return r;
}

MC_HD Philox4 keyed_block(uint64_t seed, uint64_t iteration, uint32_t word2, uint32_t purpose, uint32_t stream)
{
    return philox4x32_10((uint32_t)iteration, (uint32_t)(iteration >> 32), word2, purpose | (stream << 8),
                         (uint32_t)seed, (uint32_t)(seed >> 32));
}

// [0,1): 53 random bits
MC_HD double bits_to_uniform(uint64_t x) { return (double)(x >> 11) * 0x1.0p-53; }
// (0,1): 52 random bits, never 0 or 1
MC_HD double bits_to_uniform_pos(uint64_t x) { return ((double)(x >> 12) + 0.5) * 0x1.0p-52; }

struct EventStream {
    uint64_t state;
    MC_HD uint64_t next()
    {
        uint64_t z = (state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    MC_HD double uniform() { return bits_to_uniform(next()); }
    MC_HD double uniform_pos() { return bits_to_uniform_pos(next()); }
};

// The random stream as an INPUT (mcrat_hip_set_rng_tape; SURVEY.md section 8c, "tape"): a recorded sequence of the doubles MCRaT's generator
// returned (gsl_rng_type::get_double of ranlxs0, in [0,1)), consumed strictly in the reference's call order -- one gsl_rng_uniform_pos per located
// slot in ascending slot order (mclib.c:646-675; kernels.hip, tape_draw_kernel), then photonEvent's draws one after the other (electron.c:81,196,
// 217-233; mcrat_scattering.c:519-574) from the position the pass has reached.  gsl_rng_uniform_pos redraws while it gets 0 and gsl_ran_gaussian's
// polar method takes as many pairs as it needs: both fall out of reading the tape sequentially.  A tape that runs out raises *error.
struct TapeDev {
    const double *u;
    long long n;
    long long *cursor;           // the next unread entry (device word: the loop's kernels advance it in stream order)
    int *error;
};

struct TapeStream {
    const double *u;
    long long n, pos;
    int *error;
    MC_HD double next()
    {
        if (pos >= n) {                                // the tape has run out: say so, and go on with a low-discrepancy filler so that every
            *error = 1;                                // rejection loop still ends (the results are meaningless from here on)
            const double g = 0.6180339887498949 * (double)(++pos - n);
            return g - (double)(long long)g;
        }
        return u[pos++];
    }
    MC_HD double uniform() { return next(); }                                  // gsl_rng_uniform: [0,1)
    MC_HD double uniform_pos()                                                 // gsl_rng_uniform_pos: redraw while 0
    {
        double x = next();
        while (x == 0.0 && pos < n) x = next();
        return x;
    }
};

MC_HD EventStream event_stream(uint64_t seed, uint64_t iteration, uint32_t slot, uint32_t stream)
{
    const Philox4 b = keyed_block(seed, iteration, slot, RNG_EVENT, stream);
    EventStream s;
    s.state = (uint64_t)b.w[0] | ((uint64_t)b.w[1] << 32);
    return s;
}

}  // namespace mcrat
