/* oracle_rng.c -- see oracle_rng.h.  TEST INFRASTRUCTURE ONLY. */
#include "oracle_rng.h"
#include <math.h>

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int round = 0; round < 10; round++) {
        uint64_t prod0 = (uint64_t)0xD2511F53u * c0;
        uint64_t prod1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(prod1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)prod1;
        uint32_t n2 = (uint32_t)(prod0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)prod0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint64_t orc_splitmix64_next(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

double orc_bits_to_uniform(uint64_t x)
{
    return (double)(x >> 11) * 0x1.0p-53;
}

double orc_bits_to_uniform_pos(uint64_t x)
{
    return ((double)(x >> 12) + 0.5) * 0x1.0p-52;
}

void orc_rng_init(orc_rng *r, uint64_t seed, uint32_t stream)
{
    r->seed = seed;
    r->iteration = 0;
    r->stream = stream;
    r->ev_state = 0;
    r->n_draws = 0;
    r->tape = 0;
    r->tape_n = r->tape_pos = 0;
    r->tape_error = 0;
}

void orc_rng_init_tape(orc_rng *r, const double *tape, int64_t n)
{
    orc_rng_init(r, 0, 0);
    r->tape = tape;
    r->tape_n = n;
}

static double tape_next(orc_rng *r)
{
    r->n_draws++;
    if (r->tape_pos >= r->tape_n) {          /* run out: flag it, go on with a filler that lets every rejection loop end */
        r->tape_error = 1;
        const double g = 0.6180339887498949 * (double)(++r->tape_pos - r->tape_n);
        return g - (double)(long long)g;
    }
    return r->tape[r->tape_pos++];
}

double orc_rng_freepath_draw(orc_rng *r, uint32_t slot)
{
    if (r->tape) return orc_rng_uniform_pos(r);
    return orc_rng_freepath_upos(r, slot);
}

void orc_rng_set_iteration(orc_rng *r, uint64_t k)
{
    r->iteration = k;
}

static void keyed_block(const orc_rng *r, uint32_t word2, uint32_t purpose, uint32_t out[4])
{
    uint32_t ctr[4], key[2];
    ctr[0] = (uint32_t)(r->iteration & 0xffffffffu);
    ctr[1] = (uint32_t)(r->iteration >> 32);
    ctr[2] = word2;
    ctr[3] = purpose | (r->stream << 8);
    key[0] = (uint32_t)(r->seed & 0xffffffffu);
    key[1] = (uint32_t)(r->seed >> 32);
    orc_philox4x32_10(ctr, key, out);
}

uint64_t orc_rng_freepath_bits(const orc_rng *r, uint32_t slot)
{
    uint32_t w[4];
    keyed_block(r, slot >> 1, ORC_PURPOSE_FREEPATH, w);
    uint32_t lo = w[2u * (slot & 1u)], hi = w[2u * (slot & 1u) + 1u];
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

double orc_rng_freepath_upos(const orc_rng *r, uint32_t slot)
{
    return orc_bits_to_uniform_pos(orc_rng_freepath_bits(r, slot));
}

void orc_rng_event_begin(orc_rng *r, uint32_t slot)
{
    uint32_t w[4];
    if (r->tape) return;                     /* one sequential stream: nothing to open */
    keyed_block(r, slot, ORC_PURPOSE_EVENT, w);
    r->ev_state = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    r->n_draws = 0;
}

void orc_rng_stream_begin(orc_rng *r, uint32_t word2, uint32_t purpose)
{
    uint32_t w[4];
    if (r->tape) return;
    keyed_block(r, word2, purpose, w);
    r->ev_state = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    r->n_draws = 0;
}

double orc_rng_uniform(orc_rng *r)
{
    if (r->tape) return tape_next(r);                       /* gsl_rng_uniform: [0,1) */
    r->n_draws++;
    return orc_bits_to_uniform(orc_splitmix64_next(&r->ev_state));
}

double orc_rng_uniform_pos(orc_rng *r)
{
    if (r->tape) {                                          /* gsl_rng_uniform_pos: redraw while 0 */
        double x;
        do { x = tape_next(r); } while (x == 0.0 && !r->tape_error);
        return x;
    }
    r->n_draws++;
    return orc_bits_to_uniform_pos(orc_splitmix64_next(&r->ev_state));
}

double orc_rng_gaussian(orc_rng *r, double sigma)
{
    double x, y, r2;
    do {
        x = -1.0 + 2.0 * orc_rng_uniform_pos(r);
        y = -1.0 + 2.0 * orc_rng_uniform_pos(r);
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    return sigma * y * sqrt(-2.0 * log(r2) / r2);
}
