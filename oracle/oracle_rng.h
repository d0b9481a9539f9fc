/*
 * oracle_rng.h -- random source of the CPU oracle.   TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  The product (mcrat_amd/, include/) never links it.
 *
 * Why not ranlxs0: the reference draws from one sequential gsl_rng_ranlxs0
 * stream (/root/reference/Src/mcrat.c:99-103).  GSL (version unpinned by the
 * reference, README.md:113) is absent from this image and a sequential
 * lagged-Fibonacci stream cannot be evaluated per lane on a GPU anyway.  The
 * random source is an INPUT of the hot path, so both sides of every parity
 * comparison use the generator defined here, in the reference's draw ORDER
 * (SURVEY.md section 8a, row R):
 *
 *   free-path draw of photon slot i in global iteration k
 *       (mclib.c:675, one gsl_rng_uniform_pos per slot with idx != -1):
 *       Philox4x32-10( ctr = {k_lo, k_hi, i>>1, PURPOSE_FREEPATH | stream<<8},
 *                      key = {seed_lo, seed_hi} ),
 *       slot i takes output words {2*(i&1), 2*(i&1)+1} as one 64-bit value.
 *   event draws of candidate photon slot i in iteration k
 *       (electron.c:81,196,217,219,233; mcrat_scattering.c:519,535-536,561,573-574):
 *       a SplitMix64 stream whose initial state is the first 64 bits of
 *       Philox4x32-10( ctr = {k_lo, k_hi, i, PURPOSE_EVENT | stream<<8}, key ),
 *       consumed sequentially in the reference's call order.
 *
 * uniform      in [0,1):  (x >> 11) * 2^-53
 * uniform_pos  in (0,1):  ((x >> 12) + 0.5) * 2^-52
 * gaussian: Marsaglia polar method, one value per call, as published for
 *           gsl_ran_gaussian (GSL randist/gauss.c), drawing uniform_pos pairs.
 *
 * Philox4x32-10: Salmon et al., SC'11 (Random123); SplitMix64: Steele, Lea,
 * Flood, OOPSLA'14 / Vigna's reference constants.  Known-answer vectors for
 * both are checked in tests/test_oracle_kat.py.
 */
#ifndef ORACLE_RNG_H
#define ORACLE_RNG_H

#include <stdint.h>

#define ORC_PURPOSE_FREEPATH 0u
#define ORC_PURPOSE_EVENT    1u
/* photonInjection (mclib.c:9-300): the Poisson photon count of hydro cell i in weight-adjustment attempt a is drawn from
 * the SplitMix64 stream keyed {ctr = {a_lo, a_hi, i, INJECT_COUNT | stream<<8}}, the draws of injected photon number k
 * (counted over the whole injection) from the stream keyed {ctr = {0, 0, k, INJECT_PHOTON | stream<<8}} */
#define ORC_PURPOSE_INJECT_COUNT  2u
#define ORC_PURPOSE_INJECT_PHOTON 3u

typedef struct orc_rng {
    uint64_t seed;       /* frame seed (reference: gsl_rng_get at mcrat.c:701)   */
    uint64_t iteration;  /* global while-loop iteration k (mcrat.c:761)           */
    uint32_t stream;     /* virtual-rank id (0 for a single photon list)          */
    uint64_t ev_state;   /* SplitMix64 state of the current event stream          */
    uint64_t n_draws;    /* draws consumed from the current event stream (stats)  */
    /* TAPE source (orc_rng_init_tape): the stream is an input -- a recorded sequence of the doubles MCRaT's generator returned
     * (gsl_rng_type::get_double of ranlxs0, in [0,1)), consumed strictly in the reference's call order: one gsl_rng_uniform_pos per
     * located slot in ascending slot order (mclib.c:646-675), then photonEvent's draws (electron.c:81,196,217-233;
     * mcrat_scattering.c:519-574).  gsl_rng_uniform_pos redraws while it gets 0 (GSL rng/gsl_rng.h) and gsl_ran_gaussian's polar method
     * takes as many pairs as it needs (GSL randist/gauss.c): both fall out of reading the tape sequentially.  A tape that runs out sets
     * tape_error (and a filler sequence lets the loops end).  tools/ref_harness records such tapes from the unmodified reference. */
    const double *tape;  /* NULL: the keyed source above */
    int64_t tape_n, tape_pos;
    int tape_error;
} orc_rng;

void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
uint64_t orc_splitmix64_next(uint64_t *state);

void   orc_rng_init(orc_rng *r, uint64_t seed, uint32_t stream);
void   orc_rng_init_tape(orc_rng *r, const double *tape, int64_t n);
/* the free-path draw of slot i in the loop's ascending pass over the slots: keyed by (iteration, slot), or the next of the tape */
double orc_rng_freepath_draw(orc_rng *r, uint32_t slot);
void   orc_rng_set_iteration(orc_rng *r, uint64_t k);

/* free-path uniform_pos of photon slot i in the current iteration */
double orc_rng_freepath_upos(const orc_rng *r, uint32_t slot);
/* raw 64-bit value behind the above (exposed for tests) */
uint64_t orc_rng_freepath_bits(const orc_rng *r, uint32_t slot);

/* open the event stream of candidate slot i in the current iteration */
void   orc_rng_event_begin(orc_rng *r, uint32_t slot);
/* open the stream {iteration field = r->iteration, word2, purpose}: the generalisation the injection uses */
void   orc_rng_stream_begin(orc_rng *r, uint32_t word2, uint32_t purpose);
double orc_rng_uniform(orc_rng *r);       /* [0,1) */
double orc_rng_uniform_pos(orc_rng *r);   /* (0,1) */
double orc_rng_gaussian(orc_rng *r, double sigma);

double orc_bits_to_uniform(uint64_t x);
double orc_bits_to_uniform_pos(uint64_t x);

#endif
