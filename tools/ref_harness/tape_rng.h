/* tape_rng.h -- a GSL generator that wraps gsl_rng_ranlxs0 and records every double it hands out (see README.md). */
#ifndef MCRAT_REF_TAPE_RNG_H
#define MCRAT_REF_TAPE_RNG_H
#include <gsl/gsl_rng.h>
#include <stddef.h>

/* allocate with gsl_rng_alloc(tape_recorder_type()); MCRaT's functions take it as their gsl_rng* */
const gsl_rng_type *tape_recorder_type(void);
/* the doubles returned so far (gsl_rng_uniform, gsl_rng_uniform_pos and gsl_ran_gaussian all go through get_double) */
const double *tape_recorded(const gsl_rng *r, size_t *n);
void tape_recorder_free(gsl_rng *r);      /* the inner generator and the log; then gsl_rng_free(r) */
#endif
