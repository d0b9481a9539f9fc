/* tape_rng.c -- see tape_rng.h.  Not MCRaT code; uses only GSL's public generator interface (gsl_rng_type). */
#include "tape_rng.h"
#include <stdio.h>
#include <stdlib.h>

typedef struct {
    gsl_rng *inner;            /* gsl_rng_ranlxs0, as Src/mcrat.c:99-103 allocates */
    double *log;
    size_t n, cap;
} tape_state;

static void tape_set(void *vs, unsigned long int seed)
{
    tape_state *s = (tape_state *)vs;
    if (!s->inner) s->inner = gsl_rng_alloc(gsl_rng_ranlxs0);
    gsl_rng_set(s->inner, seed);                       /* (gsl_rng_alloc seeds with gsl_rng_default_seed through this, like MCRaT's own rng) */
}

static unsigned long int tape_get(void *vs)
{
    tape_state *s = (tape_state *)vs;
    return gsl_rng_get(s->inner);                      /* the per-frame reseed of mcrat.c:701 reads this; it is not a uniform, so not on the tape */
}

static double tape_get_double(void *vs)
{
    tape_state *s = (tape_state *)vs;
    const double x = s->inner->type->get_double(s->inner->state);
    if (s->n == s->cap) {
        s->cap = s->cap ? 2 * s->cap : (size_t)1 << 20;
        s->log = (double *)realloc(s->log, s->cap * sizeof(double));
        if (!s->log) { fprintf(stderr, "tape_rng: out of memory\n"); exit(2); }
    }
    s->log[s->n++] = x;
    return x;
}

static const gsl_rng_type tape_type = {"ranlxs0-recorded", 0x00ffffffUL, 0, sizeof(tape_state), &tape_set, &tape_get, &tape_get_double};

const gsl_rng_type *tape_recorder_type(void) { return &tape_type; }

const double *tape_recorded(const gsl_rng *r, size_t *n)
{
    const tape_state *s = (const tape_state *)r->state;
    *n = s->n;
    return s->log;
}

void tape_recorder_free(gsl_rng *r)
{
    tape_state *s = (tape_state *)r->state;
    if (s->inner) gsl_rng_free(s->inner);
    free(s->log);
    s->inner = NULL; s->log = NULL; s->n = s->cap = 0;
}
