/* harness_funcs.c -- calls MCRaT's OWN functions one by one on the inputs of functions.in and records what they return (README.md): the function-level
 * vectors G1-G7 of SURVEY.md section 8c, for the (DIMENSIONS, GEOMETRY, STOKES_SWITCH) this harness was compiled with.
 *
 * Compiled by the maintainer against an unmodified MCRaT checkout (Makefile).  This file holds no MCRaT code; it calls, with the signatures of MCRaT's
 * headers: kleinNishinaCrossSection (Src/mcrat_scattering.h:22), lorentzBoost (Src/mclib.h:4), findXY / findPhi / mullerMatrixRotation / stokesRotation
 * (Src/mcrat_scattering.h:8-14), singleThermalElectron (Src/electron.h:7), singleScatter (Src/mcrat_scattering.h:16), kleinNishinaScatter (:20),
 * mcratCoordinateToHydroCoordinate (Src/geometry.h:9).  The calls that draw random numbers get the recording generator (tape_rng.c), and the position of
 * the tape before every case is written out, so that the case can be replayed from exactly the doubles MCRaT consumed.
 *
 * functions.in (little-endian, written by make_inputs.py):
 *   int32 magic 0x4D435246, n_kn, n;  double kn_eps[n_kn];  double boost_beta[n][3], boost_p[n][4];  double stokes_v[n][3], stokes_k[n][3], stokes_kb[n][3],
 *   stokes_in[n][4];  double xy_v[n][3], xy_ref[n][3], muller_theta[n];  double scatter_temp[n], scatter_ph_in[n][4], scatter_stokes_in[n][4];
 *   double kns_p0[n], kns_q[n], kns_u[n];  double coord_xyz[n][3]
 * functions_<case>.out:
 *   int32 magic, DIMENSIONS, GEOMETRY, STOKES_SWITCH == ON, n_kn, n;  double kn_sigma[n_kn];  double boost_out_photon[n][4], boost_out_electron[n][4];
 *   double stokes_out[n][4];  double xy_x[n][3], xy_y[n][3], xy_phi[n], muller_out[n][4];
 *   double scatter_electron[n][4], scatter_ph_out[n][4], scatter_stokes_out[n][4];  int32 scatter_occurred[n];  int64 scatter_tape_at[n + 1];
 *   double kns_theta[n], kns_phi[n];  int32 kns_ok[n];  int64 kns_tape_at[n + 1];  double coord_out[n][3];  int64 tape_n;  double tape[tape_n]
 */
#include "mcrat.h"
#include "tape_rng.h"

#define FUNCS_MAGIC 0x4D435246

static void die(const char *what) { fprintf(stderr, "harness_funcs: %s\n", what); exit(1); }
static void rd(void *p, size_t sz, size_t n, FILE *f) { if (fread(p, sz, n, f) != n) die("short read"); }
static void wr(const void *p, size_t sz, size_t n, FILE *f) { if (fwrite(p, sz, n, f) != n) die("short write"); }
static double *get(size_t n, FILE *f) { double *a = (double *)malloc(sizeof(double) * (n ? n : 1)); if (!a) die("out of memory"); rd(a, sizeof(double), n, f); return a; }
static double *room(size_t n) { double *a = (double *)calloc(n ? n : 1, sizeof(double)); if (!a) die("out of memory"); return a; }

int main(int argc, char **argv)
{
    if (argc != 3) die("usage: harness_funcs functions.in functions_<case>.out");
    MPI_Init(&argc, &argv);
    FILE *in = fopen(argv[1], "rb");
    if (!in) die("cannot open the input");
    int h[3];
    rd(h, sizeof(int), 3, in);
    if (h[0] != FUNCS_MAGIC) die("not a functions.in");
    const size_t nk = (size_t)h[1], n = (size_t)h[2];
    double *kn_eps = get(nk, in);
    double *boost_beta = get(3 * n, in), *boost_p = get(4 * n, in);
    double *st_v = get(3 * n, in), *st_k = get(3 * n, in), *st_kb = get(3 * n, in), *st_in = get(4 * n, in);
    double *xy_v = get(3 * n, in), *xy_ref = get(3 * n, in), *mu_theta = get(n, in);
    double *sc_temp = get(n, in), *sc_ph = get(4 * n, in), *sc_s = get(4 * n, in);
    double *kns_p0 = get(n, in), *kns_q = get(n, in), *kns_u = get(n, in);
    double *xyz = get(3 * n, in);
    fclose(in);
    FILE *fPtr = fopen("/dev/null", "w");
    gsl_rng *rng = gsl_rng_alloc(tape_recorder_type());

    double *kn_sigma = room(nk);
    for (size_t i = 0; i < nk; ++i) kn_sigma[i] = kleinNishinaCrossSection(kn_eps[i]);                                  /* G1 */
    double *bo_p = room(4 * n), *bo_e = room(4 * n);
    for (size_t i = 0; i < n; ++i) {                                                                                      /* G2 */
        lorentzBoost(boost_beta + 3 * i, boost_p + 4 * i, bo_p + 4 * i, 'p', fPtr);
        lorentzBoost(boost_beta + 3 * i, boost_p + 4 * i, bo_e + 4 * i, 'e', fPtr);
    }
    double *st_out = room(4 * n), *xy_x = room(3 * n), *xy_y = room(3 * n), *xy_phi = room(n), *mu_out = room(4 * n);
    for (size_t i = 0; i < n; ++i) {                                                                                      /* G3 */
        memcpy(st_out + 4 * i, st_in + 4 * i, 4 * sizeof(double));
        stokesRotation(st_v + 3 * i, st_k + 3 * i, st_kb + 3 * i, st_out + 4 * i, fPtr);
        double x2[3], y2[3];
        findXY(xy_v + 3 * i, xy_ref + 3 * i, xy_x + 3 * i, xy_y + 3 * i);
        findXY(xy_v + 3 * i, st_v + 3 * i, x2, y2);
        xy_phi[i] = findPhi(xy_x + 3 * i, xy_y + 3 * i, x2, y2);
        memcpy(mu_out + 4 * i, st_in + 4 * i, 4 * sizeof(double));
        mullerMatrixRotation(mu_theta[i], mu_out + 4 * i, fPtr);
    }
    double *sc_el = room(4 * n), *sc_out = room(4 * n), *sc_sout = room(4 * n);
    int *sc_ok = (int *)calloc(n ? n : 1, sizeof(int));
    long long *sc_at = (long long *)calloc(n + 1, sizeof(long long));
    size_t at = 0;
    for (size_t i = 0; i < n; ++i) {                                                                                      /* G4 + G6 */
        (void)tape_recorded(rng, &at);
        sc_at[i] = (long long)at;
        memcpy(sc_out + 4 * i, sc_ph + 4 * i, 4 * sizeof(double));
        memcpy(sc_sout + 4 * i, sc_s + 4 * i, 4 * sizeof(double));
        singleThermalElectron(sc_el + 4 * i, sc_temp[i], sc_out + 4 * i, rng, fPtr);
        double el[4];
        memcpy(el, sc_el + 4 * i, sizeof el);
        sc_ok[i] = singleScatter(el, sc_out + 4 * i, sc_sout + 4 * i, rng, fPtr);
    }
    (void)tape_recorded(rng, &at);
    sc_at[n] = (long long)at;
    double *kns_theta = room(n), *kns_phi = room(n);
    int *kns_ok = (int *)calloc(n ? n : 1, sizeof(int));
    long long *kns_at = (long long *)calloc(n + 1, sizeof(long long));
    for (size_t i = 0; i < n; ++i) {                                                                                      /* G5 */
        (void)tape_recorded(rng, &at);
        kns_at[i] = (long long)at;
        kns_ok[i] = kleinNishinaScatter(&kns_theta[i], &kns_phi[i], kns_p0[i], kns_q[i], kns_u[i], rng, fPtr);
    }
    (void)tape_recorded(rng, &at);
    kns_at[n] = (long long)at;
    double *coord = room(3 * n);
    for (size_t i = 0; i < n; ++i) mcratCoordinateToHydroCoordinate(coord + 3 * i, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);   /* G7 */

    FILE *out = fopen(argv[2], "wb");
    if (!out) die("cannot open the output");
    const int oh[6] = {FUNCS_MAGIC, DIMENSIONS, GEOMETRY, STOKES_SWITCH == ON, (int)nk, (int)n};
    wr(oh, sizeof(int), 6, out);
    wr(kn_sigma, sizeof(double), nk, out);
    wr(bo_p, sizeof(double), 4 * n, out); wr(bo_e, sizeof(double), 4 * n, out);
    wr(st_out, sizeof(double), 4 * n, out);
    wr(xy_x, sizeof(double), 3 * n, out); wr(xy_y, sizeof(double), 3 * n, out); wr(xy_phi, sizeof(double), n, out); wr(mu_out, sizeof(double), 4 * n, out);
    wr(sc_el, sizeof(double), 4 * n, out); wr(sc_out, sizeof(double), 4 * n, out); wr(sc_sout, sizeof(double), 4 * n, out);
    wr(sc_ok, sizeof(int), n, out); wr(sc_at, sizeof(long long), n + 1, out);
    wr(kns_theta, sizeof(double), n, out); wr(kns_phi, sizeof(double), n, out); wr(kns_ok, sizeof(int), n, out); wr(kns_at, sizeof(long long), n + 1, out);
    wr(coord, sizeof(double), 3 * n, out);
    size_t n_tape = 0;
    const double *tape = tape_recorded(rng, &n_tape);
    const long long nt = (long long)n_tape;
    wr(&nt, sizeof nt, 1, out);
    wr(tape, sizeof(double), n_tape, out);
    fclose(out);
    fprintf(stderr, "harness_funcs: %zu + %zu cases, %lld uniforms recorded\n", nk, n, nt);
    tape_recorder_free(rng);
    gsl_rng_free(rng);
    MPI_Finalize();
    return 0;
}
