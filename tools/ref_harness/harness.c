/* harness.c -- drives MCRaT's OWN photon-loop functions on inputs written by make_inputs.py and records what they did (see README.md).
 *
 * Compiled by the maintainer against an unmodified MCRaT checkout (Makefile: links to $REF/Src, generated mcrat_input.h).  This file holds no MCRaT
 * code: it fills MCRaT's structs (Src/mcrat.h:142-244), calls MCRaT's functions with the signatures of Src/mclib.h:8-29 in the order main() calls
 * them inside `while (remaining_time > 0)` (Src/mcrat.c:761-851), and writes out the photons, the clock, the counters and the random stream.
 *
 * <case>.in  (little-endian, written by make_inputs.py):
 *   int32  magic 0x4D435248, dims, geometry, stokes, M, N, passes;  double fps, time_now, remaining_time, r0_domain[2], r1_domain[2], r2_domain[2]
 *   double[M] x 16: r0 r1 r2 r0_size r1_size r2_size r theta v0 v1 v2 dens dens_lab pres temp gamma
 *   per photon: the 19 doubles p0 p1 p2 p3 comv_p0..3 r0 r1 r2 s0 s1 s2 s3 num_scatt weight time_to_scatter total_optical_depth,
 *               then int32 nearest_block_index, recalc_properties, type
 * <case>.out:
 *   int32  magic, N, passes_done, frame_scatt_cnt, num_photons_find_new_element, last ph_scatt_index;  double time_now, remaining_time
 *   per photon as above;  int64 tape_n;  double[tape_n];  double min_r, max_r, min_theta, max_theta (phMinMax), avg_scatt, avg_r (phScattStats),
 *   avg_energy (averagePhotonEnergy);  int32 max_scatt, min_scatt
 */
#include "mcrat.h"
#include "tape_rng.h"

#define HARNESS_MAGIC 0x4D435248

static void die(const char *what) { fprintf(stderr, "harness: %s\n", what); exit(1); }
static void rd(void *p, size_t sz, size_t n, FILE *f) { if (fread(p, sz, n, f) != n) die("short read"); }
static void wr(const void *p, size_t sz, size_t n, FILE *f) { if (fwrite(p, sz, n, f) != n) die("short write"); }

static void read_photon(struct photon *ph, FILE *f)
{
    double d[19];
    int k[3];
    rd(d, sizeof(double), 19, f);
    rd(k, sizeof(int), 3, f);
    memset(ph, 0, sizeof *ph);
    ph->p0 = d[0]; ph->p1 = d[1]; ph->p2 = d[2]; ph->p3 = d[3];
    ph->comv_p0 = d[4]; ph->comv_p1 = d[5]; ph->comv_p2 = d[6]; ph->comv_p3 = d[7];
    ph->r0 = d[8]; ph->r1 = d[9]; ph->r2 = d[10];
    ph->s0 = d[11]; ph->s1 = d[12]; ph->s2 = d[13]; ph->s3 = d[14];
    ph->num_scatt = d[15]; ph->weight = d[16]; ph->time_to_scatter = d[17]; ph->total_optical_depth = d[18];
    ph->nearest_block_index = k[0]; ph->recalc_properties = k[1]; ph->type = (char)k[2];
}

static void write_photon(const struct photon *ph, FILE *f)
{
    const double d[19] = {ph->p0, ph->p1, ph->p2, ph->p3, ph->comv_p0, ph->comv_p1, ph->comv_p2, ph->comv_p3, ph->r0, ph->r1, ph->r2,
                          ph->s0, ph->s1, ph->s2, ph->s3, ph->num_scatt, ph->weight, ph->time_to_scatter, ph->total_optical_depth};
    const int k[3] = {ph->nearest_block_index, ph->recalc_properties, (int)ph->type};
    wr(d, sizeof(double), 19, f);
    wr(k, sizeof(int), 3, f);
}

int main(int argc, char **argv)
{
    if (argc != 3) die("usage: harness <case>.in <case>.out");
    MPI_Init(&argc, &argv);                               /* (MCRaT's translation units are MPI programs; nothing here communicates) */
    FILE *in = fopen(argv[1], "rb");
    if (!in) die("cannot open the input");
    int h[7];
    double g[9];
    rd(h, sizeof(int), 7, in);
    rd(g, sizeof(double), 9, in);
    if (h[0] != HARNESS_MAGIC) die("not a harness input");
    const int dims = h[1], geometry = h[2], stokes = h[3], M = h[4], N = h[5], passes = h[6];
    if (dims != DIMENSIONS || geometry != GEOMETRY || stokes != (STOKES_SWITCH == ON))
        die("this harness was compiled for another (DIMENSIONS, GEOMETRY, STOKES_SWITCH) than the input asks for");

    struct hydro_dataframe hydrodata;
    memset(&hydrodata, 0, sizeof hydrodata);
    hydrodata.num_elements = M;
    double **cols[16] = {&hydrodata.r0, &hydrodata.r1, &hydrodata.r2, &hydrodata.r0_size, &hydrodata.r1_size, &hydrodata.r2_size, &hydrodata.r,
                         &hydrodata.theta, &hydrodata.v0, &hydrodata.v1, &hydrodata.v2, &hydrodata.dens, &hydrodata.dens_lab, &hydrodata.pres,
                         &hydrodata.temp, &hydrodata.gamma};
    for (int k = 0; k < 16; ++k) {
        *cols[k] = (double *)malloc(sizeof(double) * (size_t)M);
        if (!*cols[k]) die("out of memory");
        rd(*cols[k], sizeof(double), (size_t)M, in);
    }
    hydrodata.fps = g[0];
    hydrodata.r0_domain[0] = g[3]; hydrodata.r0_domain[1] = g[4];
    hydrodata.r1_domain[0] = g[5]; hydrodata.r1_domain[1] = g[6];
    hydrodata.r2_domain[0] = g[7]; hydrodata.r2_domain[1] = g[8];
    hydrodata.increment_scatt_frame = 1; hydrodata.increment_inj_frame = 1;
    hydrodata.grid = NULL;                                /* as getHydroData leaves it (Src/mcrat_io.c:1985) */

    struct photonList photon_list;
    initalizePhotonList(&photon_list);
    struct photon *arr = (struct photon *)malloc(sizeof(struct photon) * (size_t)N);
    if (!arr) die("out of memory");
    for (int i = 0; i < N; ++i) read_photon(&arr[i], in);
    fclose(in);
    setPhotonList(&photon_list, arr, N);

    gsl_rng *rng = gsl_rng_alloc(tape_recorder_type());   /* ranlxs0 with GSL's default seed, as Src/mcrat.c:99-103 -- every double recorded */
    FILE *fPtr = fopen("/dev/null", "w");

    /* the body of Src/mcrat.c:754-851 (CYCLOSYNCHROTRON_SWITCH OFF), for at most `passes` passes */
    double time_now = g[1], remaining_time = g[2], time_step = 0;
    int frame_scatt_cnt = 0, frame_abs_cnt = 0, ph_scatt_index = -1, find_nearest_grid_switch = 1, num_photons_find_new_element = 0, done = 0;
    while (remaining_time > 0 && done < passes) {
        num_photons_find_new_element += findContainingHydroCell(&photon_list, &hydrodata, find_nearest_grid_switch, rng, fPtr);
        calcMeanFreePath(&photon_list, &hydrodata, rng, fPtr);
        find_nearest_grid_switch = 0;
        if (getPhoton(&photon_list, photon_list.sorted_indexes[0])->time_to_scatter < remaining_time) {
            time_step = photonEvent(&photon_list, remaining_time, &hydrodata, &ph_scatt_index, &frame_scatt_cnt, &frame_abs_cnt, rng, fPtr);
            time_now += time_step;
            remaining_time -= time_step;
        } else {
            time_now += remaining_time;
            updatePhotonPosition(&photon_list, remaining_time, fPtr);
            time_step = remaining_time;
            remaining_time = 0;
        }
        done += 1;
    }

    FILE *out = fopen(argv[2], "wb");
    if (!out) die("cannot open the output");
    const int oh[6] = {HARNESS_MAGIC, photon_list.list_capacity, done, frame_scatt_cnt, num_photons_find_new_element, ph_scatt_index};
    const double og[2] = {time_now, remaining_time};
    wr(oh, sizeof(int), 6, out);
    wr(og, sizeof(double), 2, out);
    for (int i = 0; i < photon_list.list_capacity; ++i) write_photon(getPhoton(&photon_list, i), out);
    size_t n_tape = 0;
    const double *tape = tape_recorded(rng, &n_tape);
    const long long nt = (long long)n_tape;
    wr(&nt, sizeof nt, 1, out);
    wr(tape, sizeof(double), n_tape, out);
    {   /* G10: the per-frame reductions main() logs (Src/mcrat.c:704,881; Src/mclib.h:25-29) on the end state */
        double red[7] = {0, 0, 0, 0, 0, 0, 0};
        int mm[2] = {0, 0};
        phMinMax(&photon_list, &red[0], &red[1], &red[2], &red[3], fPtr);
        phScattStats(&photon_list, &mm[0], &mm[1], &red[4], &red[5], fPtr);
        red[6] = averagePhotonEnergy(&photon_list);
        wr(red, sizeof(double), 7, out);
        wr(mm, sizeof(int), 2, out);
    }
    fclose(out);
    fprintf(stderr, "harness: %d passes, %d scatterings, %lld uniforms recorded, time_now %.17g\n", done, frame_scatt_cnt, nt, time_now);
    tape_recorder_free(rng);
    gsl_rng_free(rng);
    MPI_Finalize();
    return 0;
}
