/*
 * mcrat_oracle.h -- CPU oracle for MCRaT's per-timestep photon loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may import, call, link or execute anything
 * under oracle/ -- and there only as the checker or the timed CPU baseline,
 * never as the thing shipped.  The product path (include/, mcrat_amd/) does
 * not link this library and fails loudly when its HIP extension is missing.
 *
 * PARITY UNPINNED: the reference (lazzati-astro/MCRaT @ /root/reference) ships
 * no tests, golden vectors or fixtures for this path and
 * cannot be compiled in this image: every translation unit includes 22 GSL
 * headers (Src/mcrat.h:91-125) and GSL is not installed; writing stand-ins for
 * it is not allowed.  This file is therefore a plain-C restatement of the
 * reference's algorithm, function by function with the reference file:line
 * cited on each, pinned only by closed-form known-answer tests
 * (tests/test_oracle_kat.py) and self-consistency properties.
 *
 * Deviations from the reference, all deliberate and documented in DESIGN.md:
 *   - random source: oracle_rng.h (the reference's ranlxs0 stream is an input);
 *   - the compile-time switches of mcrat_input.h are run-time fields of
 *     orc_config so one library serves every configuration;
 *   - argsort ties (qsort_r is unstable, mclib.c:723) break by lowest slot;
 *   - a photon whose free time is the 1e12/c "never scatters" default is not
 *     offered to the scattering routine even if dt_max were larger (the
 *     reference would index hydro arrays with -1 there, mclib.c:1146-1148);
 *   - sampleElectronTheta's cosine (electron.c:177-200) is clamped to [-1, 1]: for a uniform of exactly 0 rounding can leave it an ulp
 *     outside, where the reference's acos gives NaN;
 *   - fprintf logging is replaced by counters in orc_stats.
 */
#ifndef MCRAT_ORACLE_H
#define MCRAT_ORACLE_H

#include <stdint.h>
#include <stdio.h>
#include "oracle_rng.h"

/* values mirror Src/mcrat.h:36-44 */
#define ORC_CARTESIAN   0
#define ORC_SPHERICAL   1
#define ORC_CYLINDRICAL 2
#define ORC_POLAR       3
#define ORC_TWO             0
#define ORC_TWO_POINT_FIVE  1
#define ORC_THREE           2

/* Src/mcrat.h:52-57 */
#define ORC_INJECTED_PHOTON      'i'
#define ORC_COMPTONIZED_PHOTON   'k'
#define ORC_CS_POOL_PHOTON       'p'
#define ORC_UNABSORBED_CS_PHOTON 'c'
#define ORC_REBINNED_PHOTON      'r'
#define ORC_NULL_PHOTON          'N'

/* Src/mclib.c:4-5, verbatim values */
#define ORC_A_RAD       7.56e-15
#define ORC_C_LIGHT     2.99792458e10
#define ORC_PL_CONST    6.6260755e-27
#define ORC_K_B         1.380658e-16
#define ORC_M_P         1.6726231e-24
#define ORC_THOM_X_SECT 6.65246e-25
#define ORC_M_EL        9.1093879e-28

/* Src/mcrat.h:142-171, thermal-only build: 176 bytes on x86-64 */
typedef struct orc_photon {
    char   type;
    double p0, p1, p2, p3;
    double comv_p0, comv_p1, comv_p2, comv_p3;
    double r0, r1, r2;
    double s0, s1, s2, s3;
    double num_scatt;
    int    recalc_properties;
    double weight;
    int    nearest_block_index;
    double time_to_scatter;
    double total_optical_depth;
} orc_photon;

/* Src/mcrat.h:173-180 */
typedef struct orc_photon_list {
    orc_photon *photons;
    int *sorted_indexes;
    int num_photons;
    int num_null_photons;
    int list_capacity;
} orc_photon_list;

/* the fields of struct hydro_dataframe (Src/mcrat.h:194-244) the path reads */
typedef struct orc_hydro {
    int num_elements;
    double *r0, *r1, *r2;
    double *r0_size, *r1_size, *r2_size;
    double *v0, *v1, *v2;
    double *dens_lab, *temp, *gamma;
    double r0_domain[2], r1_domain[2], r2_domain[2];
    double fps;
} orc_hydro;

#define ORC_TAU_DIRECT 1
#define ORC_TAU_TABLE  2
typedef struct orc_config {
    int dimensions;     /* ORC_TWO / ORC_TWO_POINT_FIVE / ORC_THREE  (mcrat_input.h DIMENSIONS) */
    int geometry;       /* ORC_CARTESIAN ...                          (mcrat_input.h GEOMETRY)   */
    int stokes_switch;  /* STOKES_SWITCH                                                        */
    int tau_calculation;/* TAU_CALCULATION: ORC_TAU_DIRECT (also when 0) or ORC_TAU_TABLE        */
    /* TAU_CALCULATION == TABLE: thermal_table[i][j] of hot_x_section.c (i: photon energy, j: temperature; values are
     * log10 of the cross section over sigma_T) on the grid of hot_x_section.h:2-10.  The table is an INPUT here: its
     * creation (hot_x_section.c:82-133, GSL Monte-Carlo integration) is host-side work of the reference. */
    const double *hot_table;
    int n_ph_e, n_t;                /* N_PH_E, N_T: the table has (n_ph_e + 1) x (n_t + 1) entries */
    double log_ph_e_min, log_ph_e_max, log_t_min, log_t_max;
    /* 0: the reference's algorithm as it is (linear cell search, full argsort every pass) -- the checker and the
     * "port" baseline.  1: the same arithmetic and the same results, bit for bit, with the two costs a careful CPU
     * author would remove: the cell search goes through an exact bucket grid (orc_grid_attach) and only the prefix of
     * the sorted order that photonEvent consumes is produced (bench.py "cpu_optimised"). */
    int optimised;
    int fallback_calls; /* TABLE: samples of the integral a look-up off the table takes (0: the reference's 500 000, hot_x_section.c:348) */
} orc_config;
/* builds (and replaces) the bucket grid for `h`; used by orc_findContainingBlock when c->optimised and the frame matches */
void orc_grid_attach(const orc_config *c, const orc_hydro *h);
void orc_grid_detach(void);

typedef struct orc_stats {
    long long iterations;
    long long photon_steps;          /* sum over iterations of list_capacity          */
    long long frame_scatt_cnt;       /* mclib.c:1318                                   */
    long long num_photons_find_new_element; /* mclib.c:579                             */
    long long not_found;             /* "Hydro grid index not found" lines mclib.c:583 */
    long long kn_rejections;         /* candidates that drew an electron but did not scatter */
    long long event_draws;           /* event-stream draws consumed                    */
    int    last_scattered_index;     /* *scattered_ph_index, mclib.c:1341              */
    double last_time_step;
    double remaining_time;
    double time_now;
    long long table_fallbacks;          /* TABLE: look-ups off the table, integrated afresh (orc_tableFallbackCrossSection) */
} orc_stats;

/* ---- photonInjection ------------------------------------ */
/* mclib.c:9-300.  Allocates *out (malloc) with *n_out photons; *weight_out is ph_weight_adjusted.  The Poisson sampler
 * is the oracle's own (gsl_ran_poisson's algorithm lives in GSL; the count is a random input of the path): Knuth's
 * product method below a mean of 30, Hormann's PTRS transformed rejection (1993) above. */
long long orc_poisson(orc_rng *r, double mean);
int    orc_photonInjection(const orc_config *c, orc_photon **out, int *n_out, double *weight_out, double r_inj, double ph_weight,
                           int min_photons, int max_photons, char spect, double theta_min, double theta_max,
                           const orc_hydro *h, uint64_t seed, uint32_t stream);
void   orc_free(void *p);
void   orc_hydroCoordinateToSpherical(const orc_config *c, double *r, double *theta, double r0, double r1, double r2);   /* geometry.c:66 */
void   orc_hydroCoordinateToMcratCoordinate(const orc_config *c, double out[3], double r0, double r1, double r2);         /* geometry.c:108 */

/* ---- L1 maths -------------------------------------------------------- */
void   orc_lorentzBoost(const double boost[3], const double p[4], double result[4], char object); /* mclib.c:302 */
void   orc_zeroNorm(double p[4]);                                                                /* mclib.c:409 */
double orc_dnrm2(const double *x, int n);                                /* reference BLAS dnrm2 (gsl_blas_dnrm2) */
void   orc_mcratCoordinateToHydroCoordinate(const orc_config *c, double out[3], double x, double y, double z); /* geometry.c:15 */
void   orc_hydroVectorToCartesian(const orc_config *c, double out[3], double v0, double v1, double v2,
                                  double x0, double x1, double x2);       /* geometry.c:189 */
int    orc_checkInBlock(const orc_config *c, double a0, double a1, double a2, const orc_hydro *h, int idx); /* geometry.c:394 */
int    orc_findContainingBlock(const orc_config *c, double a0, double a1, double a2, const orc_hydro *h);   /* geometry.c:350 */
double orc_hydroElementVolume(const orc_config *c, const orc_hydro *h, int idx); /* geometry.c:255 */
void   orc_mullerMatrixRotation(double theta, double s[4]);              /* mcrat_scattering.c:10 */
void   orc_findXY(const double v_ph[3], const double ref[3], double x[3], double y[3]); /* mcrat_scattering.c:41 */
double orc_findPhi(const double x_old[3], const double y_old[3], const double x_new[3], const double y_new[3]); /* :67 */
void   orc_stokesRotation(const double v[3], const double v_ph[3], const double v_ph_boosted[3], double s[4]);  /* :103 */
double orc_kleinNishinaCrossSection(double energy_ratio);                /* mcrat_scattering.c:597 */
double orc_bessel_K2(double x);                                          /* stands for gsl_sf_bessel_Kn(2,x), electron.c:221 */

/* ---- sampling -------------------------------------------------------- */
int    orc_kleinNishinaScatter(const orc_config *c, double *theta, double *phi, double p0, double q, double u, orc_rng *rng); /* :509 */
double orc_sampleThermalElectron(double temp, orc_rng *rng);             /* electron.c:202 */
double orc_sampleElectronTheta(double beta, orc_rng *rng);               /* electron.c:177 */
void   orc_rotateElectron(double el_p[4], const double ph_p[4]);         /* electron.c:126 */
void   orc_singleThermalElectron(double el_p[4], double temp, const double ph_p[4], orc_rng *rng); /* electron.c:70 */
int    orc_singleScatter(const orc_config *c, double el_comov[4], double ph_comov[4], double s[4], orc_rng *rng); /* mcrat_scattering.c:151 */

/* ---- the loop (reference signatures minus gsl_rng*, FILE*) ------------ */
void   orc_calculateOpticalDepth(const orc_config *c, orc_photon *ph, const orc_hydro *h);       /* optical_depth.c:7 */
double orc_getThermalCrossSection(const orc_config *c, double photon_comv_e, double fluid_temp, int *miss); /* optical_depth.c:132 */
/* createHotCrossSection's integrals (hot_x_section.c:324-400, electron.c:538-561).  gsl_monte_plain_integrate lives in GSL:
 * its published algorithm is volume x sample mean of the integrand at points x = xl + uniform_pos * (xu - xl) drawn
 * coordinate by coordinate; the points here come from oracle_rng.h's keyed streams {iteration = entry, word2 = k % 256,
 * purpose = 4}, sample k being the (k / 256)-th of its stream -- the source the engine uses. */
double orc_singleMaxwellJuttner(double gamma, double theta);                               /* electron.c:538 */
double orc_boostedCrossSection(double norm_ph_comv, double mu, double gamma);              /* hot_x_section.c:370 */
double orc_calculateTotalThermalCrossSection(double ph_comv, double theta, long long calls, uint64_t seed, int entry); /* :324 */
void   orc_createHotCrossSection(double *thermal_table, int n_ph_e, int n_t, double log_ph_e_min, double log_ph_e_max,
                                 double log_t_min, double log_t_max, long long calls, uint64_t seed);  /* :82-107 */
double orc_tableFallbackCrossSection(const orc_config *c, double eps, double theta, const orc_rng *rng, uint32_t slot); /* hot_x_section.c:563-599 */
double orc_getThermalCrossSection_keyed(const orc_config *c, double photon_comv_e, double fluid_temp, const orc_rng *rng, uint32_t slot, int *miss);
void   orc_calculateOpticalDepth_keyed(const orc_config *c, orc_photon *ph, const orc_hydro *h, const orc_rng *rng, uint32_t slot);
int    orc_findContainingHydroCell_keyed(const orc_config *c, orc_photon_list *l, const orc_hydro *h, int find_nearest_block_switch, orc_stats *st,
                                         const orc_rng *rng);
long long orc_table_fallbacks(void);   /* look-ups off the table (integrated afresh) since orc_reset_table_fallbacks() */
void   orc_reset_table_fallbacks(void);
int    orc_findContainingHydroCell(const orc_config *c, orc_photon_list *l, const orc_hydro *h,
                                   int find_nearest_block_switch, orc_stats *st);                /* mclib.c:436 */
void   orc_calcMeanFreePath(const orc_config *c, orc_photon_list *l, const orc_hydro *h, orc_rng *rng); /* mclib.c:617 */
void   orc_updatePhotonPosition(orc_photon_list *l, double t);                                   /* mclib.c:1054 */
double orc_photonEvent(const orc_config *c, orc_photon_list *l, double dt_max, const orc_hydro *h,
                       int *scattered_ph_index, long long *frame_scatt_cnt, orc_rng *rng, orc_stats *st); /* mclib.c:1107 */
double orc_averagePhotonEnergy(const orc_photon_list *l);                                         /* mclib.c:1358 */
void   orc_phScattStats(const orc_photon_list *l, int *max, int *min, double *avg, double *r_avg); /* mclib.c:1385 */
void   orc_phMinMax(const orc_photon_list *l, double *min, double *max, double *min_theta, double *max_theta); /* mclib.c:1465 */

/* while (remaining_time>0) of mcrat.c:754-851, optionally bounded by max_iterations (<=0: unbounded).
 * iteration_base offsets the RNG iteration counter so a frame can be run in several calls.
 * find_nearest_grid_switch is the value at entry (1 on the first call of a frame, mcrat.c:756). */
void   orc_photon_loop(const orc_config *c, orc_photon_list *l, const orc_hydro *h, orc_rng *rng,
                       double *time_now, double *remaining_time, int *find_nearest_grid_switch,
                       long long max_iterations, uint64_t iteration_base, orc_stats *st);

/* ---- hydro ingest (oracle_ingest.c) ---------------------- */
/* every column of struct hydro_dataframe (Src/mcrat.h:194-244) a thermal build fills; malloc'ed, orc_frame_free() */
typedef struct orc_frame {
    int num_elements;
    double *r0, *r1, *r2, *r0_size, *r1_size, *r2_size;
    double *v0, *v1, *v2;
    double *dens, *dens_lab, *pres, *temp, *gamma;
    double *r, *theta;
} orc_frame;
/* the arguments of getHydroData that select the slab (mcrat_io.h:26) + hydro_data->fps */
typedef struct orc_slab {
    double r_inj;
    int    ph_inj_switch;
    double min_r, max_r, min_theta, max_theta;
    double fps;
} orc_slab;
/* a FLASH checkpoint's datasets as H5Dread leaves them (mclib_flash.c:143-193) */
typedef struct orc_flash_blocks {
    int n_blocks, coord_stride, bsize_stride;
    const double *coordinates, *block_size;
    const int *node_type;
    const double *velx, *vely, *dens, *pres;     /* [n_blocks][1][8][8] */
    double l_scale, d_scale, p_scale;            /* HYDRO_L_SCALE, HYDRO_D_SCALE, HYDRO_P_SCALE */
    int cyclosynchrotron;                        /* CYCLOSYNCHROTRON_SWITCH: elem_factor starts at 2 */
} orc_flash_blocks;
/* a PLUTO .dbl frame: readGridFile's arrays and the variable blocks (mclib_pluto.c:1085-1128) */
typedef struct orc_pluto_grid {
    int nx, ny, nz;
    const double *x1, *dx1, *x2, *dx2, *x3, *dx3;
    const double *rho, *vx1, *vx2, *vx3, *prs;   /* [nz][ny][nx] */
    double l_scale, d_scale, p_scale;
    int cyclosynchrotron;
} orc_pluto_grid;
/* a PLUTO-Chombo frame as readPlutoChombo holds it after its H5Dread / H5Aread calls (mclib_pluto.c:60-430) */
typedef struct orc_chombo_level {
    int n_boxes;
    const int *boxes;            /* n_boxes x {lo_i, lo_j, [lo_k], hi_i, hi_j, [hi_k]}: "boxes" */
    const int *box_offsets;      /* "data:offsets=0": where each box's data starts within the level, in doubles */
    long long data_len;          /* length of "data:datatype=0" */
    int prob_domain[6];          /* same member order as a box */
    int ref_ratio, logr;
    double dx, dombeg1, dombeg2, dombeg3, g_x2stretch, g_x3stretch;
} orc_chombo_level;
typedef struct orc_chombo {
    int num_levels, num_vars;
    const orc_chombo_level *levels;
    const char *const *var_names;    /* component_0 ... */
    const double *data;              /* the levels' "data:datatype=0", level 0 first (all_data, :151-155,:360) */
    double l_scale, d_scale, p_scale;
    int cyclosynchrotron;
} orc_chombo;
int  orc_chombo_select(const orc_config *c, const orc_chombo *h, const orc_slab *s, int max_elem_factor, orc_frame *out, int *elem_factor_out);

#define ORC_SCIENCE                       0
#define ORC_CYLINDRICAL_OUTFLOW           1
#define ORC_SPHERICAL_OUTFLOW             2
#define ORC_STRUCTURED_SPHERICAL_OUTFLOW  3
typedef struct orc_outflow {
    int simulation_type;
    double gamma_infinity;      /* gamma_0 of the structured fireball */
    double lumi, r00;
    double t_comov, ddensity;   /* cylindrical outflow */
    double theta_j, p;          /* structured fireball */
} orc_outflow;
int  orc_flash_select(const orc_config *c, const orc_flash_blocks *b, const orc_slab *s, int max_elem_factor, orc_frame *out, int *elem_factor_out);
int  orc_pluto_select(const orc_config *c, const orc_pluto_grid *g, const orc_slab *s, int max_elem_factor, orc_frame *out, int *elem_factor_out);
void orc_frame_free(orc_frame *f);
void orc_fillHydroCoordinateToSpherical(const orc_config *c, orc_frame *f);      /* geometry.c:156 */
void orc_outflow_defaults(int simulation_type, orc_outflow *o);
void orc_cylindricalPrep(const orc_config *c, const orc_outflow *o, orc_frame *f);        /* analytic_outflows.c:3 */
void orc_sphericalPrep(const orc_config *c, const orc_outflow *o, orc_frame *f);          /* analytic_outflows.c:63 */
void orc_structuredFireballPrep(const orc_config *c, const orc_outflow *o, orc_frame *f); /* analytic_outflows.c:138 */
void orc_hydro_post_read(const orc_config *c, const orc_outflow *o, orc_frame *f);        /* mcrat_io.c:1962-1975 */

/* ---- cyclo-synchrotron, first part (oracle_cyclosynch.c) ---------------------------------- */
#define ORC_B_INTERNAL_E 0      /* B_FIELD_CALC, mcrat.h:47-49 */
#define ORC_B_TOTAL_E    1
#define ORC_B_SIMULATION 2
typedef struct orc_cs {
    int    b_field_calc;
    double epsilon_b;           /* EPSILON_B (mcrat.h:334 default 0.5) */
    double rebin_e_perc;        /* CYCLOSYNCHROTRON_REBIN_E_PERC (mcrat.h:311 default 0.1) */
    const double *dens;         /* hydro_data->dens (comoving density), per cell */
    const double *B0, *B1, *B2; /* hydro_data->B0-2 when B_FIELD_CALC == SIMULATION */
    int    scatt_frame_number, inj_frame_number;   /* hydro_data->scatt_frame_number / ->inj_frame_number */
    double rebin_ang, rebin_ang_phi;               /* CYCLOSYNCHROTRON_REBIN_ANG (0.5 deg), _ANG_PHI (10 deg), mcrat.h:315-321 */
} orc_cs;
typedef double (*orc_integrand)(double x, void *ctx);
void   orc_list_init(orc_photon_list *l);                                         /* photons.c:3 */
void   orc_list_free(orc_photon_list *l);                                         /* photons.c:12 */
int    orc_list_set(orc_photon_list *l, const orc_photon *ph, int n);             /* setPhotonList, photons.c:82 */
int    orc_list_realloc(orc_photon_list *l, int new_capacity);                    /* photons.c:37 */
int    orc_list_add(orc_photon_list *l, const orc_photon *ph, int num);           /* addToPhotonList, photons.c:108 */
int    orc_list_set_null(orc_photon_list *l, int index);                          /* setNullPhoton, photons.c:210 */
double orc_calcCyclotronFreq(double magnetic_field);                              /* mc_cyclosynch.c:30 */
double orc_calcEB(double magnetic_field);
double orc_calcDimlessTheta(double temp);
double orc_calcBoundaryE(double magnetic_field, double temp);
double orc_calcB(const orc_cs *cs, double el_dens, double temp);                  /* :54 */
double orc_getMagneticFieldMagnitude(const orc_config *c, const orc_cs *cs, const orc_hydro *h, int i);   /* :78 */
double orc_blackbody_ph_spect(double nu, double temp);                            /* :185 */
double orc_calcCyclosynchRLimits(int frame_scatt, int frame_inj, double fps, double r_inj, int want_max);  /* :225 */
void   orc_qk21(orc_integrand f, void *ctx, double a, double b, double *result, double *abserr, double *resabs, double *resasc);
int    orc_qags(orc_integrand f, void *ctx, double a, double b, double epsabs, double epsrel, int limit, double *result, double *abserr,
                int *used_fallback);
int    orc_photonEmitCyclosynch(const orc_config *c, const orc_cs *cs, orc_photon_list *l, double r_inj, double ph_weight, int maximum_photons,
                                double theta_min, double theta_max, const orc_hydro *h, orc_rng *rng, int inject_single_switch, int scatt_ph_index,
                                double *weight_out, int *used_fallback_out);     /* :1176 */
int    orc_rebinCyclosynchCompPhotons(const orc_config *c, const orc_cs *cs, orc_photon_list *l, int *num_cyclosynch_ph_emit,
                                      int *scatt_cyclosynch_num_ph, int max_photons);   /* mc_cyclosynch.c:610 */
typedef struct orc_cs_counts {          /* main()'s cyclo-synchrotron counters for one scatter frame */
    int    num_cyclosynch_ph_emit, scatt_cyclosynch_num_ph, frame_abs_cnt, rebins, error;
    double n_comptonized, pool_weight;
} orc_cs_counts;
void   orc_scatter_frame_cs(const orc_config *c, orc_cs *cs, orc_photon_list *l, const orc_hydro *h, orc_rng *rng, double *time_now,
                            double remaining_time, double r_inj, double ph_weight_suggest, int max_photons, double theta_jmin_thread,
                            double theta_jmax_thread, int emit_pool, long long max_iterations, orc_stats *st, orc_cs_counts *cnt);   /* mcrat.c:706-878 */
int    orc_saveCheckpoint_convert(orc_photon_list *l);                            /* mcrat_io.c:896-900: 'k' with weight -> 'c' */
double orc_phAbsCyclosynch(const orc_config *c, const orc_cs *cs, orc_photon_list *l, const orc_hydro *h, int *num_abs_ph,
                           int *scatt_cyclosynch_num_ph);                        /* :1571 */

/* helpers for ctypes */
int    orc_sizeof_photon(void);

#endif
