/*
 * mcrat_hip.h -- C ABI of the MI355X photon-loop engine (libmcrat_hip.so).
 *
 * Drop-in boundary for ONE path of lazzati-astro/MCRaT: the per-timestep photon
 * loop `while (remaining_time > 0)` of Src/mcrat.c:761-851 and the functions it
 * calls (findContainingHydroCell mclib.c:436, calcMeanFreePath mclib.c:617,
 * photonEvent mclib.c:1107, updatePhotonPosition mclib.c:1054) plus the
 * per-frame reductions around it (phMinMax mclib.c:1465, phScattStats
 * mclib.c:1385, averagePhotonEnergy mclib.c:1358).
 *
 * The reference has no plugin/FFI interface: its boundary is the set of C
 * functions main() calls on caller-owned structs (Src/mclib.h:8-29).  This
 * header keeps those structs' layouts (so MCRaT's own photonInjection,
 * saveCheckpoint, printPhotons keep working on the host side) and replaces the
 * loop by frame-granular calls, because a host<->device round trip per scatter
 * event would cost more than the event.  INTEGRATION.md shows the edit to
 * mcrat.c; DESIGN.md the device side.
 *
 * Conventions (mirroring the reference where it has any):
 *   - plain C, plain pointers and sizes; the library never takes ownership of
 *     caller memory and never calls exit(); every entry point returns 0 on
 *     success or a negative MCRAT_HIP_E* code (the caller keeps the reference's
 *     fatal behaviour, e.g. mcrat.c:904-915);
 *   - one context per MPI rank / GPU; a context is not thread safe (the
 *     reference runs one thread per rank);
 *   - the "cell not found" log lines of geometry.c:382 / mclib.c:583 are
 *     reported as counters in mcrat_hip_frame_stats;
 *   - there is NO CPU fallback: if the device or the code object is missing
 *     mcrat_hip_init fails with MCRAT_HIP_ENODEV.
 */
#ifndef MCRAT_HIP_H
#define MCRAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: these declarations are its whole surface */
#endif

#define MCRAT_HIP_ABI_VERSION 1

/* error codes */
#define MCRAT_HIP_OK        0
#define MCRAT_HIP_EINVAL   (-1)  /* bad argument / unsupported switch combination        */
#define MCRAT_HIP_ENODEV   (-2)  /* no usable gfx950 device or code object                */
#define MCRAT_HIP_ENOMEM   (-3)  /* device or host allocation failed                      */
#define MCRAT_HIP_EHIP     (-4)  /* a HIP runtime call failed (see mcrat_hip_last_error)  */
#define MCRAT_HIP_ESTATE   (-5)  /* call out of order (e.g. run before set_hydro)         */
#define MCRAT_HIP_EREFUSED (-6)  /* the reference refuses this too and goes on (rebinCyclosynchCompPhotons' early returns); nothing changed */

/* switch values: identical to Src/mcrat.h:36-44,64-65 so a -D of mcrat_input.h maps 1:1 */
#define MCRAT_HIP_CARTESIAN        0
#define MCRAT_HIP_SPHERICAL        1
#define MCRAT_HIP_CYLINDRICAL      2
#define MCRAT_HIP_POLAR            3
#define MCRAT_HIP_TWO              0
#define MCRAT_HIP_TWO_POINT_FIVE   1
#define MCRAT_HIP_THREE            2
#define MCRAT_HIP_TAU_DIRECT       1
#define MCRAT_HIP_TAU_TABLE        2   /* needs mcrat_hip_set_hot_cross_section before the first frame */

/* the compile-time switches of Src/mcrat_input.h:49-71 that the loop depends on */
typedef struct mcrat_hip_config {
    int abi_version;             /* MCRAT_HIP_ABI_VERSION                                     */
    int dimensions;              /* DIMENSIONS                                                */
    int geometry;                /* GEOMETRY                                                  */
    int stokes_switch;           /* STOKES_SWITCH   (0/1)                                     */
    int tau_calculation;         /* TAU_CALCULATION: MCRAT_HIP_TAU_DIRECT or MCRAT_HIP_TAU_TABLE */
    int cyclosynchrotron_switch; /* CYCLOSYNCHROTRON_SWITCH; 1 needs virtual_rank_photons = 0  */
    int device;                  /* HIP device ordinal                                        */
    void *stream;                /* hipStream_t to launch on, or NULL for a private stream    */
    uint32_t rng_stream;         /* virtual-rank id mixed into every RNG counter (rank id)    */
    int iterations_per_sync;     /* loop iterations queued between host status reads (0: 256) */
    int use_graph;               /* replay the iteration batch as a hipGraph (0/1)            */
    int profile;                 /* bracket every step-kernel launch with HIP events (0/1)    */
    int virtual_rank_photons;    /* 0: the photon list is ONE list with one clock (one MPI rank of the reference).
                                    n > 0: the list is split into ceil(capacity/n) *virtual ranks* of n consecutive
                                    slots, each an independent photon list with its own clock and RNG stream
                                    (rng_stream + r) -- the reference's many-small-ranks run shape
                                    (Doc/mcrat_doc.tex:165-166,222) on one GPU; see DESIGN.md section 2 */
} mcrat_hip_config;

/* == struct photon, Src/mcrat.h:142-171 (thermal-only build): 176 bytes on x86-64,
 * offsets type@0 p0@8 ... num_scatt@128 recalc_properties@136 weight@144
 * nearest_block_index@152 time_to_scatter@160 total_optical_depth@168 */
typedef struct mcrat_hip_photon {
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
} mcrat_hip_photon;

/* == struct photonList, Src/mcrat.h:173-180 */
typedef struct mcrat_hip_photon_list {
    mcrat_hip_photon *photons;
    int *sorted_indexes;
    int num_photons;
    int num_null_photons;
    int list_capacity;
} mcrat_hip_photon_list;

/* column form of the same record (host pointers, n entries each); any pointer may be
 * NULL on get (column skipped).  On set, NULL comv_p, s, time_to_scatter and
 * total_optical_depth columns mean zeros; the other columns are required. */
typedef struct mcrat_hip_photon_soa {
    int n;
    char   *type;
    double *p0, *p1, *p2, *p3;
    double *comv_p0, *comv_p1, *comv_p2, *comv_p3;
    double *r0, *r1, *r2;
    double *s0, *s1, *s2, *s3;
    double *num_scatt;
    int    *recalc_properties;
    double *weight;
    int    *nearest_block_index;
    double *time_to_scatter;
    double *total_optical_depth;
} mcrat_hip_photon_soa;

/* the members of struct hydro_dataframe (Src/mcrat.h:194-244) the loop reads; host pointers
 * of num_elements doubles.  r2, r2_size, v2 may be NULL in 2-D. */
typedef struct mcrat_hip_hydro {
    int num_elements;
    const double *r0, *r1, *r2;
    const double *r0_size, *r1_size, *r2_size;
    const double *v0, *v1, *v2;
    const double *dens_lab, *temp, *gamma;
    double r0_domain[2], r1_domain[2], r2_domain[2];
    double fps;
} mcrat_hip_hydro;

/* what main() logs after the loop (mcrat.c:810-817,881-892) plus throughput counters */
typedef struct mcrat_hip_frame_stats {
    long long iterations;                    /* passes of the while loop                         */
    long long photon_steps;                  /* iterations x list_capacity                       */
    long long frame_scatt_cnt;               /* mclib.c:1318                                     */
    long long num_photons_find_new_element;  /* mclib.c:579,608-611                              */
    long long not_found;                     /* "Hydro grid index not found" events, mclib.c:583 */
    long long kn_rejections;                 /* candidates that drew an electron but did not scatter */
    long long rescans;                       /* times the candidate list had to be refilled      */
    int    last_scattered_index;             /* *scattered_ph_index, mclib.c:1341                */
    double last_scattered_temp;              /* hydro temp of its cell, mcrat.c:813              */
    double last_time_step;                   /* time_step, mcrat.c:781,844                       */
    double remaining_time;
    double time_now;
    double step_kernel_ms;                   /* profile=1: summed duration of step-kernel launches */
    long long step_kernel_launches;
    double event_kernel_ms;                  /* profile=1: summed duration of event-kernel launches */
    long long table_fallbacks;                  /* TAU_CALCULATION == TABLE: look-ups off the table whose cross section was integrated afresh (mcrat_hip_set_hot_cross_section) */
    long long slot_steps;                    /* slots actually taken through a pass, summed over the passes: = photon_steps, except that a
                                                cyclo-synchrotron list's settled null slots behind its last photon (the half a doubled list consists of,
                                                Src/photons.c:112-121) take no part in a pass here although the reference walks them (Src/mclib.c:620,684) --
                                                the figure a roofline is computed from */
} mcrat_hip_frame_stats;

typedef struct mcrat_hip_ctx mcrat_hip_ctx;

/* lifetime ------------------------------------------------------------------ */
int  mcrat_hip_init(mcrat_hip_ctx **ctx, const mcrat_hip_config *cfg);
void mcrat_hip_destroy(mcrat_hip_ctx *ctx);
const char *mcrat_hip_version(void);
const char *mcrat_hip_strerror(int code);
const char *mcrat_hip_last_error(const mcrat_hip_ctx *ctx);   /* text of the last HIP failure */
/* A context belongs to the device it was created on (mcrat_hip_config.device), and HIP's "current device" is a property of the host thread:
 * a thread other than the creating one calls this once before it uses the context (two pools per GPU with a host thread each: INTEGRATION.md).
 * One context, one thread at a time. */
int mcrat_hip_bind_thread(mcrat_hip_ctx *ctx);

/* staging: once per hydro frame (after getHydroData, mcrat.c:721) ------------- */
int mcrat_hip_set_hydro(mcrat_hip_ctx *ctx, const mcrat_hip_hydro *hydro);

/* getHydroData (mcrat_io.c:1898-1990; mcrat.c:640,721) on the device, file reading excluded: the caller hands over the
 * buffers a reader holds right after its H5Dread / fread calls, in code units; the expansion to cells, the unit
 * scaling, the slab selection with the reader's elem_factor loop, the derived columns (gamma, dens_lab, temp),
 * fillHydroCoordinateToSpherical (geometry.c:156) and the SIMULATION_TYPE overwrite (analytic_outflows.c) run on the
 * device, in the reference's cell order, and the result is staged exactly as mcrat_hip_set_hydro would stage it -- no
 * host copy of the selected frame is made (mcrat_hip_get_hydro returns one when the caller needs it).
 *   mcrat_hip_ingest_flash  replaces readAndDecimate (mclib_flash.c:60-431) after line 197 (datasets read)
 *   mcrat_hip_ingest_pluto  replaces readPluto (mclib_pluto.c:1058-1459) after line 1128 (file read)
 *   mcrat_hip_ingest_chombo replaces readPlutoChombo (mclib_pluto.c:12-801) after its HDF5 reads
 * mcrat_host_read_pluto (mcrat_amd/host) parses grid.out, dbl.out and the .dbl file into mcrat_hip_pluto_grid. */
typedef struct mcrat_hip_slab {           /* the selecting arguments of getHydroData (mcrat_io.h:26) + frame constants */
    double r_inj;
    int    ph_inj_switch;                 /* 1: every cell with r > 0.95 r_inj (injection frame); 0: the photons' slab */
    double min_r, max_r, min_theta, max_theta;   /* phMinMax of the photon list (mcrat.c:716) */
    double fps;                           /* hydro_data->fps */
    double r0_domain[2], r1_domain[2], r2_domain[2];   /* hydro_data->r*_domain (mc.par / mcrat_input.h) */
} mcrat_hip_slab;

typedef struct mcrat_hip_flash_blocks {   /* the datasets of a FLASH checkpoint, mclib_flash.c:143-193 */
    int n_blocks;                         /* dims[0] of "coordinates" */
    int coord_stride, bsize_stride;       /* doubles per row of "coordinates" / "block size" */
    const double *coordinates, *block_size;
    const int    *node_type;              /* leaf blocks have 1 */
    const double *velx, *vely, *dens, *pres;   /* [n_blocks][1][8][8] */
    double l_scale, d_scale, p_scale;     /* HYDRO_L_SCALE, HYDRO_D_SCALE, HYDRO_P_SCALE */
} mcrat_hip_flash_blocks;

typedef struct mcrat_hip_pluto_grid {     /* a PLUTO .dbl frame: readGridFile's arrays + the variable blocks */
    int nx, ny, nz;                       /* nz ignored unless DIMENSIONS == THREE */
    const double *x1, *dx1, *x2, *dx2, *x3, *dx3;   /* cell centres and widths per axis, code units (mclib_pluto.c:951-971) */
    const double *rho, *vx1, *vx2, *vx3, *prs;      /* [nz][ny][nx]; vx3 may be NULL in 2-D */
    double l_scale, d_scale, p_scale;
} mcrat_hip_pluto_grid;

typedef struct mcrat_hip_chombo_level {   /* one "level_<i>" group of a PLUTO-Chombo file, mclib_pluto.c:206-276,349-430 */
    int n_boxes;
    const int *boxes;                     /* "boxes": n_boxes x {lo_i, lo_j, [lo_k], hi_i, hi_j, [hi_k]} (the compound type of :48-58) */
    const int *box_offsets;               /* "data:offsets=0": start of each box's data within the level, in doubles */
    long long data_len;                   /* length of "data:datatype=0" */
    int prob_domain[6];                   /* attribute prob_domain, same member order as a box */
    int ref_ratio, logr;                  /* attributes ref_ratio, logr */
    double dx, dombeg1, dombeg2, dombeg3, g_x2stretch, g_x3stretch;   /* attributes dx, domBeg1-3, g_x2stretch, g_x3stretch */
} mcrat_hip_chombo_level;

typedef struct mcrat_hip_chombo {         /* a PLUTO-Chombo AMR frame after readPlutoChombo's HDF5 reads */
    int num_levels, num_vars;             /* attributes num_levels, num_components */
    const mcrat_hip_chombo_level *levels; /* level 0 (coarsest) first */
    const char *const *var_names;         /* attributes component_0 ... ("rho", "vx1", "vx2", "vx3", "prs", ...) */
    const double *data;                   /* the levels' "data:datatype=0", level 0 first (all_data, :151-155,:360) */
    double l_scale, d_scale, p_scale;
} mcrat_hip_chombo;

#define MCRAT_HIP_SCIENCE                       0   /* SIMULATION_TYPE, mcrat.h:30-33 */
#define MCRAT_HIP_CYLINDRICAL_OUTFLOW           1
#define MCRAT_HIP_SPHERICAL_OUTFLOW             2
#define MCRAT_HIP_STRUCTURED_SPHERICAL_OUTFLOW  3
typedef struct mcrat_hip_outflow {        /* the constants analytic_outflows.c hard-codes (lines 5, 65, 140-141) */
    int simulation_type;
    double gamma_infinity;                /* gamma_0 of the structured fireball */
    double lumi, r00;
    double t_comov, ddensity;             /* cylindrical outflow */
    double theta_j, p;                    /* structured fireball */
} mcrat_hip_outflow;
void mcrat_hip_outflow_defaults(int simulation_type, mcrat_hip_outflow *out);   /* the reference's values */

typedef struct mcrat_hip_ingest_result {
    int num_elements;                     /* hydro_data->num_elements */
    int elem_factor;                      /* the reader's log line "Elem factor: %d" */
    long long cells_read;                 /* cells examined (leaf blocks x 64, or nx ny nz) */
} mcrat_hip_ingest_result;

/* outflow may be NULL (SCIENCE).  MCRAT_HIP_EINVAL with last_error set when no elem_factor up to 1000 selects a cell
 * (the reference would loop forever). */
int mcrat_hip_ingest_flash(mcrat_hip_ctx *ctx, const mcrat_hip_flash_blocks *blocks, const mcrat_hip_slab *slab,
                           const mcrat_hip_outflow *outflow, mcrat_hip_ingest_result *result);
int mcrat_hip_ingest_pluto(mcrat_hip_ctx *ctx, const mcrat_hip_pluto_grid *grid, const mcrat_hip_slab *slab,
                           const mcrat_hip_outflow *outflow, mcrat_hip_ingest_result *result);

/* replaces readPlutoChombo (mclib_pluto.c:12-801) after line 430 (levels read).  Cells are numbered level by level, box
 * by box, x fastest, as the reader numbers them; coarse cells covered by a finer level are dropped exactly where the
 * reference drops them -- in injection frames (ph_inj_switch != 0, :672,:745) and not in the photons'-slab branch
 * (:659,:703), which does not consult good_node_buffer.  The per-level 1-D coordinate arrays (:446-517, with libm's exp
 * for logarithmic radial grids) are computed on the host; boxes must lie inside their level's prob_domain and their
 * data must follow one another in "data:offsets=0" order. */
int mcrat_hip_ingest_chombo(mcrat_hip_ctx *ctx, const mcrat_hip_chombo *frame, const mcrat_hip_slab *slab,
                            const mcrat_hip_outflow *outflow, mcrat_hip_ingest_result *result);

/* every column of the staged frame (struct hydro_dataframe, mcrat.h:194-244), host pointers of num_elements doubles
 * allocated by the caller; NULL pointers are skipped.  After mcrat_hip_set_hydro the columns that call did not carry
 * (dens, pres) read as zero. */
typedef struct mcrat_hip_hydro_columns {
    int num_elements;                     /* in: capacity of the arrays; out: elements written */
    double *r0, *r1, *r2, *r0_size, *r1_size, *r2_size;
    double *v0, *v1, *v2;
    double *dens, *dens_lab, *pres, *temp, *gamma;
    double *r, *theta;
} mcrat_hip_hydro_columns;
int mcrat_hip_get_hydro(mcrat_hip_ctx *ctx, mcrat_hip_hydro_columns *out);

/* The consumers after a frame: what printPhotons and saveCheckpoint read from the photon list.
 *   mcrat_hip_get_output         printPhotons' gathering loop (mcrat_io.c:137-181): the photons with weight != 0, in slot
 *                                order, as the arrays it hands to H5Dwrite (datasets P0-3, COMV_P0-3, R0-2, S0-3, NS, PW, PT),
 *                                compacted on the device.  count: in, capacity of the arrays; out, photons written.  With
 *                                every pointer NULL the call only reports the count (MCRAT_HIP_EINVAL + the needed count if
 *                                the capacity is too small).  NULL columns are skipped (COMV_SWITCH / STOKES_SWITCH / SAVE_TYPE
 *                                OFF builds).  The reference holds these arrays on the stack (mcrat_io.c:123-124).
 *   mcrat_hip_get_photons_range  struct photon records of the slots first .. first+count-1 (saveCheckpoint's fwrite loop,
 *                                mcrat_io.c:883-896, in pieces: a 10^8-photon list is 17.6 GB as records).  Bytes between
 *                                the members are zero. */
typedef struct mcrat_hip_output_columns {
    int count;
    double *p0, *p1, *p2, *p3;
    double *comv_p0, *comv_p1, *comv_p2, *comv_p3;
    double *r0, *r1, *r2;
    double *s0, *s1, *s2, *s3;
    double *num_scatt, *weight;
    char   *type;
} mcrat_hip_output_columns;
int mcrat_hip_get_output(mcrat_hip_ctx *ctx, mcrat_hip_output_columns *out);
/* saveCheckpoint's in-place type conversion in CYCLOSYNCHROTRON_SWITCH builds (mcrat_io.c:896-900, :951-955, :991-995), on the resident
 * list: every comptonised photon 'k' with weight != 0 becomes an unabsorbed one 'c' -- before the records are streamed out, and for
 * good, so that the next frame's phAbsCyclosynch (mc_cyclosynch.c:1607) and printPhotons' PT see what the reference's would.
 * mcrat_host_save_checkpoint calls it when its cyclosynchrotron_switch argument is on.  *num_converted may be NULL. */
int mcrat_hip_convert_comptonized(mcrat_hip_ctx *ctx, int *num_converted);
int mcrat_hip_get_photons_range(mcrat_hip_ctx *ctx, int first, int count, mcrat_hip_photon *records);

/* The same two consumers WHILE THE NEXT FRAME RUNS.  The reference calls saveCheckpoint (mcrat.c:902; mcrat_io.c:838-1009) and printPhotons
 * (mcrat.c:907; mcrat_io.c:114-836) at the end of every frame with the rank idle; an outbox takes what they read off the device without
 * holding the loop up:
 *   mcrat_hip_outbox_post   stages the records of ALL slots (want_records) and/or the compacted output columns (want_output; what
 *                           mcrat_hip_get_output returns) in device memory of the outbox, in the context's stream order, and starts their copy
 *                           into pinned host memory on a stream of the outbox's own.  When it returns the photons may change: begin the next
 *                           frame.  Posting again waits for the previous post's copy.
 *   mcrat_hip_outbox_wait   blocks until the copy has landed; may be called from another thread (a writer).  The pointers address the outbox's
 *                           pinned memory and stay valid until the next post on the same outbox (columns: all 17 + type; count photons).
 * Two outboxes give a double buffer: frame f written from one while frame f+1 is posted into the other (mcrat_host_run_ranks does this). */
typedef struct mcrat_hip_outbox mcrat_hip_outbox;
int  mcrat_hip_outbox_create(mcrat_hip_ctx *ctx, mcrat_hip_outbox **box);
void mcrat_hip_outbox_destroy(mcrat_hip_outbox *box);
int  mcrat_hip_outbox_post(mcrat_hip_ctx *ctx, mcrat_hip_outbox *box, int want_records, int want_output);
int  mcrat_hip_outbox_wait(mcrat_hip_outbox *box, const mcrat_hip_photon **records, int *n_records, mcrat_hip_output_columns *cols);

/* Cyclo-synchrotron: the stages of a scatter frame with CYCLOSYNCHROTRON_SWITCH on.  They work on whatever photon
 * types the list holds; mcrat_hip_scatter_frame_cyclosynch (below) chains them as mcrat.c:706-878 does and needs a context created
 * with cyclosynchrotron_switch = 1.
 *   mcrat_hip_set_hydro_extras   the columns of struct hydro_dataframe the magnetic field needs and mcrat_hip_hydro does not carry:
 *                                dens (comoving density; B_FIELD_CALC INTERNAL_E / TOTAL_E, mc_cyclosynch.c:82-83) and B0-2
 *                                (B_FIELD_CALC == SIMULATION).  NULL pointers are skipped; after mcrat_hip_ingest_* dens is there already.
 *   mcrat_hip_absorb_cyclosynch  phAbsCyclosynch (mc_cyclosynch.c:1571-1623): every photon with weight != 0 and a cell whose comoving
 *                                frequency is at or below its cell's cyclotron frequency, and every pool photon, becomes a null
 *                                slot (setNullPhoton, photons.c:210-250); returns the count, the number of comptonised / unabsorbed
 *                                photons left (:1610-1614) and the weight of the absorbed 'i' / 'c' photons (the return value). */
typedef struct mcrat_hip_cyclosynch {
    int    b_field_calc;                  /* B_FIELD_CALC: 0 INTERNAL_E, 1 TOTAL_E, 2 SIMULATION (mcrat.h:47-49) */
    double epsilon_b;                     /* EPSILON_B */
    double rebin_e_perc, rebin_ang, rebin_ang_phi;   /* CYCLOSYNCHROTRON_REBIN_* (mcrat.h:310-321); unused by the absorption */
    int    scatt_frame_number, inj_frame_number;     /* hydro_data->scatt_frame_number / ->inj_frame_number; unused by the absorption */
} mcrat_hip_cyclosynch;
int mcrat_hip_set_hydro_extras(mcrat_hip_ctx *ctx, const double *dens, const double *B0, const double *B1, const double *B2);
/*   mcrat_hip_emit_cyclosynch_pool  photonEmitCyclosynch with inject_single_switch == 0 (mc_cyclosynch.c:1200-1460, mcrat.c:741) on the staged
 *                                frame: pool photons (type 'p') at the centres of the cells of the shell the injected photons occupy
 *                                (calcCyclosynchRLimits with cs->scatt_frame_number / inj_frame_number), each at its cell's cyclotron
 *                                frequency, the common weight adjusted until 1 <= N <= rebin_e_perc * maximum_photons; they go into
 *                                the list's null slots in slot order, the list doubling first when it has none (addToPhotonList,
 *                                photons.c:108-208; MCRAT_HIP_EINVAL where the reference exits with "Adding to the photon list has
 *                                failed").  The Poisson mean is gsl_integration_qags of the Planck photon density from 10 Hz to the
 *                                cyclotron frequency (:1276): QUADPACK's first 21-point rule, which is where QAGS stops while the
 *                                cyclotron frequency lies in the Rayleigh-Jeans tail (every cell of a GRB jet); a cell whose rule
 *                                does not meet QAGS' first-step test is integrated by bisection of the interval with the largest
 *                                error estimate, as the oracle does (oracle_cyclosynch.c, orc_qags: no epsilon extrapolation), with at
 *                                most 64 intervals.  *integrals_not_converged counts the cells that ran out of intervals; if there are
 *                                any the call returns MCRAT_HIP_EREFUSED and emits nothing. */
int mcrat_hip_emit_cyclosynch_pool(mcrat_hip_ctx *ctx, const mcrat_hip_cyclosynch *cs, double r_inj, double ph_weight, int maximum_photons,
                                   double theta_min, double theta_max, double fps, uint64_t seed, int *num_emitted, double *ph_weight_adjusted,
                                   int *integrals_not_converged);
/*   mcrat_hip_rebin_cyclosynch   rebinCyclosynchCompPhotons (mc_cyclosynch.c:246-712): every photon that is neither null, pool nor injected is
 *                                replaced by one 'k' photon per non-empty (log10 energy, polar angle of the position[, azimuth]) bin with the
 *                                bin's weight and weighted averages (bins filled in slot order, so the sums are the reference's sums);
 *                                returns the number of empty bins in *empty_bins and the two counters the reference updates
 *                                (:689-690).  MCRAT_HIP_EREFUSED on the reference's own refusals, which leave the list as it was: nothing to
 *                                rebin, more bins than max_photons, zero bins along an axis, a photon outside the histograms, too few null
 *                                slots (MCRAT_HIP_EINVAL is a bad argument). */
int mcrat_hip_rebin_cyclosynch(mcrat_hip_ctx *ctx, const mcrat_hip_cyclosynch *cs, int max_photons, int *empty_bins, int *num_cyclosynch_ph_emit,
                               int *scatt_cyclosynch_num_ph);
/*   mcrat_hip_scatter_frame_cyclosynch   the scatter-frame body of main() with CYCLOSYNCHROTRON_SWITCH on (mcrat.c:706-878, from the pool emission
 *                                to the absorption; the hydro frame is staged and the extras set): emit_pool is the condition
 *                                `(scatt_frame != scatt_framestart) || (restrt == CONTINUE)` of :707,:727,:855; inside the loop a pool
 *                                photon that photonEvent reports becomes a comptonised one and is replaced by a new pool photon of its
 *                                weight in its cell (:786-795; photonEmitCyclosynch with inject_single_switch = 1, the list doubling
 *                                when it has no null slot), and every 1000 scatterings the comptonised photons are rebinned once there
 *                                are more than max_photons of them (:797-808).  Needs a context created with
 *                                cyclosynchrotron_switch = 1 (one list per context -- many lists: mcrat_hip_pool_scatter_frames_cyclosynch --, no shared clock; mcrat_hip_run
 *                                and mcrat_hip_propagate_frame refuse such a context).  max_iterations <= 0: until the frame is over. */
typedef struct mcrat_hip_cyclosynch_counts {
    int    num_cyclosynch_ph_emit, scatt_cyclosynch_num_ph, frame_abs_cnt;   /* the counters of mcrat.c:735-878.  scatt_cyclosynch_num_ph is IN/OUT:
                                             main() carries it from one scatter frame of an injection to the next (set at mcrat.c:873, reset at
                                             :921) -- pass the previous frame's value in (0 for the first frame; zero the struct before use) */
    int    rebins;                        /* successful rebinCyclosynchCompPhotons calls */
    int    integrals_not_converged;       /* see mcrat_hip_emit_cyclosynch_pool */
    int    pad;
    double n_comptonized;                 /* mcrat.c:788,871 */
    double pool_weight;                   /* ph_weight_adjusted of the pool emission */
} mcrat_hip_cyclosynch_counts;
int mcrat_hip_scatter_frame_cyclosynch(mcrat_hip_ctx *ctx, const mcrat_hip_cyclosynch *cs, double *time_now, double remaining_time, uint64_t seed,
                                       double r_inj, double ph_weight_suggest, int max_photons, double theta_min, double theta_max, double fps,
                                       int emit_pool, long long max_iterations, mcrat_hip_frame_stats *stats, mcrat_hip_cyclosynch_counts *counts);
int mcrat_hip_num_photon_slots(const mcrat_hip_ctx *ctx);   /* photon_list->list_capacity as the device holds it (it grows when the pool does not fit) */
int mcrat_hip_absorb_cyclosynch(mcrat_hip_ctx *ctx, const mcrat_hip_cyclosynch *cs, int *num_abs_ph, int *scatt_cyclosynch_num_ph,
                                double *abs_weight);

/* photonInjection (mclib.c:9-300; mcrat.c:645) on the device, from the staged hydro frame: afterwards the context holds
 * the new photons (*num_photons of them, all of weight *ph_weight_adjusted -- the reference's min/max-photons loop of
 * mclib.c:87-136 runs on the device counts) exactly as if they had been injected on the host and handed to
 * mcrat_hip_set_photons; mcrat_hip_get_photons returns them as struct photon records (the caller allocates
 * *num_photons of them, as setPhotonList would).  spect: 'b' black body, 'w' Wien (mc.par); fps as in hydro_dataframe.
 * The random numbers are the engine's keyed source with the reference's draw order (seed: any 64-bit value, e.g. the
 * gsl_rng_get() of the rank); the Poisson counts come from the engine's own sampler. */
int mcrat_hip_inject_photons(mcrat_hip_ctx *ctx, double r_inj, double ph_weight, int min_photons, int max_photons, char spect,
                             double theta_min, double theta_max, double fps, uint64_t seed, int *num_photons, double *ph_weight_adjusted);

/* TAU_CALCULATION == TABLE, once per run (after initalizeHotCrossSection, hot_x_section.c:29-80): the table
 * getThermalCrossSection interpolates (optical_depth.c:132-149).  thermal_table is the reference's global
 * thermal_table[N_PH_E + 1][N_T + 1] (hot_x_section.c; log10 of the cross section over sigma_T, photon-energy index
 * first), the four bounds are LOG_PH_E_MIN/MAX and LOG_T_MIN/MAX of hot_x_section.h:2-10.  Creating the table
 * (createHotCrossSection, GSL Monte-Carlo integration) stays host-side work of MCRaT; mcrat_host_read_hot_cross_section
 * (mcrat_amd/host) reads the file MCRaT writes.  A lookup outside the table gets what the reference's fallback returns where that is closed-form
 * (interpolateThermalHotCrossSection -> calculateTotalThermalCrossSection, hot_x_section.c:563-599,324-356): below LOG_T_MIN -- every cell colder
 * than 5.9e5 K with the reference's bounds -- the Klein-Nishina cross section of the photon's comoving energy, or 1 when that is below LOG_PH_E_MIN
 * too (:337-340).  The remaining cases (a photon energy beyond the table at a tabulated temperature, or a temperature above LOG_T_MAX) are the
 * reference's Monte-Carlo integral of the cross section at that (energy, temperature) -- calculateTotalThermalCrossSection's 500 000 samples, here
 * from the keyed source: the integral of (pass, slot) has its own 256 substreams, so its value does not depend on which kernel or how many lanes
 * compute it (one wavefront per look-up in the rank pool's loop: milliseconds; one lane in list, shared-clock and FAST mode: a fifth of a second --
 * like the reference, which reports every such look-up on stderr, the loop treats them as rare).  mcrat_hip_frame_stats.table_fallbacks counts them. */
int mcrat_hip_set_hot_cross_section(mcrat_hip_ctx *ctx, const double *thermal_table, int n_ph_e, int n_t,
                                    double log_ph_e_min, double log_ph_e_max, double log_t_min, double log_t_max);

/* `calls` of that integral (hot_x_section.c:348: 500 000, the default); calls <= 0 only asks.  Returns the value in force (a pool's lists follow
 * their pool).  For tests, and for runs that would rather take a coarser integral than wait. */
int mcrat_hip_table_fallback_calls(mcrat_hip_ctx *ctx, int calls);

/* createHotCrossSection (hot_x_section.c:82-133) on the device: fills thermal_table[(n_ph_e + 1) * (n_t + 1)] (photon-energy
 * index first, log10 of the cross section over sigma_T) with the Monte-Carlo integrals of calculateTotalThermalCrossSection
 * (:324-357; `calls` samples per entry, 500 000 in the reference) over the Maxwell-Juttner electrons of
 * singleMaxwellJuttner (electron.c:538) and boostedCrossSection (:370-400).  The samples come from the engine's keyed random
 * source (the reference's are its GSL generator's: the table agrees within the Monte-Carlo error, ~1e-3), entry by entry
 * independent of the launch shape.  The table is returned to the host -- write it with mcrat_host_write_hot_cross_section
 * for later runs, hand it to mcrat_hip_set_hot_cross_section for this one.  221 x 81 entries x 500 000 samples: under a
 * second (a quarter of an hour on one host core). */
int mcrat_hip_create_hot_cross_section(mcrat_hip_ctx *ctx, double *thermal_table, int n_ph_e, int n_t, double log_ph_e_min,
                                       double log_ph_e_max, double log_t_min, double log_t_max, long long calls, uint64_t seed);

/* photons host -> device (after photonInjection mcrat.c:645 / readCheckpoint) and back
 * (before saveCheckpoint mcrat.c:902 / printPhotons mcrat.c:907).  NULL-photon slots
 * (type 'N', weight 0, index -1: photons.c:208) are carried through unchanged.  list_capacity must equal
 * mcrat_hip_num_photon_slots on get; a context with cyclosynchrotron_switch on also sets num_photons and num_null_photons
 * from what the list holds now (photons.c:252-275). */
int mcrat_hip_set_photons(mcrat_hip_ctx *ctx, const mcrat_hip_photon_list *list);
/* Optional, once per allocation (after allocatePhotonListMemory, photons.c:28 / reallocatePhotonListMemory): page-lock the caller's
 * array so that set_photons / get_photons move it by direct DMA at the link's rate -- pageable memory goes through the runtime's
 * staging buffers at a third of that.  Unregister before free()/realloc(). */
int mcrat_hip_register_host(mcrat_hip_ctx *ctx, void *ptr, size_t bytes);
int mcrat_hip_unregister_host(mcrat_hip_ctx *ctx, void *ptr);
int mcrat_hip_get_photons(mcrat_hip_ctx *ctx, mcrat_hip_photon_list *list);
int mcrat_hip_set_photons_soa(mcrat_hip_ctx *ctx, const mcrat_hip_photon_soa *soa);
int mcrat_hip_get_photons_soa(mcrat_hip_ctx *ctx, const mcrat_hip_photon_soa *soa);
int mcrat_hip_num_photon_slots(const mcrat_hip_ctx *ctx);

/* Several contexts on one GPU that are in the same hydro frame -- rank pools on their own HIP streams with a host thread each (INTEGRATION.md,
 * "Two pools per GPU") -- need one copy of the staged frame, its per-cell records, its lookup grid and the cross-section table, not one each:
 * after this call `ctx` reads `owner`'s.  The owner must keep that frame (no re-staging, no mcrat_hip_destroy) while others read it; staging a
 * frame on `ctx` itself (mcrat_hip_set_hydro, mcrat_hip_ingest_*) or sharing again ends the arrangement.  Same DIMENSIONS / GEOMETRY /
 * TAU_CALCULATION and device on both. */
int mcrat_hip_share_hydro(mcrat_hip_ctx *ctx, mcrat_hip_ctx *owner);

/* the loop -------------------------------------------------------------------- */
/* replaces mcrat.c:754-851 for one hydro frame: sets find_nearest_grid_switch=1,
 * runs until remaining_time is used up, updates *time_now, fills *stats.
 * `seed` is the per-frame seed; the reference draws one at mcrat.c:701
 * (gsl_rng_set(rng, gsl_rng_get(rng))) -- pass that gsl_rng_get value. */
int mcrat_hip_propagate_frame(mcrat_hip_ctx *ctx, double *time_now, double remaining_time,
                              uint64_t seed, mcrat_hip_frame_stats *stats);

/* SURVEY.md 8(b)'s `mode` argument.  MCRAT_HIP_MODE_EXACT is mcrat_hip_propagate_frame: the event-driven loop of mcrat.c:761-851,
 * one scattering per pass per list, comparable pass by pass with the reference.  MCRAT_HIP_MODE_FAST runs every photon through the
 * frame on its own clock with per-photon keyed random numbers (photons are independent within a frozen frame and exponential free
 * paths are memoryless): statistically equivalent, NOT sequence-equivalent -- spectra, scattering counts and polarisation agree with
 * the exact mode within Monte-Carlo error (tests/test_gpu_fast_mode.py), single photons do not.  fast_windows is how often
 * per frame a photon's cell and optical depth are refreshed besides after its own scatterings; the reference refreshes them whenever
 * any photon of the rank scatters, i.e. -- for its ranks of about 1000 photons -- as often as the frame has scatterings per 1000 photons.
 * fast_windows <= 0 follows that: the scatterings per 1000 photons of the LIST's previous FAST frame (a pool's list: kept on its view, so that a
 * list's photons do not depend on which other lists share the pool; mcrat_hip_fast_cadence reads and sets it -- a driver that restarts a run hands
 * the value back, or the restarted list begins at 32 again), between 8 and 2048 (32 for the
 * first frame); a fixed 8 is biased by about a per cent in frames with tens of scatterings per photon.  Works on a single list, virtual ranks or a rank pool alike (the lists do not matter to it);
 * refuses cyclo-synchrotron contexts and an attached shared clock.  stats: iterations = passes of the longest-running workgroup,
 * photon_steps = free-path draws, frame_scatt_cnt, kn_rejections, num_photons_find_new_element, not_found. */
#define MCRAT_HIP_MODE_EXACT 0
#define MCRAT_HIP_MODE_FAST  1
int mcrat_hip_propagate_frame_mode(mcrat_hip_ctx *ctx, double *time_now, double remaining_time, uint64_t seed, int mode, int fast_windows,
                                   mcrat_hip_frame_stats *stats);
/* FAST mode for the lists of a rank pool in one launch ([n_ranks] arrays, as mcrat_hip_pool_begin_frames): list r with open[r] != 0 runs its
 * frame of remaining_time[r] with its own seed, stream and list-local slot numbers in its keys -- exactly what
 * mcrat_hip_propagate_frame_mode(view r, ..., seeds[r], MCRAT_HIP_MODE_FAST, ...) gives, so a list's photons do not depend on which other
 * lists share the pool; stats[r]: the list's counters, time_now[r] + remaining_time[r].  mcrat_hip_propagate_frame_mode(pool, FAST) is this
 * for every list that holds photons with one seed and one frame time. */
int mcrat_hip_pool_propagate_frames_fast(mcrat_hip_ctx *pool, const int *open, const uint64_t *seeds, const double *time_now,
                                         const double *remaining_time, int fast_windows, mcrat_hip_frame_stats *stats);
/* the learnt refresh cadence of a list (a context holding one list, or a view of a pool): returns it; set_windows > 0 sets it first (clamped to 8..2048) */
int mcrat_hip_fast_cadence(mcrat_hip_ctx *ctx, int set_windows);

/* the same loop in pieces, for bounded runs (benchmarks, tests, progress logging):
 * begin_frame resets the per-frame state; each run executes at most max_iterations
 * passes (<= 0: until the frame time is used up) and is synchronous on return. */
int mcrat_hip_begin_frame(mcrat_hip_ctx *ctx, uint64_t seed, double time_now, double remaining_time);
int mcrat_hip_run(mcrat_hip_ctx *ctx, long long max_iterations, mcrat_hip_frame_stats *stats);

/* keep / bring back a device-side copy of the resident photons (repeatable frames without a host round trip) */
int mcrat_hip_snapshot_photons(mcrat_hip_ctx *ctx);
int mcrat_hip_restore_photons(mcrat_hip_ctx *ctx);

/* virtual-rank mode: number of lists, and the loop statistics / clock of one of them */
int mcrat_hip_num_virtual_ranks(const mcrat_hip_ctx *ctx);
int mcrat_hip_rank_stats(mcrat_hip_ctx *ctx, int rank, mcrat_hip_frame_stats *stats);

/* Rank pool: the reference's run shape on one GPU ------------------------------------------------------------------
 * MCRaT is run as 150-600 MPI ranks of 10^3 - 5*10^3 photons (Doc/mcrat_doc.tex:165-166,222): every rank owns an
 * (angle bin, injection-frame range) of mcrat.c:139-164,457-479, a Poisson-sized photon list (mclib.c:87-136), its own
 * generator (mcrat.c:99-103,701), its own clock, and its own mc_proc_<rank>.h5 / mc_chkpt_<rank>.dat
 * (mcrat_io.c:199-200,871-903); ranks never talk inside the loop.  A pool lets ONE process per GPU adopt R such ranks:
 *   mcrat_hip_pool_create   turns `pool` into R lists of up to slots_per_rank slots each (slots_per_rank >= the longest list
 *                           a rank will ever hold: max_photons, times the doublings cyclo-synchrotron emission may need)
 *   mcrat_hip_pool_rank     the *view* of list `rank`: a context whose photons, loop state and clock are that list's and
 *                           whose hydro frame is the pool's.  Every per-list entry point of this header works on a view
 *                           exactly as on a context of its own holding only that list with rng_stream `rng_stream`:
 *                           mcrat_hip_set_photons / _inject_photons (the rank joins), _begin_frame (its seed and clock),
 *                           _ph_minmax, _scatt_stats, _get_output, _get_photons_range, _get_photons ... A view is
 *                           destroyed with mcrat_hip_destroy or with its pool; set_hydro / ingest go to the pool.
 *   mcrat_hip_run(pool)     the loop of mcrat.c:761-851 for every list whose view has an open frame (mcrat_hip_begin_frame
 *                           on the view), all lists in ONE launch, one workgroup per list (rank_loop_kernel) -- each list
 *                           sees exactly the arithmetic and the random numbers of a context of its own.  The stats are the
 *                           totals; mcrat_hip_frame_statistics(view) gives a list's own.  mcrat_hip_begin_frame(pool)
 *                           opens a frame for all lists at once with one seed and clock.
 *   mcrat_hip_pool_summaries  phMinMax, phScattStats, averagePhotonEnergy and printPhotons' photon count for every list in
 *                           one launch (lists that do not exist: list_capacity 0).
 * mcrat_host_run_ranks (mcrat_amd/host) is main()'s rank logic on top of this; INTEGRATION.md shows the edit. */
typedef struct mcrat_hip_rank_summary {
    double min_r, max_r, min_theta, max_theta;   /* phMinMax, mclib.c:1465 */
    double avg_scatt, avg_r;                     /* phScattStats, mclib.c:1385 */
    double avg_energy;                           /* averagePhotonEnergy, mclib.c:1358 */
    int    max_scatt, min_scatt;
    int    num_output;                           /* photons with weight != 0: what printPhotons writes (mcrat_io.c:137-181) */
    int    list_capacity;                        /* photon_list->list_capacity of this rank (0: no list) */
} mcrat_hip_rank_summary;
/* mcrat_hip_config.profile = 1: the summed duration (HIP events on the context's stream) and the number of the loop kernel's launches since
 * the context was created -- mcrat_hip_run's and mcrat_hip_pool_scatter_frames_cyclosynch's (there: rank_loop_kernel with the hook inside;
 * emission, rebinning and absorption are NOT in it).  MCRAT_HIP_ESTATE without profile. */
int mcrat_hip_profile_totals(const mcrat_hip_ctx *ctx, double *loop_kernel_ms, long long *loop_kernel_launches);
int mcrat_hip_pool_create(mcrat_hip_ctx *pool, int n_ranks, int slots_per_rank);
int mcrat_hip_pool_rank(mcrat_hip_ctx *pool, int rank, uint32_t rng_stream, mcrat_hip_ctx **view);
int mcrat_hip_pool_summaries(mcrat_hip_ctx *pool, mcrat_hip_rank_summary *out /* [n_ranks] */);
/* the same for hundreds of lists at once, so that a frame of the pool costs a handful of launches, not a handful per list:
 *   mcrat_hip_pool_begin_frames  mcrat_hip_begin_frame for every list with open[r] != 0, each with its own seed and clock ([n_ranks] arrays)
 *   mcrat_hip_pool_frame_stats   mcrat_hip_frame_statistics of every list ([n_ranks]; lists without a frame: what they last had)
 *   mcrat_hip_pool_layout        the lists' places in the pool's own slot numbering: list r occupies the slots r * slots_per_rank ...,
 *                                which mcrat_hip_get_photons_range / mcrat_hip_get_output on the POOL context address (slots beyond a
 *                                list's length hold no photon: type 0, weight 0) -- one transfer for all lists' records or columns */
/* CYCLOSYNCHROTRON_SWITCH on (the pool created from a context with cyclosynchrotron_switch = 1): mcrat_hip_scatter_frame_cyclosynch for
 * every list with lists[r].open != 0 -- pool emission per list, then the loops of ALL lists in the same launches (a list leaves the loop only
 * for the hook of mcrat.c:786-808: its scattered pool photon converted and replaced, the list doubling inside its window of the pool when
 * it has no null slot left, the rebinning trigger), rebinning and absorption at the end.  cs carries the build's switches (b_field_calc,
 * epsilon_b, rebin_*); the frame numbers come per list.  slots_per_rank of mcrat_hip_pool_create must allow for the doublings
 * (MCRAT_HIP_ENOMEM otherwise).  stats / counts: [n_ranks], per list; stats may be NULL. */
typedef struct mcrat_hip_pool_cs_list {
    int      open;
    int      emit_pool;                   /* (scatt_frame != scatt_framestart) || (restrt == CONTINUE), mcrat.c:707 */
    int      scatt_frame_number, inj_frame_number;
    uint64_t seed;
    double   time_now, remaining_time;
    double   r_inj, ph_weight_suggest, theta_min, theta_max;
} mcrat_hip_pool_cs_list;
int mcrat_hip_pool_scatter_frames_cyclosynch(mcrat_hip_ctx *pool, const mcrat_hip_cyclosynch *cs, int max_photons, double fps,
                                             const mcrat_hip_pool_cs_list *lists, mcrat_hip_frame_stats *stats, mcrat_hip_cyclosynch_counts *counts);
/* photonInjection (mcrat_hip_inject_photons) for every list of the pool with inject != 0, a handful of launches for all of them: each list gets
 * exactly the photons mcrat_hip_inject_photons(view, ...) would give it (same keys, same order); num_photons / ph_weight_adjusted are filled.
 * lists: [n_ranks]; the lists' views must exist (mcrat_hip_pool_rank). */
typedef struct mcrat_hip_pool_inject_list {
    int      inject;
    char     spect;                        /* 'b' or 'w' */
    int      min_photons, max_photons;
    double   r_inj, ph_weight, theta_min, theta_max;
    uint64_t seed;
    int      num_photons;                  /* out */
    double   ph_weight_adjusted;           /* out */
    int      status;                       /* out: MCRAT_HIP_OK, or why THIS list was not injected (the other lists are; the call returns the first
                                            *      such code).  Arguments are validated for all lists before any window is touched. */
} mcrat_hip_pool_inject_list;
int mcrat_hip_pool_inject_photons(mcrat_hip_ctx *pool, double fps, mcrat_hip_pool_inject_list *lists);
/* mcrat_hip_set_photons for `count` lists of the pool at once -- a CONTINUE run's restart, every adopted rank's checkpointed list (readCheckpoint,
 * Src/mcrat_io.c:1011): the records cross PCIe in one copy and reach their windows in one launch.  rank_of[j]: the pool rank of lists[j] (its view
 * exists, no rank twice).  Each list ends up exactly as mcrat_hip_set_photons(view, &lists[j]) leaves it; everything is validated first. */
int mcrat_hip_pool_set_photons(mcrat_hip_ctx *pool, int count, const int *rank_of, const mcrat_hip_photon_list *lists);
int mcrat_hip_pool_begin_frames(mcrat_hip_ctx *pool, const int *open, const uint64_t *seeds, const double *time_now, const double *remaining_time);
int mcrat_hip_pool_frame_stats(mcrat_hip_ctx *pool, mcrat_hip_frame_stats *out /* [n_ranks] */);
int mcrat_hip_pool_layout(const mcrat_hip_ctx *pool, int *n_ranks, int *slots_per_rank);
/* The frame queue: every list of the pool through SEVERAL hydro frames in ONE launch.  The reference's ranks are asynchronous processes, each in its
 * own frame loop (Src/mcrat.c:457-479, :566-934: nothing makes rank A wait for rank B at the end of a hydro frame); one launch per hydro frame does,
 * and a third of such a launch is the tail of its last lists.  Here the loop kernel's workgroups are persistent and take
 * (frame, list) items in frame-major order: a list that is through frame f starts f + 1 while others are still in f.  Every list runs exactly the
 * frames mcrat_hip_pool_begin_frames + mcrat_hip_run would have given it one at a time -- the same seeds, clocks, passes, photons, bit for bit.
 * All arrays are [n_frames * n_ranks], frame-major (item f * n_ranks + r); a list's open frames must be consecutive (a rank joins at its injection
 * frame and stays, mcrat.c:566-700).  chain_clock: a list's clock carries over from its previous frame of the call -- time_now = where that frame ended,
 * remaining_time = frame_end - time_now (mcrat.c:757 with frame_end = (scatt_frame + increment_scatt_frame) / fps) -- and time_now / remaining_time
 * are used for its first frame only; else every frame takes its own time_now and remaining_time.  restore_each_frame: every frame starts from
 * the lists as mcrat_hip_snapshot_photons saved them (benchmarks: the same work every frame).  hydro: NULL, or [n_frames] contexts holding the
 * staged hydro frame of frame f (mcrat_hip_set_hydro / mcrat_hip_ingest_* on them, same device and switches as the pool; NULL entry: the pool's own
 * frame; the contexts keep their frames staged while the call runs) -- a real run stages frame f + 1 for the slab the photons can reach from frame
 * f (phMinMax widened by c / fps) before the launch: a list in frame f + 1 then finds its photons in the cells of THAT frame, exactly as after
 * mcrat_hip_share_hydro(pool, hydro[f + 1]) + one launch per frame.  stats: [n_frames * n_ranks], what
 * mcrat_hip_pool_frame_stats would have reported after each frame (lists that sat a frame out: zeros).  Not with the cyclo-synchrotron switch (its
 * hook needs the host between passes).  Afterwards the pool is as after the last frame's mcrat_hip_run. */
typedef struct mcrat_hip_frame_plan {
    int n_frames;
    int chain_clock;
    int restore_each_frame;
    int capture_frames;                   /* 1: the lists as every frame but the last leaves them are kept for mcrat_hip_pool_select_frame */
    const int      *open;
    const uint64_t *seeds;
    const double   *time_now;
    const double   *remaining_time;
    const double   *frame_end;            /* chain_clock only */
    mcrat_hip_ctx *const *hydro;          /* NULL: every frame through the pool's own staged frame */
} mcrat_hip_frame_plan;
int mcrat_hip_pool_run_frames(mcrat_hip_ctx *pool, const mcrat_hip_frame_plan *plan, mcrat_hip_frame_stats *stats /* [n_frames * n_ranks] */);

/* The outputs of the frames of a plan (phScattStats, saveCheckpoint, printPhotons: mcrat.c:881-915 need every list as ITS frame left it, and in a
 * queue launch a list is moved on by its next frame at once).  With plan.capture_frames = 1 each list is copied aside at the end of every frame but
 * the last (the lists that were open in it -- every other slot of a capture reads as an empty one, weight 0; one copy of the pool per such frame in HBM); mcrat_hip_pool_select_frame(pool, f) then makes the pool's
 * read entry points -- mcrat_hip_pool_summaries, mcrat_hip_outbox_post, mcrat_hip_get_photons_range, mcrat_hip_get_output -- look at frame f's copy,
 * until mcrat_hip_pool_select_frame(pool, -1) (the live lists = the last frame).  While a frame is selected mcrat_hip_run and
 * mcrat_hip_pool_run_frames refuse.  The captures are valid until the next mcrat_hip_pool_run_frames. */
int mcrat_hip_pool_select_frame(mcrat_hip_ctx *pool, int frame);

/* function-granular A/B entry points (one kernel each, for parity tests against the
 * reference functions): the findContainingHydroCell + calcMeanFreePath half of an
 * iteration and the photonEvent half.  mcrat_amd/host/mcrat_hip_host.h wraps them in shims with the reference's
 * own signatures (mclib.h:8-29) for A/B runs function by function. */
int mcrat_hip_step_locate_sample(mcrat_hip_ctx *ctx, int find_nearest_block_switch);
int mcrat_hip_frame_statistics(mcrat_hip_ctx *ctx, mcrat_hip_frame_stats *stats);   /* the loop counters so far (synchronises) */
int mcrat_hip_update_photon_position(mcrat_hip_ctx *ctx, double t);   /* updatePhotonPosition, mclib.c:1054-1100, on the resident photons */
int mcrat_hip_step_event(mcrat_hip_ctx *ctx, mcrat_hip_frame_stats *stats);

/* ONE photon list split over several GPUs, ONE clock ------------------------------------------------------------
 * The reference never does this: its ranks own disjoint photons AND disjoint clocks (mcrat.c:761-851 runs per rank;
 * SURVEY.md 8e).  This mode keeps the event order of a single list -- what one rank holding all the photons would
 * compute -- while the N-wide work is spread over `world` GPUs: every GPU owns the contiguous global slots
 * [slot_base, slot_base + n), with n the capacity of its own mcrat_hip_set_photons list.  A round is
 *
 *     mcrat_hip_shared_clock_propose(ctx);          findContainingHydroCell + calcMeanFreePath on the own slots, then
 *                                                    the own earliest candidates (the argsort prefix of mclib.c:702-712)
 *                                                    with the photon data photonEvent needs -> send buffer
 *     all-gather send -> recv over the GPUs          the min-reduction of SURVEY.md 8e; bytes_per_rank each; the host's
 *                                                    job: ncclAllGather / MPI_Allgather on the context's stream
 *     mcrat_hip_shared_clock_resolve(ctx);          photonEvent (mclib.c:1107) on the merged candidates, identically on
 *                                                    every GPU; the owner of the scattered photon stores it
 *
 * and mcrat_hip_shared_clock_poll tells when the frame is over (rounds after that are no-ops, so polling every few
 * dozen rounds is enough).  All contexts of the group use the same rng_stream, seed, time_now and remaining_time;
 * the photons end up bit-identical to a single context holding the concatenated list.  In this mode
 * last_scattered_index in the stats is a GLOBAL slot, num_photons_find_new_element / not_found count the own slots,
 * and `rescans` counts the rounds that continued an undecided iteration.
 * send/recv: device buffers of bytes_per_rank and world * bytes_per_rank bytes; pass NULL to let the library
 * allocate them (read them back with mcrat_hip_shared_clock_buffers); with world == 1 recv may equal send. */
size_t mcrat_hip_shared_clock_bytes_per_rank(void);
int mcrat_hip_shared_clock_attach(mcrat_hip_ctx *ctx, int world, int rank, long long slot_base, void *send, void *recv);
int mcrat_hip_shared_clock_buffers(mcrat_hip_ctx *ctx, void **send, void **recv);
/* The exchange done by the GPUs themselves (one-shot peer writes and a local wait instead of a collective launched by the host):
 *   mcrat_hip_shared_clock_attach_device   attach as above with library-owned, fine-grained (peer-coherent) buffers: a receive buffer of
 *                                          2 x world proposals (the rounds alternate between its halves) and world stamps;
 *   mcrat_hip_shared_clock_peer_buffers    their addresses and sizes, for the caller to hand to the other ranks (hipIpcGetMemHandle between
 *                                          processes, the pointers themselves between contexts of one process or with peer access enabled);
 *   mcrat_hip_shared_clock_set_peers       [world] arrays with every rank's receive buffer and stamps as THIS process addresses them (the own
 *                                          entries are ignored);
 *   mcrat_hip_shared_clock_exchange        between propose and resolve: a kernel writes this rank's proposal into every peer's buffer and
 *                                          stamps it (_push), a second one waits for the stamps of all ranks (_wait; bounded: a peer that
 *                                          never arrives makes the next poll fail with MCRAT_HIP_EHIP instead of hanging the GPU).  A wait
 *                                          that gives up copies nothing and parks the loop: the photons stay at the last pass every rank
 *                                          completed, and the rounds still queued (a captured batch) return at once instead of each
 *                                          spinning its own budget;
 *   mcrat_hip_shared_clock_reset_exchange  after such a failure: clears this rank's give-up word, round number and stamps.  Every rank calls
 *                                          it, then the ranks synchronise (any barrier), then begin_frame as usual.
 * Since round 3 the push runs at the end of the propose kernel and the wait at the head of the resolve kernel (two launches less per round); the
 * three exchange calls then only check the state and return, so a host loop written as propose / exchange / resolve stays as it is.  The peers
 * must therefore be set before the first propose.  MCRAT_HIP_SC_FOLD=0 (environment, read at attach) keeps push and wait as kernels of their own.
 * All on the context's stream and without per-round arguments (the round number lives on the device, the wait kernel copies the round's
 * proposals to the fixed place resolve reads), so propose / exchange / resolve capture into a hipGraph like the collective.  Every rank must run the same
 * number of rounds (rounds after the frame's end are no-ops but still exchange), as with the all-gather. */
int mcrat_hip_shared_clock_attach_device(mcrat_hip_ctx *ctx, int world, int rank, long long slot_base);
int mcrat_hip_shared_clock_peer_buffers(mcrat_hip_ctx *ctx, void **recv, size_t *recv_bytes, void **flags, size_t *flags_bytes);
int mcrat_hip_shared_clock_set_peers(mcrat_hip_ctx *ctx, void *const *peer_recv, void *const *peer_flags);
int mcrat_hip_shared_clock_exchange_push(mcrat_hip_ctx *ctx);
int mcrat_hip_shared_clock_exchange_wait(mcrat_hip_ctx *ctx);
int mcrat_hip_shared_clock_exchange(mcrat_hip_ctx *ctx);
int mcrat_hip_shared_clock_reset_exchange(mcrat_hip_ctx *ctx);
int mcrat_hip_shared_clock_propose(mcrat_hip_ctx *ctx);
int mcrat_hip_shared_clock_resolve(mcrat_hip_ctx *ctx);
int mcrat_hip_shared_clock_poll(mcrat_hip_ctx *ctx, int *frame_done, mcrat_hip_frame_stats *stats);   /* synchronises the stream */
int mcrat_hip_shared_clock_finish(mcrat_hip_ctx *ctx, mcrat_hip_frame_stats *stats);   /* apply the pending advance, final stats */

/* The random stream as an INPUT ("tape").  MCRaT draws everything from one sequential gsl_rng_ranlxs0 stream
 * (Src/mcrat.c:99-103, reseeded per frame :701).  The engine's default source is its own keyed generator (every comparison in tests/ uses it on both
 * sides); with a tape the caller supplies the stream instead -- the doubles MCRaT's generator returned (gsl_rng_type::get_double, in [0,1)), recorded
 * in call order by tools/ref_harness from the UNMODIFIED reference -- and the loop consumes it exactly as MCRaT does:
 *   per pass, one gsl_rng_uniform_pos for every slot with a cell (nearest_block_index != -1), in ascending slot order   (Src/mclib.c:646-675)
 *   then photonEvent's draws for each candidate it tries, in its call order                       (Src/electron.c:81,196,217-233; Src/mcrat_scattering.c:519-574)
 * gsl_rng_uniform_pos skips zeros, gsl_ran_gaussian (polar method) takes as many pairs as it needs: as GSL publishes them.  The photons after a frame
 * can then be held against MCRaT's own, photon for photon ("Pinning parity with your GSL").  STATUS: the tape path is validated against
 * this repository's CPU restatement of MCRaT only (tests/test_gpu_tape.py: engine == oracle on the same tape, zeros on the tape included) -- no tape
 * recorded from MCRaT itself exists yet (the reference needs GSL, which the development image lacks; tools/ref_harness has not been compiled
 * against it).  The product build's arithmetic differs from the reference's IEEE sequence in the last place or two (reciprocals and roots by
 * v_rcp / v_rsq + Newton steps, physics.hpp), so a comparison decided within a few ulp can fall the other way than MCRaT's; a maintainer who wants
 * correctly rounded operands under every decision builds the kernels with -DMCRAT_IEEE_ARITH=1 (slower; physics.hpp).  A validation mode: one list per context
 * (refused on rank pools, virtual ranks, the shared clock and with the cyclo-synchrotron switch), the free-path draws of a pass taken by one
 * workgroup.  uniforms == NULL or n == 0 returns to the keyed source.  While a tape is set the seed of begin_frame / propagate_frame is ignored and
 * the tape is read on across frames (MCRaT's reseeding is part of the recorded stream).
 *   mcrat_hip_rng_tape_position   entries read so far; *ran_out != 0 if the loop needed more than the tape holds (results then meaningless). */
int mcrat_hip_set_rng_tape(mcrat_hip_ctx *ctx, const double *uniforms, long long n);
int mcrat_hip_rng_tape_position(mcrat_hip_ctx *ctx, long long *position, int *ran_out);

/* The device functions of the path, one at a time, on arrays -- for function-level parity tests against the reference functions
 * (tests/test_gpu_functions.py; the loop does not use this entry).  in / out: n rows of doubles, row layouts:
 *   KN_CROSS_SECTION       in  energy_ratio                          out sigma / sigma_T        kleinNishinaCrossSection, mcrat_scattering.c:597
 *   LORENTZ_BOOST_PHOTON   in  beta[3], p[4]                         out p'[4]                  lorentzBoost(.., 'p'), mclib.c:302 (zeroNorm'ed)
 *   LORENTZ_BOOST_ELECTRON in  beta[3], p[4]                         out p'[4]                  lorentzBoost(.., 'e')
 *   STOKES_ROTATION        in  v[3], v_ph[3], v_ph_boosted[3], s[4]  out s[4]                   stokesRotation, mcrat_scattering.c:103
 *   THERMAL_ELECTRON       in  temp, ph[4]                           out el[4]                  singleThermalElectron, electron.c:70
 *   THERMAL_ELECTRON_WAVE  the same through the wavefront-wide sampler the event walk uses (one wavefront per row)
 *   ELECTRON_AND_SCATTER   in  temp, ph[4], s[4]                     out el[4], ph'[4], s'[4], occurred    singleThermalElectron then
 *                                                                    singleScatter (mcrat_scattering.c:151) on one stream; STOKES as the context's
 * Random numbers of row i: the engine's event stream {seed, iteration i, slot 0, the context's rng_stream} (rng.hpp). */
#define MCRAT_HIP_FN_KN_CROSS_SECTION       1
#define MCRAT_HIP_FN_LORENTZ_BOOST_PHOTON   2
#define MCRAT_HIP_FN_LORENTZ_BOOST_ELECTRON 3
#define MCRAT_HIP_FN_STOKES_ROTATION        4
#define MCRAT_HIP_FN_THERMAL_ELECTRON       5
#define MCRAT_HIP_FN_THERMAL_ELECTRON_WAVE  6
#define MCRAT_HIP_FN_ELECTRON_AND_SCATTER   7
int mcrat_hip_eval_function(mcrat_hip_ctx *ctx, int fn, int n, const double *in, double *out, uint64_t seed);

/* per-frame reductions on the resident photons ------------------------------- */
int mcrat_hip_ph_minmax(mcrat_hip_ctx *ctx, double *min_r, double *max_r, double *min_theta, double *max_theta); /* mclib.c:1465 */
int mcrat_hip_scatt_stats(mcrat_hip_ctx *ctx, int *max_scatt, int *min_scatt, double *avg_scatt, double *avg_r); /* mclib.c:1385 */
int mcrat_hip_avg_energy(mcrat_hip_ctx *ctx, double *erg);                                                    /* mclib.c:1358 */

/* introspection used by bench.py / tests -------------------------------------- */
int mcrat_hip_synchronize(mcrat_hip_ctx *ctx);
size_t mcrat_hip_device_bytes(const mcrat_hip_ctx *ctx);   /* HBM held by the context */
int mcrat_hip_lookup_cell(mcrat_hip_ctx *ctx, int n, const double *a0, const double *a1, const double *a2, int *cell_out); /* device cell search == findContainingBlock geometry.c:350 */

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MCRAT_HIP_H */
