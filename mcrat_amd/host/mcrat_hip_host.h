/*
 * mcrat_hip_host.h -- host-side C (gcc) on top of the C ABI of include/mcrat_hip.h.
 *
 * MCRaT's host code is C and stays C (BASELINE.json north_star): this file mirrors, for the accelerated path,
 * the pieces of the reference's driver a maintainer would otherwise have to re-type:
 *   - mcrat_host_scatter_frame(): the body of the scatter-frame loop of Src/mcrat.c:754-892 -- remaining_time
 *     bookkeeping (:758), the while loop (:761-851, now one library call), phScattStats (:881) and the log
 *     lines of :883-890, byte for byte in the reference's format;
 *   - mcrat_host_read_mcpar(): the positional mc.par grammar of readMcPar (Src/mcrat_io.c:1136-1237), so the
 *     run-size surface stays compatible (fps, last frame, domains, angle bins, spectrum, photon counts, i/c);
 *   - mcrat_host_read_hot_cross_section(): the thermal cross-section table file of TAU_CALCULATION == TABLE builds;
 *   - mcrat_host_read_pluto(): PLUTO's grid.out / dbl.out / .dbl files into the buffers mcrat_hip_ingest_pluto takes.
 * It links against libmcrat_hip.so only; nothing here computes photon physics on the CPU.
 */
#ifndef MCRAT_HIP_HOST_H
#define MCRAT_HIP_HOST_H

#include <stdio.h>
#include "mcrat_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* what readMcPar fills (Src/mcrat_io.c:1136): block 1 into hydro_dataframe, the rest into main()'s locals */
typedef struct mcrat_host_mcpar {
    double fps;
    int    last_frame;
    double r0_domain[2], r1_domain[2], r2_domain[2];
    double theta_jmin, theta_j;      /* degrees, as the reference keeps them */
    int    n_theta_j;
    int   *frm0;                     /* [n_theta_j] first injection frame per angle bin                */
    int   *frm2;                     /* [n_theta_j] frm0 + number of frames (mcrat_io.c:1201)          */
    double *inj_radius;              /* [n_theta_j]                                                    */
    char   spect;                    /* 'b' or 'w'                                                     */
    int    min_photons, max_photons;
    char   restart;                  /* 'i' or 'c'                                                     */
} mcrat_host_mcpar;

/* 0 on success, -1 if the file cannot be opened, -2 on a malformed file */
int  mcrat_host_read_mcpar(const char *path, mcrat_host_mcpar *out);
void mcrat_host_free_mcpar(mcrat_host_mcpar *p);

/* thermal_hot_x_section.dat as createHotCrossSection writes it and readHotCrossSection reads it
 * (Src/hot_x_section.c:107-133, 208-254): header lines up to a line of dashes, then rows
 * "i <TAB> j <TAB> log10(energy) <TAB> log10(theta) <TAB> log10(cross section)".  Fills table[(n_ph_e+1)*(n_t+1)]
 * (photon-energy index first, the layout mcrat_hip_set_hot_cross_section takes); every entry must be present.
 * 0 on success, -1 if the file cannot be opened, -2 on a malformed or incomplete file or an index outside the bounds
 * (the reference exits with "The bounds of the input file exceed what MCRaT has been compiled with"). */
int mcrat_host_read_hot_cross_section(const char *path, double *table, int n_ph_e, int n_t);
/* the same file as createHotCrossSection writes it (hot_x_section.c:109-132): four header lines, then one row per entry,
 * "%d\t%d\t%g\t%g\t%15.10g" -- for a table made by mcrat_hip_create_hot_cross_section.  0, or -1 if the file cannot be written. */
int mcrat_host_write_hot_cross_section(const char *path, const double *table, int n_ph_e, int n_t, double log_ph_e_min,
                                       double log_ph_e_max, double log_t_min, double log_t_max);

/* A PLUTO .dbl frame from disk: what readPluto holds after readGridFile, readDblOutFile and its fread
 * (Src/mclib_pluto.c:852-1128), ready for mcrat_hip_ingest_pluto.
 *   grid_out   PLUTO's grid.out: '#' header whose "# X1: [ a, b], N point(s), ..." lines carry the cell counts, then per
 *              axis a count line and N rows "index left right"; centre = (left+right)/2, width = right-left (:951-971)
 *   dbl_out    PLUTO's dbl.out: its first line ends with the variable names in file order (:990-1056)
 *   dbl_file   the frame, num_vars blocks of nx*ny*nz doubles (single_file layout; mcrat_host_pluto_name builds the name)
 *   three_dimensional   DIMENSIONS == THREE: read the X3 axis (otherwise nz = 1)
 * The variables are picked by name (rho, vx1, vx2, vx3, prs; :1147-1212); grid.rho etc. point into `data`.
 * 0 on success, -1 if a file cannot be opened, -2 on a malformed file, a short .dbl file or a missing variable. */
typedef struct mcrat_host_pluto {
    mcrat_hip_pluto_grid grid;
    int     num_vars;
    char  **var_names;       /* [num_vars] */
    double *axes;            /* the six 1-D arrays grid.x1 ... grid.dx3 point into */
    double *data;            /* [num_vars][nz][ny][nx], the file as read */
} mcrat_host_pluto;
int  mcrat_host_read_pluto(const char *grid_out, const char *dbl_out, const char *dbl_file, int three_dimensional,
                           double l_scale, double d_scale, double p_scale, mcrat_host_pluto *out);
void mcrat_host_free_pluto(mcrat_host_pluto *p);
/* modifyPlutoName (mclib_pluto.c:803-850): prefix + frame zero-padded to four digits + ".dbl" */
void mcrat_host_pluto_name(char *out, size_t n, const char *prefix, int frame);

/* saveCheckpoint (Src/mcrat_io.c:838-1009): mc_chkpt_<angle_rank>.dat in `dir`, byte for byte the reference's layout --
 * angle_size (int), restart flag (char), frame, frame2 (int) and, unless this is the frame after the last one,
 * scatt_frame (int), time_now (double), list_capacity (int) -- followed by the struct photon records (176 B each).  The
 * previous file is kept as <file>_old except when scatt_frame == frame, where it is removed first (:849,:901,:951).
 * With ctx != NULL the records stream from the device in pieces (mcrat_hip_get_photons_range): no host copy of the list;
 * with ctx == NULL they are `list`'s.  cyclosynchrotron_switch (CYCLOSYNCHROTRON_SWITCH of mcrat_input.h) on: every comptonised
 * photon 'k' with weight != 0 becomes an unabsorbed one 'c' before its record is written, and stays so in the list -- on the
 * device (mcrat_hip_convert_comptonized) or in `list` -- as the reference converts it in place (:896-900, :951-955, :991-995).
 * Returns 0, or 1 if the file cannot be written (as the reference). */
int mcrat_host_save_checkpoint(const char *dir, int frame, int frame2, int scatt_frame, double time_now, mcrat_hip_ctx *ctx,
                               mcrat_hip_photon_list *list, int list_capacity, int last_frame, int angle_rank, int angle_size,
                               int cyclosynchrotron_switch);
/* readCheckpoint (mcrat_io.c:1011-1134): fills the scalars with the reference's "+1" conventions; for a 'c' file
 * list->photons is malloc'ed with list_capacity records (free() it) ready for mcrat_hip_set_photons.  A missing file is
 * not an error (restart = 'i', scatt_framestart = framestart, :1127-1131).  Returns 0, -2 on a truncated file. */
int mcrat_host_read_checkpoint(const char *dir, mcrat_hip_photon_list *list, int *frame2, int *framestart, int *scatt_framestart,
                               char *restart, double *time, int angle_rank, int *angle_size);

/* printPhotons (Src/mcrat_io.c:114-836; mcrat_hip_host_h5.c, built into libmcrat_hip_host_h5.so when the HDF5 C library is
 * present): appends the frame's photons (weight != 0, compacted on the device by mcrat_hip_get_output) to group "<frame>" of
 * <dir>mc_proc_<angle_rank>.h5 as the datasets P0-3, [COMV_P0-3], R0-2, [S0-3], NS, PW, [PT] -- chunked, unlimited, extended
 * when the group already exists, as the reference writes them.  comv_switch / stokes_switch / save_type: COMV_SWITCH,
 * STOKES_SWITCH, SAVE_TYPE of mcrat_input.h.  mcrat_host_h5_read reads one of them back (tests; dirFileMerge's per-dataset read). */
int mcrat_host_print_photons(mcrat_hip_ctx *ctx, int frame, const char *dir, int angle_rank, int comv_switch, int stokes_switch,
                             int save_type, FILE *fPtr);
int mcrat_host_h5_read(const char *file, const char *group, const char *name, int is_char, void *data, int cap, int *n);
/* HDF5's default build is not thread-safe, and mcrat_host_run_ranks with asynchronous output has its writer thread in print_photons (frame F)
 * while the calling thread is in get_hydro (frame F + 1) -- with several pools per process, several of each.  Every function of
 * libmcrat_hip_host_h5 that enters HDF5 (mcrat_host_print_photons, _print_photon_arrays, _h5_read, _read_flash, _read_chombo) therefore takes ONE
 * process-wide recursive lock.  A caller whose own get_hydro / print_photons callbacks call HDF5 directly must bracket those calls with
 * mcrat_host_h5_lock() / mcrat_host_h5_unlock() (or run with sync_output = 1 and one pool per process).  mcrat_host_h5_threadsafe(): what
 * H5is_library_threadsafe says about the library this build links. */
void mcrat_host_h5_lock(void);
void mcrat_host_h5_unlock(void);
int  mcrat_host_h5_threadsafe(void);
/* the HDF5 half alone, on arrays the caller holds (cols->count photons, NULL columns skipped): what mcrat_host_run_ranks calls per rank
 * with slices of ONE mcrat_hip_get_output of the whole pool */
int mcrat_host_print_photon_arrays(const mcrat_hip_output_columns *cols, int frame, const char *dir, int angle_rank, FILE *fPtr);

/* The HDF5 reads of the two HDF5-based readers, and nothing else of them (same file, mcrat_hip_host_h5.c):
 *   mcrat_host_read_flash    readAndDecimate's H5Dread calls (mclib_flash.c:95-197): "coordinates", "block size", "node type",
 *                            "velx", "vely", "dens", "pres" -> the buffers mcrat_hip_ingest_flash takes
 *   mcrat_host_read_chombo   readPlutoChombo's (mclib_pluto.c:44-430): attributes num_levels, num_components, component_<k>;
 *                            per level_<i>: "boxes", "data:offsets=0", "data:datatype=0", prob_domain, ref_ratio, dx, logr,
 *                            domBeg1-3, g_x2stretch, g_x3stretch -> mcrat_hip_ingest_chombo's
 * mcrat_host_flash_name is modifyFlashName (mclib_flash.c:15-58).  0, -1 if the file cannot be opened, -2 if a dataset or
 * attribute is missing or has an unexpected shape. */
typedef struct mcrat_host_flash {
    mcrat_hip_flash_blocks blocks;       /* owns its arrays: mcrat_host_free_flash */
} mcrat_host_flash;
int  mcrat_host_read_flash(const char *file, double l_scale, double d_scale, double p_scale, mcrat_host_flash *out);
void mcrat_host_free_flash(mcrat_host_flash *f);
void mcrat_host_flash_name(char *out, size_t n, const char *prefix, int frame);
typedef struct mcrat_host_chombo {
    mcrat_hip_chombo frame;              /* points into the members below */
    mcrat_hip_chombo_level *levels;
    char  **var_names;
    double *data;
} mcrat_host_chombo;
int  mcrat_host_read_chombo(const char *file, int three_dimensional, double l_scale, double d_scale, double p_scale, mcrat_host_chombo *out);
void mcrat_host_free_chombo(mcrat_host_chombo *h);

/* One scatter frame on the device (replaces Src/mcrat.c:754-892 between getHydroData and saveCheckpoint).
 *   list       caller-owned photon list; uploaded, propagated, downloaded in place
 *   hydro      the frame getHydroData just produced
 *   time_now   in/out, as in main()
 *   scatt_frame, increment_scatt_frame, fps   give remaining_time = (scatt_frame+increment)/fps - time_now (:758)
 *   seed       per-frame seed (the reference reseeds with gsl_rng_get at :701; pass that value)
 *   fPtr       the rank's log file (may be NULL)
 * Returns 0 or a negative MCRAT_HIP_E* code; *stats (may be NULL) receives the loop statistics. */
int mcrat_host_scatter_frame(mcrat_hip_ctx *ctx, mcrat_hip_photon_list *list, const mcrat_hip_hydro *hydro,
                             double *time_now, int scatt_frame, int increment_scatt_frame, double fps,
                             uint64_t seed, FILE *fPtr, mcrat_hip_frame_stats *stats);

/* The same loop body with the photons resident on the device from injection to the last frame (no list crosses PCIe between
 * frames): phMinMax on the device (mcrat.c:704) -> the caller's reader with the photons' slab (mcrat.c:721; `get_hydro` is
 * expected to end in mcrat_hip_ingest_flash / _pluto / _chombo or mcrat_hip_set_hydro for scatt_frame) -> the loop for this
 * hydro frame -> phScattStats and the log lines.  saveCheckpoint / printPhotons afterwards: mcrat_host_save_checkpoint(ctx, ...)
 * and mcrat_host_print_photons.  inj_radius and the domains are getHydroData's other arguments (mcrat_io.h:26, mc.par). */
typedef int (*mcrat_host_get_hydro_fn)(void *user, mcrat_hip_ctx *ctx, int scatt_frame, const mcrat_hip_slab *slab);
int mcrat_host_scatter_frame_resident(mcrat_hip_ctx *ctx, mcrat_host_get_hydro_fn get_hydro, void *user, double inj_radius,
                                      const double r0_domain[2], const double r1_domain[2], const double r2_domain[2],
                                      double *time_now, int scatt_frame, int increment_scatt_frame, double fps, uint64_t seed,
                                      FILE *fPtr, mcrat_hip_frame_stats *stats);

/* ---- rank pool: one process per GPU adopts R of the reference's MPI ranks (mcrat_hip_pool_*, include/mcrat_hip.h) ----
 * What main() decides per MPI rank between mcrat.c:116 and :483 -- its angle bin (:139-164), its block of injection frames
 * (:457-479), its directory, log file and generator -- lives in one mcrat_host_rank per adopted rank; mcrat_host_split_ranks
 * fills them for an 'i' run the way MPI_Comm_split and the frame arithmetic would.  mcrat_host_run_ranks is then the rest of
 * main() (:609-924) for all of them at once: for every rank the outer loop over its injection frames and the inner loop
 * over the scatter frames, with the SAME per-rank results and files -- mc_proc_<angle_id>.h5, mc_chkpt_<angle_id>.dat and
 * mc_output_<angle_id>.log in the rank's mc_dir -- but scheduled by hydro frame: ranks are independent (they never talk
 * inside the loop), so the driver takes every rank's k-th injection batch together, walks the hydro frames
 * once per batch, lets each rank join at its own injection frame, reads every hydro frame once for all ranks that are in
 * it (the slab is the union of the ranks' phMinMax slabs, mcrat.c:704-721) and propagates all their lists in one launch.
 * The photons stay on the device from injection to the last frame; what crosses PCIe per frame is what the reference
 * writes per frame (checkpoint records, printPhotons' columns). */
typedef struct mcrat_host_rank {
    /* what the rank is (filled by mcrat_host_split_ranks, or by hand) */
    int      myid;                       /* world rank this entry stands for (log lines only) */
    int      angle_id, angle_procs;      /* rank and size in its angle communicator (mcrat.c:147-148); angle_procs is saveCheckpoint's angle_size */
    char     mc_dir[1024];               /* mcrat.c:155 */
    double   theta_jmin_thread, theta_jmax_thread;   /* radians, mcrat.c:152-153 */
    double   inj_radius;                 /* mcrat.c:159 */
    double   ph_weight_suggest;          /* mcrat.c:162 (ph_weight_default) */
    int      framestart, frm2;           /* first and last injection frame of this rank, mcrat.c:472-483 */
    uint64_t rng_seed;                   /* the rank's generator: seed k of the rank is mcrat_host_rank_seed(rng_seed, k), standing for the
                                            k-th gsl_rng_get of mcrat.c:701 (and of photonInjection's draws) */
    uint32_t rng_stream;                 /* its stream id in the engine's keyed source */
    FILE    *fPtr;                       /* mc_output_<angle_id>.log, may be NULL */
    /* a CONTINUE run (mc.par 'c'): what mcrat_host_read_checkpoint returned for this rank (mcrat.c:487) -- restrt 'c', the list (uploaded
     * at the start; the caller keeps ownership), scatt_framestart and time_now_start; framestart / frm2 as the checkpoint gives them.
     * restrt 0 or 'i': the rank starts by injecting at framestart */
    char     restrt;
    int      scatt_framestart;
    double   time_now_start;
    const mcrat_hip_photon_list *restart_list;
    int      fast_cadence_start;         /* MCRAT_HIP_MODE_FAST with fast_windows <= 0: the list's learnt refresh cadence when the run was interrupted
                                            (fast_cadence below, as the previous run left it) -- handed back so that a restarted run goes on as the
                                            uninterrupted one would have; 0: none known (a fresh list starts at 32 windows) */
    /* progress (driver-owned; zero it before the first call) */
    mcrat_hip_ctx *view;
    int      frame, scatt_frame;         /* the loop variables of mcrat.c:609,:664 */
    double   time_now;
    int      num_photons;                /* photon_list.num_photons after the injection */
    double   ph_weight;                  /* the adjusted weight of the injection */
    long long seeds_drawn;
    long long frame_scatt_cnt_total;     /* scatterings over all frames (for callers' accounting) */
    int      fast_cadence;               /* FAST mode: the list's learnt refresh cadence after its last frame (mcrat_hip_fast_cadence) */
    int      scatt_cyclosynch_num_ph;    /* main()'s counter of comptonised photons, carried over the scatter frames of an injection (mcrat.c:873,921) */
    int      first_scatt_frame;          /* scatt_framestart of the running batch: no pool emission in that frame (mcrat.c:707) */
    long long cyclosynch_emitted_total, cyclosynch_absorbed_total;
    int      state;                      /* 0 waiting for its injection frame, 1 scattering, 2 batch finished, 3 all batches done,
                                            4 restarted from a checkpoint, waiting for scatt_framestart */
} mcrat_host_rank;

uint64_t mcrat_host_rank_seed(uint64_t rng_seed, long long k);

/* the rank/angle split of an INITALIZE run for the world ranks first_rank .. first_rank + n_adopt - 1 of numprocs
 * (mcrat.c:116-164,457-483): numprocs must be a multiple of par->n_theta_j.  base_dir is FILEPATH MC_PATH.  rng_seed /
 * rng_stream are set to (base_seed, myid): every rank its own stream (the reference gives all ranks GSL's default seed,
 * SURVEY.md section 0 fact 7; distinct streams are what a user wants and what the engine's virtual ranks always used).
 * 0, or -1 on a bad split. */
int mcrat_host_split_ranks(const mcrat_host_mcpar *par, int numprocs, int first_rank, int n_adopt, const char *base_dir,
                           double ph_weight_default, uint64_t base_seed, mcrat_host_rank *out);

typedef int (*mcrat_host_print_arrays_fn)(const mcrat_hip_output_columns *cols, int frame, const char *dir, int angle_rank, FILE *fPtr);
typedef struct mcrat_host_pool_config {
    double fps;                          /* mc.par */
    int    last_frm;
    double r0_domain[2], r1_domain[2], r2_domain[2];
    char   spect;
    int    min_photons, max_photons;
    int    slots_per_rank;               /* 0: max_photons */
    mcrat_host_get_hydro_fn get_hydro;   /* the reader: stages hydro frame `scatt_frame` for `slab` on the POOL context (getHydroData) */
    void  *user;
    int    write_checkpoints;            /* saveCheckpoint per rank and frame, mcrat.c:902 (0: skip -- benchmarks).  With cyclosynchrotron_switch the
                                          * conversion saveCheckpoint performs on the list ('k' -> 'c', mcrat_io.c:896-900) happens every frame either way:
                                          * it is what the next frame's absorption and the PT column expect */
    mcrat_host_print_arrays_fn print_photons;    /* mcrat_host_print_photon_arrays (the HDF5 build), or NULL to skip printPhotons */
    int    comv_switch, stokes_switch, save_type;
    int    max_frames;                   /* > 0: stop after this many hydro frames in total (tests, benchmarks) */
    /* Hydro frames are stepped by ONE at the constant `fps` (time_now = frame / fps).  The legacy RIKEN schedule of mcrat.c:551-556,609-621 --
     * increment 10 and fps 1 from frame 3000 on -- is not implemented here (the RIKEN reader is out of scope): such data go
     * through the single-context shims, which take increment_scatt_frame as an argument. */
    /* CYCLOSYNCHROTRON_SWITCH ON (the pool context created with cyclosynchrotron_switch = 1): the frame is mcrat.c:706-878 per rank --
     * pool emission from a rank's second scatter frame on (:707), the hook inside the loop, rebinning, absorption -- through
     * mcrat_hip_pool_scatter_frames_cyclosynch; the slab read for a frame also covers the emission shell (calcCyclosynchRLimits,
     * :708-720); saveCheckpoint converts 'k' -> 'c' on the device.  The reader callback must hand the magnetic-field columns over
     * (mcrat_hip_set_hydro_extras) when B_FIELD_CALC needs them.  slots_per_rank must allow for the lists' doublings. */
    int    cyclosynchrotron_switch;
    mcrat_hip_cyclosynch cs;             /* B_FIELD_CALC, EPSILON_B, CYCLOSYNCHROTRON_REBIN_* (the frame numbers are filled per rank) */
    /* MCRAT_HIP_MODE_EXACT (0): the event-driven loop of mcrat.c:761-851 per rank.  MCRAT_HIP_MODE_FAST: every photon on its own clock
     * (mcrat_hip_pool_propagate_frames_fast; statistically equivalent, see mcrat_hip.h) -- each rank still with its own per-frame seed and
     * stream, so its files do not depend on which other ranks the process adopted.  Not with cyclosynchrotron_switch. */
    int    mode, fast_windows;
    /* saveCheckpoint (mcrat.c:902) and printPhotons (:907) of frame f are written by a writer thread from pinned host memory while frame f+1
     * propagates (mcrat_hip_outbox_*): two outboxes of the pool's records + output columns each (176 B + 137 B per slot, device and pinned host).
     * The files and their bytes are those of the synchronous path; a rank's log file gets the writer's lines ("Making checkpoint file", printPhotons'
     * own) when they are written, i.e. possibly after the next frame's first lines.  An error of the writer ends the run at the next frame.
     * sync_output = 1: the reference's order -- the loop waits for the files (records staged in pieces of 2^20 slots: the choice for pools whose
     * records do not fit pinned memory twice).  output_threads: threads that share the ranks' checkpoint files (0: 4); printPhotons is always
     * called from one thread, one rank after the other, and only after every checkpoint of the frame has been written (the reference's order,
     * mcrat.c:902-915: a failed checkpoint means no mc_proc data of that frame for any rank).  CONCURRENCY: with sync_output = 0 the caller's
     * print_photons (writer thread, frame F) and get_hydro (calling thread, frame F + 1) run at the same time; the library's own HDF5 functions
     * serialise on one lock, callbacks that use HDF5 themselves take mcrat_host_h5_lock (above). */
    int    sync_output, output_threads;
    /* Optional: a second context (the pool's switches and device) for the NEXT hydro frame.  With it, in EXACT mode without the cyclo-synchrotron switch,
     * the driver stages frame F + 1 beside frame F (get_hydro is called with this context and a slab widened by c / fps) and takes every list
     * through both frames in one launch (mcrat_hip_pool_run_frames: a list that is through F goes on in F + 1 at once) whenever no rank joins at
     * F + 1; each frame's checkpoint and mc_proc data are written from the lists as that frame left them.  Files and logs are those of the
     * one-frame-per-launch run (tests/test_gpu_rank_pool_host.py). */
    mcrat_hip_ctx *stage_ctx;
    /* out */
    long long hydro_frames_read;         /* get_hydro calls */
    long long launches;                  /* mcrat_hip_run / mcrat_hip_pool_run_frames calls */
    long long two_frame_launches;        /* of which took two hydro frames */
    double ms_propagate, ms_hydro, ms_output;   /* wall time of the calling thread in the loop, in the reader callback, in checkpoint + printPhotons
                                                 * (asynchronous output: posting the outbox, waiting for a free one, the final drain) */
    double ms_output_writer;                    /* asynchronous output: what the writer thread spent on the frames' files (copy wait + writing) ... */
    double ms_output_blocked;                   /* ... and how much of that the calling thread had to wait for: the rest was hidden behind the loop */
} mcrat_host_pool_config;
/* pool: a context of the run's DIMENSIONS / GEOMETRY / STOKES_SWITCH; the driver creates the pool layout and the views.
 * Returns 0 or the first negative MCRAT_HIP_E* code (1: a checkpoint could not be written, as saveCheckpoint). */
int mcrat_host_run_ranks(mcrat_hip_ctx *pool, mcrat_host_rank *ranks, int n_ranks, mcrat_host_pool_config *cfg);
/* What the file system alone asks for a frame's checkpoints, measured by the same C: per frame, n_files files of bytes_each bytes in `dir`, each
 * renamed to <name>_old, opened "wb", written and closed (saveCheckpoint's sequence, mcrat_io.c:846-900, without a byte of photon data moving),
 * shared by `threads` threads.  *ms_per_frame: wall time per frame.  0, or 1 when a file could not be written. */
int mcrat_host_output_floor(const char *dir, int n_files, size_t bytes_each, int frames, int threads, double *ms_per_frame);

/* ---- one list over several GPUs, one clock (mcrat_hip_shared_clock_*, include/mcrat_hip.h): the host loop in C ----------------
 * One process per GPU; every process calls this with the same seed, time_now and remaining_time and its own rank / slot_base
 * (the number of slots on the lower ranks, even).  A round is propose -> exchange -> resolve; `exchange` is the all-gather of the
 * round's proposals (bytes_per_rank from every rank, in rank order, into recv) ON THE CONTEXT'S STREAM -- mcrat_host_allgather_rccl
 * (mcrat_hip_host_rccl.c) is ncclAllGather; an MPI build passes MPI_Allgather on device pointers after a stream synchronise.  The
 * state is polled every rounds_per_poll rounds (rounds after the frame's end are no-ops).  The context must have its photons and
 * hydro frame set; it is attached on the first call (library-owned exchange buffers).  0 or a negative MCRAT_HIP_E* code. */
typedef int (*mcrat_host_allgather_fn)(void *user, const void *send, void *recv, size_t bytes_per_rank, void *stream);
int mcrat_host_shared_clock_frame(mcrat_hip_ctx *ctx, int world, int rank, long long slot_base, mcrat_host_allgather_fn exchange, void *user,
                                  void *stream, double *time_now, double remaining_time, uint64_t seed, int rounds_per_poll,
                                  mcrat_hip_frame_stats *stats);
/* the exchange done by the GPUs themselves (mcrat_hip_shared_clock_attach_device / _set_peers in mcrat_hip.h): pass this as `exchange` with
 * user = the context; no collective is launched, a kernel writes the proposal into every peer's buffer and a second one waits for theirs */
int mcrat_host_exchange_device(void *user /* mcrat_hip_ctx * */, const void *send, void *recv, size_t bytes_per_rank, void *stream);
/* (in libmcrat_hip_host_rccl.so, which links HIP) a device allocation as a 64-byte handle another process can map, for the peer buffers:
 * hipIpcGetMemHandle / hipIpcOpenMemHandle / hipIpcCloseMemHandle.  0 or MCRAT_HIP_EHIP. */
int  mcrat_host_ipc_export(void *device_ptr, unsigned char handle[64]);
int  mcrat_host_ipc_import(const unsigned char handle[64], void **device_ptr);
void mcrat_host_ipc_close(void *device_ptr);
/* mcrat_hip_host_rccl.c (libmcrat_hip_host_rccl.so, links librccl): the exchange as ncclAllGather -- `user` points at the caller's
 * ncclComm_t --, and the same frame with its rounds captured in a hipGraph (the forced first round is launched eagerly, then
 * rounds_per_graph rounds of {propose kernels, ncclAllGather, resolve kernel} replay as one graph launch between two polls).
 * mcrat_host_rccl_comm_single makes a one-rank communicator on the current device (tests, one-GPU runs). */
int mcrat_host_allgather_rccl(void *user, const void *send, void *recv, size_t bytes_per_rank, void *stream);
int mcrat_host_shared_clock_frame_graph(mcrat_hip_ctx *ctx, int world, int rank, long long slot_base, void *nccl_comm, void *stream,
                                        double *time_now, double remaining_time, uint64_t seed, int rounds_per_graph,
                                        mcrat_hip_frame_stats *stats);
int  mcrat_host_rccl_comm_single(void **nccl_comm);
void mcrat_host_rccl_comm_destroy(void *nccl_comm);

/* ---- A/B shims: the reference's loop functions with their own argument order (Src/mclib.h:8-29) ----------------
 * For checking the engine against the CPU functions one call at a time inside MCRaT's own loop (mcrat.c:761-851):
 * replace `findContainingHydroCell(&photon_list, &hydrodata, sw, rng, fPtr)` by
 * `mcrat_ab_findContainingHydroCell(&pl, &hy, sw, &ab, fPtr)` and so on.  gsl_rng* becomes mcrat_ab_rng*: the engine's
 * random source is keyed by (frame seed, loop iteration), both held by the context, so the handle carries the context
 * instead of a generator state.  Within a frame the photons live on the device; every shim copies them back into the
 * caller's list afterwards (host <- device per call: a debugging path; production is mcrat_host_scatter_frame).
 * The engine fuses findContainingHydroCell and calcMeanFreePath into one kernel: the first shim runs it (and already
 * fills time_to_scatter), the second fills sorted_indexes as the argsort of mclib.c:702-712 does (ties by slot).
 * The reference's functions report no status; the shims leave the MCRAT_HIP_E* code of the last call in last_rc. */
typedef struct mcrat_ab_rng {
    mcrat_hip_ctx *ctx;
    int last_rc;
    long long relocated_seen;      /* internal: counters already reported */
    long long scatt_seen;
} mcrat_ab_rng;

/* once per scatter frame, where mcrat.c:754-758 sets find_nearest_grid_switch = 1 and remaining_time: stages the hydro
 * frame and the photon list and opens the frame with the seed main() draws at mcrat.c:701 */
int    mcrat_ab_begin_frame(mcrat_ab_rng *rng, mcrat_hip_ctx *ctx, const mcrat_hip_photon_list *ph, const mcrat_hip_hydro *hydro,
                            uint64_t seed, double time_now, double remaining_time);
int    mcrat_ab_findContainingHydroCell(mcrat_hip_photon_list *ph, const mcrat_hip_hydro *hydro, int find_nearest_block_switch,
                                        mcrat_ab_rng *rng, FILE *fPtr);                                   /* mclib.c:436 */
void   mcrat_ab_calcMeanFreePath(mcrat_hip_photon_list *ph, const mcrat_hip_hydro *hydro, mcrat_ab_rng *rng, FILE *fPtr);   /* mclib.c:617 */
double mcrat_ab_photonEvent(mcrat_hip_photon_list *ph, double dt_max, const mcrat_hip_hydro *hydro, int *scattered_ph_index,
                            int *frame_scatt_cnt, int *frame_abs_cnt, mcrat_ab_rng *rng, FILE *fPtr);    /* mclib.c:1107 */
void   mcrat_ab_updatePhotonPosition(mcrat_hip_photon_list *ph, double t, mcrat_ab_rng *rng, FILE *fPtr);  /* mclib.c:1054 (+ the handle) */

#ifdef __cplusplus
}
#endif
#endif
