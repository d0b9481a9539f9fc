/* mcrat_hip_host_h5.c -- printPhotons (Src/mcrat_io.c:114-836) on top of mcrat_hip_get_output: the per-frame photon
 * datasets of mc_proc_<angle_rank>.h5, same file, group and dataset names, chunking and append behaviour, so that
 * dirFileMerge / MERGE / ProcessMCRaT read the files unchanged.  Plain C99 + the HDF5 C library (the one dependency of
 * this file; the rest of the host mirror does not need it).  The arrays come from the device already compacted
 * (photons with weight != 0, slot order) and live on the heap -- the reference keeps 18 arrays of num_photons doubles on
 * the stack (:130-131), which is what limits it to ~10^5 photons per rank with default stack sizes. */
#include "mcrat_hip_host.h"

#include <hdf5.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ONE lock around everything in this file that enters the HDF5 library.  HDF5's default build is not thread-safe (H5is_library_threadsafe), and
 * mcrat_host_run_ranks' writer thread calls print_photons for frame F while the calling thread reads frame F + 1 (get_hydro -> mcrat_host_read_flash /
 * mcrat_host_read_chombo) -- with two pools per process two writers and two readers: four threads in a library that keeps global ID tables.  The lock
 * is recursive (mcrat_host_print_photons -> mcrat_host_print_photon_arrays) and exported: a caller whose own get_hydro / print_photons callbacks
 * call HDF5 directly brackets those calls with mcrat_host_h5_lock / _unlock. */
static pthread_mutex_t g_h5_mutex;
static pthread_once_t g_h5_once = PTHREAD_ONCE_INIT;
static void h5_mutex_init(void)
{
    pthread_mutexattr_t a;
    pthread_mutexattr_init(&a);
    pthread_mutexattr_settype(&a, PTHREAD_MUTEX_RECURSIVE);
    pthread_mutex_init(&g_h5_mutex, &a);
    pthread_mutexattr_destroy(&a);
}
void mcrat_host_h5_lock(void)
{
    pthread_once(&g_h5_once, h5_mutex_init);
    pthread_mutex_lock(&g_h5_mutex);
}
void mcrat_host_h5_unlock(void) { pthread_mutex_unlock(&g_h5_mutex); }
int mcrat_host_h5_threadsafe(void)
{
    hbool_t ts = 0;
    mcrat_host_h5_lock();
    const herr_t e = H5is_library_threadsafe(&ts);
    mcrat_host_h5_unlock();
    return (e >= 0 && ts) ? 1 : 0;
}

/* one dataset of the frame's group: created chunked and unlimited on first use (:252-262), otherwise extended by `n` and
 * written at the old end (:402-424) */
static int put(hid_t group, const char *name, hid_t type, const void *data, hsize_t n)
{
    hsize_t dims[1] = {n}, maxdims[1] = {H5S_UNLIMITED}, old[1] = {0}, size[1], offset[1];
    herr_t st = 0;
    if (H5Lexists(group, name, H5P_DEFAULT) <= 0) {
        hid_t prop = H5Pcreate(H5P_DATASET_CREATE);
        H5Pset_chunk(prop, 1, dims);
        hid_t space = H5Screate_simple(1, dims, maxdims);
        hid_t dset = H5Dcreate2(group, name, type, space, H5P_DEFAULT, prop, H5P_DEFAULT);
        if (dset < 0) st = -1;
        else { st = H5Dwrite(dset, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data); H5Dclose(dset); }
        H5Sclose(space);
        H5Pclose(prop);
        return st < 0 ? -1 : 0;
    }
    hid_t dset = H5Dopen2(group, name, H5P_DEFAULT);
    if (dset < 0) return -1;
    hid_t space = H5Dget_space(dset);
    H5Sget_simple_extent_dims(space, old, NULL);
    H5Sclose(space);
    size[0] = old[0] + n;
    st = H5Dset_extent(dset, size);
    hid_t fspace = H5Dget_space(dset);
    offset[0] = old[0];
    H5Sselect_hyperslab(fspace, H5S_SELECT_SET, offset, NULL, dims, NULL);
    hid_t mspace = H5Screate_simple(1, dims, NULL);
    if (st >= 0) st = H5Dwrite(dset, type, mspace, fspace, H5P_DEFAULT, data);
    H5Sclose(mspace);
    H5Sclose(fspace);
    H5Dclose(dset);
    return st < 0 ? -1 : 0;
}

/* the HDF5 half of printPhotons (mcrat_io.c:183-836) on arrays the caller holds: cols->count photons; NULL columns are not written */
int mcrat_host_print_photon_arrays(const mcrat_hip_output_columns *cols, int frame, const char *dir, int angle_rank, FILE *fPtr)
{
    if (!cols || !dir || cols->count < 0) return MCRAT_HIP_EINVAL;
    const int n = cols->count;
    if (n == 0) return MCRAT_HIP_OK;                              /* an empty H5Dcreate with chunk 0 is an error; nothing to write */
    if (fPtr) fprintf(fPtr, "num_ph %d\nAllocated weight to be %d values large and other arrays to be %d\n", n, n, n);
    char file[2000], group[64];
    snprintf(file, sizeof file, "%s%s%d%s", dir, "mc_proc_", angle_rank, ".h5");
    snprintf(group, sizeof group, "%d", frame);
    mcrat_host_h5_lock();
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fcreate(file, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);           /* :199-206 */
    if (f < 0) f = H5Fopen(file, H5F_ACC_RDWR, H5P_DEFAULT);
    if (f < 0) { mcrat_host_h5_unlock(); return MCRAT_HIP_EINVAL; }
    hid_t g = (H5Lexists(f, group, H5P_DEFAULT) > 0) ? H5Gopen2(f, group, H5P_DEFAULT) : H5Gcreate2(f, group, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    int bad = g < 0;
    static const char *names[17] = {"P0", "P1", "P2", "P3", "COMV_P0", "COMV_P1", "COMV_P2", "COMV_P3", "R0", "R1", "R2",
                                    "S0", "S1", "S2", "S3", "NS", "PW"};
    const double *col[17] = {cols->p0, cols->p1, cols->p2, cols->p3, cols->comv_p0, cols->comv_p1, cols->comv_p2, cols->comv_p3,
                             cols->r0, cols->r1, cols->r2, cols->s0, cols->s1, cols->s2, cols->s3, cols->num_scatt, cols->weight};
    for (int k = 0; !bad && k < 17; k++)
        if (col[k]) bad = put(g, names[k], H5T_NATIVE_DOUBLE, col[k], (hsize_t)n) != 0;
    if (!bad && cols->type) bad = put(g, "PT", H5T_NATIVE_CHAR, cols->type, (hsize_t)n) != 0;
    if (g >= 0) H5Gclose(g);
    H5Fclose(f);
    mcrat_host_h5_unlock();
    return bad ? MCRAT_HIP_EINVAL : MCRAT_HIP_OK;
}

int mcrat_host_print_photons(mcrat_hip_ctx *ctx, int frame, const char *dir, int angle_rank, int comv_switch, int stokes_switch,
                             int save_type, FILE *fPtr)
{
    if (!ctx || !dir) return MCRAT_HIP_EINVAL;
    mcrat_hip_output_columns o;
    memset(&o, 0, sizeof o);
    int rc = mcrat_hip_get_output(ctx, &o);                       /* the count */
    if (rc) return rc;
    const int n = o.count;
    if (n == 0) return MCRAT_HIP_OK;                              /* an empty H5Dcreate with chunk 0 is an error; nothing to write */
    const int ncol = 17;
    double *buf = (double *)malloc(sizeof(double) * (size_t)n * ncol);
    char *type = (char *)malloc((size_t)n);
    if (!buf || !type) { free(buf); free(type); return MCRAT_HIP_ENOMEM; }
    double **slot[17] = {&o.p0, &o.p1, &o.p2, &o.p3, &o.comv_p0, &o.comv_p1, &o.comv_p2, &o.comv_p3, &o.r0, &o.r1, &o.r2,
                         &o.s0, &o.s1, &o.s2, &o.s3, &o.num_scatt, &o.weight};
    for (int k = 0; k < ncol; k++) {
        const int is_comv = k >= 4 && k < 8, is_stokes = k >= 11 && k < 15;
        *slot[k] = ((is_comv && !comv_switch) || (is_stokes && !stokes_switch)) ? NULL : buf + (size_t)k * n;
    }
    o.type = save_type ? type : NULL;
    o.count = n;
    rc = mcrat_hip_get_output(ctx, &o);
    if (rc) { free(buf); free(type); return rc; }
    rc = mcrat_host_print_photon_arrays(&o, frame, dir, angle_rank, fPtr);
    free(buf);
    free(type);
    return rc;
}

/* a dataset of a frame back into memory (what dirFileMerge does per dataset, mcrat_io.c:1500-1560); *n receives its length;
 * up to cap values are copied (data may be NULL to ask for the length only) */
int mcrat_host_h5_read(const char *file, const char *group, const char *name, int is_char, void *data, int cap, int *n)
{
    if (!file || !group || !name || !n) return MCRAT_HIP_EINVAL;
    mcrat_host_h5_lock();
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fopen(file, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) { mcrat_host_h5_unlock(); return MCRAT_HIP_EINVAL; }
    int rc = MCRAT_HIP_EINVAL;
    hid_t g = H5Gopen2(f, group, H5P_DEFAULT);
    if (g >= 0) {
        hid_t d = H5Dopen2(g, name, H5P_DEFAULT);
        if (d >= 0) {
            hsize_t dims[1] = {0};
            hid_t space = H5Dget_space(d);
            H5Sget_simple_extent_dims(space, dims, NULL);
            H5Sclose(space);
            *n = (int)dims[0];
            rc = MCRAT_HIP_OK;
            if (data && (int)dims[0] <= cap && dims[0] > 0)
                rc = H5Dread(d, is_char ? H5T_NATIVE_CHAR : H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0 ? MCRAT_HIP_EINVAL : MCRAT_HIP_OK;
            H5Dclose(d);
        }
        H5Gclose(g);
    }
    H5Fclose(f);
    mcrat_host_h5_unlock();
    return rc;
}

/* ---- the HDF5 file reads of readAndDecimate (mclib_flash.c:95-197) and readPlutoChombo (mclib_pluto.c:44-430) -------------
 * Only the reads: everything those functions do afterwards is mcrat_hip_ingest_flash / mcrat_hip_ingest_chombo. */
void mcrat_host_flash_name(char *out, size_t n, const char *prefix, int frame)
{
    snprintf(out, n, "%s%04d", prefix, frame);          /* modifyFlashName, mclib_flash.c:15-58: FILEPATH FILEROOT + four digits */
}

static void *read_dataset(hid_t file, const char *name, hid_t memtype, size_t elem, hsize_t dims_out[4], int *rank_out)
{
    hid_t d = H5Dopen2(file, name, H5P_DEFAULT);
    if (d < 0) return NULL;
    hid_t sp = H5Dget_space(d);
    hsize_t dims[8] = {0};
    const int rank = H5Sget_simple_extent_ndims(sp);
    void *buf = NULL;
    if (rank >= 1 && rank <= 8) {
        H5Sget_simple_extent_dims(sp, dims, NULL);
        size_t count = 1;
        for (int k = 0; k < rank; k++) count *= (size_t)dims[k];
        buf = malloc(elem * (count ? count : 1));
        if (buf && H5Dread(d, memtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) { free(buf); buf = NULL; }
        for (int k = 0; k < 4; k++) dims_out[k] = k < rank ? dims[k] : 1;
        if (rank_out) *rank_out = rank;
    }
    H5Sclose(sp);
    H5Dclose(d);
    return buf;
}

void mcrat_host_free_flash(mcrat_host_flash *f)
{
    if (!f) return;
    free((void *)f->blocks.coordinates); free((void *)f->blocks.block_size); free((void *)f->blocks.node_type);
    free((void *)f->blocks.velx); free((void *)f->blocks.vely); free((void *)f->blocks.dens); free((void *)f->blocks.pres);
    memset(f, 0, sizeof *f);
}

static int read_flash_locked(const char *file, double l_scale, double d_scale, double p_scale, mcrat_host_flash *out)
{
    if (!file || !out) return -2;
    memset(out, 0, sizeof *out);
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fopen(file, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) return -1;
    hsize_t dc[4], db[4], dn[4], dv[4];
    mcrat_hip_flash_blocks *b = &out->blocks;
    b->coordinates = (const double *)read_dataset(f, "coordinates", H5T_NATIVE_DOUBLE, sizeof(double), dc, NULL);
    b->block_size = (const double *)read_dataset(f, "block size", H5T_NATIVE_DOUBLE, sizeof(double), db, NULL);
    b->node_type = (const int *)read_dataset(f, "node type", H5T_NATIVE_INT, sizeof(int), dn, NULL);
    const char *names[4] = {"velx", "vely", "dens", "pres"};
    const double **slots[4] = {&b->velx, &b->vely, &b->dens, &b->pres};
    int ok = b->coordinates && b->block_size && b->node_type && dc[0] > 0 && db[0] == dc[0] && dn[0] == dc[0] && dc[1] >= 2 && db[1] >= 2 &&
             dc[0] <= 0x7fffffffu / 64;
    for (int k = 0; k < 4 && ok; k++) {
        *slots[k] = (const double *)read_dataset(f, names[k], H5T_NATIVE_DOUBLE, sizeof(double), dv, NULL);
        ok = *slots[k] && dv[0] == dc[0] && dv[1] * dv[2] * dv[3] == 64;          /* PROP_DIM1 x PROP_DIM2 x PROP_DIM3, mclib_flash.c:10-12 */
    }
    H5Fclose(f);
    if (!ok) { mcrat_host_free_flash(out); return -2; }
    b->n_blocks = (int)dc[0];
    b->coord_stride = (int)dc[1];
    b->bsize_stride = (int)db[1];
    b->l_scale = l_scale; b->d_scale = d_scale; b->p_scale = p_scale;
    return 0;
}

void mcrat_host_free_chombo(mcrat_host_chombo *h)
{
    if (!h) return;
    for (int i = 0; h->levels && i < h->frame.num_levels; i++) { free((void *)h->levels[i].boxes); free((void *)h->levels[i].box_offsets); }
    for (int i = 0; h->var_names && i < h->frame.num_vars; i++) free(h->var_names[i]);
    free(h->levels); free(h->var_names); free(h->data);
    memset(h, 0, sizeof *h);
}

static int attr_read(hid_t obj, const char *name, hid_t type, void *out)
{
    hid_t a = H5Aopen(obj, name, H5P_DEFAULT);
    if (a < 0) return -1;
    const herr_t st = H5Aread(a, type, out);
    H5Aclose(a);
    return st < 0 ? -1 : 0;
}

static int read_chombo_locked(const char *file, int three_dimensional, double l_scale, double d_scale, double p_scale, mcrat_host_chombo *out)
{
    if (!file || !out) return -2;
    memset(out, 0, sizeof *out);
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fopen(file, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) return -1;
    const int nd = three_dimensional ? 3 : 2, bi = 2 * nd;
    int rc = -2, num_levels = 0, num_vars = 0;
    /* the box compound of mclib_pluto.c:48-58, packed as the ints mcrat_hip_chombo_level.boxes expects */
    static const char *members3[6] = {"lo_i", "lo_j", "lo_k", "hi_i", "hi_j", "hi_k"}, *members2[4] = {"lo_i", "lo_j", "hi_i", "hi_j"};
    hid_t box_t = H5Tcreate(H5T_COMPOUND, sizeof(int) * (size_t)bi);
    for (int k = 0; k < bi; k++) H5Tinsert(box_t, three_dimensional ? members3[k] : members2[k], sizeof(int) * (size_t)k, H5T_NATIVE_INT);
    if (attr_read(f, "num_levels", H5T_NATIVE_INT, &num_levels) || attr_read(f, "num_components", H5T_NATIVE_INT, &num_vars) ||
        num_levels <= 0 || num_vars <= 0 || num_levels > 64 || num_vars > 256)
        goto done;
    out->frame.num_levels = num_levels;
    out->frame.num_vars = num_vars;
    out->levels = (mcrat_hip_chombo_level *)calloc((size_t)num_levels, sizeof(mcrat_hip_chombo_level));
    out->var_names = (char **)calloc((size_t)num_vars, sizeof(char *));
    if (!out->levels || !out->var_names) goto done;
    for (int k = 0; k < num_vars; k++) {                                    /* component_<k>, :98-113 */
        char an[64];
        snprintf(an, sizeof an, "component_%d", k);
        hid_t a = H5Aopen(f, an, H5P_DEFAULT);
        if (a < 0) goto done;
        hid_t ft = H5Aget_type(a);
        const size_t sdim = H5Tget_size(ft) + 1;
        out->var_names[k] = (char *)calloc(sdim, 1);
        hid_t mt = H5Tcopy(H5T_C_S1);
        H5Tset_size(mt, sdim);
        const herr_t st = out->var_names[k] ? H5Aread(a, mt, out->var_names[k]) : -1;
        H5Tclose(mt); H5Tclose(ft); H5Aclose(a);
        if (st < 0) goto done;
    }
    long long total = 0;
    for (int i = 0; i < num_levels; i++) {                                  /* sizes first (:128-155), level 0 first in all_data */
        char gn[64];
        snprintf(gn, sizeof gn, "level_%d", i);
        hid_t g = H5Gopen2(f, gn, H5P_DEFAULT);
        if (g < 0) goto done;
        hid_t d = H5Dopen2(g, "data:datatype=0", H5P_DEFAULT);
        hsize_t dims[1] = {0};
        if (d >= 0) { hid_t sp = H5Dget_space(d); H5Sget_simple_extent_dims(sp, dims, NULL); H5Sclose(sp); H5Dclose(d); }
        H5Gclose(g);
        if (d < 0) goto done;
        out->levels[i].data_len = (long long)dims[0];
        total += (long long)dims[0];
    }
    out->data = (double *)malloc(sizeof(double) * (size_t)(total > 0 ? total : 1));
    if (!out->data) goto done;
    long long offset = 0;
    for (int i = 0; i < num_levels; i++) {                                  /* :349-430 */
        mcrat_hip_chombo_level *L = &out->levels[i];
        char gn[64];
        snprintf(gn, sizeof gn, "level_%d", i);
        hid_t g = H5Gopen2(f, gn, H5P_DEFAULT);
        if (g < 0) goto done;
        int bad = 0;
        hid_t d = H5Dopen2(g, "data:datatype=0", H5P_DEFAULT);
        bad |= d < 0 || H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, out->data + offset) < 0;
        if (d >= 0) H5Dclose(d);
        hsize_t dims[4];
        L->box_offsets = (const int *)read_dataset(g, "data:offsets=0", H5T_NATIVE_INT, sizeof(int), dims, NULL);
        hsize_t nb[4];
        L->boxes = (const int *)read_dataset(g, "boxes", box_t, sizeof(int) * (size_t)bi, nb, NULL);
        bad |= !L->box_offsets || !L->boxes || dims[0] < nb[0];
        L->n_boxes = (int)nb[0];
        int pd[6] = {0, 0, 0, 0, 0, 0};
        bad |= attr_read(g, "prob_domain", box_t, pd);
        for (int k = 0; k < bi; k++) L->prob_domain[k] = pd[k];
        L->ref_ratio = 2;
        (void)attr_read(g, "ref_ratio", H5T_NATIVE_INT, &L->ref_ratio);     /* the finest level's is never used (:209-242) */
        bad |= attr_read(g, "dx", H5T_NATIVE_DOUBLE, &L->dx) || attr_read(g, "logr", H5T_NATIVE_INT, &L->logr) ||
               attr_read(g, "domBeg1", H5T_NATIVE_DOUBLE, &L->dombeg1) || attr_read(g, "g_x2stretch", H5T_NATIVE_DOUBLE, &L->g_x2stretch) ||
               attr_read(g, "domBeg2", H5T_NATIVE_DOUBLE, &L->dombeg2);
        if (three_dimensional)
            bad |= attr_read(g, "g_x3stretch", H5T_NATIVE_DOUBLE, &L->g_x3stretch) || attr_read(g, "domBeg3", H5T_NATIVE_DOUBLE, &L->dombeg3);
        H5Gclose(g);
        if (bad) goto done;
        offset += L->data_len;
    }
    out->frame.levels = out->levels;
    out->frame.var_names = (const char *const *)out->var_names;
    out->frame.data = out->data;
    out->frame.l_scale = l_scale; out->frame.d_scale = d_scale; out->frame.p_scale = p_scale;
    rc = 0;
done:
    H5Tclose(box_t);
    H5Fclose(f);
    if (rc) mcrat_host_free_chombo(out);
    return rc;
}

/* (the public readers: the same under the library-wide HDF5 lock, see the top of this file) */
int mcrat_host_read_flash(const char *file, double l_scale, double d_scale, double p_scale, mcrat_host_flash *out)
{
    mcrat_host_h5_lock();
    const int rc = read_flash_locked(file, l_scale, d_scale, p_scale, out);
    mcrat_host_h5_unlock();
    return rc;
}

int mcrat_host_read_chombo(const char *file, int three_dimensional, double l_scale, double d_scale, double p_scale, mcrat_host_chombo *out)
{
    mcrat_host_h5_lock();
    const int rc = read_chombo_locked(file, three_dimensional, l_scale, d_scale, p_scale, out);
    mcrat_host_h5_unlock();
    return rc;
}
