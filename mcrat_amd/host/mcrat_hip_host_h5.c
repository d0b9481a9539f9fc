/* mcrat_hip_host_h5.c -- printPhotons (Src/mcrat_io.c:114-836) on top of mcrat_hip_get_output: the per-frame photon
 * datasets of mc_proc_<angle_rank>.h5, same file, group and dataset names, chunking and append behaviour, so that
 * dirFileMerge / MERGE / ProcessMCRaT read the files unchanged.  Plain C99 + the HDF5 C library (the one dependency of
 * this file; the rest of the host mirror does not need it).  The arrays come from the device already compacted
 * (photons with weight != 0, slot order) and live on the heap -- the reference keeps 18 arrays of num_photons doubles on
 * the stack (:130-131), which is what limits it to ~10^5 photons per rank with default stack sizes. */
#include "mcrat_hip_host.h"

#include <hdf5.h>
#include <stdlib.h>
#include <string.h>

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
    if (fPtr) fprintf(fPtr, "num_ph %d\nAllocated weight to be %d values large and other arrays to be %d\n", n, n, n);

    char file[2000], group[64];
    snprintf(file, sizeof file, "%s%s%d%s", dir, "mc_proc_", angle_rank, ".h5");
    snprintf(group, sizeof group, "%d", frame);
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fcreate(file, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT);           /* :199-206 */
    if (f < 0) f = H5Fopen(file, H5F_ACC_RDWR, H5P_DEFAULT);
    if (f < 0) { free(buf); free(type); return MCRAT_HIP_EINVAL; }
    hid_t g = (H5Lexists(f, group, H5P_DEFAULT) > 0) ? H5Gopen2(f, group, H5P_DEFAULT) : H5Gcreate2(f, group, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    int bad = g < 0;
    static const char *names[17] = {"P0", "P1", "P2", "P3", "COMV_P0", "COMV_P1", "COMV_P2", "COMV_P3", "R0", "R1", "R2",
                                    "S0", "S1", "S2", "S3", "NS", "PW"};
    for (int k = 0; !bad && k < ncol; k++)
        if (*slot[k]) bad = put(g, names[k], H5T_NATIVE_DOUBLE, *slot[k], (hsize_t)n) != 0;
    if (!bad && save_type) bad = put(g, "PT", H5T_NATIVE_CHAR, type, (hsize_t)n) != 0;
    if (g >= 0) H5Gclose(g);
    H5Fclose(f);
    free(buf);
    free(type);
    return bad ? MCRAT_HIP_EINVAL : MCRAT_HIP_OK;
}

/* a dataset of a frame back into memory (what dirFileMerge does per dataset, mcrat_io.c:1500-1560); *n receives its length;
 * up to cap values are copied (data may be NULL to ask for the length only) */
int mcrat_host_h5_read(const char *file, const char *group, const char *name, int is_char, void *data, int cap, int *n)
{
    if (!file || !group || !name || !n) return MCRAT_HIP_EINVAL;
    H5Eset_auto2(H5E_DEFAULT, NULL, NULL);
    hid_t f = H5Fopen(file, H5F_ACC_RDONLY, H5P_DEFAULT);
    if (f < 0) return MCRAT_HIP_EINVAL;
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
    return rc;
}
