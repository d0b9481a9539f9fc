/* Test-only: writes files with the dataset / attribute layout readAndDecimate (Src/mclib_flash.c:95-197) and readPlutoChombo
 * (Src/mclib_pluto.c:44-430) open, from the synthetic buffers of mcrat_amd.synth -- so that mcrat_host_read_flash /
 * mcrat_host_read_chombo are tested on real HDF5 files.  Compiled by tests/test_h5_readers.py. */
#include <hdf5.h>
#include <stdio.h>
#include <string.h>
#include "mcrat_hip.h"

static int put(hid_t loc, const char *name, hid_t type, int rank, const hsize_t *dims, const void *data)
{
    hid_t sp = H5Screate_simple(rank, dims, NULL);
    hid_t d = H5Dcreate2(loc, name, type, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    const herr_t st = d < 0 ? -1 : H5Dwrite(d, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
    if (d >= 0) H5Dclose(d);
    H5Sclose(sp);
    return st < 0 ? -1 : 0;
}

static int put_attr(hid_t loc, const char *name, hid_t type, const void *v)
{
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(loc, name, type, sp, H5P_DEFAULT, H5P_DEFAULT);
    const herr_t st = a < 0 ? -1 : H5Awrite(a, type, v);
    if (a >= 0) H5Aclose(a);
    H5Sclose(sp);
    return st < 0 ? -1 : 0;
}

int fixture_write_flash(const char *file, const mcrat_hip_flash_blocks *b)
{
    hid_t f = H5Fcreate(file, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) return -1;
    hsize_t dc[2] = {(hsize_t)b->n_blocks, (hsize_t)b->coord_stride}, db[2] = {(hsize_t)b->n_blocks, (hsize_t)b->bsize_stride};
    hsize_t dn[1] = {(hsize_t)b->n_blocks}, dv[4] = {(hsize_t)b->n_blocks, 1, 8, 8};
    int bad = put(f, "coordinates", H5T_NATIVE_DOUBLE, 2, dc, b->coordinates) | put(f, "block size", H5T_NATIVE_DOUBLE, 2, db, b->block_size) |
              put(f, "node type", H5T_NATIVE_INT, 1, dn, b->node_type) | put(f, "velx", H5T_NATIVE_DOUBLE, 4, dv, b->velx) |
              put(f, "vely", H5T_NATIVE_DOUBLE, 4, dv, b->vely) | put(f, "dens", H5T_NATIVE_DOUBLE, 4, dv, b->dens) |
              put(f, "pres", H5T_NATIVE_DOUBLE, 4, dv, b->pres);
    H5Fclose(f);
    return bad ? -1 : 0;
}

int fixture_write_chombo(const char *file, int three, const mcrat_hip_chombo *h)
{
    hid_t f = H5Fcreate(file, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    if (f < 0) return -1;
    const int nd = three ? 3 : 2, bi = 2 * nd;
    static const char *m3[6] = {"lo_i", "lo_j", "lo_k", "hi_i", "hi_j", "hi_k"}, *m2[4] = {"lo_i", "lo_j", "hi_i", "hi_j"};
    hid_t box_t = H5Tcreate(H5T_COMPOUND, sizeof(int) * (size_t)bi);
    for (int k = 0; k < bi; k++) H5Tinsert(box_t, three ? m3[k] : m2[k], sizeof(int) * (size_t)k, H5T_NATIVE_INT);
    int bad = put_attr(f, "num_levels", H5T_NATIVE_INT, &h->num_levels) | put_attr(f, "num_components", H5T_NATIVE_INT, &h->num_vars);
    hid_t cg = H5Gcreate2(f, "Chombo_global", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    bad |= put_attr(cg, "SpaceDim", H5T_NATIVE_INT, &nd);
    H5Gclose(cg);
    for (int k = 0; k < h->num_vars; k++) {
        char an[64];
        snprintf(an, sizeof an, "component_%d", k);
        hid_t st = H5Tcopy(H5T_C_S1);
        H5Tset_size(st, strlen(h->var_names[k]));
        bad |= put_attr(f, an, st, h->var_names[k]);
        H5Tclose(st);
    }
    long long offset = 0;
    for (int i = 0; i < h->num_levels; i++) {
        const mcrat_hip_chombo_level *L = &h->levels[i];
        char gn[64];
        snprintf(gn, sizeof gn, "level_%d", i);
        hid_t g = H5Gcreate2(f, gn, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        hsize_t nb[1] = {(hsize_t)L->n_boxes}, nd_[1] = {(hsize_t)L->data_len};
        bad |= put(g, "boxes", box_t, 1, nb, L->boxes) | put(g, "data:offsets=0", H5T_NATIVE_INT, 1, nb, L->box_offsets) |
               put(g, "data:datatype=0", H5T_NATIVE_DOUBLE, 1, nd_, h->data + offset);
        bad |= put_attr(g, "prob_domain", box_t, L->prob_domain) | put_attr(g, "ref_ratio", H5T_NATIVE_INT, &L->ref_ratio) |
               put_attr(g, "dx", H5T_NATIVE_DOUBLE, &L->dx) | put_attr(g, "logr", H5T_NATIVE_INT, &L->logr) |
               put_attr(g, "domBeg1", H5T_NATIVE_DOUBLE, &L->dombeg1) | put_attr(g, "domBeg2", H5T_NATIVE_DOUBLE, &L->dombeg2) |
               put_attr(g, "g_x2stretch", H5T_NATIVE_DOUBLE, &L->g_x2stretch);
        if (three) bad |= put_attr(g, "domBeg3", H5T_NATIVE_DOUBLE, &L->dombeg3) | put_attr(g, "g_x3stretch", H5T_NATIVE_DOUBLE, &L->g_x3stretch);
        H5Gclose(g);
        offset += L->data_len;
    }
    H5Tclose(box_t);
    H5Fclose(f);
    return bad ? -1 : 0;
}
