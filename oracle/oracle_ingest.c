/*
 * oracle_ingest.c -- CPU oracle for the step in front of the photon loop: turning a hydro simulation's frame into the
 * struct hydro_dataframe the loop reads.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see
 * mcrat_oracle.h.
 *
 * Restated from the reference, file I/O removed (the inputs here are the buffers the reference holds right after its
 * H5Dread / fread calls):
 *   orc_flash_select   readAndDecimate, Src/mclib_flash.c:199-428 (leaf blocks -> 8x8 cells, slab selection, derived columns)
 *   orc_pluto_select   readPluto,       Src/mclib_pluto.c:1130-1456
 *   orc_fillHydroCoordinateToSpherical  Src/geometry.c:156-174
 *   orc_cylindricalPrep / orc_sphericalPrep / orc_structuredFireballPrep   Src/analytic_outflows.c:3-236 (the constants the
 *       reference hard-codes are parameters here; orc_outflow_defaults() returns the reference's values)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mcrat_oracle.h"

static double *col(int n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* allocateHydroDataFrameMemory, mcrat_io.c:1853-1896 (every column; the ones a configuration does not use stay zero) */
static void frame_alloc(orc_frame *f, int n)
{
    memset(f, 0, sizeof *f);
    f->num_elements = n;
    f->r0 = col(n); f->r1 = col(n); f->r2 = col(n);
    f->r0_size = col(n); f->r1_size = col(n); f->r2_size = col(n);
    f->v0 = col(n); f->v1 = col(n); f->v2 = col(n);
    f->dens = col(n); f->dens_lab = col(n); f->pres = col(n); f->temp = col(n); f->gamma = col(n);
    f->r = col(n); f->theta = col(n);
}

void orc_frame_free(orc_frame *f)
{
    free(f->r0); free(f->r1); free(f->r2); free(f->r0_size); free(f->r1_size); free(f->r2_size);
    free(f->v0); free(f->v1); free(f->v2); free(f->dens); free(f->dens_lab); free(f->pres); free(f->temp); free(f->gamma);
    free(f->r); free(f->theta);
    memset(f, 0, sizeof *f);
}

/* the slab test both readers apply to every cell (mclib_flash.c:288-318 == mclib_pluto.c:1271-1302) */
static int in_slab(const orc_config *c, const orc_slab *s, int elem_factor, double x0, double x1, double x2, double s0, double s1, double s2,
                   int half_as_product)
{
    double r_in = 0, th_in = 0, r_out = 0, th_out = 0;
    /* FLASH writes size/2.0, PLUTO 0.5*size: the same double either way; kept apart for the citation only */
    const double h0 = half_as_product ? 0.5 * s0 : s0 / 2.0, h1 = half_as_product ? 0.5 * s1 : s1 / 2.0,
                 h2 = half_as_product ? 0.5 * s2 : s2 / 2.0;
    if (s->ph_inj_switch == 0) {
        const double ph_rmin = s->min_r, ph_rmax = s->max_r;
        const double ph_thetamin = s->min_theta - 2 * 0.017453292519943295;   /* mclib_flash.c:84-85 */
        const double ph_thetamax = s->max_theta + 2 * 0.017453292519943295;
        if (c->dimensions == ORC_THREE) {
            orc_hydroCoordinateToSpherical(c, &r_in, &th_in, fabs(x0) - h0, fabs(x1) - h1, fabs(x2) - h2);
            orc_hydroCoordinateToSpherical(c, &r_out, &th_out, fabs(x0) + h0, fabs(x1) + h1, fabs(x2) + h2);
        } else {
            orc_hydroCoordinateToSpherical(c, &r_in, &th_in, x0 - h0, x1 - h1, 0);
            orc_hydroCoordinateToSpherical(c, &r_out, &th_out, x0 + h0, x1 + h1, 0);
        }
        return ((ph_rmin - elem_factor * ORC_C_LIGHT / s->fps) <= r_out) && (r_in <= (ph_rmax + elem_factor * ORC_C_LIGHT / s->fps)) &&
               (th_out >= ph_thetamin) && (th_in <= ph_thetamax);
    }
    if (c->dimensions == ORC_THREE) orc_hydroCoordinateToSpherical(c, &r_in, &th_in, x0, x1, x2);
    else orc_hydroCoordinateToSpherical(c, &r_in, &th_in, x0, x1, 0);
    return r_in > (0.95 * s->r_inj);
}

/* readAndDecimate, mclib_flash.c:199-428.  coord[n_blocks][coord_stride], bsize[n_blocks][bsize_stride] (COORD_DIM1 = 2 in
 * the reference), node[n_blocks], and the four variables [n_blocks][64] as H5Dread left them.  Returns 0, or -1 when no
 * elem_factor up to `max_elem_factor` selects a cell (the reference would loop forever). */
int orc_flash_select(const orc_config *c, const orc_flash_blocks *b, const orc_slab *s, int max_elem_factor, orc_frame *out, int *elem_factor_out)
{
    static const double x1[8] = {-7.0 / 16, -5.0 / 16, -3.0 / 16, -1.0 / 16, 1.0 / 16, 3.0 / 16, 5.0 / 16, 7.0 / 16};   /* mclib_flash.c:69 */
    int num_nodes = 0, count = 0, i, j, r_count, elem_factor;
    for (i = 0; i < b->n_blocks; i++)
        if (b->node_type[i] == 1) num_nodes++;                                   /* :213-219 */
    const int n = num_nodes * 64;
    double *pres = col(n), *dens = col(n), *velx = col(n), *vely = col(n), *x = col(n), *y = col(n), *r = col(n), *szx = col(n), *szy = col(n);
    for (i = 0; i < b->n_blocks; i++) {                                          /* :236-268 */
        if (b->node_type[i] != 1) continue;
        int x1_count = 0, y1_count = 0;
        const double *co = b->coordinates + (size_t)i * b->coord_stride, *bs = b->block_size + (size_t)i * b->bsize_stride;
        for (j = 0; j < 64; j++) {
            pres[count] = b->pres[(size_t)i * 64 + j] * b->p_scale;
            dens[count] = b->dens[(size_t)i * 64 + j] * b->d_scale;
            velx[count] = b->velx[(size_t)i * 64 + j];
            vely[count] = b->vely[(size_t)i * 64 + j];
            szx[count] = (bs[0] / 8) * b->l_scale;
            szy[count] = (bs[1] / 8) * b->l_scale;
            if (j % 8 == 0) x1_count = 0;
            if ((j % 8 == 0) && (j != 0)) y1_count++;
            x[count] = (co[0] + bs[0] * x1[x1_count]) * b->l_scale;
            y[count] = (co[1] + bs[1] * x1[y1_count]) * b->l_scale;
            x1_count++;
            count++;
        }
    }
    elem_factor = b->cyclosynchrotron ? 2 : 0;                                   /* :275-279 */
    r_count = 0;
    while (r_count == 0) {                                                       /* :281-320 */
        r_count = 0;
        elem_factor++;
        if (elem_factor > max_elem_factor) break;
        for (i = 0; i < count; i++) {
            r[i] = sqrt(x[i] * x[i] + y[i] * y[i]);
            if (in_slab(c, s, elem_factor, x[i], y[i], 0, szx[i], szy[i], 0, 0)) r_count++;
        }
    }
    int rc = 0;
    if (r_count == 0) { rc = -1; frame_alloc(out, 0); }
    else {
        frame_alloc(out, r_count);
        j = 0;
        for (i = 0; i < count; i++) {                                            /* :346-420 */
            if (!in_slab(c, s, elem_factor, x[i], y[i], 0, szx[i], szy[i], 0, 0)) continue;
            out->pres[j] = pres[i];
            out->v0[j] = velx[i];
            out->v1[j] = vely[i];
            out->dens[j] = dens[i];
            out->r0[j] = x[i];
            out->r1[j] = y[i];
            out->r[j] = r[i];
            out->r0_size[j] = szx[i];
            out->r1_size[j] = szy[i];
            out->theta[j] = atan2(x[i], y[i]);
            out->gamma[j] = 1 / sqrt(1.0 - (pow(velx[i], 2) + pow(vely[i], 2)));
            out->dens_lab[j] = dens[i] / sqrt(1.0 - (pow(velx[i], 2) + pow(vely[i], 2)));
            out->temp[j] = pow(3 * pres[i] / (ORC_A_RAD), 1.0 / 4.0);
            j++;
        }
    }
    if (elem_factor_out) *elem_factor_out = elem_factor;
    free(pres); free(dens); free(velx); free(vely); free(x); free(y); free(r); free(szx); free(szy);
    return rc;
}

/* readPluto, mclib_pluto.c:1130-1456.  The 1-D grid arrays are readGridFile's (centre = (left+right)/2, width =
 * right-left, :951-971); the variables are the blocks of the .dbl file, [nz][ny][nx] each, picked by name from dbl.out. */
int orc_pluto_select(const orc_config *c, const orc_pluto_grid *g, const orc_slab *s, int max_elem_factor, orc_frame *out, int *elem_factor_out)
{
    const int nx = g->nx, ny = g->ny, nz = (c->dimensions == ORC_THREE) ? g->nz : 1;
    const int grid_size = nx * ny * nz;
    const int three = c->dimensions == ORC_THREE, v3 = c->dimensions != ORC_TWO;
    double *x1b = col(grid_size), *x2b = col(grid_size), *x3b = col(grid_size), *dx1b = col(grid_size), *dx2b = col(grid_size), *dx3b = col(grid_size);
    double *densb = col(grid_size), *presb = col(grid_size), *v1b = col(grid_size), *v2b = col(grid_size), *v3b = col(grid_size);
    int count = 0, i, j, k, l, r_count, elem_factor;
    for (j = 0; j < nz; j++)                                                     /* :1137-1215 */
        for (k = 0; k < ny; k++)
            for (l = 0; l < nx; l++) {
                const size_t idx = (size_t)j * nx * ny + (size_t)k * nx + l;
                densb[count] = g->rho[idx] * g->d_scale;
                x1b[count] = g->x1[l];
                x2b[count] = g->x2[k];
                dx1b[count] = g->dx1[l];
                dx2b[count] = g->dx2[k];
                if (three) { x3b[count] = g->x3[j]; dx3b[count] = g->dx3[j]; }
                x1b[count] *= g->l_scale;
                dx1b[count] *= g->l_scale;
                if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) { x2b[count] *= g->l_scale; dx2b[count] *= g->l_scale; }
                if (three && (c->geometry == ORC_CARTESIAN || c->geometry == ORC_POLAR)) { x3b[count] *= g->l_scale; dx3b[count] *= g->l_scale; }
                v1b[count] = g->vx1[idx];
                v2b[count] = g->vx2[idx];
                presb[count] = g->prs[idx] * g->p_scale;
                if (v3) v3b[count] = g->vx3[idx];
                count++;
            }
    elem_factor = g->cyclosynchrotron ? 2 : 0;                                   /* :1249-1253 */
    r_count = 0;
    while (r_count == 0) {                                                       /* :1254-1303 */
        r_count = 0;
        elem_factor++;
        if (elem_factor > max_elem_factor) break;
        for (i = 0; i < grid_size; i++)
            if (in_slab(c, s, elem_factor, x1b[i], x2b[i], x3b[i], dx1b[i], dx2b[i], dx3b[i], 1)) r_count++;
    }
    int rc = 0;
    if (r_count == 0) { rc = -1; frame_alloc(out, 0); }
    else {
        frame_alloc(out, r_count);
        j = 0;
        for (i = 0; i < grid_size; i++) {                                        /* :1341-1440 */
            if (!in_slab(c, s, elem_factor, x1b[i], x2b[i], x3b[i], dx1b[i], dx2b[i], dx3b[i], 1)) continue;
            out->pres[j] = presb[i];
            out->v0[j] = v1b[i];
            out->v1[j] = v2b[i];
            out->dens[j] = densb[i];
            out->r0[j] = x1b[i];
            out->r1[j] = x2b[i];
            out->r[j] = x1b[i];
            out->theta[j] = x2b[i];
            out->r0_size[j] = dx1b[i];
            out->r1_size[j] = dx2b[i];
            out->gamma[j] = 1 / sqrt(1.0 - (v1b[i] * v1b[i] + v2b[i] * v2b[i]));       /* gamma ignores vx3, :1362 */
            out->dens_lab[j] = densb[i] / sqrt(1.0 - (v1b[i] * v1b[i] + v2b[i] * v2b[i]));
            out->temp[j] = pow(3 * presb[i] / (ORC_A_RAD), 1.0 / 4.0);
            if (three) { out->r2[j] = x3b[i]; out->r2_size[j] = dx3b[i]; }
            if (v3) out->v2[j] = v3b[i];
            j++;
        }
    }
    if (elem_factor_out) *elem_factor_out = elem_factor;
    free(x1b); free(x2b); free(x3b); free(dx1b); free(dx2b); free(dx3b); free(densb); free(presb); free(v1b); free(v2b); free(v3b);
    return rc;
}

/* geometry.c:156-174 */
void orc_fillHydroCoordinateToSpherical(const orc_config *c, orc_frame *f)
{
    for (int i = 0; i < f->num_elements; i++) {
        double sph_r = 0, sph_theta = 0;
        if (c->dimensions == ORC_THREE) orc_hydroCoordinateToSpherical(c, &sph_r, &sph_theta, f->r0[i], f->r1[i], f->r2[i]);
        else orc_hydroCoordinateToSpherical(c, &sph_r, &sph_theta, f->r0[i], f->r1[i], 0);
        f->r[i] = sph_r;
        f->theta[i] = sph_theta;
    }
}

/* the constants analytic_outflows.c hard-codes (:5, :65, :140) */
void orc_outflow_defaults(int simulation_type, orc_outflow *o)
{
    memset(o, 0, sizeof *o);
    o->simulation_type = simulation_type;
    if (simulation_type == ORC_CYLINDRICAL_OUTFLOW) { o->gamma_infinity = 100; o->t_comov = 1e5; o->ddensity = 3e-7; }
    if (simulation_type == ORC_SPHERICAL_OUTFLOW) { o->gamma_infinity = 100; o->lumi = 1e54; o->r00 = 1e8; }
    if (simulation_type == ORC_STRUCTURED_SPHERICAL_OUTFLOW) { o->gamma_infinity = 100; o->lumi = 1e52; o->r00 = 1e8; o->theta_j = 1e-2; o->p = 4; }
}

/* the velocity block the three preps share for a radial flow (analytic_outflows.c:97-133 == :185-221) */
static void radial_velocity(const orc_config *c, orc_frame *f, int i, double vel)
{
    double r;
    if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE) {
        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) {
            r = sqrt(pow(f->r0[i], 2) + pow(f->r1[i], 2));
            f->v0[i] = (vel * f->r0[i]) / r;
            f->v1[i] = (vel * f->r1[i]) / r;
        }
        if (c->geometry == ORC_SPHERICAL) { f->v0[i] = vel; f->v1[i] = 0; }
        if (c->dimensions == ORC_TWO_POINT_FIVE) f->v2[i] = 0;
    } else {
        if (c->geometry == ORC_CARTESIAN) {
            r = sqrt(pow(f->r0[i], 2) + pow(f->r1[i], 2) + pow(f->r2[i], 2));
            f->v0[i] = (vel * f->r0[i]) / r;
            f->v1[i] = (vel * f->r1[i]) / r;
            f->v2[i] = (vel * f->r2[i]) / r;
        }
        if (c->geometry == ORC_SPHERICAL) { f->v0[i] = vel; f->v1[i] = 0; f->v2[i] = 0; }
        if (c->geometry == ORC_POLAR) {
            r = sqrt(pow(f->r0[i], 2) + pow(f->r2[i], 2));
            f->v0[i] = (vel * f->r0[i]) / r;
            f->v1[i] = 0;
            f->v2[i] = (vel * f->r2[i]) / r;
        }
    }
}

/* analytic_outflows.c:3-61 */
void orc_cylindricalPrep(const orc_config *c, const orc_outflow *o, orc_frame *f)
{
    const double gamma_infinity = o->gamma_infinity, t_comov = o->t_comov, ddensity = o->ddensity;
    const double vel = sqrt(1 - pow(gamma_infinity, -2.0)), lab_dens = gamma_infinity * ddensity;
    for (int i = 0; i < f->num_elements; i++) {
        f->gamma[i] = gamma_infinity;
        f->dens[i] = ddensity;
        f->dens_lab[i] = lab_dens;
        f->pres[i] = (ORC_A_RAD * pow(t_comov, 4.0)) / (3);
        f->temp[i] = t_comov;
        if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE) {
            if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) { f->v0[i] = 0; f->v1[i] = vel; }
            if (c->geometry == ORC_SPHERICAL) { f->v0[i] = vel * cos(f->r1[i]); f->v1[i] = -vel * sin(f->r1[i]); }
            if (c->dimensions == ORC_TWO_POINT_FIVE) f->v2[i] = 0;
        } else {
            if (c->geometry == ORC_CARTESIAN) { f->v0[i] = 0; f->v1[i] = 0; f->v2[i] = vel; }
            if (c->geometry == ORC_SPHERICAL) { f->v0[i] = vel * cos(f->r1[i]); f->v1[i] = -vel * sin(f->r1[i]); f->v2[i] = 0; }
            if (c->geometry == ORC_POLAR) { f->v0[i] = 0; f->v1[i] = 0; f->v2[i] = vel; }
        }
    }
}

/* analytic_outflows.c:63-136 */
void orc_sphericalPrep(const orc_config *c, const orc_outflow *o, orc_frame *f)
{
    const double gamma_infinity = o->gamma_infinity, lumi = o->lumi, r00 = o->r00;
    for (int i = 0; i < f->num_elements; i++) {
        if (f->r[i] >= (r00 * gamma_infinity)) {
            f->gamma[i] = gamma_infinity;
            f->pres[i] = (lumi * pow(r00, 2.0 / 3.0) * pow(f->r[i], -8.0 / 3.0)) / (12.0 * M_PI * ORC_C_LIGHT * pow(gamma_infinity, 4.0 / 3.0));
        } else {
            f->gamma[i] = f->r[i] / r00;
            f->pres[i] = (lumi * pow(r00, 2.0)) / (12.0 * M_PI * ORC_C_LIGHT * pow(f->r[i], 4.0));
        }
        f->dens[i] = lumi / (4 * M_PI * pow(f->r[i], 2.0) * pow(ORC_C_LIGHT, 3.0) * gamma_infinity * f->gamma[i]);
        f->dens_lab[i] = f->dens[i] * f->gamma[i];
        f->temp[i] = pow(3 * f->pres[i] / (ORC_A_RAD), 1.0 / 4.0);
        radial_velocity(c, f, i, sqrt(1 - pow(f->gamma[i], -2.0)));
    }
}

/* analytic_outflows.c:138-236 */
void orc_structuredFireballPrep(const orc_config *c, const orc_outflow *o, orc_frame *f)
{
    const double gamma_0 = o->gamma_infinity, lumi = o->lumi, r00 = o->r00, theta_j = o->theta_j, p = o->p;
    const double T_0 = pow(lumi / (4 * M_PI * r00 * r00 * ORC_A_RAD * ORC_C_LIGHT), 1.0 / 4.0);
    for (int i = 0; i < f->num_elements; i++) {
        const double theta_ratio = f->theta[i] / theta_j;
        double eta = gamma_0 / sqrt(1 + pow(theta_ratio, 2 * p));
        if (f->theta[i] >= theta_j * pow(gamma_0 / 2, 1.0 / p)) eta = 2.0;
        const double r_sat = eta * r00;
        if (f->r[i] >= r_sat) {
            f->gamma[i] = eta;
            f->temp[i] = T_0 * pow(r_sat / f->r[i], 2.0 / 3.0) / eta;
        } else {
            f->gamma[i] = f->r[i] / r_sat;
            f->temp[i] = T_0;
        }
        const double vel = sqrt(1 - pow(f->gamma[i], -2.0));
        f->dens[i] = ORC_M_P * lumi / (4 * M_PI * ORC_M_P * ORC_C_LIGHT * ORC_C_LIGHT * ORC_C_LIGHT * eta * vel * f->gamma[i] * f->r[i] * f->r[i]);
        f->dens_lab[i] = f->dens[i] * f->gamma[i];
        f->pres[i] = (ORC_A_RAD * pow(f->temp[i], 4.0)) / (3);
        radial_velocity(c, f, i, vel);
    }
}

/* what getHydroData does after the reader returns (mcrat_io.c:1962-1975) */
void orc_hydro_post_read(const orc_config *c, const orc_outflow *o, orc_frame *f)
{
    orc_fillHydroCoordinateToSpherical(c, f);
    if (!o) return;
    if (o->simulation_type == ORC_CYLINDRICAL_OUTFLOW) orc_cylindricalPrep(c, o, f);
    else if (o->simulation_type == ORC_SPHERICAL_OUTFLOW) orc_sphericalPrep(c, o, f);
    else if (o->simulation_type == ORC_STRUCTURED_SPHERICAL_OUTFLOW) orc_structuredFireballPrep(c, o, f);
}

/* readPlutoChombo, mclib_pluto.c:12-801, after its HDF5 reads: per level the "boxes", "data:offsets=0" and
 * "data:datatype=0" datasets and the attributes prob_domain, ref_ratio, dx, logr, domBeg1-3, g_x2stretch, g_x3stretch; the
 * component names.  Cells are numbered level by level (level 0 first, :151-155), box by box, x fastest (:520-545).
 * Reference behaviour kept on purpose: cells of a level that a finer level covers (good_node_buffer == 0, :206-345) are
 * dropped only in the injection-frame branch (ph_inj_switch != 0, :672 / :745); the photons'-slab branch (:659 / :703)
 * does not look at the mask, so covered coarse cells stay in the frame there. */
int orc_chombo_select(const orc_config *c, const orc_chombo *h, const orc_slab *s, int max_elem_factor, orc_frame *out, int *elem_factor_out)
{
    const int three = c->dimensions == ORC_THREE, v3 = c->dimensions != ORC_TWO, num_vars = h->num_vars, nl = h->num_levels;
    const int bi = three ? 6 : 4;                       /* ints per box: lo_i, lo_j, [lo_k], hi_i, hi_j, [hi_k] */
    long long total_size = 0;
    long long *start_displacement = (long long *)calloc((size_t)nl, sizeof(long long));
    int i, j, k, l, m, n;
    for (i = 0; i < nl; i++) { start_displacement[i] = total_size; total_size += h->levels[i].data_len; }      /* :128-155 */
    const long long cells = total_size / num_vars;
    double *x1b = col((int)cells), *x2b = col((int)cells), *x3b = col((int)cells), *dx1b = col((int)cells), *dx2b = col((int)cells), *dx3b = col((int)cells);
    double *densb = col((int)cells), *presb = col((int)cells), *v1b = col((int)cells), *v2b = col((int)cells), *v3b = col((int)cells);
    int *good = (int *)malloc(sizeof(int) * (size_t)(cells > 0 ? cells : 1));
    for (long long q = 0; q < cells; q++) good[q] = 1;                                                           /* :195-198 */
#define LO(b, a) ((b)[(a)])
#define HI(b, a) ((b)[(three ? 3 : 2) + (a)])
    for (i = nl - 2; i >= 0; i--) {                                                                              /* :206-345 */
        const orc_chombo_level *Li = &h->levels[i], *Lp = &h->levels[i + 1];
        const int ref_ratio = Li->ref_ratio;
        const long long offset = start_displacement[i];
        for (j = 0; j < Li->n_boxes; j++) {
            const int *bj = Li->boxes + (size_t)j * bi;
            const int nbx = HI(bj, 0) - LO(bj, 0) + 1, nby = HI(bj, 1) - LO(bj, 1) + 1, nbz = three ? HI(bj, 2) - LO(bj, 2) + 1 : 1;
            for (k = 0; k < Lp->n_boxes; k++) {
                const int *bk = Lp->boxes + (size_t)k * bi;
                int overlap = (ref_ratio * HI(bj, 0) >= LO(bk, 0)) && (ref_ratio * LO(bj, 0) <= HI(bk, 0)) &&
                              (ref_ratio * HI(bj, 1) >= LO(bk, 1)) && (ref_ratio * LO(bj, 1) <= HI(bk, 1));
                if (three) overlap = overlap && (ref_ratio * HI(bj, 2) >= LO(bk, 2)) && (ref_ratio * LO(bj, 2) <= HI(bk, 2));
                if (!overlap) continue;
                for (l = 0; l < nbz; l++)
                    for (m = 0; m < nby; m++)
                        for (n = 0; n < nbx; n++) {
                            const int idx1 = ref_ratio * (LO(bj, 0) + n), idx2 = ref_ratio * (LO(bj, 1) + m), idx3 = three ? ref_ratio * (LO(bj, 2) + l) : 0;
                            int inside = (HI(bk, 0) >= idx1) && (LO(bk, 0) <= idx1) && (HI(bk, 1) >= idx2) && (LO(bk, 1) <= idx2);
                            if (three) inside = inside && (HI(bk, 2) >= idx3) && (LO(bk, 2) <= idx3);
                            if (inside) good[(offset + Li->box_offsets[j]) / num_vars + (long long)l * nbx * nby + (long long)m * nbx + n] = 0;
                        }
            }
        }
    }
    int kv[5] = {-1, -1, -1, -1, -1};                   /* rho, vx1, vx2, vx3, prs by component name (:547-590) */
    for (k = 0; k < num_vars; k++) {
        if (strcmp(h->var_names[k], "rho") == 0) kv[0] = k;
        else if (strcmp(h->var_names[k], "vx1") == 0) kv[1] = k;
        else if (strcmp(h->var_names[k], "vx2") == 0) kv[2] = k;
        else if (strcmp(h->var_names[k], "vx3") == 0) kv[3] = k;
        else if (strcmp(h->var_names[k], "prs") == 0) kv[4] = k;
    }
    for (i = nl - 1; i >= 0; i--) {                                                                              /* :349-623 */
        const orc_chombo_level *L = &h->levels[i];
        const long long offset = start_displacement[i];
        const int n1 = L->prob_domain[three ? 3 : 2] - L->prob_domain[0] + 1, n2 = L->prob_domain[(three ? 3 : 2) + 1] - L->prob_domain[1] + 1;
        const int n3 = three ? L->prob_domain[5] - L->prob_domain[2] + 1 : 1;
        double *x1a = col(n1), *dx1a = col(n1), *x2a = col(n2), *dx2a = col(n2), *x3a = col(n3), *dx3a = col(n3);
        for (j = 0; j < n1; j++) {                                                                               /* :446-466 */
            if (L->logr == 0) {
                x1a[j] = L->dombeg1 + L->dx * (L->prob_domain[0] + j + 0.5);
                dx1a[j] = L->dx;
            } else {
                x1a[j] = L->dombeg1 * 0.5 * (exp(L->dx * (L->prob_domain[0] + j + 1)) + exp(L->dx * (L->prob_domain[0] + j)));
                dx1a[j] = L->dombeg1 * (exp(L->dx * (L->prob_domain[0] + j + 1)) - exp(L->dx * (L->prob_domain[0] + j)));
            }
        }
        for (j = 0; j < n2; j++) {                                                                               /* :477-480 */
            x2a[j] = L->dombeg2 + L->dx * L->g_x2stretch * (L->prob_domain[1] + j + 0.5);
            dx2a[j] = L->dx * L->g_x2stretch;
        }
        for (j = 0; three && j < n3; j++) {                                                                      /* :499-502 */
            x3a[j] = L->dombeg3 + L->dx * L->g_x3stretch * (L->prob_domain[2] + j + 0.5);
            dx3a[j] = L->dx * L->g_x3stretch;
        }
        for (j = 0; j < L->n_boxes; j++) {                                                                       /* :520-612 */
            const int *b = L->boxes + (size_t)j * bi;
            const int nbx = HI(b, 0) - LO(b, 0) + 1, nby = HI(b, 1) - LO(b, 1) + 1, nbz = three ? HI(b, 2) - LO(b, 2) + 1 : 1;
            for (l = 0; l < nbz; l++)
                for (m = 0; m < nby; m++)
                    for (n = 0; n < nbx; n++) {
                        const long long q = (offset + L->box_offsets[j]) / num_vars + (long long)l * nbx * nby + (long long)m * nbx + n;
                        const long long d = offset + L->box_offsets[j] + (long long)l * nbx * nby + (long long)m * nbx + n;
                        const long long vs = (long long)nbx * nby * nbz;
                        x1b[q] = x1a[LO(b, 0) + n]; x2b[q] = x2a[LO(b, 1) + m];
                        dx1b[q] = dx1a[LO(b, 0) + n]; dx2b[q] = dx2a[LO(b, 1) + m];
                        if (three) { x3b[q] = x3a[LO(b, 2) + l]; dx3b[q] = dx3a[LO(b, 2) + l]; }
                        x1b[q] *= h->l_scale; dx1b[q] *= h->l_scale;
                        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) { x2b[q] *= h->l_scale; dx2b[q] *= h->l_scale; }
                        if (three && (c->geometry == ORC_CARTESIAN || c->geometry == ORC_POLAR)) { x3b[q] *= h->l_scale; dx3b[q] *= h->l_scale; }
                        if (kv[0] >= 0) densb[q] = h->data[d + kv[0] * vs] * h->d_scale;
                        if (kv[1] >= 0) v1b[q] = h->data[d + kv[1] * vs];
                        if (kv[2] >= 0) v2b[q] = h->data[d + kv[2] * vs];
                        if (kv[4] >= 0) presb[q] = h->data[d + kv[4] * vs] * h->p_scale;
                        if (v3 && kv[3] >= 0) v3b[q] = h->data[d + kv[3] * vs];
                    }
        }
        free(x1a); free(dx1a); free(x2a); free(dx2a); free(x3a); free(dx3a);
    }
#undef LO
#undef HI
    int elem_factor = h->cyclosynchrotron ? 2 : 0, r_count = 0;                                                  /* :635-687 */
    while (r_count == 0) {
        r_count = 0;
        elem_factor++;
        if (elem_factor > max_elem_factor) break;
        for (long long q = 0; q < cells; q++)
            if (in_slab(c, s, elem_factor, x1b[q], x2b[q], x3b[q], dx1b[q], dx2b[q], dx3b[q], 1) && (s->ph_inj_switch == 0 || good[q] != 0)) r_count++;
    }
    int rc = 0;
    if (r_count == 0) { rc = -1; frame_alloc(out, 0); }
    else {
        frame_alloc(out, r_count);
        j = 0;
        for (long long q = 0; q < cells; q++) {                                                                  /* :691-793 */
            if (!(in_slab(c, s, elem_factor, x1b[q], x2b[q], x3b[q], dx1b[q], dx2b[q], dx3b[q], 1) && (s->ph_inj_switch == 0 || good[q] != 0))) continue;
            out->pres[j] = presb[q];
            out->v0[j] = v1b[q];
            out->v1[j] = v2b[q];
            out->dens[j] = densb[q];
            out->r0[j] = x1b[q];
            out->r1[j] = x2b[q];
            out->r[j] = x1b[q];
            out->theta[j] = x2b[q];
            out->r0_size[j] = dx1b[q];
            out->r1_size[j] = dx2b[q];
            out->gamma[j] = 1 / sqrt(1.0 - (v1b[q] * v1b[q] + v2b[q] * v2b[q]));
            out->dens_lab[j] = densb[q] / sqrt(1.0 - (v1b[q] * v1b[q] + v2b[q] * v2b[q]));
            out->temp[j] = pow(3 * presb[q] / (ORC_A_RAD), 1.0 / 4.0);
            if (three) { out->r2[j] = x3b[q]; out->r2_size[j] = dx3b[q]; }
            if (v3) out->v2[j] = v3b[q];
            j++;
        }
    }
    if (elem_factor_out) *elem_factor_out = elem_factor;
    free(start_displacement); free(good);
    free(x1b); free(x2b); free(x3b); free(dx1b); free(dx2b); free(dx3b); free(densb); free(presb); free(v1b); free(v2b); free(v3b);
    return rc;
}
