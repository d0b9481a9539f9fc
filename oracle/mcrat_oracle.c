/*
 * mcrat_oracle.c -- plain-C restatement of MCRaT's per-timestep photon loop.
 * TEST INFRASTRUCTURE ONLY; see mcrat_oracle.h for the rules and the
 * "parity unpinned" statement.  Every function cites the reference lines it
 * restates (paths relative to /root/reference/Src).
 */
#define _GNU_SOURCE
#include "mcrat_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <limits.h>

int orc_sizeof_photon(void) { return (int)sizeof(orc_photon); }

/* ------------------------------------------------------------------ */
/* small dense helpers standing for the GSL BLAS calls of the reference */

/* gsl_blas_dnrm2 == reference BLAS DNRM2 (scaled sum of squares) */
double orc_dnrm2(const double *x, int n)
{
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; i++) {
        if (x[i] != 0.0) {
            double ax = fabs(x[i]);
            if (scale < ax) {
                ssq = 1.0 + ssq * (scale / ax) * (scale / ax);
                scale = ax;
            } else {
                ssq += (ax / scale) * (ax / scale);
            }
        }
    }
    return scale * sqrt(ssq);
}

/* y = A x for a row-major n x n matrix, accumulation order of the reference
 * CBLAS dgemv (temp += x[j]*A[i][j], j ascending), alpha=1, beta=0 */
static void matvec(int n, const double *A, const double *x, double *y)
{
    for (int i = 0; i < n; i++) {
        double temp = 0.0;
        for (int j = 0; j < n; j++) temp += x[j] * A[i * n + j];
        y[i] = temp;
    }
}

static double dot3(const double a[3], const double b[3])
{
    double r = 0.0;
    for (int i = 0; i < 3; i++) r += a[i] * b[i];
    return r;
}

/* ------------------------------------------------------------------ */
/* mclib.c:409-434 */
void orc_zeroNorm(double p[4])
{
    double nrm = orc_dnrm2(p + 1, 3);
    if (p[0] != nrm) {
        p[1] = (p[1] / nrm) * p[0];
        p[2] = (p[2] / nrm) * p[0];
        p[3] = (p[3] / nrm) * p[0];
    }
}

/* mclib.c:302-407: general boost into the frame moving with velocity `boost`
 * (units of c); photons ('p') are re-normalised to a null vector afterwards */
void orc_lorentzBoost(const double boost[3], const double p[4], double result[4], char object)
{
    double out[4];
    double beta = orc_dnrm2(boost, 3);
    if (beta > 0) {
        double gamma = 1.0 / sqrt(1 - beta * beta);
        double L[16];
        memset(L, 0, sizeof L);
        L[0] = gamma;
        L[1] = -1 * boost[0] * gamma;
        L[2] = -1 * boost[1] * gamma;
        L[3] = -1 * boost[2] * gamma;
        L[5]  = 1 + ((gamma - 1) * (boost[0] * boost[0]) / (beta * beta));
        L[6]  = ((gamma - 1) * (boost[0] * boost[1] / (beta * beta)));
        L[7]  = ((gamma - 1) * (boost[0] * boost[2] / (beta * beta)));
        L[10] = 1 + ((gamma - 1) * (boost[1] * boost[1]) / (beta * beta));
        L[11] = ((gamma - 1) * (boost[1] * boost[2]) / (beta * beta));
        L[15] = 1 + ((gamma - 1) * (boost[2] * boost[2]) / (beta * beta));
        L[4] = L[1]; L[8] = L[2]; L[12] = L[3];
        L[9] = L[6]; L[13] = L[7]; L[14] = L[11];
        matvec(4, L, p, out);
        if (object == 'p') orc_zeroNorm(out);
    } else {
        memcpy(out, p, sizeof out);
        if (object == 'p') orc_zeroNorm(out);
    }
    memcpy(result, out, sizeof out);
}

/* geometry.c:15-64 */
void orc_mcratCoordinateToHydroCoordinate(const orc_config *c, double out[3], double x, double y, double z)
{
    double r0 = -1, r1 = -1, r2 = -1;
    if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE) {
        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) {
            r0 = sqrt(x * x + y * y);
            r1 = z;
        }
        if (c->geometry == ORC_SPHERICAL) {
            r0 = sqrt(x * x + y * y + z * z);
            r1 = acos(z / r0);
        }
    } else {
        if (c->geometry == ORC_CARTESIAN) { r0 = x; r1 = y; r2 = z; }
        if (c->geometry == ORC_SPHERICAL) {
            r0 = sqrt(x * x + y * y + z * z);
            r1 = acos(z / r0);
            r2 = fmod(atan2(y, x) * 180.0 / M_PI + 360.0, 360.0) * M_PI / 180;
        }
        if (c->geometry == ORC_POLAR) {
            r0 = sqrt(x * x + y * y);
            r1 = fmod(atan2(y, x) * 180.0 / M_PI + 360.0, 360.0) * M_PI / 180;
            r2 = z;
        }
    }
    out[0] = r0; out[1] = r1; out[2] = r2;
}

/* geometry.c:189-253 */
void orc_hydroVectorToCartesian(const orc_config *c, double out[3], double v0, double v1, double v2,
                                double x0, double x1, double x2)
{
    double t0 = 0, t1 = 0, t2 = 0;
    (void)x0;
    if (c->dimensions == ORC_TWO) {
        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) {
            t0 = v0 * cos(x2);
            t1 = v0 * sin(x2);
            t2 = v1;
        }
        if (c->geometry == ORC_SPHERICAL) {
            v2 = 0;
            t0 = v0 * sin(x1) * cos(x2) + v1 * cos(x1) * cos(x2) - v2 * sin(x2);
            t1 = v0 * sin(x1) * sin(x2) + v1 * cos(x1) * sin(x2) + v2 * cos(x2);
            t2 = v0 * cos(x1) - v1 * sin(x1);
        }
    } else if (c->dimensions == ORC_TWO_POINT_FIVE) {
        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) {
            t0 = v0 * cos(x2) - v2 * sin(x2);
            t1 = v0 * sin(x2) + v2 * cos(x2);
            t2 = v1;
        }
        if (c->geometry == ORC_SPHERICAL) {
            t0 = v0 * sin(x1) * cos(x2) + v1 * cos(x1) * cos(x2) - v2 * sin(x2);
            t1 = v0 * sin(x1) * sin(x2) + v1 * cos(x1) * sin(x2) + v2 * cos(x2);
            t2 = v0 * cos(x1) - v1 * sin(x1);
        }
    } else {
        if (c->geometry == ORC_CARTESIAN) { t0 = v0; t1 = v1; t2 = v2; }
        if (c->geometry == ORC_SPHERICAL) {
            t0 = v0 * sin(x1) * cos(x2) + v1 * cos(x1) * cos(x2) - v2 * sin(x2);
            t1 = v0 * sin(x1) * sin(x2) + v1 * cos(x1) * sin(x2) + v2 * cos(x2);
            t2 = v0 * cos(x1) - v1 * sin(x1);
        }
        if (c->geometry == ORC_POLAR) {
            t0 = v0 * cos(x1) - v1 * sin(x1);
            t1 = v0 * sin(x1) + v1 * cos(x1);
            t2 = v2;
        }
    }
    out[0] = t0; out[1] = t1; out[2] = t2;
}

/* the fluid velocity of cell `idx` as a Cartesian 3-vector for a photon at
 * azimuth ph_phi: the three call shapes at mclib.c:546-555,1167-1174 and
 * optical_depth.c:27-36 */
static void cell_beta_cartesian(const orc_config *c, const orc_hydro *h, int idx, double ph_phi, double out[3])
{
    if (c->dimensions == ORC_THREE)
        orc_hydroVectorToCartesian(c, out, h->v0[idx], h->v1[idx], h->v2[idx], h->r0[idx], h->r1[idx], h->r2[idx]);
    else if (c->dimensions == ORC_TWO_POINT_FIVE)
        orc_hydroVectorToCartesian(c, out, h->v0[idx], h->v1[idx], h->v2[idx], h->r0[idx], h->r1[idx], ph_phi);
    else
        orc_hydroVectorToCartesian(c, out, h->v0[idx], h->v1[idx], 0, h->r0[idx], h->r1[idx], ph_phi);
}

/* geometry.c:394-417 */
int orc_checkInBlock(const orc_config *c, double a0, double a1, double a2, const orc_hydro *h, int idx)
{
    int in;
    if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE)
        in = (2 * fabs(a0 - h->r0[idx]) - h->r0_size[idx] <= 0) && (2 * fabs(a1 - h->r1[idx]) - h->r1_size[idx] <= 0);
    else
        in = (2 * fabs(a0 - h->r0[idx]) - h->r0_size[idx] <= 0) && (2 * fabs(a1 - h->r1[idx]) - h->r1_size[idx] <= 0)
             && (2 * fabs(a2 - h->r2[idx]) - h->r2_size[idx] <= 0);
    return in ? 1 : 0;
}

/* geometry.c:350-391 (the path taken because hydro_data->grid == NULL,
 * mcrat_io.c:1985 -> geometry.c:426-430): lowest-index containing cell or -1 */
/* ---- optimised mode only: an exact accelerator of the linear search.  Uniform buckets per axis over the mesh's extent;
 * every cell is entered, in ascending cell order, into all buckets its closed extent (widened by 1e-9 relative) touches;
 * a point's bucket list therefore holds every cell that can contain it, lowest index first, and the reference's own
 * closed-interval test picks the first -- the answer of the linear scan. */
typedef struct orc_grid {
    const orc_hydro *h;
    int naxes, dim[3];
    double lo[3], inv[3];
    int *start, *cells;
} orc_grid;
static orc_grid g_grid;

void orc_grid_detach(void)
{
    free(g_grid.start); free(g_grid.cells);
    memset(&g_grid, 0, sizeof g_grid);
}

static int grid_bucket_1d(const orc_grid *g, int k, double x)
{
    double f = floor((x - g->lo[k]) * g->inv[k]);
    if (!(f == f) || f < 0) return 0;
    if (f > g->dim[k] - 1) return g->dim[k] - 1;
    return (int)f;
}

void orc_grid_attach(const orc_config *c, const orc_hydro *h)
{
    orc_grid_detach();
    const int M = h->num_elements, naxes = (c->dimensions == ORC_THREE) ? 3 : 2;
    const double *cc[3] = {h->r0, h->r1, h->r2}, *ss[3] = {h->r0_size, h->r1_size, h->r2_size};
    orc_grid *g = &g_grid;
    g->naxes = naxes;
    long long nb = 1;
    const int per_axis = (int)fmax(1.0, floor(pow((double)M, 1.0 / naxes)));
    for (int k = 0; k < 3; k++) {
        g->dim[k] = 1; g->lo[k] = 0; g->inv[k] = 0;
        if (k >= naxes) continue;
        double lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < M; i++) { lo = fmin(lo, cc[k][i] - 0.5 * ss[k][i]); hi = fmax(hi, cc[k][i] + 0.5 * ss[k][i]); }
        g->dim[k] = per_axis;
        g->lo[k] = lo;
        g->inv[k] = per_axis / (hi - lo);
        nb *= per_axis;
    }
    g->start = (int *)calloc((size_t)nb + 1, sizeof(int));
    for (int pass = 0; pass < 2; pass++) {
        int *cursor = pass ? (int *)malloc(sizeof(int) * (size_t)nb) : NULL;
        if (pass) { memcpy(cursor, g->start, sizeof(int) * (size_t)nb); g->cells = (int *)malloc(sizeof(int) * (size_t)(g->start[nb] > 0 ? g->start[nb] : 1)); }
        for (int i = 0; i < M; i++) {
            int b0[3] = {0, 0, 0}, b1[3] = {0, 0, 0};
            for (int k = 0; k < naxes; k++) {
                const double m = 1e-9 * (fabs(cc[k][i]) + ss[k][i]);
                b0[k] = grid_bucket_1d(g, k, cc[k][i] - 0.5 * ss[k][i] - m);
                b1[k] = grid_bucket_1d(g, k, cc[k][i] + 0.5 * ss[k][i] + m);
            }
            for (int z = b0[2]; z <= b1[2]; z++)
                for (int y = b0[1]; y <= b1[1]; y++)
                    for (int x = b0[0]; x <= b1[0]; x++) {
                        const long long b = ((long long)z * g->dim[1] + y) * g->dim[0] + x;
                        if (pass) g->cells[cursor[b]++] = i; else g->start[b + 1]++;
                    }
        }
        if (!pass) for (long long b = 0; b < nb; b++) g->start[b + 1] += g->start[b];
        free(cursor);
    }
    g->h = h;
}

int orc_findContainingBlock(const orc_config *c, double a0, double a1, double a2, const orc_hydro *h)
{
    if (c->optimised && g_grid.h == h) {
        const orc_grid *g = &g_grid;
        const double a[3] = {a0, a1, a2};
        long long b = 0;
        for (int k = g->naxes - 1; k >= 0; k--) b = b * g->dim[k] + grid_bucket_1d(g, k, a[k]);
        for (int e = g->start[b]; e < g->start[b + 1]; e++)
            if (orc_checkInBlock(c, a0, a1, a2, h, g->cells[e])) return g->cells[e];
        return -1;
    }
    for (int i = 0; i < h->num_elements; i++)
        if (orc_checkInBlock(c, a0, a1, a2, h, i)) return i;
    return -1;
}

/* geometry.c:255-296 */
double orc_hydroElementVolume(const orc_config *c, const orc_hydro *h, int idx)
{
    double V = 0;
    double r0_max = h->r0[idx] + 0.5 * h->r0_size[idx], r0_min = h->r0[idx] - 0.5 * h->r0_size[idx];
    double r1_max = h->r1[idx] + 0.5 * h->r1_size[idx], r1_min = h->r1[idx] - 0.5 * h->r1_size[idx];
    if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE) {
        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL)
            V = M_PI * (r0_max * r0_max - r0_min * r0_min) * h->r1_size[idx];
        if (c->geometry == ORC_SPHERICAL)
            V = (2.0 * M_PI / 3.0) * (r0_max * r0_max * r0_max - r0_min * r0_min * r0_min) * (cos(r1_min) - cos(r1_max));
    } else {
        double r2_max = h->r2[idx] + 0.5 * h->r2_size[idx], r2_min = h->r2[idx] - 0.5 * h->r2_size[idx];
        if (c->geometry == ORC_CARTESIAN) V = h->r0_size[idx] * h->r1_size[idx] * h->r2_size[idx];
        if (c->geometry == ORC_SPHERICAL)
            V = (1.0 / 3.0) * (r0_max * r0_max * r0_max - r0_min * r0_min * r0_min) * (cos(r1_min) - cos(r1_max)) * (r2_max - r2_min);
        if (c->geometry == ORC_POLAR)
            V = 0.5 * (r0_max * r0_max - r0_min * r0_min) * h->r1_size[idx] * h->r2_size[idx];
    }
    return V;
}

/* ------------------------------------------------------------------ */
/* Stokes helpers */

/* mcrat_scattering.c:10-39: Q' = Q cos2t - U sin2t, U' = Q sin2t + U cos2t */
void orc_mullerMatrixRotation(double theta, double s[4])
{
    double M[16], out[4];
    memset(M, 0, sizeof M);
    M[0] = 1; M[15] = 1;
    M[5] = cos(2 * theta);
    M[10] = cos(2 * theta);
    M[6] = -1 * sin(2 * theta);
    M[9] = sin(2 * theta);
    matvec(4, M, s, out);
    memcpy(s, out, sizeof out);
}

/* mcrat_scattering.c:41-65 */
void orc_findXY(const double v[3], const double ref[3], double x[3], double y[3])
{
    double norm;
    y[0] = (v[1] * ref[2] - v[2] * ref[1]);
    y[1] = -1 * (v[0] * ref[2] - v[2] * ref[0]);
    y[2] = (v[0] * ref[1] - v[1] * ref[0]);
    norm = 1.0 / sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    y[0] *= norm; y[1] *= norm; y[2] *= norm;

    x[0] = y[1] * v[2] - y[2] * v[1];
    x[1] = -1 * (y[0] * v[2] - y[2] * v[0]);
    x[2] = y[0] * v[1] - y[1] * v[0];
    norm = 1.0 / sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    x[0] *= norm; x[1] *= norm; x[2] *= norm;
}

/* mcrat_scattering.c:67-101 */
double orc_findPhi(const double x_old[3], const double y_old[3], const double x_new[3], const double y_new[3])
{
    double factor, d;
    (void)x_new;
    d = dot3(x_old, y_new);
    if (d > 0) factor = 1;
    else if (d < 0) factor = -1;
    else factor = 0;
    d = dot3(y_old, y_new);
    if ((d < -1) || (d > 1)) d = round(d);
    return -1 * factor * acos(d);
}

/* mcrat_scattering.c:103-149 */
void orc_stokesRotation(const double v[3], const double v_ph[3], const double v_ph_boosted[3], double s[4])
{
    const double z_hat[3] = {0, 0, 1};
    double x[3], y[3], x_new[3], y_new[3], phi;
    orc_findXY(v_ph, z_hat, x, y);
    orc_findXY(v_ph, v, x_new, y_new);
    phi = orc_findPhi(x, y, x_new, y_new);
    orc_mullerMatrixRotation(phi, s);
    orc_findXY(v_ph_boosted, v, x, y);
    orc_findXY(v_ph_boosted, z_hat, x_new, y_new);
    phi = orc_findPhi(x, y, x_new, y_new);
    orc_mullerMatrixRotation(phi, s);
}

/* ------------------------------------------------------------------ */
/* mcrat_scattering.c:597-623 */
double orc_kleinNishinaCrossSection(double e)
{
    if (e >= 1e-3)
        return (3. / 4.) * (2. / (e * e) + (1. / (2. * e) - (1. + e) / (e * e * e)) * log(1. + 2. * e)
                            + (1. + e) / ((1. + 2. * e) * (1. + 2. * e)));
    return (1. - 2. * e);
}

/* mcrat_scattering.c:509-595 */
int orc_kleinNishinaScatter(const orc_config *c, double *theta, double *phi, double p0, double q, double u, orc_rng *rng)
{
    double phi_dum = 0, cos_theta_dum = 0, f_phi_dum = 0, f_cos_theta_dum = 0, f_theta_dum = 0;
    double phi_y_dum = 1, cos_theta_y_dum = 1, mu, phi_max, norm;
    double energy_ratio = p0 / (ORC_M_EL * ORC_C_LIGHT);
    double kn = orc_kleinNishinaCrossSection(energy_ratio);
    double rand_num = orc_rng_uniform(rng);

    if (!(rand_num <= kn)) return 0;

    while (cos_theta_y_dum > f_cos_theta_dum) {
        cos_theta_y_dum = orc_rng_uniform(rng) * 2;
        cos_theta_dum = orc_rng_uniform(rng) * 2 - 1;
        f_cos_theta_dum = pow((1 + energy_ratio * (1 - cos_theta_dum)), -2)
                          * (energy_ratio * (1 - cos_theta_dum) + (1 / (1 + energy_ratio * (1 - cos_theta_dum)))
                             + cos_theta_dum * cos_theta_dum);
    }
    *theta = acos(cos_theta_dum);
    mu = 1 + energy_ratio * (1 - cos(*theta));
    f_theta_dum = (pow(mu, -1.0) + pow(mu, -3.0) - pow(mu, -2.0) * sin(*theta) * sin(*theta)) * sin(*theta);

    while (phi_y_dum > f_phi_dum) {
        if (!c->stokes_switch || (u == 0 && q == 0)) {
            phi_dum = orc_rng_uniform(rng) * 2 * M_PI;
            phi_y_dum = -1;
        } else {
            phi_max = fabs(atan2(-u, q)) / 2.0;
            norm = (f_theta_dum + pow(mu, -2.0) * sin(*theta) * sin(*theta) * sin(*theta)
                                  * (q * cos(2 * phi_max) - u * sin(2 * phi_max)));
            phi_y_dum = orc_rng_uniform(rng);
            phi_dum = orc_rng_uniform(rng) * 2 * M_PI;
            f_phi_dum = (f_theta_dum + pow(mu, -2.0) * sin(*theta) * sin(*theta) * sin(*theta)
                                       * (q * cos(2 * phi_dum) - u * sin(2 * phi_dum))) / norm;
        }
    }
    *phi = phi_dum;
    return 1;
}

/* Modified Bessel function K_2(x), standing for gsl_sf_bessel_Kn(2, x) at
 * electron.c:221.  Evaluated from the integral representation
 *   K_nu(x) = int_0^inf exp(-x cosh t) cosh(nu t) dt        (DLMF 10.32.9)
 * with the trapezoid rule, which converges geometrically for this analytic,
 * doubly-exponentially decaying integrand.  Checked against scipy.special.kn
 * in tests/test_oracle_kat.py (relative error < 1e-13 on [0.05, 700]). */
double orc_bessel_K2(double x)
{
    double h = 0.35 / sqrt(x);
    if (h > 0.125) h = 0.125;
    double sum = 0.5;               /* t = 0 term: exp(0)*cosh(0), half weight */
    for (int k = 1; k < 100000; k++) {
        double t = k * h;
        double sh = sinh(0.5 * t);
        double e = -x * 2.0 * sh * sh;   /* -x (cosh t - 1) */
        double term = exp(e) * cosh(2.0 * t);
        sum += term;
        if (e + 2.0 * t < -80.0) break;
    }
    return exp(-x) * sum * h;
}

/* electron.c:202-237 */
double orc_sampleThermalElectron(double temp, orc_rng *rng)
{
    double gamma = 1, factor, x_dum = 0, y_dum = 1, f_x_dum = 0, beta_x_dum;
    if (temp >= 1e7) {
        factor = ORC_K_B * temp / (ORC_M_EL * ORC_C_LIGHT * ORC_C_LIGHT);
        double k2 = orc_bessel_K2(1.0 / factor);  /* loop-invariant; the reference re-evaluates it each pass */
        while (isnan(f_x_dum) || (y_dum > f_x_dum)) {
            x_dum = orc_rng_uniform_pos(rng) * (1 + 100 * factor);
            beta_x_dum = sqrt(1 - (1 / (x_dum * x_dum)));
            y_dum = orc_rng_uniform(rng) / 2.0;
            f_x_dum = x_dum * x_dum * (beta_x_dum / k2) * exp(-1 * x_dum / factor);
        }
        gamma = x_dum;
    } else {
        factor = sqrt(ORC_K_B * temp / ORC_M_EL);
        double g1 = orc_rng_gaussian(rng, factor) / ORC_C_LIGHT;
        double g2 = orc_rng_gaussian(rng, factor) / ORC_C_LIGHT;
        double g3 = orc_rng_gaussian(rng, factor) / ORC_C_LIGHT;
        gamma = 1.0 / sqrt(1 - (pow(g1, 2) + pow(g2, 2) + pow(g3, 2)));
    }
    return gamma;
}

/* electron.c:177-200 (eq. 56 of the RAIKOU paper) */
double orc_sampleElectronTheta(double beta, orc_rng *rng)
{
    /* the argument is -1 for a uniform of exactly 0 and +1 towards 1, and rounding can leave it an ulp outside [-1,1] there (acos -> NaN in the
     * reference; a 24-bit ranlxs0 returns exactly 0 once in 1.7e7 draws): clamped, as in the engine (documented deviation, mcrat_oracle.h) */
    double c = (1 - sqrt(1 + beta * beta + 2 * beta - 4 * beta * orc_rng_uniform(rng))) / beta;
    if (c < -1.0) c = -1.0;
    if (c > 1.0) c = 1.0;
    return acos(c);
}

/* electron.c:126-175 */
void orc_rotateElectron(double el_p[4], const double ph_p[4])
{
    double rot[9], tmp[3], out[3];
    double ph_phi = atan2(ph_p[2], ph_p[3]);
    double ph_theta = atan2(sqrt(pow(ph_p[2], 2) + pow(ph_p[3], 2)), ph_p[1]);

    memset(rot, 0, sizeof rot);
    rot[4] = 1;
    rot[8] = cos(ph_theta);
    rot[0] = cos(ph_theta);
    rot[2] = -sin(ph_theta);
    rot[6] = sin(ph_theta);
    matvec(3, rot, el_p + 1, tmp);

    memset(rot, 0, sizeof rot);
    rot[0] = 1;
    rot[4] = cos(-ph_phi);
    rot[8] = cos(-ph_phi);
    rot[5] = -sin(-ph_phi);
    rot[7] = sin(-ph_phi);
    matvec(3, rot, tmp, out);
    el_p[1] = out[0]; el_p[2] = out[1]; el_p[3] = out[2];
}

/* electron.c:70-94 (generateSingleElectron, electron.c:7-13, reduces to this
 * when NONTHERMAL_E_DIST == OFF) */
void orc_singleThermalElectron(double el_p[4], double temp, const double ph_p[4], orc_rng *rng)
{
    double gamma = orc_sampleThermalElectron(temp, rng);
    double beta = sqrt(1 - (1 / (gamma * gamma)));
    double phi = orc_rng_uniform(rng) * 2 * M_PI;
    double theta = orc_sampleElectronTheta(beta, rng);
    el_p[0] = gamma * ORC_M_EL * ORC_C_LIGHT;
    el_p[1] = gamma * ORC_M_EL * ORC_C_LIGHT * beta * cos(theta);
    el_p[2] = gamma * ORC_M_EL * ORC_C_LIGHT * beta * sin(theta) * sin(phi);
    el_p[3] = gamma * ORC_M_EL * ORC_C_LIGHT * beta * sin(theta) * cos(phi);
    orc_rotateElectron(el_p, ph_p);
}

/* mcrat_scattering.c:151-485 */
int orc_singleScatter(const orc_config *c, double el_comov[4], double ph_comov[4], double s[4], orc_rng *rng)
{
    const double z_axis[3] = {0, 0, 1};
    double el_v[3], neg_el_v[3], ph_pr[4], el_pr[4], ph_orig[4];
    double rot0[9], rot1[9], res0[3], res1[3], result[4];
    double phi0, phi1, phi = 0, theta = 0;
    int occurred;

    el_v[0] = el_comov[1] / el_comov[0];
    el_v[1] = el_comov[2] / el_comov[0];
    el_v[2] = el_comov[3] / el_comov[0];

    orc_lorentzBoost(el_v, el_comov, el_pr, 'e');   /* :217 */
    orc_lorentzBoost(el_v, ph_comov, ph_pr, 'p');   /* :218 */

    if (c->stokes_switch) orc_stokesRotation(el_v, ph_comov + 1, ph_pr + 1, s); /* :225 */

    memcpy(ph_orig, ph_pr, sizeof ph_orig);          /* :236-239 */

    phi0 = atan2(ph_pr[2], ph_pr[1]);                 /* :244 */
    memset(rot0, 0, sizeof rot0);
    rot0[8] = 1;
    rot0[0] = cos(-phi0);
    rot0[4] = cos(-phi0);
    rot0[1] = -sin(-phi0);
    rot0[3] = sin(-phi0);
    matvec(3, rot0, ph_pr + 1, res0);                 /* :252 */
    ph_pr[1] = res0[0];
    ph_pr[2] = 0;
    ph_pr[3] = res0[2];

    phi1 = atan2(res0[2], res0[0]);                   /* :269 */
    memset(rot1, 0, sizeof rot1);
    rot1[4] = 1;
    rot1[0] = cos(-phi1);
    rot1[8] = cos(-phi1);
    rot1[2] = -sin(-phi1);
    rot1[6] = sin(-phi1);
    matvec(3, rot1, ph_pr + 1, res1);                 /* :283 */
    ph_pr[1] = ph_pr[0];                              /* :294 */
    ph_pr[2] = res1[1];
    ph_pr[3] = 0;

    occurred = orc_kleinNishinaScatter(c, &theta, &phi, ph_pr[0], s[1], s[2], rng); /* :307 */

    if (occurred == 1) {
        result[0] = ph_pr[0] / (1 + ((ph_pr[0] * (1 - cos(theta))) / (ORC_M_EL * ORC_C_LIGHT))); /* :322 */
        result[1] = result[0] * cos(theta);
        result[2] = result[0] * sin(theta) * sin(phi);
        result[3] = result[0] * sin(theta) * cos(phi);
        /* :342-348 update the electron 4-momentum, which is never read again */

        memcpy(ph_pr, result, sizeof result);         /* :356-359 */
        memset(rot1, 0, sizeof rot1);
        rot1[4] = 1;
        rot1[0] = cos(-phi1);
        rot1[8] = cos(-phi1);
        rot1[2] = sin(-phi1);
        rot1[6] = -sin(-phi1);
        matvec(3, rot1, ph_pr + 1, res1);             /* :366 */
        ph_pr[1] = res1[0]; ph_pr[2] = res1[1]; ph_pr[3] = res1[2];

        memset(rot0, 0, sizeof rot0);
        rot0[8] = 1;
        rot0[0] = cos(-phi0);
        rot0[4] = cos(-phi0);
        rot0[1] = sin(-phi0);
        rot0[3] = -sin(-phi0);
        matvec(3, rot0, ph_pr + 1, res0);             /* :386 */

        if (c->stokes_switch) {
            double xt[3], yt[3], xn[3], yn[3], scatt[16], sres[4];
            orc_findXY(ph_orig + 1, z_axis, xt, yt);  /* :402 */
            orc_findXY(res0, ph_orig + 1, xn, yn);    /* :403 */
            phi = orc_findPhi(xt, yt, xn, yn);
            orc_mullerMatrixRotation(phi, s);

            theta = acos((ph_orig[1] * res0[0] + ph_orig[2] * res0[1] + ph_orig[3] * res0[2]) / (ph_orig[0] * ph_pr[0])); /* :408 */

            memset(scatt, 0, sizeof scatt);           /* :411-416, Fano's matrix */
            scatt[0]  = 1.0 + pow(cos(theta), 2.0) + ((1 - cos(theta)) * (ph_orig[0] - result[0]) / (ORC_M_EL * ORC_C_LIGHT));
            scatt[1]  = sin(theta) * sin(theta);
            scatt[4]  = sin(theta) * sin(theta);
            scatt[5]  = 1.0 + cos(theta) * cos(theta);
            scatt[10] = 2.0 * cos(theta);
            scatt[15] = 2.0 * cos(theta) + ((cos(theta)) * (1 - cos(theta)) * (ph_orig[0] - result[0]) / (ORC_M_EL * ORC_C_LIGHT));
            matvec(4, scatt, s, sres);                /* :418 */
            s[0] = sres[0] / sres[0];                 /* :430-433 */
            s[1] = sres[1] / sres[0];
            s[2] = sres[2] / sres[0];
            s[3] = sres[3] / sres[0];

            orc_findXY(res0, ph_orig + 1, xt, yt);    /* :438 */
            orc_findXY(res0, z_axis, xn, yn);         /* :441 */
            phi = orc_findPhi(xt, yt, xn, yn);
            orc_mullerMatrixRotation(phi, s);         /* :447 */
        }

        ph_pr[1] = res0[0]; ph_pr[2] = res0[1]; ph_pr[3] = res0[2]; /* :452-454 */
        neg_el_v[0] = -1 * el_v[0];
        neg_el_v[1] = -1 * el_v[1];
        neg_el_v[2] = -1 * el_v[2];
        orc_lorentzBoost(neg_el_v, ph_pr, ph_comov, 'p');            /* :465 */
        if (c->stokes_switch) orc_stokesRotation(neg_el_v, ph_pr + 1, ph_comov + 1, s); /* :473 */
    }
    return occurred;
}

/* ------------------------------------------------------------------ */
/* getThermalCrossSection, optical_depth.c:132-149: 10^interp(log10(h nu'/m_e c^2), log10(kT/m_e c^2)), the
 * interpolation being gsl_spline2d_eval_e on a gsl_interp2d_bilinear spline over the grid of hot_x_section.c:461-502.
 * GSL is not in this image; its published bilinear scheme (interp2d/bilinear.c of GSL 2.x: cell found by bisection
 * with x[i] <= x < x[i+1], the last cell closed; t = (x-x_i)/(x_{i+1}-x_i), u likewise;
 * z = (1-t)(1-u) z00 + t(1-u) z10 + (1-t)u z01 + t u z11) is restated here.
 * Outside the tabulated range GSL reports GSL_EDOM and the reference integrates the cross section afresh with
 * gsl_monte_plain (hot_x_section.c:563-599 -> :324-356, 2 x 500 000 numbers from the rank's generator): orc_tableFallbackCrossSection
 * below, from the keyed source (the look-up is counted in table_fallbacks). */
/* electron.c:538-561 */
double orc_singleMaxwellJuttner(double gamma, double theta)
{
    double normalization;
    if (theta > 1.e-2) normalization = orc_bessel_K2(1. / theta) * exp(1. / theta);
    else normalization = sqrt(M_PI * theta / 2.);
    return ((gamma * sqrt(gamma * gamma - 1.) / (theta * normalization)) * exp(-(gamma - 1.) / theta));
}

/* hot_x_section.c:370-400 */
double orc_boostedCrossSection(double norm_ph_comv, double mu, double gamma)
{
    const double beta = sqrt(gamma * gamma - 1.) / gamma;
    const double norm_ph_e = norm_ph_comv * gamma * (1. - mu * beta);
    return orc_kleinNishinaCrossSection(norm_ph_e) * (1. - mu * beta);
}

/* hot_x_section.c:324-357, with the plain Monte-Carlo rule written out (see mcrat_oracle.h) */
double orc_calculateTotalThermalCrossSection(double ph_comv, double theta, long long calls, uint64_t seed, int entry)
{
    const double xl[2] = {1, -1}, xu[2] = {1. + 12 * theta, 1};
    double total = 0;
    orc_rng r;
    orc_rng_init(&r, seed, 0);
    orc_rng_set_iteration(&r, (uint64_t)entry);
    for (int s = 0; s < 256 && s < calls; s++) {
        orc_rng_stream_begin(&r, (uint32_t)s, 4u);
        double sum = 0;
        for (long long k = s; k < calls; k += 256) {
            const double gamma = xl[0] + orc_rng_uniform_pos(&r) * (xu[0] - xl[0]);
            const double mu = xl[1] + orc_rng_uniform_pos(&r) * (xu[1] - xl[1]);
            sum += orc_singleMaxwellJuttner(gamma, theta) * orc_boostedCrossSection(ph_comv, mu, gamma);
        }
        total += sum;
    }
    const double result = ((xu[0] - xl[0]) * (xu[1] - xl[1])) * (total / (double)calls);
    return 0.5 * result;
}

/* hot_x_section.c:82-107 */
void orc_createHotCrossSection(double *thermal_table, int n_ph_e, int n_t, double log_ph_e_min, double log_ph_e_max,
                               double log_t_min, double log_t_max, long long calls, uint64_t seed)
{
    const double dt = (log_t_max - log_t_min) / n_t, dph_e = (log_ph_e_max - log_ph_e_min) / n_ph_e;
    for (int i = 0; i <= n_ph_e; i++)
        for (int j = 0; j <= n_t; j++) {
            const double comv_ph_e = pow(10., log_ph_e_min + i * dph_e), theta = pow(10., log_t_min + j * dt);
            thermal_table[(size_t)i * (n_t + 1) + j] = log10(orc_calculateTotalThermalCrossSection(comv_ph_e, theta, calls, seed, i * (n_t + 1) + j));
        }
}

/* interpolateThermalHotCrossSection's GSL_EDOM branch inside the loop (hot_x_section.c:563-599): calculateTotalThermalCrossSection (:324-356) at
 * (eps, theta) with the plain Monte-Carlo rule of orc_calculateTotalThermalCrossSection above, its 256 substreams keyed
 * {iteration = pass | (s + 1) << 48, word2 = slot, purpose = 9, the list's stream} -- the engine's physics.hpp:table_fallback_* -- and returned as
 * getThermalCrossSection returns it: 10^log10(integral) (:588, optical_depth.c:143).  (The normalisation of singleMaxwellJuttner depends on theta
 * only and is taken once.) */
double orc_tableFallbackCrossSection(const orc_config *c, double eps, double theta, const orc_rng *rng, uint32_t slot)
{
    const long long calls = c->fallback_calls > 0 ? c->fallback_calls : 500000;              /* :348 */
    const double xl[2] = {1, -1}, xu[2] = {1. + 12 * theta, 1};
    double normalization;
    if (theta > 1.e-2) normalization = orc_bessel_K2(1. / theta) * exp(1. / theta);
    else normalization = sqrt(M_PI * theta / 2.);
    orc_rng r;
    orc_rng_init(&r, rng ? rng->seed : 0, rng ? rng->stream : 0);
    const uint64_t pass = rng ? rng->iteration : 0;
    double total = 0;
    for (int s = 0; s < 256 && s < calls; s++) {
        orc_rng_set_iteration(&r, pass | ((uint64_t)(s + 1) << 48));
        orc_rng_stream_begin(&r, slot, 9u);
        double sum = 0;
        for (long long k = s; k < calls; k += 256) {
            const double gamma = xl[0] + orc_rng_uniform_pos(&r) * (xu[0] - xl[0]);
            const double mu = xl[1] + orc_rng_uniform_pos(&r) * (xu[1] - xl[1]);
            const double mj = ((gamma * sqrt(gamma * gamma - 1.) / (theta * normalization)) * exp(-(gamma - 1.) / theta));   /* electron.c:560 */
            sum += mj * orc_boostedCrossSection(eps, mu, gamma);
        }
        total += sum;
    }
    const double result = 0.5 * (((xu[0] - xl[0]) * (xu[1] - xl[1])) * (total / (double)calls));
    return pow(10.0, log10(result));
}

static long long g_table_fallbacks = 0;
long long orc_table_fallbacks(void) { return g_table_fallbacks; }
void orc_reset_table_fallbacks(void) { g_table_fallbacks = 0; }

static int bisect_cell(double x0, double dx, int n_cells, double x)
{
    /* gsl_interp_bsearch(xa, x, 0, n_cells) on xa[i] = x0 + i*dx */
    int ilo = 0, ihi = n_cells;
    while (ihi > ilo + 1) {
        int i = (ihi + ilo) / 2;
        if (x0 + i * dx > x) ihi = i; else ilo = i;
    }
    return ilo;
}

double orc_getThermalCrossSection(const orc_config *c, double photon_comv_e, double fluid_temp, int *miss)
{
    return orc_getThermalCrossSection_keyed(c, photon_comv_e, fluid_temp, NULL, 0, miss);
}

/* rng: the list's generator in the pass the look-up happens in, slot: the photon's index -- the key of the integral a look-up off the table takes */
double orc_getThermalCrossSection_keyed(const orc_config *c, double photon_comv_e, double fluid_temp, const orc_rng *rng, uint32_t slot, int *miss)
{
    if (c->tau_calculation != ORC_TAU_TABLE) return 1;                         /* optical_depth.c:125-127,147 */
    const double normalized_photon_comv_e = photon_comv_e / (ORC_M_EL * ORC_C_LIGHT);     /* :139 */
    const double theta = ORC_K_B * fluid_temp / (ORC_M_EL * ORC_C_LIGHT * ORC_C_LIGHT);   /* calcDimlessTheta, mc_cyclosynch.c:48-52 */
    const double x = log10(normalized_photon_comv_e), y = log10(theta);
    const double dx = (c->log_ph_e_max - c->log_ph_e_min) / c->n_ph_e;        /* hot_x_section.c:464 */
    const double dy = (c->log_t_max - c->log_t_min) / c->n_t;
    const double x_hi = c->log_ph_e_min + c->n_ph_e * dx, y_hi = c->log_t_min + c->n_t * dy;
    if (!(x >= c->log_ph_e_min) || x > x_hi || !(y >= c->log_t_min) || y > y_hi) {          /* gsl_spline2d_eval_e: GSL_EDOM */
        /* interpolateThermalHotCrossSection's fallback (hot_x_section.c:563-599): calculateTotalThermalCrossSection (:324-356) -- cold plasma below the
         * table is 1 / the Klein-Nishina cross section (:337-340); else its Monte-Carlo integral at 10^x, 10^y (:584-588) */
        const double theta_min = pow(10.0, c->log_t_min), e_min = pow(10.0, c->log_ph_e_min);
        if (theta < theta_min) return (normalized_photon_comv_e < e_min) ? 1.0 : orc_kleinNishinaCrossSection(normalized_photon_comv_e);
        g_table_fallbacks += 1;
        if (miss) *miss += 1;
        return orc_tableFallbackCrossSection(c, pow(10.0, x), pow(10.0, y), rng, slot);
    }
    const int xi = bisect_cell(c->log_ph_e_min, dx, c->n_ph_e, x);
    const int yi = bisect_cell(c->log_t_min, dy, c->n_t, y);
    const double xmin = c->log_ph_e_min + xi * dx, xmax = c->log_ph_e_min + (xi + 1) * dx;
    const double ymin = c->log_t_min + yi * dy, ymax = c->log_t_min + (yi + 1) * dy;
    const int ny = c->n_t + 1;
    const double zminmin = c->hot_table[xi * ny + yi], zminmax = c->hot_table[xi * ny + yi + 1];
    const double zmaxmin = c->hot_table[(xi + 1) * ny + yi], zmaxmax = c->hot_table[(xi + 1) * ny + yi + 1];
    const double t = (x - xmin) / (xmax - xmin), u = (y - ymin) / (ymax - ymin);
    const double z = (1. - t) * (1. - u) * zminmin + t * (1. - u) * zmaxmin + (1. - t) * u * zminmax + t * u * zmaxmax;
    return pow(10.0, z);                                                        /* :143 */
}

/* optical_depth.c:7-59 (getCrossSection :117-130: 1 in DIRECT, the table in TABLE) */
void orc_calculateOpticalDepth(const orc_config *c, orc_photon *ph, const orc_hydro *h)
{
    orc_calculateOpticalDepth_keyed(c, ph, h, NULL, 0);
}

void orc_calculateOpticalDepth_keyed(const orc_config *c, orc_photon *ph, const orc_hydro *h, const orc_rng *rng, uint32_t slot)
{
    int idx = ph->nearest_block_index;
    double fluid_beta[3];
    double ph_phi = atan2(ph->r1, ph->r0);
    cell_beta_cartesian(c, h, idx, ph_phi, fluid_beta);

    double fl_v_x = fluid_beta[0], fl_v_y = fluid_beta[1], fl_v_z = fluid_beta[2];
    double fl_v_norm = sqrt(fl_v_x * fl_v_x + fl_v_y * fl_v_y + fl_v_z * fl_v_z);
    double ph_v_norm = sqrt(ph->p1 * ph->p1 + ph->p2 * ph->p2 + ph->p3 * ph->p3);
    double n_cosangle = ((fl_v_x * ph->p1) + (fl_v_y * ph->p2) + (fl_v_z * ph->p3)) / (fl_v_norm * ph_v_norm);
    double beta = sqrt(1.0 - 1.0 / (h->gamma[idx] * h->gamma[idx]));
    double fluid_factor = (1.0 - beta * n_cosangle);
    double thermal_n_dens_lab = h->dens_lab[idx] / ORC_M_P;
    double norm_cross_section = orc_getThermalCrossSection_keyed(c, ph->comv_p0, h->temp[idx], rng, slot, NULL);   /* :58 */
    ph->total_optical_depth = (thermal_n_dens_lab) * (ORC_THOM_X_SECT * norm_cross_section) * fluid_factor;
}

/* mclib.c:436-615 */
int orc_findContainingHydroCell(const orc_config *c, orc_photon_list *l, const orc_hydro *h,
                                int find_nearest_block_switch, orc_stats *st)
{
    return orc_findContainingHydroCell_keyed(c, l, h, find_nearest_block_switch, st, NULL);
}

/* (the reference's takes gsl_rng *rand for the look-ups off the table, mclib.c:436; `rng` here keys them) */
int orc_findContainingHydroCell_keyed(const orc_config *c, orc_photon_list *l, const orc_hydro *h,
                                      int find_nearest_block_switch, orc_stats *st, const orc_rng *rng)
{
    int n_new = 0;
    for (int i = 0; i < l->list_capacity; i++) {
        orc_photon *ph = &l->photons[i];
        int ph_block_index = (find_nearest_block_switch == 0) ? ph->nearest_block_index : 0;
        double hc[3];
        orc_mcratCoordinateToHydroCoordinate(c, hc, ph->r0, ph->r1, ph->r2);

        int inside;
        if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE)
            inside = (hc[1] < h->r1_domain[1]) && (hc[1] > h->r1_domain[0]) &&
                     (hc[0] < h->r0_domain[1]) && (hc[0] > h->r0_domain[0]);
        else
            inside = (hc[2] < h->r2_domain[1]) && (hc[2] > h->r2_domain[0]) &&
                     (hc[1] < h->r1_domain[1]) && (hc[1] > h->r1_domain[0]) &&
                     (hc[0] < h->r0_domain[1]) && (hc[0] > h->r0_domain[0]);

        if (inside && (ph->nearest_block_index != -1)) {
            int is_in_block = orc_checkInBlock(c, hc[0], hc[1], hc[2], h, ph_block_index);
            if (find_nearest_block_switch == 1 || !is_in_block) {
                int min_index = orc_findContainingBlock(c, hc[0], hc[1], hc[2], h);
                ph->nearest_block_index = min_index;
                if (min_index != -1) {
                    double ph_p[4] = {ph->p0, ph->p1, ph->p2, ph->p3}, ph_p_comv[4], fluid_beta[3];
                    double ph_phi = atan2(ph->r1, ph->r0);
                    cell_beta_cartesian(c, h, min_index, ph_phi, fluid_beta);
                    orc_lorentzBoost(fluid_beta, ph_p, ph_p_comv, 'p');
                    ph->comv_p0 = ph_p_comv[0];
                    ph->comv_p1 = ph_p_comv[1];
                    ph->comv_p2 = ph_p_comv[2];
                    ph->comv_p3 = ph_p_comv[3];
                    orc_calculateOpticalDepth_keyed(c, ph, h, rng, (uint32_t)i);
                    if (ph->recalc_properties == 1) ph->recalc_properties = 0;
                    n_new += 1;
                } else if (st) {
                    st->not_found += 1;
                }
            }
        } else {
            ph->nearest_block_index = -1;
        }
    }
    if (find_nearest_block_switch != 0) n_new = 0;   /* mclib.c:608-611 */
    return n_new;
}

/* argsort comparator of mclib.c:753-763 with a total order on ties */
static int cmp_time_then_slot(const void *a, const void *b, void *ctx)
{
    const orc_photon *ph = (const orc_photon *)ctx;
    int aa = *(const int *)a, bb = *(const int *)b;
    double ta = ph[aa].time_to_scatter, tb = ph[bb].time_to_scatter;
    if (isnan(ta)) ta = INFINITY;
    if (isnan(tb)) tb = INFINITY;
    if (ta < tb) return -1;
    if (ta > tb) return 1;
    return (aa > bb) - (aa < bb);
}

/* optimised mode: photonEvent only ever reads sorted_indexes[0 .. k] for the few candidates it tries (mclib.c:1128-1133), so
 * only that prefix of the sorted order is produced: the ORC_PREFIX smallest (time, slot) pairs by one pass with a small
 * insertion buffer; if the walk runs past it (a long Klein-Nishina rejection chain) the full sort is done after all.
 * sorted_valid[thread] says how much of sorted_indexes is in its final place. */
#define ORC_PREFIX 8
static _Thread_local int sorted_valid = 0;

static int before(const orc_photon *ph, int a, int b)      /* the order of cmp_time_then_slot */
{
    return cmp_time_then_slot(&a, &b, (void *)ph) < 0;
}

static void sorted_prefix(orc_photon_list *l, int need)
{
    const int n = l->list_capacity;
    if (need >= ORC_PREFIX || n <= ORC_PREFIX) {
        for (int i = 0; i < n; i++) l->sorted_indexes[i] = i;
        qsort_r(l->sorted_indexes, (size_t)n, sizeof(int), cmp_time_then_slot, l->photons);
        sorted_valid = n;
        return;
    }
    int best[ORC_PREFIX], k = 0;
    for (int i = 0; i < n; i++) {
        if (k == ORC_PREFIX && !before(l->photons, i, best[k - 1])) continue;
        int j = (k < ORC_PREFIX) ? k++ : k - 1;
        while (j > 0 && before(l->photons, i, best[j - 1])) { best[j] = best[j - 1]; j--; }
        best[j] = i;
    }
    for (int i = 0; i < k; i++) l->sorted_indexes[i] = best[i];
    sorted_valid = k;
}

/* mclib.c:617-714 */
void orc_calcMeanFreePath(const orc_config *c, orc_photon_list *l, const orc_hydro *h, orc_rng *rng)
{
    const double default_mfp = 1e12;
    for (int i = 0; i < l->list_capacity; i++) {
        orc_photon *ph = &l->photons[i];
        double mfp;
        if (ph->nearest_block_index != -1) {
            if (ph->recalc_properties == 1) {
                orc_calculateOpticalDepth_keyed(c, ph, h, rng, (uint32_t)i);
                ph->recalc_properties = 0;
            }
            double rnd = orc_rng_freepath_draw(rng, (uint32_t)i);
            mfp = (-1.0 / ph->total_optical_depth) * log(rnd);
        } else {
            mfp = default_mfp;
        }
        ph->time_to_scatter = mfp / ORC_C_LIGHT;
    }
    if (c->optimised) { sorted_prefix(l, 0); return; }
    for (int i = 0; i < l->list_capacity; i++) l->sorted_indexes[i] = i;
    qsort_r(l->sorted_indexes, (size_t)l->list_capacity, sizeof(int), cmp_time_then_slot, l->photons);
}

/* mclib.c:1054-1100 (the two position norms computed there are unused) */
void orc_updatePhotonPosition(orc_photon_list *l, double t)
{
    for (int i = 0; i < l->list_capacity; i++) {
        orc_photon *ph = &l->photons[i];
        if ((ph->type != ORC_CS_POOL_PHOTON) && (ph->weight != 0)) {
            double divide_p0 = 1.0 / ph->p0;
            ph->r0 += ph->p1 * divide_p0 * ORC_C_LIGHT * t;
            ph->r1 += ph->p2 * divide_p0 * ORC_C_LIGHT * t;
            ph->r2 += ph->p3 * divide_p0 * ORC_C_LIGHT * t;
        }
    }
}

/* mclib.c:1107-1356 */
double orc_photonEvent(const orc_config *c, orc_photon_list *l, double dt_max, const orc_hydro *h,
                       int *scattered_ph_index, long long *frame_scatt_cnt, orc_rng *rng, orc_stats *st)
{
    int i = 0, ph_index = 0, event_did_occur = 0;
    double scatt_time = 0, old_scatt_time = 0;

    while (i < l->list_capacity && event_did_occur == 0) {
        if (c->optimised && i >= sorted_valid) sorted_prefix(l, i);
        ph_index = l->sorted_indexes[i];
        orc_photon *ph = &l->photons[ph_index];
        scatt_time = ph->time_to_scatter;

        if (scatt_time < dt_max) {
            orc_updatePhotonPosition(l, scatt_time - old_scatt_time);
            int index = ph->nearest_block_index;
            if (index != -1) {   /* documented deviation: see header */
                double fluid_temp = h->temp[index];
                double ph_phi = atan2(ph->r1, ph->r0);
                double fluid_beta[3], negative_fluid_beta[3];
                cell_beta_cartesian(c, h, index, ph_phi, fluid_beta);

                double ph_p[4] = {ph->p0, ph->p1, ph->p2, ph->p3};
                double ph_p_comov[4] = {ph->comv_p0, ph->comv_p1, ph->comv_p2, ph->comv_p3};
                double s[4] = {ph->s0, ph->s1, ph->s2, ph->s3};
                double el_p_comov[4];

                if (c->stokes_switch) orc_stokesRotation(fluid_beta, ph_p + 1, ph_p_comov + 1, s); /* :1227 */

                orc_rng_event_begin(rng, (uint32_t)ph_index);
                orc_singleThermalElectron(el_p_comov, fluid_temp, ph_p_comov, rng);            /* :1234 */
                event_did_occur = orc_singleScatter(c, el_p_comov, ph_p_comov, s, rng);        /* :1245 */
                if (st) st->event_draws += (long long)rng->n_draws;

                if (event_did_occur == 1) {
                    negative_fluid_beta[0] = -1 * fluid_beta[0];
                    negative_fluid_beta[1] = -1 * fluid_beta[1];
                    negative_fluid_beta[2] = -1 * fluid_beta[2];
                    orc_lorentzBoost(negative_fluid_beta, ph_p_comov, ph_p, 'p');             /* :1265 */
                    if (c->stokes_switch) {
                        orc_stokesRotation(negative_fluid_beta, ph_p_comov + 1, ph_p + 1, s);  /* :1280 */
                        ph->s0 = s[0]; ph->s1 = s[1]; ph->s2 = s[2]; ph->s3 = s[3];
                    }
                    ph->p0 = ph_p[0]; ph->p1 = ph_p[1]; ph->p2 = ph_p[2]; ph->p3 = ph_p[3];
                    ph->comv_p0 = ph_p_comov[0]; ph->comv_p1 = ph_p_comov[1];
                    ph->comv_p2 = ph_p_comov[2]; ph->comv_p3 = ph_p_comov[3];
                    ph->num_scatt += 1;
                    *frame_scatt_cnt += 1;
                    ph->recalc_properties = 1;
                } else if (st) {
                    st->kn_rejections += 1;
                }
            }
        } else {
            scatt_time = dt_max;
            orc_updatePhotonPosition(l, scatt_time - old_scatt_time);
            event_did_occur = 1;
        }
        old_scatt_time = scatt_time;
        i++;
    }
    *scattered_ph_index = ph_index;
    return scatt_time;
}

/* mclib.c:1358-1383 (CYCLOSYNCHROTRON_SWITCH OFF: no weight filter) */
double orc_averagePhotonEnergy(const orc_photon_list *l)
{
    double e_sum = 0, w_sum = 0;
    for (int i = 0; i < l->list_capacity; i++) {
        e_sum += l->photons[i].p0 * l->photons[i].weight;
        w_sum += l->photons[i].weight;
    }
    return (e_sum * ORC_C_LIGHT) / w_sum;
}

/* mclib.c:1385-1462 (CYCLOSYNCHROTRON_SWITCH OFF) */
void orc_phScattStats(const orc_photon_list *l, int *max, int *min, double *avg, double *r_avg)
{
    int temp_max = 0, temp_min = INT_MAX, count = 0;
    double sum = 0, avg_r_sum = 0;
    for (int i = 0; i < l->list_capacity; i++) {
        const orc_photon *ph = &l->photons[i];
        sum += ph->num_scatt;
        avg_r_sum += sqrt(ph->r0 * ph->r0 + ph->r1 * ph->r1 + ph->r2 * ph->r2);
        if (ph->num_scatt > temp_max) temp_max = (int)ph->num_scatt;
        if (ph->num_scatt < temp_min) temp_min = (int)ph->num_scatt;
        count++;
    }
    *avg = sum / count;
    *r_avg = avg_r_sum / count;
    *max = temp_max;
    *min = temp_min;
}

/* mclib.c:1465-1515 */
void orc_phMinMax(const orc_photon_list *l, double *min, double *max, double *min_theta, double *max_theta)
{
    double r_max = 0, r_min = DBL_MAX, th_max = 0, th_min = DBL_MAX;
    for (int i = 0; i < l->list_capacity; i++) {
        const orc_photon *ph = &l->photons[i];
        if (ph->weight != 0) {
            double r = sqrt(ph->r0 * ph->r0 + ph->r1 * ph->r1 + ph->r2 * ph->r2);
            double th = acos(ph->r2 / r);
            if (r > r_max) r_max = r;
            if (r < r_min) r_min = r;
            if (th > th_max) th_max = th;
            if (th < th_min) th_min = th;
        }
    }
    *max = r_max; *min = r_min; *max_theta = th_max; *min_theta = th_min;
}

/* mcrat.c:754-851 */
void orc_photon_loop(const orc_config *c, orc_photon_list *l, const orc_hydro *h, orc_rng *rng,
                     double *time_now, double *remaining_time, int *find_nearest_grid_switch,
                     long long max_iterations, uint64_t iteration_base, orc_stats *st)
{
    long long it = 0;
    double time_step = 0;
    const long long misses0 = g_table_fallbacks;
    while (*remaining_time > 0 && (max_iterations <= 0 || it < max_iterations)) {
        orc_rng_set_iteration(rng, iteration_base + (uint64_t)it);
        st->num_photons_find_new_element += orc_findContainingHydroCell_keyed(c, l, h, *find_nearest_grid_switch, st, rng);
        orc_calcMeanFreePath(c, l, h, rng);
        *find_nearest_grid_switch = 0;

        if (l->photons[l->sorted_indexes[0]].time_to_scatter < *remaining_time) {
            time_step = orc_photonEvent(c, l, *remaining_time, h, &st->last_scattered_index,
                                        &st->frame_scatt_cnt, rng, st);
            *time_now += time_step;
            *remaining_time -= time_step;
        } else {
            *time_now += *remaining_time;
            orc_updatePhotonPosition(l, *remaining_time);
            time_step = *remaining_time;
            *remaining_time = 0;
        }
        it++;
        st->photon_steps += l->list_capacity;
    }
    st->iterations += it;
    st->last_time_step = time_step;
    st->remaining_time = *remaining_time;
    st->time_now = *time_now;
    st->table_fallbacks += g_table_fallbacks - misses0;
}

/* ------------------------------------------------------------------ */
/* photonInjection and its helpers */

/* geometry.c:66-106 */
void orc_hydroCoordinateToSpherical(const orc_config *c, double *r, double *theta, double r0, double r1, double r2)
{
    double sph_r = 0, sph_theta = 0;
    if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE) {
        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) {
            sph_r = sqrt(r0 * r0 + r1 * r1);
            sph_theta = atan2(r0, r1);
        }
        if (c->geometry == ORC_SPHERICAL) { sph_r = r0; sph_theta = r1; }
    } else {
        if (c->geometry == ORC_CARTESIAN) { sph_r = sqrt(r0 * r0 + r1 * r1 + r2 * r2); sph_theta = acos(r2 / sph_r); }
        if (c->geometry == ORC_SPHERICAL) { sph_r = r0; sph_theta = r1; }
        if (c->geometry == ORC_POLAR) { sph_r = sqrt(r0 * r0 + r2 * r2); sph_theta = acos(r2 / sph_r); }
    }
    *r = sph_r;
    *theta = sph_theta;
}

/* geometry.c:108-156 */
void orc_hydroCoordinateToMcratCoordinate(const orc_config *c, double out[3], double r0, double r1, double r2)
{
    double x = 0, y = 0, z = 0;
    if (c->dimensions == ORC_TWO || c->dimensions == ORC_TWO_POINT_FIVE) {
        if (c->geometry == ORC_CARTESIAN || c->geometry == ORC_CYLINDRICAL) { x = r0 * cos(r2); y = r0 * sin(r2); z = r1; }
        if (c->geometry == ORC_SPHERICAL) { x = r0 * sin(r1) * cos(r2); y = r0 * sin(r1) * sin(r2); z = r0 * cos(r1); }
    } else {
        if (c->geometry == ORC_CARTESIAN) { x = r0; y = r1; z = r2; }
        if (c->geometry == ORC_SPHERICAL) { x = r0 * sin(r1) * cos(r2); y = r0 * sin(r1) * sin(r2); z = r0 * cos(r1); }
        if (c->geometry == ORC_POLAR) { x = r0 * cos(r1); y = r0 * sin(r1); z = r2; }
    }
    out[0] = x; out[1] = y; out[2] = z;
}

/* stands for gsl_ran_poisson (mclib.c:110): Knuth's product of uniforms below a mean of 30, PTRS (W. Hormann, "The
 * transformed rejection method for generating Poisson random variables", Insurance: Mathematics and Economics 12, 1993)
 * above */
long long orc_poisson(orc_rng *r, double mean)
{
    if (!(mean > 0)) return 0;
    if (mean < 30.0) {
        const double L = exp(-mean);
        long long k = 0;
        double p = 1.0;
        do {
            k += 1;
            p *= orc_rng_uniform_pos(r);
        } while (p > L);
        return k - 1;
    }
    const double smu = sqrt(mean);
    const double b = 0.931 + 2.53 * smu;
    const double a = -0.059 + 0.02483 * b;
    const double inv_alpha = 1.1239 + 1.1328 / (b - 3.4);
    const double v_r = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < (1 << 22); ++it) {
        const double U = orc_rng_uniform(r) - 0.5;
        const double V = orc_rng_uniform_pos(r);
        const double us = 0.5 - fabs(U);
        const double kf = floor((2.0 * a / us + b) * U + mean + 0.43);
        if (us >= 0.07 && V <= v_r) return (long long)kf;
        if (kf < 0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(inv_alpha) - log(a / (us * us) + b) <= -mean + kf * log(mean) - lgamma(kf + 1.0)) return (long long)kf;
    }
    return (long long)mean;
}

void orc_free(void *p) { free(p); }

static int cell_in_injection_slab(const orc_config *c, const orc_hydro *h, int i, double rmin, double rmax, double theta_min, double theta_max)
{
    double r_in, th_in, r_out, th_out;
    if (c->dimensions == ORC_THREE) {                                   /* mclib.c:42-49 */
        orc_hydroCoordinateToSpherical(c, &r_in, &th_in, fabs(h->r0[i]) - 0.5 * h->r0_size[i], fabs(h->r1[i]) - 0.5 * h->r1_size[i],
                                       fabs(h->r2[i]) - 0.5 * h->r2_size[i]);
        orc_hydroCoordinateToSpherical(c, &r_out, &th_out, fabs(h->r0[i]) + 0.5 * h->r0_size[i], fabs(h->r1[i]) + 0.5 * h->r1_size[i],
                                       fabs(h->r2[i]) + 0.5 * h->r2_size[i]);
    } else {                                                            /* mclib.c:51-52 */
        orc_hydroCoordinateToSpherical(c, &r_in, &th_in, h->r0[i] - 0.5 * h->r0_size[i], h->r1[i] - 0.5 * h->r1_size[i], 0);
        orc_hydroCoordinateToSpherical(c, &r_out, &th_out, h->r0[i] + 0.5 * h->r0_size[i], h->r1[i] + 0.5 * h->r1_size[i], 0);
    }
    return (rmin <= r_out) && (r_in <= rmax) && (th_out >= theta_min) && (th_in <= theta_max);   /* mclib.c:57 */
}

/* mclib.c:9-300 */
int orc_photonInjection(const orc_config *c, orc_photon **out, int *n_out, double *weight_out, double r_inj, double ph_weight,
                        int min_photons, int max_photons, char spect, double theta_min, double theta_max,
                        const orc_hydro *h, uint64_t seed, uint32_t stream)
{
    const float num_dens_coeff = (spect == 'w') ? 8.44f : 20.29f;      /* :23-32: a float in the reference */
    const double rmin = r_inj - 0.5 * ORC_C_LIGHT / h->fps, rmax = r_inj + 0.5 * ORC_C_LIGHT / h->fps;
    const int M = h->num_elements;
    int *ph_dens = (int *)malloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    if (!ph_dens) return -1;
    orc_rng rng;
    orc_rng_init(&rng, seed, stream);
    long long ph_tot = 0;
    double ph_weight_adjusted = ph_weight;
    uint64_t attempt = 0;
    while ((ph_tot > max_photons) || (ph_tot < min_photons)) {          /* :87-136 */
        ph_tot = 0;
        orc_rng_set_iteration(&rng, attempt);
        for (int i = 0; i < M; i++) {
            ph_dens[i] = 0;
            if (cell_in_injection_slab(c, h, i, rmin, rmax, theta_min, theta_max)) {
                const double ph_dens_calc = (4.0 / 3.0) * orc_hydroElementVolume(c, h, i) *
                                            (((h->gamma)[i] * num_dens_coeff * (h->temp)[i] * (h->temp)[i] * (h->temp)[i]) / ph_weight_adjusted);
                orc_rng_stream_begin(&rng, (uint32_t)i, ORC_PURPOSE_INJECT_COUNT);
                {   /* :114; a count beyond int range (a weight far too small) is held at INT_MAX, the total is summed in 64 bits */
                    const long long kk = orc_poisson(&rng, ph_dens_calc);
                    ph_dens[i] = (int)(kk > INT_MAX ? INT_MAX : kk);
                }
                ph_tot += ph_dens[i];
            }
        }
        if (ph_tot > max_photons) ph_weight_adjusted *= 10;
        else if (ph_tot < min_photons) ph_weight_adjusted *= 0.5;
        attempt += 1;
        if (attempt > 200) { free(ph_dens); return -2; }
    }
    orc_photon *ph = (orc_photon *)calloc((size_t)(ph_tot > 0 ? ph_tot : 1), sizeof(orc_photon));
    if (!ph) { free(ph_dens); return -1; }
    orc_rng_set_iteration(&rng, 0);
    long long k = 0;
    for (int i = 0; i < M; i++) {
        for (int j = 0; j < ph_dens[i]; j++, k++) {
            double fr_dum = 0;
            orc_rng_stream_begin(&rng, (uint32_t)k, ORC_PURPOSE_INJECT_PHOTON);
            if (spect == 'w') {                                         /* :175-190 */
                double y_dum = 1, yfr_dum = 0;
                while (y_dum > yfr_dum) {
                    fr_dum = orc_rng_uniform_pos(&rng) * 6.3e11 * ((h->temp)[i]);
                    y_dum = orc_rng_uniform_pos(&rng);
                    yfr_dum = (1.0 / (1.29e31)) * pow((fr_dum / ((h->temp)[i])), 3.0) / (exp((ORC_PL_CONST * fr_dum) / (ORC_K_B * ((h->temp)[i]))) - 1);
                }
            } else {                                                    /* :199-214, Bjorkman & Wood 2001 */
                double test = 0, test_cnt = 0;
                const double test_rand1 = orc_rng_uniform_pos(&rng), test_rand2 = orc_rng_uniform_pos(&rng), test_rand3 = orc_rng_uniform_pos(&rng),
                             test_rand4 = orc_rng_uniform_pos(&rng), test_rand5 = orc_rng_uniform_pos(&rng);
                while (test < M_PI * M_PI * M_PI * M_PI * test_rand1 / 90.0) {
                    test_cnt += 1;
                    test += 1 / (test_cnt * test_cnt * test_cnt * test_cnt);
                }
                fr_dum = -log(test_rand2 * test_rand3 * test_rand4 * test_rand5) / test_cnt;
                fr_dum *= ORC_K_B * ((h->temp)[i]) / ORC_PL_CONST;
            }
            double position_phi = 0;
            if (c->dimensions != ORC_THREE) position_phi = orc_rng_uniform(&rng) * 2 * M_PI;      /* :223-227 */
            const double com_v_phi = orc_rng_uniform(&rng) * 2 * M_PI;
            const double com_v_theta = acos((orc_rng_uniform(&rng) * 2) - 1);
            double p_comv[4], boost[3], l_boost[4];
            p_comv[0] = ORC_PL_CONST * fr_dum / ORC_C_LIGHT;                                      /* :232-235 */
            p_comv[1] = (ORC_PL_CONST * fr_dum / ORC_C_LIGHT) * sin(com_v_theta) * cos(com_v_phi);
            p_comv[2] = (ORC_PL_CONST * fr_dum / ORC_C_LIGHT) * sin(com_v_theta) * sin(com_v_phi);
            p_comv[3] = (ORC_PL_CONST * fr_dum / ORC_C_LIGHT) * cos(com_v_theta);
            cell_beta_cartesian(c, h, i, position_phi, boost);                                      /* :239-246 */
            boost[0] *= -1; boost[1] *= -1; boost[2] *= -1;
            orc_lorentzBoost(boost, p_comv, l_boost, 'p');                                          /* :252 */
            orc_photon *q = &ph[k];
            q->p0 = l_boost[0]; q->p1 = l_boost[1]; q->p2 = l_boost[2]; q->p3 = l_boost[3];
            q->comv_p0 = p_comv[0]; q->comv_p1 = p_comv[1]; q->comv_p2 = p_comv[2]; q->comv_p3 = p_comv[3];
            const double position_rand = orc_rng_uniform_pos(&rng) * ((h->r0_size)[i]) - 0.5 * ((h->r0_size)[i]);   /* :265-266 */
            const double position2_rand = orc_rng_uniform_pos(&rng) * ((h->r1_size)[i]) - 0.5 * ((h->r1_size)[i]);
            double xyz[3];
            if (c->dimensions == ORC_THREE) {
                const double position3_rand = orc_rng_uniform_pos(&rng) * ((h->r2_size)[i]) - 0.5 * ((h->r2_size)[i]);
                orc_hydroCoordinateToMcratCoordinate(c, xyz, (h->r0)[i] + position_rand, (h->r1)[i] + position2_rand, (h->r2)[i] + position3_rand);
            } else {
                orc_hydroCoordinateToMcratCoordinate(c, xyz, (h->r0)[i] + position_rand, (h->r1)[i] + position2_rand, position_phi);
            }
            q->r0 = xyz[0]; q->r1 = xyz[1]; q->r2 = xyz[2];
            q->s0 = 1; q->s1 = 0; q->s2 = 0; q->s3 = 0;                                             /* :281-291 */
            q->num_scatt = 0;
            q->weight = ph_weight_adjusted;
            q->nearest_block_index = 0;
            q->type = ORC_INJECTED_PHOTON;
            q->recalc_properties = 1;
        }
    }
    free(ph_dens);
    *out = ph;
    *n_out = (int)ph_tot;
    *weight_out = ph_weight_adjusted;
    return 0;
}
