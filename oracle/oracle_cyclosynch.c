/*
 * oracle_cyclosynch.c -- CPU oracle for the cyclo-synchrotron row, first part: the photon-list
 * operations it needs, the magnetic-field helpers, the emission of the pool photons and of a single replacement photon,
 * and the absorption at the end of a frame.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED -- see mcrat_oracle.h.  The device
 * side of this row is not built yet: this file is the checker it will be built against.
 * orc_scatter_frame_cs at the end of the file is the scatter-frame body of main() with the switch ON (mcrat.c:706-878).
 *
 * Restated from the reference:
 *   list operations          Src/photons.c:3-285 (exit(1) paths become error returns)
 *   calcCyclotronFreq ... getMagneticFieldMagnitude   Src/mc_cyclosynch.c:30-92
 *   blackbody_ph_spect       :185-196;   calcCyclosynchRLimits :225-244
 *   photonEmitCyclosynch     :1176-1569 (both inject_single_switch branches)
 *   phAbsCyclosynch          :1571-1623
 *   rebinCyclosynchCompPhotons :246-712 (the gsl_histogram2d objects only find bin indexes; restated)
 * Third-party arithmetic: gsl_integration_qags (:1276) lives in GSL (unpinned version).  Its published algorithm (QUADPACK
 * QAGS) starts with one 21-point Gauss-Kronrod rule on the whole interval and returns at once when that rule's error
 * estimate meets the tolerance; the integrand here -- the Planck photon number density from 10 Hz to the cyclotron
 * frequency, far down the Rayleigh-Jeans tail, with epsrel = 1e-2 -- is linear in nu to many digits, so QAGS never gets
 * past that first rule.  orc_qags restates the rule (QUADPACK's nodes, weights and error heuristic) and the first-step test;
 * should a caller hand it an integrand that does not converge at once it falls back to plain bisection of the worst
 * interval WITHOUT the epsilon-algorithm extrapolation and reports that through *used_fallback (tests assert it stays 0).
 * gsl_ran_poisson: the oracle's own sampler, as for photonInjection (orc_poisson).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mcrat_oracle.h"

#define ORC_CHARGE_EL 4.8032068e-10       /* Src/mclib.c:4-5 */

/* ---- Src/photons.c ------------------------------------------------------------------------------------------- */
void orc_list_init(orc_photon_list *l) { memset(l, 0, sizeof *l); }                       /* :3-10 */

void orc_list_free(orc_photon_list *l)                                                     /* :12-21 */
{
    free(l->photons); free(l->sorted_indexes);
    memset(l, 0, sizeof *l);
}

static int verify_num(const orc_photon_list *l) { return (l->num_photons + l->num_null_photons != l->list_capacity) ? -1 : 0; }   /* :277-285 */

int orc_list_set_null(orc_photon_list *l, int index)                                       /* setNullPhoton :210-250 */
{
    orc_photon *p = &l->photons[index];
    p->type = ORC_NULL_PHOTON;
    p->weight = 0;
    p->nearest_block_index = -1;
    p->recalc_properties = 0;
    p->p0 = p->p1 = p->p2 = p->p3 = 0;
    p->comv_p0 = p->comv_p1 = p->comv_p2 = p->comv_p3 = 0;
    p->r0 = p->r1 = p->r2 = 0;
    p->s0 = p->s1 = p->s2 = p->s3 = 0;
    p->num_scatt = 0;
    p->total_optical_depth = 0;
    l->num_photons -= 1;                                                                   /* incrementNullPhotonNum :268-274 */
    l->num_null_photons += 1;
    return verify_num(l);
}

int orc_list_realloc(orc_photon_list *l, int new_capacity)                                 /* :37-80 */
{
    const int old = l->list_capacity;
    orc_photon *np = (orc_photon *)realloc(l->photons, (size_t)new_capacity * sizeof(orc_photon));
    int *ns = (int *)realloc(l->sorted_indexes, (size_t)new_capacity * sizeof(int));
    if (!np || !ns) return -2;
    l->photons = np; l->sorted_indexes = ns;
    l->list_capacity = new_capacity;
    l->num_photons += (new_capacity - old);        /* the new slots count as real ... */
    for (int i = old; i < new_capacity; i++)
        if (orc_list_set_null(l, i)) return -1;    /* ... until setNullPhoton turns each into a null one */
    return 0;
}

int orc_list_set(orc_photon_list *l, const orc_photon *ph, int n)                          /* setPhotonList :82-106 */
{
    if (l->photons) orc_list_free(l);
    l->photons = (orc_photon *)malloc((size_t)(n > 0 ? n : 1) * sizeof(orc_photon));
    l->sorted_indexes = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
    if (!l->photons || !l->sorted_indexes) return -2;
    memcpy(l->photons, ph, (size_t)n * sizeof(orc_photon));
    l->list_capacity = n;
    l->num_photons = n;                            /* as the reference: null photons are counted in num_photons here too */
    int nulls = 0;
    for (int i = 0; i < n; i++) nulls += l->photons[i].type == ORC_NULL_PHOTON;
    l->num_null_photons = nulls;
    return 0;
}

/* addToPhotonList :108-208.  Returns 0, -1 on the conservation error, -3 where the reference prints "Adding to the photon list
 * has failed" and exits (fewer null slots than photons to add, and no growth) */
int orc_list_add(orc_photon_list *l, const orc_photon *ph, int num)
{
    if ((l->num_photons >= l->list_capacity) && (l->num_null_photons <= num)) {
        int new_capacity;
        if (l->list_capacity * 2 > l->list_capacity + num) new_capacity = l->list_capacity * 2;
        else new_capacity = l->list_capacity * (num / l->list_capacity);
        const int rc = orc_list_realloc(l, new_capacity);
        if (rc) return rc;
    }
    if (num == 1) {
        int idx = 0;
        if (l->num_null_photons == 0) idx = l->num_photons;
        else
            for (int i = 0; i < l->list_capacity; i++)
                if (l->photons[i].type == ORC_NULL_PHOTON) { idx = i; break; }
        if (idx < 0 || idx >= l->list_capacity) return -3;     /* the reference would write past the array here */
        memcpy(&l->photons[idx], ph, sizeof(orc_photon));
        l->num_photons += 1; l->num_null_photons -= 1;          /* incrementPhotonNum :260-266 */
        return verify_num(l);
    }
    if (num > l->num_null_photons) return -3;
    int *nulls = (int *)malloc(sizeof(int) * (size_t)(l->num_null_photons > 0 ? l->num_null_photons : 1)), j = 0;
    for (int i = 0; i < l->list_capacity; i++)
        if (l->photons[i].type == ORC_NULL_PHOTON) nulls[j++] = i;
    int rc = 0;
    for (int i = 0; i < num && !rc; i++)
        if (ph[i].type != ORC_NULL_PHOTON) {
            memcpy(&l->photons[nulls[i]], &ph[i], sizeof(orc_photon));
            l->num_photons += 1; l->num_null_photons -= 1;
            rc = verify_num(l);
        }
    free(nulls);
    return rc;
}

/* ---- mc_cyclosynch.c:30-92 ----------------------------------------------------------------------------------- */
double orc_calcCyclotronFreq(double magnetic_field) { return ORC_CHARGE_EL * magnetic_field / (2 * M_PI * ORC_M_EL * ORC_C_LIGHT); }
double orc_calcEB(double magnetic_field) { return ORC_PL_CONST * orc_calcCyclotronFreq(magnetic_field); }
double orc_calcDimlessTheta(double temp) { return ORC_K_B * temp / (ORC_M_EL * ORC_C_LIGHT * ORC_C_LIGHT); }
double orc_calcBoundaryE(double magnetic_field, double temp)
{
    return 14 * pow(ORC_M_EL * ORC_C_LIGHT * ORC_C_LIGHT, 1.0 / 10.0) * pow(orc_calcEB(magnetic_field), 9.0 / 10.0) * pow(orc_calcDimlessTheta(temp), 3.0 / 10.0);
}

double orc_calcB(const orc_cs *cs, double el_dens, double temp)                            /* :54-76 */
{
    if (cs->b_field_calc == ORC_B_INTERNAL_E) return sqrt(cs->epsilon_b * 8 * M_PI * 3 * el_dens * ORC_K_B * temp / 2);
    if (cs->b_field_calc == ORC_B_TOTAL_E)
        return sqrt(8 * M_PI * cs->epsilon_b * (el_dens * ORC_M_P * ORC_C_LIGHT * ORC_C_LIGHT + 4 * ORC_A_RAD * temp * temp * temp * temp / 3));
    return 0;
}

double orc_getMagneticFieldMagnitude(const orc_config *c, const orc_cs *cs, const orc_hydro *h, int i)   /* :78-92 */
{
    if (cs->b_field_calc == ORC_B_INTERNAL_E || cs->b_field_calc == ORC_B_TOTAL_E) return orc_calcB(cs, cs->dens[i] / ORC_M_P, h->temp[i]);
    if (c->dimensions == ORC_TWO) return sqrt(cs->B0[i] * cs->B0[i] + cs->B1[i] * cs->B1[i]);          /* vectorMagnitude, geometry.c:176-187 */
    return sqrt(cs->B0[i] * cs->B0[i] + cs->B1[i] * cs->B1[i] + cs->B2[i] * cs->B2[i]);
}

double orc_blackbody_ph_spect(double nu, double temp)                                       /* :185-196 */
{
    return (8 * M_PI * nu * nu) / (exp(ORC_PL_CONST * nu / (ORC_K_B * temp)) - 1) / (ORC_C_LIGHT * ORC_C_LIGHT * ORC_C_LIGHT);
}

double orc_calcCyclosynchRLimits(int frame_scatt, int frame_inj, double fps, double r_inj, int want_max)   /* :225-244 */
{
    double val = r_inj;
    if (!want_max) val += (ORC_C_LIGHT * (frame_scatt - frame_inj) / fps - 0.5 * ORC_C_LIGHT / fps);
    else val += (ORC_C_LIGHT * (frame_scatt - frame_inj) / fps + 0.5 * ORC_C_LIGHT / fps);
    return val;
}

/* ---- QUADPACK's 21-point Gauss-Kronrod rule (dqk21) and QAGS' first step (see the head of this file) ---------- */
static const double XGK[11] = {0.995657163025808080735527280689003, 0.973906528517171720077964012084452, 0.930157491355708226001207180059508,
                               0.865063366688984510732096688423493, 0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
                               0.562757134668604683339000099272694, 0.433395394129247190799265943165784, 0.294392862701460198131126603103866,
                               0.148874338981631210884826001129720, 0.0};
static const double WGK[11] = {0.011694638867371874278064396062192, 0.032558162307964727478818972459390, 0.054755896574351996031381300244580,
                               0.075039674810919952767043140916190, 0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
                               0.123491976262065851077958109585166, 0.134709217311473325928054001771707, 0.142775938577060080797094273138717,
                               0.147739104901338491374841515972068, 0.149445554002916905664936468389821};
static const double WG[5] = {0.066671344308688137593568809893332, 0.149451349150580593145776339657697, 0.219086362515982043995534934228163,
                             0.269266719309996355091226921569469, 0.295524224714752870173815619188769};

void orc_qk21(orc_integrand f, void *ctx, double a, double b, double *result, double *abserr, double *resabs, double *resasc)
{
    const double centr = 0.5 * (a + b), hlgth = 0.5 * (b - a), dhlgth = fabs(hlgth);
    const double fc = f(centr, ctx);
    double fv1[10], fv2[10], resg = 0, resk = WGK[10] * fc, rabs = fabs(resk);
    for (int j = 0; j < 5; j++) {
        const int jtw = 2 * j + 1;
        const double absc = hlgth * XGK[jtw], f1 = f(centr - absc, ctx), f2 = f(centr + absc, ctx);
        fv1[jtw] = f1; fv2[jtw] = f2;
        resg += WG[j] * (f1 + f2);
        resk += WGK[jtw] * (f1 + f2);
        rabs += WGK[jtw] * (fabs(f1) + fabs(f2));
    }
    for (int j = 0; j < 5; j++) {
        const int jtwm1 = 2 * j;
        const double absc = hlgth * XGK[jtwm1], f1 = f(centr - absc, ctx), f2 = f(centr + absc, ctx);
        fv1[jtwm1] = f1; fv2[jtwm1] = f2;
        resk += WGK[jtwm1] * (f1 + f2);
        rabs += WGK[jtwm1] * (fabs(f1) + fabs(f2));
    }
    const double reskh = resk * 0.5;
    double rasc = WGK[10] * fabs(fc - reskh);
    for (int j = 0; j < 10; j++) rasc += WGK[j] * (fabs(fv1[j] - reskh) + fabs(fv2[j] - reskh));
    double err = fabs((resk - resg) * hlgth);
    *result = resk * hlgth;
    rabs *= dhlgth; rasc *= dhlgth;
    if (rasc != 0 && err != 0) { const double s = pow(200 * err / rasc, 1.5); err = (s < 1) ? rasc * s : rasc; }
    if (rabs > DBL_MIN / (50 * DBL_EPSILON)) { const double m = 50 * DBL_EPSILON * rabs; if (m > err) err = m; }
    *abserr = err; *resabs = rabs; *resasc = rasc;
}

int orc_qags(orc_integrand f, void *ctx, double a, double b, double epsabs, double epsrel, int limit, double *result, double *abserr,
             int *used_fallback)
{
    double res, err, rabs, rasc;
    if (used_fallback) *used_fallback = 0;
    orc_qk21(f, ctx, a, b, &res, &err, &rabs, &rasc);
    double tol = fmax(epsabs, epsrel * fabs(res));
    *result = res; *abserr = err;
    if (err <= 100 * DBL_EPSILON * rabs && err > tol) return 1;                    /* GSL_EROUND */
    if ((err <= tol && err != rasc) || err == 0.0) return 0;
    if (limit <= 1) return 2;
    /* not QAGS any more: bisect the interval with the largest error until the sum of the errors meets the tolerance */
    if (used_fallback) *used_fallback = 1;
    double *al = (double *)malloc(sizeof(double) * 4 * (size_t)limit), *bl = al + limit, *rl = bl + limit, *el = rl + limit;
    int n = 1;
    al[0] = a; bl[0] = b; rl[0] = res; el[0] = err;
    int rc = 2;
    while (n < limit) {
        int worst = 0;
        for (int k = 1; k < n; k++) if (el[k] > el[worst]) worst = k;
        const double mid = 0.5 * (al[worst] + bl[worst]);
        double r1, e1, r2, e2, t1, t2;
        orc_qk21(f, ctx, al[worst], mid, &r1, &e1, &t1, &t2);
        orc_qk21(f, ctx, mid, bl[worst], &r2, &e2, &t1, &t2);
        al[n] = mid; bl[n] = bl[worst]; rl[n] = r2; el[n] = e2;
        bl[worst] = mid; rl[worst] = r1; el[worst] = e1;
        n++;
        double sr = 0, se = 0;
        for (int k = 0; k < n; k++) { sr += rl[k]; se += el[k]; }
        *result = sr; *abserr = se;
        tol = fmax(epsabs, epsrel * fabs(sr));
        if (se <= tol) { rc = 0; break; }
    }
    free(al);
    return rc;
}

static double planck_tail(double nu, void *ctx) { return orc_blackbody_ph_spect(nu, *(const double *)ctx); }

/* ---- photonEmitCyclosynch, mc_cyclosynch.c:1176-1569 ---------------------------------------------------------- */
#define RNG_CS_COUNT 5u
#define RNG_CS_PHOTON 6u
#define RNG_CS_SINGLE 7u

static int in_emission_slab(const orc_config *c, const orc_hydro *h, int i, double rmin, double rmax, double theta_min, double theta_max)   /* :1215-1226 */
{
    double r_in, th_in, r_out, th_out;
    if (c->dimensions == ORC_THREE) {
        orc_hydroCoordinateToSpherical(c, &r_in, &th_in, fabs(h->r0[i]) - 0.5 * h->r0_size[i], fabs(h->r1[i]) - 0.5 * h->r1_size[i], fabs(h->r2[i]) - 0.5 * h->r2_size[i]);
        orc_hydroCoordinateToSpherical(c, &r_out, &th_out, fabs(h->r0[i]) + 0.5 * h->r0_size[i], fabs(h->r1[i]) + 0.5 * h->r1_size[i], fabs(h->r2[i]) + 0.5 * h->r2_size[i]);
    } else {
        orc_hydroCoordinateToSpherical(c, &r_in, &th_in, h->r0[i] - 0.5 * h->r0_size[i], h->r1[i] - 0.5 * h->r1_size[i], 0);
        orc_hydroCoordinateToSpherical(c, &r_out, &th_out, h->r0[i] + 0.5 * h->r0_size[i], h->r1[i] + 0.5 * h->r1_size[i], 0);
    }
    return (rmin <= r_out) && (r_in < rmax) && (th_out >= theta_min) && (th_in < theta_max);
}

/* one pool photon at the centre of cell i with the frequency nu_c, direction from three (two in 3-D) uniform draws (:1380-1440) */
static void emit_one(const orc_config *c, const orc_hydro *h, int i, double nu_c, double weight, int block_index, orc_rng *rng, orc_photon *out,
                     double *position_phi_out)
{
    const double fr_dum = nu_c;
    double position_phi = 0;
    if (c->dimensions != ORC_THREE) position_phi = orc_rng_uniform(rng) * 2 * M_PI;
    const double com_v_phi = orc_rng_uniform(rng) * 2 * M_PI;
    const double com_v_theta = orc_rng_uniform(rng) * M_PI;
    double p_comv[4], boost[3], l_boost[4], pos[3];
    p_comv[0] = ORC_PL_CONST * fr_dum / ORC_C_LIGHT;
    p_comv[1] = (ORC_PL_CONST * fr_dum / ORC_C_LIGHT) * sin(com_v_theta) * cos(com_v_phi);
    p_comv[2] = (ORC_PL_CONST * fr_dum / ORC_C_LIGHT) * sin(com_v_theta) * sin(com_v_phi);
    p_comv[3] = (ORC_PL_CONST * fr_dum / ORC_C_LIGHT) * cos(com_v_theta);
    if (c->dimensions == ORC_THREE) orc_hydroVectorToCartesian(c, boost, h->v0[i], h->v1[i], h->v2[i], h->r0[i], h->r1[i], h->r2[i]);
    else if (c->dimensions == ORC_TWO_POINT_FIVE) orc_hydroVectorToCartesian(c, boost, h->v0[i], h->v1[i], h->v2[i], h->r0[i], h->r1[i], position_phi);
    else orc_hydroVectorToCartesian(c, boost, h->v0[i], h->v1[i], 0, h->r0[i], h->r1[i], position_phi);
    boost[0] *= -1; boost[1] *= -1; boost[2] *= -1;
    orc_lorentzBoost(boost, p_comv, l_boost, 'p');
    memset(out, 0, sizeof *out);
    out->p0 = l_boost[0]; out->p1 = l_boost[1]; out->p2 = l_boost[2]; out->p3 = l_boost[3];
    out->comv_p0 = p_comv[0]; out->comv_p1 = p_comv[1]; out->comv_p2 = p_comv[2]; out->comv_p3 = p_comv[3];
    if (c->dimensions == ORC_THREE) orc_hydroCoordinateToMcratCoordinate(c, pos, h->r0[i], h->r1[i], h->r2[i]);
    else orc_hydroCoordinateToMcratCoordinate(c, pos, h->r0[i], h->r1[i], position_phi);
    out->r0 = pos[0]; out->r1 = pos[1]; out->r2 = pos[2];
    out->s0 = 1; out->s1 = 0; out->s2 = 0; out->s3 = 0;
    out->num_scatt = 0;
    out->weight = weight;
    out->nearest_block_index = block_index;
    out->type = ORC_CS_POOL_PHOTON;
    out->recalc_properties = 1;
    if (position_phi_out) *position_phi_out = position_phi;
}

/* Returns the number of photons emitted (>= 0) or a negative orc_list_add error.  *weight_out receives ph_weight_adjusted (pool mode).
 * rng: pool mode uses its seed / stream for the keyed streams described in the header; single mode draws from the stream
 * {rng's current iteration, scatt_ph_index, CS_SINGLE}. */
int orc_photonEmitCyclosynch(const orc_config *c, const orc_cs *cs, orc_photon_list *l, double r_inj, double ph_weight, int maximum_photons,
                             double theta_min, double theta_max, const orc_hydro *h, orc_rng *rng, int inject_single_switch, int scatt_ph_index,
                             double *weight_out, int *used_fallback_out)
{
    int ph_tot = 0, fallback_any = 0;
    orc_photon *ph_emit = NULL;
    if (inject_single_switch == 0) {
        const double max_photons = cs->rebin_e_perc * maximum_photons;                      /* :1178 */
        const double rmin = orc_calcCyclosynchRLimits(cs->scatt_frame_number, cs->inj_frame_number, h->fps, r_inj, 0);
        const double rmax = orc_calcCyclosynchRLimits(cs->scatt_frame_number, cs->inj_frame_number, h->fps, r_inj, 1);
        int block_cnt = 0, min_photons = 1;
        for (int i = 0; i < h->num_elements; i++) block_cnt += in_emission_slab(c, h, i, rmin, rmax, theta_min, theta_max);
        if (block_cnt == 0) min_photons = 0;                                                /* :1236-1239 */
        int *ph_dens = (int *)calloc((size_t)(block_cnt > 0 ? block_cnt : 1), sizeof(int));
        double ph_weight_adjusted = ph_weight;
        ph_tot = -1;
        for (uint64_t attempt = 0; (ph_tot > max_photons) || (ph_tot < min_photons); attempt++) {   /* :1244-1296 */
            int j = 0;
            ph_tot = 0;
            orc_rng_set_iteration(rng, attempt);
            for (int i = 0; i < h->num_elements; i++) {
                if (!in_emission_slab(c, h, i, rmin, rmax, theta_min, theta_max)) continue;
                const double b_field = orc_getMagneticFieldMagnitude(c, cs, h, i);
                const double nu_c = orc_calcCyclotronFreq(b_field);
                double temp = h->temp[i], ph_dens_calc = 0, error = 0;
                int fb = 0;
                (void)orc_qags(planck_tail, &temp, 10, nu_c, 0, 1e-2, 10000, &ph_dens_calc, &error, &fb);   /* :1276 */
                fallback_any |= fb;
                ph_dens_calc *= orc_hydroElementVolume(c, h, i) / ph_weight_adjusted;
                orc_rng_stream_begin(rng, (uint32_t)i, RNG_CS_COUNT);
                const long long k = orc_poisson(rng, ph_dens_calc);
                ph_dens[j] = (k > 2147483647LL) ? 2147483647 : (int)k;
                ph_tot += ph_dens[j];
                j++;
            }
            if (ph_tot > max_photons) ph_weight_adjusted *= 10;
            else if (ph_tot < min_photons) ph_weight_adjusted *= 0.5;
            if (attempt > 400) { free(ph_dens); return -4; }
        }
        ph_emit = (orc_photon *)calloc((size_t)(ph_tot > 0 ? ph_tot : 1), sizeof(orc_photon));
        const int net_ph = ph_tot;
        int k = 0;
        ph_tot = 0;
        orc_rng_set_iteration(rng, 0);
        for (int i = 0; i < h->num_elements && ph_tot < net_ph; i++) {                       /* :1340-1455 */
            if (!in_emission_slab(c, h, i, rmin, rmax, theta_min, theta_max)) continue;
            const double nu_c = orc_calcCyclotronFreq(orc_getMagneticFieldMagnitude(c, cs, h, i));
            for (int j = 0; j < ph_dens[k] && ph_tot < net_ph; j++) {
                orc_rng_stream_begin(rng, (uint32_t)ph_tot, RNG_CS_PHOTON);
                emit_one(c, h, i, nu_c, ph_weight_adjusted, 0, rng, &ph_emit[ph_tot], NULL);   /* nearest_block_index = 0, :1436 */
                ph_tot++;
            }
            k++;
        }
        free(ph_dens);
        if (weight_out) *weight_out = ph_weight_adjusted;
    } else {                                                                                /* :1467-1558 */
        ph_tot = 1;
        ph_emit = (orc_photon *)calloc(1, sizeof(orc_photon));
        orc_photon *tmp = &l->photons[scatt_ph_index];
        const int i = tmp->nearest_block_index;
        const double nu_c = orc_calcCyclotronFreq(orc_getMagneticFieldMagnitude(c, cs, h, i));
        double position_phi = 0, pos[3];
        orc_rng_stream_begin(rng, (uint32_t)scatt_ph_index, RNG_CS_SINGLE);
        emit_one(c, h, i, nu_c, tmp->weight, i, rng, &ph_emit[0], &position_phi);
        /* the photon that just scattered is moved to a random place in its cell (:1540-1556) */
        const double position_rand = orc_rng_uniform_pos(rng) * h->r0_size[i] - h->r0_size[i] / 2.0;
        const double position2_rand = orc_rng_uniform_pos(rng) * h->r1_size[i] - h->r1_size[i] / 2.0;
        if (c->dimensions == ORC_THREE) {
            const double position3_rand = orc_rng_uniform_pos(rng) * h->r2_size[i] - h->r2_size[i] / 2.0;
            orc_hydroCoordinateToMcratCoordinate(c, pos, h->r0[i] + position_rand, h->r1[i] + position2_rand, h->r2[i] + position3_rand);
        } else {
            orc_hydroCoordinateToMcratCoordinate(c, pos, h->r0[i] + position_rand, h->r1[i] + position2_rand, position_phi);
        }
        tmp->r0 = pos[0]; tmp->r1 = pos[1]; tmp->r2 = pos[2];
    }
    const int rc = (ph_tot > 0) ? orc_list_add(l, ph_emit, ph_tot) : 0;                     /* :1560 */
    free(ph_emit);
    if (used_fallback_out) *used_fallback_out = fallback_any;
    return rc ? rc : ph_tot;
}

/* ---- phAbsCyclosynch, mc_cyclosynch.c:1571-1623 ---------------------------------------------------------------- */
double orc_phAbsCyclosynch(const orc_config *c, const orc_cs *cs, orc_photon_list *l, const orc_hydro *h, int *num_abs_ph, int *scatt_cyclosynch_num_ph)
{
    int abs_ph_count = 0;
    double abs_count = 0;
    *scatt_cyclosynch_num_ph = 0;
    for (int i = 0; i < l->list_capacity; i++) {
        orc_photon *ph = &l->photons[i];
        if ((ph->weight != 0) && (ph->nearest_block_index != -1)) {
            const double b_field = orc_getMagneticFieldMagnitude(c, cs, h, ph->nearest_block_index);
            const double nu_c = orc_calcCyclotronFreq(b_field);
            if ((ph->comv_p0 * ORC_C_LIGHT / ORC_PL_CONST <= nu_c) || (ph->type == ORC_CS_POOL_PHOTON)) {
                abs_ph_count++;
                if ((ph->type == ORC_INJECTED_PHOTON) || (ph->type == ORC_UNABSORBED_CS_PHOTON)) {
                    abs_count += ph->weight;
                    ph->p0 = -1;                      /* overwritten by setNullPhoton right below, as in the reference */
                }
                (void)orc_list_set_null(l, i);
            } else if ((ph->type == ORC_COMPTONIZED_PHOTON) || (ph->type == ORC_UNABSORBED_CS_PHOTON)) {
                *scatt_cyclosynch_num_ph += 1;
            }
        }
    }
    *num_abs_ph = abs_ph_count;
    return abs_count;
}

/* ---- rebinCyclosynchCompPhotons, mc_cyclosynch.c:246-712 ----------------------------------------------------------
 * Every photon that is neither null, nor a pool photon, nor an injected one ('k', 'c', 'r') is replaced by one photon per
 * non-empty (energy, theta[, phi]) bin carrying the bin's weight and its weighted averages.  No random numbers.  The three
 * gsl_histogram2d objects serve only to find bin indexes: gsl_histogram2d_set_ranges_uniform puts edge i at
 * ((n - i) / n) lo + (i / n) hi and gsl_histogram2d_find returns the i with edge[i] <= x < edge[i + 1] (GSL's published
 * behaviour); both restated below. */
#define ORC_DEG_TO_RAD (M_PI / 180.0)          /* mcrat.h:80-81 */
#define ORC_RAD_TO_DEG (180.0 / M_PI)

typedef struct { double lo, hi; int n; } uniform_axis;

static double axis_edge(const uniform_axis *a, int i)
{
    const double f1 = ((double)(a->n - i)) / (double)a->n, f2 = ((double)i) / (double)a->n;
    return f1 * a->lo + f2 * a->hi;
}

static int axis_find(const uniform_axis *a, double x)            /* -1: outside [lo, hi) -- GSL_EDOM */
{
    if (!(x >= axis_edge(a, 0)) || !(x < axis_edge(a, a->n))) return -1;
    int lo = 0, hi = a->n;                                       /* edge[lo] <= x < edge[hi] */
    while (hi - lo > 1) {
        const int mid = (lo + hi) / 2;
        if (x >= axis_edge(a, mid)) lo = mid; else hi = mid;
    }
    return lo;
}

static void photon_position(const orc_config *c, const orc_photon *ph, double *r, double *theta, double *phi)   /* :246-270 */
{
    const double x = ph->r0, y = ph->r1, z = ph->r2;
    *r = sqrt(x * x + y * y + z * z);
    *phi = 0;
    if (*r < DBL_MIN) { *theta = 0.0; return; }
    *theta = acos(z / *r);
    if (c->dimensions == ORC_THREE) *phi = fmod(atan2(y, x) * ORC_RAD_TO_DEG + 360.0, 360.0);
}

typedef struct {
    double weighted_r, weighted_theta, weighted_phi_offset, weighted_stokes[4], weighted_scatt_count, total_weight;
    double weighted_phi_dir, weighted_theta_dir, weighted_energy, weighted_phi_pos;
} bin_stats;

/* Returns the number of empty bins (>= 0) as the reference does, -1 on its error paths (no photon to rebin, more bins than
 * max_photons, a photon outside the histograms), or an orc_list_add error. */
int orc_rebinCyclosynchCompPhotons(const orc_config *c, const orc_cs *cs, orc_photon_list *l, int *num_cyclosynch_ph_emit,
                                   int *scatt_cyclosynch_num_ph, int max_photons)
{
    const int three = c->dimensions == ORC_THREE;
    /* collect_photon_statistics :273-322 */
    double p0_min = DBL_MAX, p0_max = 0, theta_min = DBL_MAX, theta_max = 0, phi_min = DBL_MAX, phi_max = 0;
    int valid = 0, synch = 0;
    for (int i = 0; i < l->list_capacity; i++) {
        const orc_photon *ph = &l->photons[i];
        if ((ph->type != ORC_NULL_PHOTON) && (ph->type != ORC_CS_POOL_PHOTON) && (ph->type != ORC_INJECTED_PHOTON)) {
            if (ph->p0 > 0) { p0_min = fmin(p0_min, ph->p0); p0_max = fmax(p0_max, ph->p0); valid++; }
            double r, theta, phi;
            photon_position(c, ph, &r, &theta, &phi);
            theta_min = fmin(theta_min, theta); theta_max = fmax(theta_max, theta);
            if (three) { phi_min = fmin(phi_min, phi); phi_max = fmax(phi_max, phi); }
        }
        if (ph->type == ORC_CS_POOL_PHOTON) synch++;
    }
    if (valid == 0) return -1;
    const double log_p0_min = (p0_min > 0 && p0_max > 0) ? log10(p0_min) : 0.0, log_p0_max = (p0_min > 0 && p0_max > 0) ? log10(p0_max) : 1.0;
    /* calculate_binning_params :324-347 */
    const int num_bins = (int)(cs->rebin_e_perc * max_photons);
    const int num_bins_theta = (int)ceil((theta_max - theta_min) / (cs->rebin_ang * ORC_DEG_TO_RAD));
    const int num_bins_phi = three ? (int)ceil((phi_max - phi_min) / cs->rebin_ang_phi) : 1;
    const int total_bins = num_bins_theta * num_bins * (three ? num_bins_phi : 1);
    if (total_bins > max_photons) return -1;                                          /* :649-654 */
    if (num_bins <= 0 || num_bins_theta <= 0 || num_bins_phi <= 0) return -1;         /* allocate_histograms :351-354 */
    /* allocate_histograms :360-391: the ranges, widened by 1e-6 of their width so that the maxima fall inside */
    const double e_eps = (log_p0_max - log_p0_min) * 1e-6, t_eps = (theta_max - theta_min) * 1e-6, p_eps = (phi_max - phi_min) * 1e-6;
    const uniform_axis ax_e = {log_p0_min, log_p0_max + e_eps, num_bins}, ax_t = {theta_min, theta_max + t_eps, num_bins_theta},
                       ax_p = {phi_min, phi_max + p_eps, num_bins_phi};
    bin_stats *stats = (bin_stats *)calloc((size_t)total_bins, sizeof(bin_stats));
    /* accumulate_bin_statistics :448-501 */
    for (int i = 0; i < l->list_capacity; i++) {
        const orc_photon *ph = &l->photons[i];
        if ((ph->type == ORC_NULL_PHOTON) || (ph->type == ORC_CS_POOL_PHOTON) || (ph->type == ORC_INJECTED_PHOTON)) continue;
        double r, theta, phi;
        photon_position(c, ph, &r, &theta, &phi);
        int idx_x = axis_find(&ax_e, log10(ph->p0)), idx_y = axis_find(&ax_t, theta), idx_z = 0;
        if (idx_x < 0 || idx_y < 0) {            /* gsl_histogram2d_find leaves both indexes untouched (0) on a domain error */
            idx_x = 0; idx_y = 0;
        }
        if (three) {
            /* the second and third find calls overwrite idx_x / idx_y / idx_z pairwise (:459-460), each all-or-nothing */
            const int ex = axis_find(&ax_e, log10(ph->p0)), pz = axis_find(&ax_p, phi);
            if (ex >= 0 && pz >= 0) { idx_x = ex; idx_z = pz; }
            const int ty = axis_find(&ax_t, theta);
            if (ty >= 0 && pz >= 0) { idx_y = ty; idx_z = pz; }
        }
        int bin_idx;                                                                  /* calculate_bin_index :432-446 */
        if (idx_x < 0 || idx_x >= num_bins || idx_y < 0 || idx_y >= num_bins_theta || (three && (idx_z < 0 || idx_z >= num_bins_phi))) bin_idx = -1;
        else bin_idx = three ? idx_z * num_bins * num_bins_theta + idx_x * num_bins_theta + idx_y : idx_x * num_bins_theta + idx_y;
        if (bin_idx < 0 || bin_idx >= total_bins) { free(stats); return -1; }         /* the reference exit(1)s */
        bin_stats *s = &stats[bin_idx];
        s->weighted_r += r * ph->weight;
        s->weighted_theta += theta * ph->weight;
        s->weighted_phi_offset += (atan2(ph->p2, ph->p1) - atan2(ph->r1, ph->r0)) * ORC_RAD_TO_DEG * ph->weight;
        s->weighted_stokes[0] += ph->s0 * ph->weight; s->weighted_stokes[1] += ph->s1 * ph->weight;
        s->weighted_stokes[2] += ph->s2 * ph->weight; s->weighted_stokes[3] += ph->s3 * ph->weight;
        s->weighted_scatt_count += ph->num_scatt * ph->weight;
        s->total_weight += ph->weight;
        const double phi_dir = fmod(atan2(ph->p2, ph->p1) * ORC_RAD_TO_DEG + 360.0, 360.0);
        const double theta_dir = acos(ph->p3 / ph->p0) * ORC_RAD_TO_DEG;
        s->weighted_phi_dir += phi_dir * ph->weight;
        s->weighted_theta_dir += theta_dir * ph->weight;
        s->weighted_energy += ph->p0 * ph->weight;
        if (three) s->weighted_phi_pos += phi * ph->weight;
    }
    /* create_rebinned_photons :504-607 */
    orc_photon *rebin_ph = (orc_photon *)calloc((size_t)total_bins, sizeof(orc_photon));
    int num_null_rebin_ph = 0;
    for (int i = 0; i < total_bins; i++) {
        const bin_stats *s = &stats[i];
        orc_photon *np = &rebin_ph[i];
        if (s->total_weight <= 0) {
            np->type = ORC_NULL_PHOTON; np->weight = 0; np->nearest_block_index = -1; np->recalc_properties = 0;
            num_null_rebin_ph++;
            continue;
        }
        np->type = ORC_COMPTONIZED_PHOTON;
        np->weight = s->total_weight;
        const double avg_energy = s->weighted_energy / s->total_weight, avg_phi_dir = s->weighted_phi_dir / s->total_weight;
        const double avg_theta_dir = s->weighted_theta_dir / s->total_weight, avg_r = s->weighted_r / s->total_weight;
        const double avg_theta_pos = s->weighted_theta / s->total_weight;
        np->p0 = avg_energy;
        np->p1 = avg_energy * sin(avg_theta_dir * ORC_DEG_TO_RAD) * cos(avg_phi_dir * ORC_DEG_TO_RAD);
        np->p2 = avg_energy * sin(avg_theta_dir * ORC_DEG_TO_RAD) * sin(avg_phi_dir * ORC_DEG_TO_RAD);
        np->p3 = avg_energy * cos(avg_theta_dir * ORC_DEG_TO_RAD);
        double pos_phi;
        if (three) pos_phi = (s->weighted_phi_pos / s->total_weight) * ORC_DEG_TO_RAD;
        else pos_phi = (avg_phi_dir - s->weighted_phi_offset / s->total_weight) * ORC_DEG_TO_RAD;
        np->r0 = avg_r * sin(avg_theta_pos) * cos(pos_phi);
        np->r1 = avg_r * sin(avg_theta_pos) * sin(pos_phi);
        np->r2 = avg_r * cos(avg_theta_pos);
        np->s0 = s->weighted_stokes[0] / s->total_weight; np->s1 = s->weighted_stokes[1] / s->total_weight;
        np->s2 = s->weighted_stokes[2] / s->total_weight; np->s3 = s->weighted_stokes[3] / s->total_weight;
        np->num_scatt = (int)(s->weighted_scatt_count / s->total_weight + 0.5);
        np->nearest_block_index = 0;
        np->recalc_properties = 1;
    }
    for (int i = 0; i < l->list_capacity; i++)
        if (l->photons[i].type == ORC_UNABSORBED_CS_PHOTON || l->photons[i].type == ORC_COMPTONIZED_PHOTON) (void)orc_list_set_null(l, i);
    const int rc = orc_list_add(l, rebin_ph, total_bins);
    free(rebin_ph); free(stats);
    if (rc) return rc;
    if (l->num_photons < total_bins) return -1;
    *scatt_cyclosynch_num_ph = total_bins - num_null_rebin_ph;                        /* :689-690 */
    *num_cyclosynch_ph_emit = total_bins + synch - num_null_rebin_ph;
    return num_null_rebin_ph;
}

/* ---- the scatter-frame body of main() with CYCLOSYNCHROTRON_SWITCH ON, mcrat.c:706-878 ---------------------------------
 * (between getHydroData and saveCheckpoint; the radius limits of :708-719 only widen getHydroData's slab and are the caller's).
 * emit_pool is the condition `(scatt_frame != scatt_framestart) || (restrt == CONTINUE)` of :707,:727,:855.
 * Reference behaviour kept: the photon photonEvent reports (:781,:786) is turned from a pool photon into a comptonised one and
 * replaced even when every candidate of the pass was Klein-Nishina-rejected, because *scattered_ph_index then names the last
 * candidate tried (mclib.c:1128,1337-1355). */
/* saveCheckpoint's in-place conversion with the switch on (mcrat_io.c:896-900, :951-955, :991-995): every comptonised photon with weight
 * becomes an unabsorbed one before its record is written -- and stays so for printPhotons (mcrat.c:907) and the next frame.  Returns the
 * number converted. */
int orc_saveCheckpoint_convert(orc_photon_list *l)
{
    int n = 0;
    for (int i = 0; i < l->list_capacity; i++)
        if (l->photons[i].type == ORC_COMPTONIZED_PHOTON && l->photons[i].weight != 0) { l->photons[i].type = ORC_UNABSORBED_CS_PHOTON; n++; }
    return n;
}

void orc_scatter_frame_cs(const orc_config *c, orc_cs *cs, orc_photon_list *l, const orc_hydro *h, orc_rng *rng, double *time_now,
                          double remaining_time, double r_inj, double ph_weight_suggest, int max_photons, double theta_jmin_thread,
                          double theta_jmax_thread, int emit_pool, long long max_iterations, orc_stats *st, orc_cs_counts *cnt)
{
    /* scatt_cyclosynch_num_ph is main()'s own counter: it lives from one scatter frame of an injection to the next (set at :873 by
     * phAbsCyclosynch or the rebinning, reset at :921 only) -- the caller hands the previous frame's value in through cnt */
    const int carried = cnt->scatt_cyclosynch_num_ph;
    memset(cnt, 0, sizeof *cnt);
    cnt->scatt_cyclosynch_num_ph = carried;
    if (emit_pool) {                                                                      /* :727-744 */
        const int n = orc_photonEmitCyclosynch(c, cs, l, r_inj, ph_weight_suggest, max_photons, theta_jmin_thread, theta_jmax_thread, h, rng, 0, 0,
                                               &cnt->pool_weight, NULL);
        cnt->num_cyclosynch_ph_emit = n > 0 ? n : 0;
        if (n < 0) cnt->error = n;
    }
    int find_nearest_grid_switch = 1;                                                     /* :756 */
    long long it = 0;
    double time_step = 0;
    while (remaining_time > 0 && (max_iterations <= 0 || it < max_iterations) && !cnt->error) {   /* :761-851 */
        orc_rng_set_iteration(rng, (uint64_t)it);
        st->num_photons_find_new_element += orc_findContainingHydroCell_keyed(c, l, h, find_nearest_grid_switch, st, rng);
        orc_calcMeanFreePath(c, l, h, rng);
        find_nearest_grid_switch = 0;
        if (l->photons[l->sorted_indexes[0]].time_to_scatter < remaining_time) {
            time_step = orc_photonEvent(c, l, remaining_time, h, &st->last_scattered_index, &st->frame_scatt_cnt, rng, st);
            *time_now += time_step;
            remaining_time -= time_step;
            orc_photon *scattered = &l->photons[st->last_scattered_index];              /* :786-795 */
            if (scattered->type == ORC_CS_POOL_PHOTON) {
                cnt->n_comptonized += scattered->weight;
                scattered->type = ORC_COMPTONIZED_PHOTON;
                const int n = orc_photonEmitCyclosynch(c, cs, l, r_inj, ph_weight_suggest, max_photons, theta_jmin_thread, theta_jmax_thread, h, rng, 1,
                                                       st->last_scattered_index, NULL, NULL);
                if (n < 0) cnt->error = n; else cnt->num_cyclosynch_ph_emit += n;
                cnt->scatt_cyclosynch_num_ph++;
            }
            if ((st->frame_scatt_cnt % 1000 == 0) && (st->frame_scatt_cnt != 0) && cnt->scatt_cyclosynch_num_ph > max_photons) {   /* :797-808 */
                const int rc = orc_rebinCyclosynchCompPhotons(c, cs, l, &cnt->num_cyclosynch_ph_emit, &cnt->scatt_cyclosynch_num_ph, max_photons);
                if (rc >= 0) cnt->rebins++;
            }
        } else {
            *time_now += remaining_time;
            orc_updatePhotonPosition(l, remaining_time);
            time_step = remaining_time;
            remaining_time = 0;
        }
        it++;
        st->photon_steps += l->list_capacity;
    }
    if (emit_pool && !cnt->error) {                                                       /* :853-878 */
        if (cnt->scatt_cyclosynch_num_ph > max_photons) {
            const int rc = orc_rebinCyclosynchCompPhotons(c, cs, l, &cnt->num_cyclosynch_ph_emit, &cnt->scatt_cyclosynch_num_ph, max_photons);
            if (rc >= 0) cnt->rebins++;
        }
        if (cnt->num_cyclosynch_ph_emit > 0)
            cnt->n_comptonized -= orc_phAbsCyclosynch(c, cs, l, h, &cnt->frame_abs_cnt, &cnt->scatt_cyclosynch_num_ph);
    }
    st->iterations += it;
    st->last_time_step = time_step;
    st->remaining_time = remaining_time;
    st->time_now = *time_now;
}
