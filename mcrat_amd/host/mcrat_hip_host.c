/* mcrat_hip_host.c -- see mcrat_hip_host.h.  Plain C99, no photon physics on the CPU. */
#include "mcrat_hip_host.h"

#include <ctype.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* next line that carries values: skips blank lines and "[Block]" headers, cuts the trailing "# comment" */
static int next_value_line(FILE *f, char *buf, size_t n)
{
    while (fgets(buf, (int)n, f)) {
        char *hash = strchr(buf, '#');
        if (hash) *hash = '\0';
        char *p = buf;
        while (*p && isspace((unsigned char)*p)) p++;
        if (*p == '\0' || *p == '[') continue;
        memmove(buf, p, strlen(p) + 1);
        return 1;
    }
    return 0;
}

static int parse_doubles(char *line, double *out, int n)
{
    char *save = NULL;
    int k = 0;
    for (char *tok = strtok_r(line, " \t\r\n", &save); tok && k < n; tok = strtok_r(NULL, " \t\r\n", &save)) {
        char *end;
        out[k] = strtod(tok, &end);
        if (end == tok) return k;
        k++;
    }
    return k;
}

int mcrat_host_read_mcpar(const char *path, mcrat_host_mcpar *out)
{
    char buf[2000];
    double v[64];
    if (!path || !out) return -2;
    memset(out, 0, sizeof *out);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int rc = -2;
    do {
        /* [Hydro/MHD Simulation Block]  mcrat_io.c:1151-1166 */
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, 1) != 1) break;
        out->fps = v[0];
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, 1) != 1) break;
        out->last_frame = (int)v[0];
        double *dom[3] = {out->r0_domain, out->r1_domain, out->r2_domain};
        int ok = 1;
        for (int a = 0; a < 3 && ok; a++) {
            ok = next_value_line(f, buf, sizeof buf) && parse_doubles(buf, v, 2) == 2;
            if (ok) { dom[a][0] = v[0]; dom[a][1] = v[1]; }
        }
        if (!ok) break;
        /* [MCRaT Injection Angles Block]  :1168-1216 */
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, 1) != 1) break;
        out->theta_jmin = v[0];
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, 1) != 1) break;
        out->theta_j = v[0];
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, 1) != 1) break;
        out->n_theta_j = (int)v[0];
        const int nb = out->n_theta_j;
        if (nb < 1 || nb > 64) break;
        out->frm0 = (int *)malloc(sizeof(int) * nb);
        out->frm2 = (int *)malloc(sizeof(int) * nb);
        out->inj_radius = (double *)malloc(sizeof(double) * nb);
        if (!out->frm0 || !out->frm2 || !out->inj_radius) break;
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, nb) != nb) break;
        for (int i = 0; i < nb; i++) out->frm0[i] = (int)v[i];
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, nb) != nb) break;
        for (int i = 0; i < nb; i++) out->frm2[i] = (int)v[i] + out->frm0[i];     /* number of frames -> last frame, :1201 */
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, nb) != nb) break;
        for (int i = 0; i < nb; i++) out->inj_radius[i] = (double)(float)v[i];    /* strtof in the reference, :1211 */
        /* [MCRaT Photon Block]  :1218-1227 */
        if (!next_value_line(f, buf, sizeof buf)) break;
        out->spect = buf[0];
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, 1) != 1) break;
        out->min_photons = (int)v[0];
        if (!next_value_line(f, buf, sizeof buf) || parse_doubles(buf, v, 1) != 1) break;
        out->max_photons = (int)v[0];
        /* [Initialization/Continuation Block]  :1229-1233 */
        if (!next_value_line(f, buf, sizeof buf)) break;
        out->restart = buf[0];
        if ((out->spect != 'b' && out->spect != 'w') || (out->restart != 'i' && out->restart != 'c')) break;
        rc = 0;
    } while (0);
    fclose(f);
    if (rc != 0) mcrat_host_free_mcpar(out);
    return rc;
}

void mcrat_host_free_mcpar(mcrat_host_mcpar *p)
{
    if (!p) return;
    free(p->frm0); free(p->frm2); free(p->inj_radius);
    p->frm0 = p->frm2 = NULL;
    p->inj_radius = NULL;
}

static int is_dash_line(const char *line)
{
    int n = 0;
    for (; *line && *line != '\r' && *line != '\n'; ++line) {
        if (*line != '-') return 0;
        n++;
    }
    return n > 0;
}

int mcrat_host_read_hot_cross_section(const char *path, double *table, int n_ph_e, int n_t)
{
    if (!path || !table || n_ph_e < 1 || n_t < 1) return -2;
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    const size_t count = (size_t)(n_ph_e + 1) * (size_t)(n_t + 1);
    unsigned char *seen = (unsigned char *)calloc(count, 1);
    if (!seen) { fclose(f); return -2; }
    char line[1024];
    int rc = -2, in_data = 0;
    size_t filled = 0;
    while (fgets(line, sizeof line, f)) {
        if (!in_data) { in_data = is_dash_line(line); continue; }     /* hot_x_section.c:229-236 */
        int i, j;
        double e, t, v;
        if (sscanf(line, "%d %d %lf %lf %lf", &i, &j, &e, &t, &v) != 5) continue;
        if (i < 0 || i > n_ph_e || j < 0 || j > n_t) { filled = count + 1; break; }   /* :242-251 */
        const size_t k = (size_t)i * (size_t)(n_t + 1) + (size_t)j;
        if (!seen[k]) { seen[k] = 1; filled++; }
        table[k] = v;
    }
    if (in_data && filled == count) rc = 0;
    free(seen);
    fclose(f);
    return rc;
}

int mcrat_host_write_hot_cross_section(const char *path, const double *table, int n_ph_e, int n_t, double log_ph_e_min,
                                       double log_ph_e_max, double log_t_min, double log_t_max)
{
    if (!path || !table || n_ph_e < 1 || n_t < 1) return -1;
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    const double dt = (log_t_max - log_t_min) / n_t, dph_e = (log_ph_e_max - log_ph_e_min) / n_ph_e;
    fprintf(fp, "The comoving photon energy and the temperatures are normalized by the electron rest mass\n");
    fprintf(fp, "The calculated hot cross sections are normalized by the thompson cross section.\n");
    fprintf(fp, "Photon index\tTheta Index\tlog10(Comoving Photon Energy)\tlog10(Theta)\tlog10(Hot Cross Section)\n");
    fprintf(fp, "------------------------------------------------\n");
    for (int i = 0; i <= n_ph_e; i++)
        for (int j = 0; j <= n_t; j++)
            fprintf(fp, "%d\t%d\t%g\t%g\t%15.10g\n", i, j, log_ph_e_min + i * dph_e, log_t_min + j * dt, table[(size_t)i * (n_t + 1) + j]);
    return fclose(fp) == 0 ? 0 : -1;
}

/* ---- PLUTO .dbl frames (mclib_pluto.c:803-1128) ---------------------------------------------------------- */
void mcrat_host_pluto_name(char *out, size_t n, const char *prefix, int frame)
{
    snprintf(out, n, "%s%04d.dbl", prefix, frame);      /* "%s%.3d%d%s" with 000 / "%.2d%d" with 00 / ... (:829-844) */
}

void mcrat_host_free_pluto(mcrat_host_pluto *p)
{
    if (!p) return;
    for (int i = 0; p->var_names && i < p->num_vars; i++) free(p->var_names[i]);
    free(p->var_names);
    free(p->axes);
    free(p->data);
    memset(p, 0, sizeof *p);
}

/* "# X1: [ 1.0,  100.0], 1024 point(s), 0 ghosts": the third comma-separated field starts with the count (:880-895) */
static int header_count(const char *line)
{
    const char *c1 = strchr(line, ','), *c2 = c1 ? strchr(c1 + 1, ',') : NULL;
    if (!c2) return -1;
    char *end;
    long v = strtol(c2 + 1, &end, 10);
    return (end == c2 + 1 || v <= 0 || v > 0x7fffffffL) ? -1 : (int)v;
}

int mcrat_host_read_pluto(const char *grid_out, const char *dbl_out, const char *dbl_file, int three_dimensional,
                          double l_scale, double d_scale, double p_scale, mcrat_host_pluto *out)
{
    if (!grid_out || !dbl_out || !dbl_file || !out) return -2;
    memset(out, 0, sizeof *out);
    char line[2000];
    int n[3] = {0, 0, 1}, naxes = three_dimensional ? 3 : 2, rc = -2;

    /* grid.out: the header */
    FILE *f = fopen(grid_out, "r");
    if (!f) return -1;
    long data_pos = 0;
    for (;;) {
        data_pos = ftell(f);
        if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
        if (line[0] != '#') break;
        for (int a = 0; a < naxes; a++) {
            char tag[8];
            snprintf(tag, sizeof tag, "X%d:", a + 1);
            if (strstr(line, tag)) n[a] = header_count(line);
        }
    }
    for (int a = 0; a < naxes; a++)
        if (n[a] <= 0) { fclose(f); return -2; }
    const size_t total_axes = 2 * ((size_t)n[0] + (size_t)n[1] + (size_t)n[2]);
    out->axes = (double *)calloc(total_axes, sizeof(double));
    if (!out->axes) { fclose(f); return -2; }
    double *centre[3], *width[3], *p = out->axes;
    for (int a = 0; a < 3; a++) { centre[a] = p; p += n[a]; width[a] = p; p += n[a]; }
    /* per axis: a count line, then rows "index left right" (:935-971) */
    fseek(f, data_pos, SEEK_SET);
    for (int a = 0; a < naxes; a++) {
        int count = 0, idx;
        double left, right;
        if (fscanf(f, "%d", &count) != 1 || count != n[a]) goto done_grid;
        for (int i = 0; i < n[a]; i++) {
            if (fscanf(f, "%d %lf %lf", &idx, &left, &right) != 3) goto done_grid;
            centre[a][i] = 0.5 * (left + right);
            width[a][i] = (right - left);
        }
    }
    rc = 0;
done_grid:
    fclose(f);
    if (rc) { mcrat_host_free_pluto(out); return -2; }
    rc = -2;

    /* dbl.out: "0 0.000000e+00 1.000000e-04 0 single_file little rho vx1 vx2 prs" (:1003-1050) */
    f = fopen(dbl_out, "r");
    if (!f) { mcrat_host_free_pluto(out); return -1; }
    if (!fgets(line, sizeof line, f)) { fclose(f); mcrat_host_free_pluto(out); return -2; }
    fclose(f);
    {
        char *save = NULL, *tok = strtok_r(line, " \t\r\n", &save);
        int field = 0, cap = 16;
        out->var_names = (char **)calloc((size_t)cap, sizeof(char *));
        for (; tok && out->var_names; tok = strtok_r(NULL, " \t\r\n", &save), field++) {
            if (field < 6) continue;                    /* index, time, dt, step, file layout, endianness */
            if (out->num_vars == cap) {
                cap *= 2;
                char **grown = (char **)realloc(out->var_names, (size_t)cap * sizeof(char *));
                if (!grown) break;
                out->var_names = grown;
            }
            out->var_names[out->num_vars] = strdup(tok);
            out->num_vars++;
        }
        if (!out->var_names || out->num_vars == 0) { mcrat_host_free_pluto(out); return -2; }
    }

    /* the frame: num_vars blocks of grid_size doubles (:1116-1128) */
    const size_t grid_size = (size_t)n[0] * (size_t)n[1] * (size_t)n[2];
    out->data = (double *)malloc(sizeof(double) * grid_size * (size_t)out->num_vars);
    if (!out->data) { mcrat_host_free_pluto(out); return -2; }
    f = fopen(dbl_file, "rb");
    if (!f) { mcrat_host_free_pluto(out); return -1; }
    const size_t got = fread(out->data, sizeof(double), grid_size * (size_t)out->num_vars, f);
    fclose(f);
    if (got != grid_size * (size_t)out->num_vars) { mcrat_host_free_pluto(out); return -2; }

    mcrat_hip_pluto_grid *g = &out->grid;
    g->nx = n[0]; g->ny = n[1]; g->nz = n[2];
    g->x1 = centre[0]; g->dx1 = width[0]; g->x2 = centre[1]; g->dx2 = width[1];
    g->x3 = three_dimensional ? centre[2] : NULL; g->dx3 = three_dimensional ? width[2] : NULL;
    g->l_scale = l_scale; g->d_scale = d_scale; g->p_scale = p_scale;
    for (int v = 0; v < out->num_vars; v++) {           /* :1147-1212 */
        const double *block = out->data + (size_t)v * grid_size;
        const char *name = out->var_names[v];
        if (strcmp(name, "rho") == 0) g->rho = block;
        else if (strcmp(name, "vx1") == 0) g->vx1 = block;
        else if (strcmp(name, "vx2") == 0) g->vx2 = block;
        else if (strcmp(name, "vx3") == 0) g->vx3 = block;
        else if (strcmp(name, "prs") == 0) g->prs = block;
    }
    if (!g->rho || !g->vx1 || !g->vx2 || !g->prs) { mcrat_host_free_pluto(out); return -2; }
    return 0;
}

/* ---- checkpoints (mcrat_io.c:838-1134) -------------------------------------------------------------------- */
static int copy_file(const char *from, const char *to)        /* "exec cp file file_old", mcrat_io.c:849 */
{
    FILE *in = fopen(from, "rb");
    if (!in) return -1;
    FILE *out = fopen(to, "wb");
    if (!out) { fclose(in); return -1; }
    char buf[1 << 16];
    size_t n;
    int rc = 0;
    while ((n = fread(buf, 1, sizeof buf, in)) > 0)
        if (fwrite(buf, 1, n, out) != n) { rc = -1; break; }
    fclose(in);
    if (fclose(out) != 0) rc = -1;
    return rc;
}

int mcrat_host_save_checkpoint(const char *dir, int frame, int frame2, int scatt_frame, double time_now, mcrat_hip_ctx *ctx,
                               mcrat_hip_photon_list *list, int list_capacity, int last_frame, int angle_rank, int angle_size,
                               int cyclosynchrotron_switch)
{
    char file[2000], old[2100];
    if (!dir || list_capacity < 0 || (!ctx && list_capacity > 0 && (!list || !list->photons))) return 1;
    snprintf(file, sizeof file, "%s%s%d%s", dir, "mc_chkpt_", angle_rank, ".dat");
    snprintf(old, sizeof old, "%s_old", file);
    const int continuing = (scatt_frame != last_frame) || (scatt_frame == frame);       /* the three cases of :846,:898,:947 */
    /* "exec cp file file_old" and then fopen(file, "wb"), which truncates the original: the same files result from a rename, without
     * reading and writing the old checkpoint once more per frame (copy_file is the fallback across file systems) */
    if (scatt_frame == frame) remove(file);
    else if (rename(file, old) != 0) (void)copy_file(file, old);
    FILE *f = fopen(file, "wb");
    if (!f) { printf("Cannot open %s to save checkpoint\n", file); return 1; }
    char restart = continuing ? 'c' : 'i';                     /* CONTINUE / INITALIZE, mcrat.h */
    int ok = fwrite(&angle_size, sizeof(int), 1, f) == 1 && fwrite(&restart, sizeof(char), 1, f) == 1 &&
             fwrite(&frame, sizeof(int), 1, f) == 1 && fwrite(&frame2, sizeof(int), 1, f) == 1;
    if (ok && continuing) {
        int ph_num = list_capacity;
        ok = fwrite(&scatt_frame, sizeof(int), 1, f) == 1 && fwrite(&time_now, sizeof(double), 1, f) == 1 &&
             fwrite(&ph_num, sizeof(int), 1, f) == 1;
    }
    if (ok && list_capacity > 0 && cyclosynchrotron_switch) {  /* :896-900, :951-955, :991-995: 'k' with weight != 0 -> 'c', in place */
        if (ctx) {
            ok = mcrat_hip_convert_comptonized(ctx, NULL) == 0;
        } else {
            for (int i = 0; i < list_capacity; i++)
                if (list->photons[i].type == 'k' && list->photons[i].weight != 0) list->photons[i].type = 'c';
        }
    }
    if (ok && list_capacity > 0) {
        if (ctx) {
            const int piece = 1 << 20;                          /* 185 MB of records at a time */
            mcrat_hip_photon *buf = (mcrat_hip_photon *)malloc(sizeof(mcrat_hip_photon) * (size_t)(list_capacity < piece ? list_capacity : piece));
            ok = buf != NULL;
            for (int first = 0; ok && first < list_capacity; first += piece) {
                const int n = list_capacity - first < piece ? list_capacity - first : piece;
                ok = mcrat_hip_get_photons_range(ctx, first, n, buf) == 0 && fwrite(buf, sizeof(mcrat_hip_photon), (size_t)n, f) == (size_t)n;
            }
            free(buf);
        } else {
            ok = fwrite(list->photons, sizeof(mcrat_hip_photon), (size_t)list_capacity, f) == (size_t)list_capacity;
        }
    }
    if (fclose(f) != 0) ok = 0;
    return ok ? 0 : 1;
}

int mcrat_host_read_checkpoint(const char *dir, mcrat_hip_photon_list *list, int *frame2, int *framestart, int *scatt_framestart,
                               char *restart, double *time, int angle_rank, int *angle_size)
{
    char file[2000];
    if (!dir || !list || !frame2 || !framestart || !scatt_framestart || !restart || !time || !angle_size) return -2;
    snprintf(file, sizeof file, "%s%s%d%s", dir, "mc_chkpt_", angle_rank, ".dat");
    FILE *f = fopen(file, "rb");
    if (!f) {                                                   /* :1127-1131 */
        *scatt_framestart = *framestart;
        *restart = 'i';
        return 0;
    }
    int rc = -2, ph_num = 0;
    if (fread(angle_size, sizeof(int), 1, f) != 1 || fread(restart, sizeof(char), 1, f) != 1 || fread(framestart, sizeof(int), 1, f) != 1 ||
        fread(frame2, sizeof(int), 1, f) != 1)
        goto done;
    if (*restart == 'c') {
        if (fread(scatt_framestart, sizeof(int), 1, f) != 1) goto done;
        *scatt_framestart += 1;                                 /* start at the frame after the interrupted one, :1052 */
        if (fread(time, sizeof(double), 1, f) != 1 || fread(&ph_num, sizeof(int), 1, f) != 1 || ph_num < 0) goto done;
        mcrat_hip_photon *ph = (mcrat_hip_photon *)calloc((size_t)(ph_num > 0 ? ph_num : 1), sizeof(mcrat_hip_photon));
        if (!ph) goto done;
        if (fread(ph, sizeof(mcrat_hip_photon), (size_t)ph_num, f) != (size_t)ph_num) { free(ph); goto done; }
        int nulls = 0;
        for (int i = 0; i < ph_num; i++) {                      /* the members the reference does not carry over (:1064-1083) */
            ph[i].recalc_properties = 1;
            ph[i].time_to_scatter = 0;
            ph[i].total_optical_depth = 0;
            nulls += ph[i].type == 'N';
        }
        list->photons = ph;
        list->sorted_indexes = NULL;
        list->list_capacity = ph_num;
        list->num_null_photons = nulls;
        list->num_photons = ph_num - nulls;
    } else {
        *framestart += 1;                                       /* :1117 */
        *scatt_framestart = *framestart;
    }
    rc = 0;
done:
    fclose(f);
    return rc;
}

/* the log lines of mcrat.c:883-890 and mclib.c:583, same wording */
static void log_frame(FILE *fPtr, const mcrat_hip_frame_stats *st, double time_now, int max_scatt, int min_scatt, double avg_scatt, double avg_r)
{
    if (!fPtr) return;
    fprintf(fPtr, "The number of scatterings in this frame is: %d\n", (int)st->frame_scatt_cnt);
    fprintf(fPtr, "The last time step was: %e.\nThe time now is: %e\n", st->last_time_step, time_now);
    fprintf(fPtr, "MCRaT had to refind the position of photons %d times in this frame.\n", (int)st->num_photons_find_new_element);
    fprintf(fPtr, "The maximum number of scatterings for a photon is: %d\nThe minimum number of scatterings for a photon is: %d\n",
            max_scatt, min_scatt);
    fprintf(fPtr, "The average number of scatterings thus far is: %lf\nThe average position of photons is %e\n", avg_scatt, avg_r);
    for (long long k = 0; k < st->not_found; k++)
        fprintf(fPtr, "Photon Hydro grid index not found, making sure it doesnt scatter.\n");
    fflush(fPtr);
}

int mcrat_host_scatter_frame(mcrat_hip_ctx *ctx, mcrat_hip_photon_list *list, const mcrat_hip_hydro *hydro,
                             double *time_now, int scatt_frame, int increment_scatt_frame, double fps,
                             uint64_t seed, FILE *fPtr, mcrat_hip_frame_stats *stats)
{
    mcrat_hip_frame_stats st;
    int rc, max_scatt = 0, min_scatt = 0;
    double avg_scatt = 0, avg_r = 0;
    if (!ctx || !list || !hydro || !time_now || !(fps > 0)) return MCRAT_HIP_EINVAL;

    if ((rc = mcrat_hip_set_hydro(ctx, hydro)) != 0) return rc;          /* after getHydroData, mcrat.c:721 */
    if ((rc = mcrat_hip_set_photons(ctx, list)) != 0) return rc;

    /* mcrat.c:758: time left in this hydro frame */
    const double remaining_time = ((scatt_frame + increment_scatt_frame) / fps) - *time_now;
    if ((rc = mcrat_hip_propagate_frame(ctx, time_now, remaining_time, seed, &st)) != 0) return rc;   /* mcrat.c:761-851 */

    if ((rc = mcrat_hip_scatt_stats(ctx, &max_scatt, &min_scatt, &avg_scatt, &avg_r)) != 0) return rc;  /* mcrat.c:881 */
    if ((rc = mcrat_hip_get_photons(ctx, list)) != 0) return rc;          /* before saveCheckpoint/printPhotons, mcrat.c:902-907 */

    log_frame(fPtr, &st, *time_now, max_scatt, min_scatt, avg_scatt, avg_r);
    if (stats) *stats = st;
    return MCRAT_HIP_OK;
}

int mcrat_host_scatter_frame_resident(mcrat_hip_ctx *ctx, mcrat_host_get_hydro_fn get_hydro, void *user, double inj_radius,
                                      const double r0_domain[2], const double r1_domain[2], const double r2_domain[2],
                                      double *time_now, int scatt_frame, int increment_scatt_frame, double fps, uint64_t seed,
                                      FILE *fPtr, mcrat_hip_frame_stats *stats)
{
    mcrat_hip_frame_stats st;
    mcrat_hip_slab slab;
    int rc, max_scatt = 0, min_scatt = 0;
    double avg_scatt = 0, avg_r = 0;
    if (!ctx || !get_hydro || !time_now || !(fps > 0) || !r0_domain || !r1_domain || !r2_domain) return MCRAT_HIP_EINVAL;

    memset(&slab, 0, sizeof slab);
    slab.r_inj = inj_radius;
    slab.ph_inj_switch = 0;
    slab.fps = fps;
    memcpy(slab.r0_domain, r0_domain, sizeof slab.r0_domain);
    memcpy(slab.r1_domain, r1_domain, sizeof slab.r1_domain);
    memcpy(slab.r2_domain, r2_domain, sizeof slab.r2_domain);
    /* mcrat.c:704: where the photons are -> mcrat.c:721: the part of the hydro frame they can reach */
    if ((rc = mcrat_hip_ph_minmax(ctx, &slab.min_r, &slab.max_r, &slab.min_theta, &slab.max_theta)) != 0) return rc;
    if ((rc = get_hydro(user, ctx, scatt_frame, &slab)) != 0) return rc;

    const double remaining_time = ((scatt_frame + increment_scatt_frame) / fps) - *time_now;                 /* mcrat.c:758 */
    if ((rc = mcrat_hip_propagate_frame(ctx, time_now, remaining_time, seed, &st)) != 0) return rc;          /* mcrat.c:761-851 */
    if ((rc = mcrat_hip_scatt_stats(ctx, &max_scatt, &min_scatt, &avg_scatt, &avg_r)) != 0) return rc;       /* mcrat.c:881 */
    log_frame(fPtr, &st, *time_now, max_scatt, min_scatt, avg_scatt, avg_r);
    if (stats) *stats = st;
    return MCRAT_HIP_OK;
}

/* ------------------------------------------------------------------ rank pool (mcrat_hip_host.h) */
uint64_t mcrat_host_rank_seed(uint64_t rng_seed, long long k)
{
    uint64_t z = rng_seed + 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1);       /* SplitMix64 */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int mcrat_host_split_ranks(const mcrat_host_mcpar *par, int numprocs, int first_rank, int n_adopt, const char *base_dir,
                           double ph_weight_default, uint64_t base_seed, mcrat_host_rank *out)
{
    if (!par || !out || !base_dir || numprocs <= 0 || first_rank < 0 || n_adopt <= 0 || first_rank + n_adopt > numprocs) return -1;
    const int num_angles = par->n_theta_j;
    if (num_angles <= 0 || numprocs % num_angles != 0) return -1;
    const double delta_theta = (par->theta_j - par->theta_jmin) / num_angles;                     /* degrees, mcrat.c:125 */
    const int procs_per_angle = numprocs / num_angles;                                            /* :139 */
    for (int k = 0; k < n_adopt; k++) {
        mcrat_host_rank *r = &out[k];
        memset(r, 0, sizeof *r);
        const int myid = first_rank + k, color = myid / procs_per_angle;                          /* MPI_Comm_split(color, key = myid), :146 */
        double thread_theta = par->theta_jmin;
        for (int j = 1; j <= color; j++) thread_theta = thread_theta + delta_theta;               /* :129-133 */
        r->myid = myid;
        r->angle_id = myid - color * procs_per_angle;
        r->angle_procs = procs_per_angle;
        r->theta_jmin_thread = thread_theta * (M_PI / 180);                                       /* :152-153 */
        r->theta_jmax_thread = r->theta_jmin_thread + (delta_theta * (M_PI / 180));
        snprintf(r->mc_dir, sizeof r->mc_dir, "%s%0.1lf-%0.1lf/", base_dir, r->theta_jmin_thread * 180 / M_PI, r->theta_jmax_thread * 180 / M_PI);
        r->inj_radius = par->inj_radius[color];
        r->ph_weight_suggest = ph_weight_default;
        const int frm0 = par->frm0[color], frm2 = par->frm2[color];
        const int proc_frame_size = (int)ceil((frm2 - frm0) / (float)r->angle_procs);             /* :457 */
        r->framestart = frm0 + r->angle_id * proc_frame_size;                                     /* :472 */
        r->frm2 = (r->angle_id != r->angle_procs - 1) ? frm0 + r->angle_id * proc_frame_size + proc_frame_size - 1 : frm2;   /* :475-482 */
        r->rng_seed = base_seed;
        r->rng_stream = (uint32_t)myid;
    }
    return 0;
}

#define MCRAT_C_LIGHT 2.99792458e10       /* C_LIGHT, Src/mclib.c:4 */

static double wall_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e3 * (double)t.tv_sec + 1e-6 * (double)t.tv_nsec;
}

#define RANK_DEG(r) (r)->theta_jmin_thread * 180 / M_PI, (r)->theta_jmax_thread * 180 / M_PI

/* ------------------------------------------------------------------ asynchronous output (mcrat_hip_outbox_*, include/mcrat_hip.h)
 * Frame f's checkpoint and mc_proc files are written by a writer thread from pinned memory while frame f+1 propagates.  The reference writes
 * them at the end of the frame with the rank idle (mcrat.c:902,907); the bytes are the same, only who waits for the file system differs.
 * Checkpoints are independent stdio files: `helpers` threads share the ranks.  printPhotons is the caller's callback (HDF5, whose library is
 * not thread-safe in its default build): the writer thread alone calls it, for one rank after the other, beside the helpers. */
typedef struct out_rank_meta {
    int active, frame, frm2, list_capacity, num_output, angle_id, angle_procs;
    double time_now, deg_lo, deg_hi;
    const char *mc_dir;
    FILE *fPtr;
} out_rank_meta;

typedef struct out_job {
    int F, n_ranks, stride, last_frm, write_checkpoints, helpers;
    out_rank_meta *meta;
    const mcrat_hip_photon *records;
    mcrat_hip_output_columns cols;
    mcrat_host_print_arrays_fn print_photons;
    int comv_switch, stokes_switch, save_type;
    volatile int rc;
} out_job;

typedef struct out_slice { out_job *job; int first, step; pthread_t thread; int started; } out_slice;

static void *checkpoint_slice(void *p)
{
    out_slice *sl = (out_slice *)p;
    out_job *j = sl->job;
    for (int r = sl->first; r < j->n_ranks && __atomic_load_n(&j->rc, __ATOMIC_RELAXED) == 0; r += sl->step) {
        const out_rank_meta *k = &j->meta[r];
        if (!k->active) continue;
        mcrat_hip_photon_list l;
        memset(&l, 0, sizeof l);
        l.photons = (mcrat_hip_photon *)(j->records + (size_t)r * (size_t)j->stride);     /* (written from, never to) */
        l.list_capacity = k->list_capacity;
        if (k->fPtr) fprintf(k->fPtr, ">> Proc %d with angles %0.1lf-%0.1lf: Making checkpoint file\n", k->angle_id, k->deg_lo, k->deg_hi);
        if (mcrat_host_save_checkpoint(k->mc_dir, k->frame, k->frm2, j->F, k->time_now, NULL, &l, l.list_capacity, j->last_frm, k->angle_id, k->angle_procs, 0) != 0) {
            if (k->fPtr) fprintf(k->fPtr, "There is an issue with opening and saving the chkpt file therefore MCRaT is not saving data to the checkpoint or mc_proc files to prevent corruption of those data.\n");
            __atomic_store_n(&j->rc, 1, __ATOMIC_RELAXED);                 /* (several helper threads may get here) */
        }
    }
    return NULL;
}

/* saveCheckpoint (mcrat.c:902-915) and printPhotons (:907) of every active rank from the host copies the job points at */
static int write_frame_files(out_job *j)
{
    out_slice sl[64];
    int T = 0;
    if (j->write_checkpoints && j->records) {
        T = j->helpers < 1 ? 1 : (j->helpers > 64 ? 64 : j->helpers);
        for (int t = 0; t < T; t++) {
            sl[t].job = j; sl[t].first = t; sl[t].step = T;
            sl[t].started = (t + 1 < T || j->print_photons) ? pthread_create(&sl[t].thread, NULL, checkpoint_slice, &sl[t]) == 0 : 0;
            if (!sl[t].started && (t + 1 < T || j->print_photons)) checkpoint_slice(&sl[t]);     /* no thread to be had: here */
        }
        if (!j->print_photons) checkpoint_slice(&sl[T - 1]);                                   /* (the last slice on this thread when it has nothing else to do) */
        /* Every checkpoint of the frame before the first mc_proc dataset of the frame, as in the reference (mcrat.c:902-915: saveCheckpoint, and
         * printPhotons only if it succeeded -- "not saving data ... to prevent corruption"): were a rank's datasets appended while another rank's
         * checkpoint fails, a restart from the older checkpoints would append frame F to that rank's file a second time. */
        for (int t = 0; t < T; t++)
            if (sl[t].started) { pthread_join(sl[t].thread, NULL); sl[t].started = 0; }
    }
    if (j->print_photons && __atomic_load_n(&j->rc, __ATOMIC_RELAXED) == 0) {
        size_t first = 0;
        for (int r = 0; r < j->n_ranks && j->rc == 0; r++) {
            const out_rank_meta *k = &j->meta[r];
            const size_t m = (size_t)k->num_output;
            if (k->active && m > 0) {
                mcrat_hip_output_columns one = j->cols;
                double **os[17] = {&one.p0, &one.p1, &one.p2, &one.p3, &one.comv_p0, &one.comv_p1, &one.comv_p2, &one.comv_p3, &one.r0, &one.r1, &one.r2,
                                   &one.s0, &one.s1, &one.s2, &one.s3, &one.num_scatt, &one.weight};
                for (int c = 0; c < 17; c++) {
                    const int is_comv = c >= 4 && c < 8, is_stokes = c >= 11 && c < 15;
                    if ((is_comv && !j->comv_switch) || (is_stokes && !j->stokes_switch)) *os[c] = NULL;
                    else if (*os[c]) *os[c] += first;
                }
                one.type = (j->save_type && one.type) ? one.type + first : NULL;
                one.count = (int)m;
                const int prc = j->print_photons(&one, j->F, k->mc_dir, k->angle_id, k->fPtr);
                if (prc) j->rc = prc;
            }
            first += m;
        }
    }
    for (int t = 0; t < T; t++)
        if (sl[t].started) pthread_join(sl[t].thread, NULL);
    return j->rc;
}

typedef struct out_writer {
    pthread_t thread;
    int started;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    out_job job[2];
    mcrat_hip_outbox *box[2];
    long long submitted, done;
    int quit, rc;
    double ms_busy;
} out_writer;

static void *writer_main(void *p)
{
    out_writer *w = (out_writer *)p;
    for (;;) {
        pthread_mutex_lock(&w->mu);
        while (w->done == w->submitted && !w->quit) pthread_cond_wait(&w->cv, &w->mu);
        if (w->done == w->submitted) { pthread_mutex_unlock(&w->mu); return NULL; }
        const int slot = (int)(w->done & 1);
        pthread_mutex_unlock(&w->mu);
        out_job *j = &w->job[slot];
        const double t0 = wall_ms();
        int n_rec = 0;
        int rc = mcrat_hip_outbox_wait(w->box[slot], &j->records, &n_rec, &j->cols);
        if (rc == 0) rc = write_frame_files(j);
        const double dt = wall_ms() - t0;
        pthread_mutex_lock(&w->mu);
        if (rc && !w->rc) w->rc = rc;
        w->ms_busy += dt;
        w->done += 1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
    }
}

/* blocks until fewer than `keep` jobs are in flight; returns the writer's first error */
static int writer_drain(out_writer *w, long long keep)
{
    pthread_mutex_lock(&w->mu);
    while (w->submitted - w->done >= keep + 1) pthread_cond_wait(&w->cv, &w->mu);
    const int rc = w->rc;
    pthread_mutex_unlock(&w->mu);
    return rc;
}

static void *floor_slice(void *p)
{
    out_slice *sl = (out_slice *)p;
    out_job *j = sl->job;                       /* (n_ranks files of `stride` bytes each in meta[0].mc_dir, from `records`) */
    char file[2000], old[2100];
    for (int r = sl->first; r < j->n_ranks; r += sl->step) {
        snprintf(file, sizeof file, "%sfloor_%d.dat", j->meta[0].mc_dir, r);
        snprintf(old, sizeof old, "%s_old", file);
        (void)rename(file, old);
        FILE *f = fopen(file, "wb");
        if (!f) { j->rc = 1; continue; }
        if (fwrite(j->records, 1, (size_t)j->stride, f) != (size_t)j->stride) j->rc = 1;
        if (fclose(f) != 0) j->rc = 1;
    }
    return NULL;
}

int mcrat_host_output_floor(const char *dir, int n_files, size_t bytes_each, int frames, int threads, double *ms_per_frame)
{
    if (!dir || n_files <= 0 || bytes_each == 0 || bytes_each > (size_t)INT_MAX || frames <= 0 || !ms_per_frame) return MCRAT_HIP_EINVAL;
    char *payload = (char *)calloc(1, bytes_each);
    if (!payload) return MCRAT_HIP_ENOMEM;
    out_rank_meta meta;
    memset(&meta, 0, sizeof meta);
    meta.mc_dir = dir;
    out_job j;
    memset(&j, 0, sizeof j);
    j.n_ranks = n_files; j.stride = (int)bytes_each; j.meta = &meta; j.records = (const mcrat_hip_photon *)payload;
    const int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    out_slice sl[64];
    const double t0 = wall_ms();
    for (int f = 0; f < frames; f++) {
        for (int t = 0; t < T; t++) {
            sl[t].job = &j; sl[t].first = t; sl[t].step = T;
            sl[t].started = t + 1 < T ? pthread_create(&sl[t].thread, NULL, floor_slice, &sl[t]) == 0 : 0;
            if (!sl[t].started) floor_slice(&sl[t]);
        }
        for (int t = 0; t < T; t++)
            if (sl[t].started) pthread_join(sl[t].thread, NULL);
    }
    *ms_per_frame = (wall_ms() - t0) / frames;
    free(payload);
    return j.rc;
}

int mcrat_host_run_ranks(mcrat_hip_ctx *pool, mcrat_host_rank *ranks, int n_ranks, mcrat_host_pool_config *cfg)
{
    if (!pool || !ranks || n_ranks <= 0 || !cfg || !cfg->get_hydro || !(cfg->fps > 0) || cfg->max_photons <= 0) return MCRAT_HIP_EINVAL;
    if (cfg->mode != MCRAT_HIP_MODE_EXACT && (cfg->mode != MCRAT_HIP_MODE_FAST || cfg->cyclosynchrotron_switch)) return MCRAT_HIP_EINVAL;
    int rc = mcrat_hip_pool_create(pool, n_ranks, cfg->slots_per_rank > 0 ? cfg->slots_per_rank : cfg->max_photons);
    if (rc) return rc;
    mcrat_hip_rank_summary *summ = (mcrat_hip_rank_summary *)calloc((size_t)n_ranks, sizeof *summ);
    mcrat_hip_frame_stats *stats = (mcrat_hip_frame_stats *)calloc((size_t)n_ranks, sizeof *stats);
    int *open = (int *)calloc((size_t)n_ranks, sizeof(int));
    mcrat_hip_pool_inject_list *inj = (mcrat_hip_pool_inject_list *)calloc((size_t)n_ranks, sizeof *inj);
    uint64_t *seeds = (uint64_t *)calloc((size_t)n_ranks, sizeof(uint64_t));
    double *t_now = (double *)calloc((size_t)n_ranks, sizeof(double)), *t_rem = (double *)calloc((size_t)n_ranks, sizeof(double));
    mcrat_hip_pool_cs_list *cs_lists = (mcrat_hip_pool_cs_list *)calloc((size_t)n_ranks, sizeof *cs_lists);
    mcrat_hip_cyclosynch_counts *cs_counts = (mcrat_hip_cyclosynch_counts *)calloc((size_t)n_ranks, sizeof *cs_counts);
    /* two hydro frames per launch (cfg->stage_ctx): the plan's arrays [2 * n_ranks] and what the second frame left for the next turn of the loop */
    mcrat_hip_frame_stats *stats2 = (mcrat_hip_frame_stats *)calloc((size_t)n_ranks * 2, sizeof *stats2);
    int *open2 = (int *)calloc((size_t)n_ranks * 2, sizeof(int));
    uint64_t *seeds2 = (uint64_t *)calloc((size_t)n_ranks * 2, sizeof(uint64_t));
    double *t_now2 = (double *)calloc((size_t)n_ranks * 2, sizeof(double)), *t_rem2 = (double *)calloc((size_t)n_ranks * 2, sizeof(double));
    double *t_end2 = (double *)calloc((size_t)n_ranks * 2, sizeof(double));
    int second_pending = 0;              /* this turn's frame was propagated by the previous turn's launch */
    mcrat_hip_photon *rec_buf = NULL;
    double *out_buf = NULL;
    char *out_type = NULL;
    size_t out_cap = 0;
    int stride = 0;
    if (!summ || !stats || !open || !inj || !seeds || !t_now || !t_rem || !cs_lists || !cs_counts || !stats2 || !open2 || !seeds2 || !t_now2 || !t_rem2 || !t_end2 ||
        mcrat_hip_pool_layout(pool, NULL, &stride) != 0) {
        free(summ); free(stats); free(open); free(inj); free(seeds); free(t_now); free(t_rem); free(cs_lists); free(cs_counts);
        free(stats2); free(open2); free(seeds2); free(t_now2); free(t_rem2); free(t_end2);
        return MCRAT_HIP_ENOMEM;
    }
    cfg->hydro_frames_read = cfg->launches = 0;
    cfg->two_frame_launches = 0;
    cfg->ms_propagate = cfg->ms_hydro = cfg->ms_output = 0;
    cfg->ms_output_writer = cfg->ms_output_blocked = 0;
    /* the writer of the frames' files (asynchronous output): two outboxes, two jobs */
    const int async_out = !cfg->sync_output && (cfg->write_checkpoints || cfg->print_photons);
    out_writer *wr = NULL;
    if (async_out) {
        wr = (out_writer *)calloc(1, sizeof *wr);
        if (!wr) rc = MCRAT_HIP_ENOMEM;
        if (rc == 0) {
            pthread_mutex_init(&wr->mu, NULL);
            pthread_cond_init(&wr->cv, NULL);
            for (int b = 0; b < 2 && rc == 0; b++) {
                wr->job[b].meta = (out_rank_meta *)calloc((size_t)n_ranks, sizeof(out_rank_meta));
                if (!wr->job[b].meta) rc = MCRAT_HIP_ENOMEM;
                else rc = mcrat_hip_outbox_create(pool, &wr->box[b]);
            }
            if (rc == 0) {
                wr->started = pthread_create(&wr->thread, NULL, writer_main, wr) == 0;
                if (!wr->started) rc = MCRAT_HIP_ENOMEM;
            }
        }
    }
    for (int r = 0; r < n_ranks && rc == 0; r++) {
        mcrat_host_rank *k = &ranks[r];
        rc = mcrat_hip_pool_rank(pool, r, k->rng_stream, &k->view);
        if (rc == 0 && k->fast_cadence_start > 0) (void)mcrat_hip_fast_cadence(k->view, k->fast_cadence_start);
        k->frame = k->framestart;
        k->state = 0;
        k->seeds_drawn = 0;
        k->frame_scatt_cnt_total = 0;
        if (rc == 0 && k->restrt == 'c') {                                                        /* readCheckpoint's photons, mcrat.c:487 (set below, all at once) */
            if (!k->restart_list) { rc = MCRAT_HIP_EINVAL; break; }
            k->state = 4;
            k->scatt_frame = k->scatt_framestart;
            k->time_now = k->time_now_start;
            k->num_photons = k->restart_list->num_photons;
        }
        if (k->fPtr) {                                                                            /* mcrat.c:560 */
            fprintf(k->fPtr, "Im Proc %d with angles %0.1lf-%0.1lf  Starting on Frame: %d scatt_framestart: %d\n", k->angle_id, RANK_DEG(k),
                    k->framestart, k->framestart);
            fflush(k->fPtr);
        }
    }
    if (rc == 0) {                        /* the restarted ranks' lists: one copy over PCIe and one launch for all of them (mcrat_hip_pool_set_photons) */
        int n_restart = 0;
        for (int r = 0; r < n_ranks; r++) n_restart += ranks[r].state == 4;
        if (n_restart > 0) {
            int *which = (int *)malloc(sizeof(int) * (size_t)n_restart);
            mcrat_hip_photon_list *lists = (mcrat_hip_photon_list *)malloc(sizeof *lists * (size_t)n_restart);
            if (!which || !lists) rc = MCRAT_HIP_ENOMEM;
            for (int r = 0, j = 0; r < n_ranks && rc == 0; r++)
                if (ranks[r].state == 4) { which[j] = r; lists[j] = *ranks[r].restart_list; j++; }
            if (rc == 0) rc = mcrat_hip_pool_set_photons(pool, n_restart, which, lists);
            free(which); free(lists);
        }
    }
    mcrat_hip_slab slab;
    memset(&slab, 0, sizeof slab);
    slab.fps = cfg->fps;
    memcpy(slab.r0_domain, cfg->r0_domain, sizeof slab.r0_domain);
    memcpy(slab.r1_domain, cfg->r1_domain, sizeof slab.r1_domain);
    memcpy(slab.r2_domain, cfg->r2_domain, sizeof slab.r2_domain);
    long long frames_done = 0;
    int stop = 0;
    while (rc == 0 && !stop) {                     /* one turn: every rank's next injection batch (the outer loop of mcrat.c:609) */
        int first = INT_MAX;
        for (int r = 0; r < n_ranks; r++)
            if (ranks[r].state == 4) { if (ranks[r].scatt_frame < first) first = ranks[r].scatt_frame; }
            else if (ranks[r].frame <= ranks[r].frm2) { ranks[r].state = 0; if (ranks[r].frame < first) first = ranks[r].frame; }
            else ranks[r].state = 3;
        if (first == INT_MAX) break;
        for (int F = first; F <= cfg->last_frm && rc == 0 && !stop; F++) {                         /* the hydro frames, read once for all ranks */
            /* ranks whose injection frame this is (mcrat.c:626-647): the injection frame staged once per injection radius (getHydroData(...,
             * inj_radius, 1, ...), :638), then photonInjection (:645) for all ranks of that radius at once (mcrat_hip_pool_inject_photons) */
            for (int r = 0; r < n_ranks; r++) inj[r].inject = 0;
            for (int r0 = 0; r0 < n_ranks && rc == 0; r0++) {
                if (ranks[r0].state != 0 || ranks[r0].frame != F) continue;                       /* the first rank still to inject at this frame ... */
                const double radius = ranks[r0].inj_radius;
                slab.r_inj = radius; slab.ph_inj_switch = 1;
                slab.min_r = slab.max_r = slab.min_theta = slab.max_theta = 0;
                {
                    const double t0 = wall_ms();
                    rc = cfg->get_hydro(cfg->user, pool, F, &slab);
                    cfg->ms_hydro += wall_ms() - t0;
                    cfg->hydro_frames_read += 1;
                    if (rc) break;
                }
                for (int r = r0; r < n_ranks; r++) {                                              /* ... and everyone else of its radius */
                    mcrat_host_rank *k = &ranks[r];
                    inj[r].inject = 0;
                    if (k->state != 0 || k->frame != F || k->inj_radius != radius) continue;
                    k->time_now = F / cfg->fps;                                                   /* :628 */
                    if (k->fPtr) {
                        fprintf(k->fPtr, ">> Im Proc: %d with angles %0.1lf - %0.1lf Working on Frame: %d\n", k->angle_id, RANK_DEG(k), F);
                        fprintf(k->fPtr, ">>  Proc: %d with angles %0.1lf-%0.1lf: Injecting photons\n", k->angle_id, RANK_DEG(k));
                    }
                    inj[r].inject = 1; inj[r].spect = cfg->spect; inj[r].min_photons = cfg->min_photons; inj[r].max_photons = cfg->max_photons;
                    inj[r].r_inj = radius; inj[r].ph_weight = k->ph_weight_suggest;
                    inj[r].theta_min = k->theta_jmin_thread; inj[r].theta_max = k->theta_jmax_thread;
                    inj[r].seed = mcrat_host_rank_seed(k->rng_seed, k->seeds_drawn++);
                }
                if ((rc = mcrat_hip_pool_inject_photons(pool, cfg->fps, inj))) break;
                for (int r = r0; r < n_ranks; r++) {
                    mcrat_host_rank *k = &ranks[r];
                    if (!inj[r].inject) continue;
                    inj[r].inject = 0;
                    k->num_photons = inj[r].num_photons;
                    k->ph_weight = inj[r].ph_weight_adjusted;
                    k->state = 1;
                    k->scatt_frame = F;                                                           /* scatt_framestart = frame, :660 */
                    k->first_scatt_frame = F;
                    k->scatt_cyclosynch_num_ph = 0;                                               /* :921 */
                }
            }
            if (rc) break;
            int n_active = 0;
            for (int r = 0; r < n_ranks; r++) {
                if (ranks[r].state == 4 && ranks[r].scatt_frame == F) { ranks[r].state = 1; ranks[r].first_scatt_frame = -1; }   /* a restarted rank picks up at
                                                                                                     its scatt_framestart; restrt == CONTINUE emits (:707) */
                n_active += ranks[r].state == 1;
            }
            if (!n_active) continue;
            /* (second_pending: this frame was propagated by the previous turn's launch, in the frame staged on cfg->stage_ctx -- nothing to read) */
            if (!second_pending) {
            /* phMinMax of every list (mcrat.c:704) -> the slab all of them fit in -> getHydroData once (:721) */
            if ((rc = mcrat_hip_pool_summaries(pool, summ))) break;
            slab.ph_inj_switch = 0;
            slab.min_r = slab.min_theta = HUGE_VAL; slab.max_r = slab.max_theta = 0;
            for (int r = 0; r < n_ranks; r++) {
                if (ranks[r].state != 1) continue;
                if (summ[r].min_r < slab.min_r) slab.min_r = summ[r].min_r;
                if (summ[r].max_r > slab.max_r) slab.max_r = summ[r].max_r;
                if (summ[r].min_theta < slab.min_theta) slab.min_theta = summ[r].min_theta;
                if (summ[r].max_theta > slab.max_theta) slab.max_theta = summ[r].max_theta;
                slab.r_inj = ranks[r].inj_radius;
                if (cfg->cyclosynchrotron_switch && F != ranks[r].first_scatt_frame) {           /* calcCyclosynchRLimits, mcrat.c:708-720 */
                    const double lo = ranks[r].inj_radius + (MCRAT_C_LIGHT * (F - ranks[r].frame) / cfg->fps - 0.5 * MCRAT_C_LIGHT / cfg->fps);
                    const double hi = ranks[r].inj_radius + (MCRAT_C_LIGHT * (F - ranks[r].frame) / cfg->fps + 0.5 * MCRAT_C_LIGHT / cfg->fps);
                    if (lo < slab.min_r) slab.min_r = lo;
                    if (hi > slab.max_r) slab.max_r = hi;
                }
            }
            {
                const double t0 = wall_ms();
                rc = cfg->get_hydro(cfg->user, pool, F, &slab);
                cfg->ms_hydro += wall_ms() - t0;
                cfg->hydro_frames_read += 1;
                if (rc) break;
            }
            }
            /* Two hydro frames in one launch (cfg->stage_ctx; EXACT mode): frame F + 1 staged beside frame F for everything the photons can reach from
             * where they are (phMinMax widened by c / fps: nobody outruns light), every list through both frames in one call of
             * mcrat_hip_pool_run_frames -- a list that is through frame F goes on in F + 1 while others are still in F, as the reference's ranks do in
             * their own frame loops (mcrat.c:566-934).  Only when no rank joins at F + 1 (injection and restart sit between two frames), and the
             * frames' outputs are taken from what each frame left (capture_frames, mcrat_hip_pool_select_frame). */
            int two = 0;
            if (!second_pending && cfg->stage_ctx && !cfg->cyclosynchrotron_switch && cfg->mode == MCRAT_HIP_MODE_EXACT && F + 1 <= cfg->last_frm &&
                !(cfg->max_frames > 0 && frames_done + 2 > cfg->max_frames)) {
                two = 1;
                for (int r = 0; r < n_ranks; r++) {
                    if (ranks[r].state == 0 && ranks[r].frame == F + 1) two = 0;                  /* injects at F + 1 */
                    if (ranks[r].state == 4 && ranks[r].scatt_frame == F + 1) two = 0;            /* a restarted rank picks up at F + 1 */
                    if (ranks[r].state == 2 || ranks[r].state == 3) continue;
                }
            }
            if (two) {
                mcrat_hip_slab next = slab;
                const double reach = MCRAT_C_LIGHT / cfg->fps;
                next.min_r = slab.min_r - reach > 0 ? slab.min_r - reach : 0;
                next.max_r = slab.max_r + reach;
                const double dth = next.min_r > 0 ? reach / next.min_r : M_PI;
                next.min_theta = slab.min_theta - dth > 0 ? slab.min_theta - dth : 0;
                next.max_theta = slab.max_theta + dth < M_PI ? slab.max_theta + dth : M_PI;
                const double t0 = wall_ms();
                rc = cfg->get_hydro(cfg->user, cfg->stage_ctx, F + 1, &next);
                cfg->ms_hydro += wall_ms() - t0;
                cfg->hydro_frames_read += 1;
                if (rc) break;
            }
            const double t_prop = wall_ms();
            for (int r = 0; r < n_ranks; r++) {
                mcrat_host_rank *k = &ranks[r];
                open[r] = k->state == 1;
                if (!open[r]) continue;
                if (k->fPtr) {
                    fprintf(k->fPtr, ">>\n>> Proc %d with angles %0.1lf-%0.1lf: Working on photons injected at frame: %d out of %d\n", k->angle_id, RANK_DEG(k),
                            k->frame, k->frm2);
                    fprintf(k->fPtr, ">> Proc %d with angles %0.1lf-%0.1lf: propagating and scattering %d photons\n", k->angle_id, RANK_DEG(k), k->num_photons);
                }
                if (second_pending) continue;                                                    /* (its seed was drawn with the launch) */
                /* the rank's per-frame seed (gsl_rng_set(rng, gsl_rng_get(rng)), :701) and its own clock (:758) */
                seeds[r] = mcrat_host_rank_seed(k->rng_seed, k->seeds_drawn++);
                t_now[r] = k->time_now;
                t_rem[r] = ((F + 1) / cfg->fps) - k->time_now;
            }
            int captured = 0;
            if (second_pending) {
                for (int r = 0; r < n_ranks; r++) stats[r] = stats2[n_ranks + r];
            } else if (two) {
                for (int r = 0; r < n_ranks; r++) {
                    mcrat_host_rank *k = &ranks[r];
                    open2[r] = open2[n_ranks + r] = open[r];
                    seeds2[r] = seeds[r];
                    seeds2[n_ranks + r] = open[r] ? mcrat_host_rank_seed(k->rng_seed, k->seeds_drawn++) : 0;     /* frame F + 1's seed: the rank's next */
                    t_now2[r] = t_now[r]; t_rem2[r] = t_rem[r];
                    t_now2[n_ranks + r] = t_rem2[n_ranks + r] = 0;                                            /* (the clock is carried: chain_clock) */
                    t_end2[r] = (F + 1) / cfg->fps; t_end2[n_ranks + r] = (F + 2) / cfg->fps;
                }
                mcrat_hip_ctx *frames[2] = {NULL, cfg->stage_ctx};
                mcrat_hip_frame_plan plan;
                memset(&plan, 0, sizeof plan);
                plan.n_frames = 2; plan.chain_clock = 1; plan.capture_frames = 1;
                plan.open = open2; plan.seeds = seeds2; plan.time_now = t_now2; plan.remaining_time = t_rem2; plan.frame_end = t_end2; plan.hydro = frames;
                if ((rc = mcrat_hip_pool_run_frames(pool, &plan, stats2))) break;
                for (int r = 0; r < n_ranks; r++) stats[r] = stats2[r];
                if ((rc = mcrat_hip_pool_select_frame(pool, 0))) break;                          /* frame F's outputs: the lists as frame F left them */
                captured = 1;
                cfg->two_frame_launches += 1;
            } else if (!cfg->cyclosynchrotron_switch && cfg->mode == MCRAT_HIP_MODE_FAST) {
                if ((rc = mcrat_hip_pool_propagate_frames_fast(pool, open, seeds, t_now, t_rem, cfg->fast_windows, stats))) break;
                for (int r = 0; r < n_ranks; r++)
                    if (open[r] && ranks[r].view) ranks[r].fast_cadence = mcrat_hip_fast_cadence(ranks[r].view, 0);
            } else if (!cfg->cyclosynchrotron_switch) {
                if ((rc = mcrat_hip_pool_begin_frames(pool, open, seeds, t_now, t_rem))) break;  /* every list's begin_frame, one launch */
                mcrat_hip_frame_stats tot;
                if ((rc = mcrat_hip_run(pool, 0, &tot))) break;                                   /* mcrat.c:761-851 for every list */
                if ((rc = mcrat_hip_pool_frame_stats(pool, stats))) break;
            } else {                                                                              /* mcrat.c:706-878 for every list */
                for (int r = 0; r < n_ranks; r++) {
                    mcrat_host_rank *k = &ranks[r];
                    memset(&cs_lists[r], 0, sizeof cs_lists[r]);
                    memset(&cs_counts[r], 0, sizeof cs_counts[r]);
                    if (!open[r]) continue;
                    cs_lists[r].open = 1;
                    cs_lists[r].emit_pool = F != k->first_scatt_frame;                            /* (scatt_frame != scatt_framestart) || CONTINUE, :707 */
                    cs_lists[r].scatt_frame_number = F; cs_lists[r].inj_frame_number = k->frame;
                    cs_lists[r].seed = seeds[r]; cs_lists[r].time_now = t_now[r]; cs_lists[r].remaining_time = t_rem[r];
                    cs_lists[r].r_inj = k->inj_radius; cs_lists[r].ph_weight_suggest = k->ph_weight_suggest;
                    cs_lists[r].theta_min = k->theta_jmin_thread; cs_lists[r].theta_max = k->theta_jmax_thread;
                    cs_counts[r].scatt_cyclosynch_num_ph = k->scatt_cyclosynch_num_ph;
                    if (cs_lists[r].emit_pool && k->fPtr) fprintf(k->fPtr, "Emitting Cyclosynchrotron Photons in frame %d\n", F);
                }
                if ((rc = mcrat_hip_pool_scatter_frames_cyclosynch(pool, &cfg->cs, cfg->max_photons, cfg->fps, cs_lists, stats, cs_counts))) break;
                for (int r = 0; r < n_ranks; r++) {
                    if (!open[r]) continue;
                    ranks[r].scatt_cyclosynch_num_ph = cs_counts[r].scatt_cyclosynch_num_ph;
                    ranks[r].cyclosynch_emitted_total += cs_counts[r].num_cyclosynch_ph_emit;
                    ranks[r].cyclosynch_absorbed_total += cs_counts[r].frame_abs_cnt;
                    if (ranks[r].fPtr) fprintf(ranks[r].fPtr, "The number of cyclosynchrotron photons absorbed in this frame is: %d\n", cs_counts[r].frame_abs_cnt);
                }
            }
            if (!second_pending) cfg->launches += 1;
            if ((rc = mcrat_hip_pool_summaries(pool, summ))) break;                               /* phScattStats, :881 */
            cfg->ms_propagate += wall_ms() - t_prop;
            const double t_out = wall_ms();
            for (int r = 0; r < n_ranks; r++) {
                mcrat_host_rank *k = &ranks[r];
                if (k->state != 1) continue;
                k->time_now = stats[r].time_now;
                k->frame_scatt_cnt_total += stats[r].frame_scatt_cnt;
                log_frame(k->fPtr, &stats[r], k->time_now, summ[r].max_scatt, summ[r].min_scatt, summ[r].avg_scatt, summ[r].avg_r);
            }
            /* saveCheckpoint (:902-915): the records of all lists come over in pieces of whole lists, one transfer per piece */
            /* saveCheckpoint's 'k' -> 'c' (mcrat_io.c:896-900), on every list at once.  In the reference the conversion sits inside saveCheckpoint's
             * successful-fopen branch, and a run cannot go on without its checkpoint (it exits) -- so every frame of a reference run ends with it.
             * Here it is therefore part of the FRAME, whatever write_checkpoints says: the next frame's phAbsCyclosynch and the PT column of
             * mc_proc see the converted types also in runs that skip the checkpoint files (benchmarks). */
            if (cfg->cyclosynchrotron_switch && (rc = mcrat_hip_convert_comptonized(pool, NULL))) break;
            if (async_out) {
                /* the files of frame F: records and columns staged on the device in stream order and on their way into pinned memory when post()
                 * returns; the writer thread takes it from there while the loop goes on with frame F + 1 */
                const double tb = wall_ms();
                rc = writer_drain(wr, 1);                                                         /* a free outbox (the one of frame F - 2) */
                cfg->ms_output_blocked += wall_ms() - tb;
                if (rc) break;
                const int slot = (int)(wr->submitted & 1);
                if ((rc = mcrat_hip_outbox_post(pool, wr->box[slot], cfg->write_checkpoints, cfg->print_photons != NULL))) break;
                out_job *j = &wr->job[slot];
                j->F = F; j->n_ranks = n_ranks; j->stride = stride; j->last_frm = cfg->last_frm; j->write_checkpoints = cfg->write_checkpoints;
                j->helpers = cfg->output_threads > 0 ? cfg->output_threads : 4;
                j->print_photons = cfg->print_photons;
                j->comv_switch = cfg->comv_switch; j->stokes_switch = cfg->stokes_switch; j->save_type = cfg->save_type;
                j->rc = 0;
                for (int r = 0; r < n_ranks; r++) {
                    const mcrat_host_rank *k = &ranks[r];
                    out_rank_meta *m = &j->meta[r];
                    m->active = k->state == 1;
                    m->frame = k->frame; m->frm2 = k->frm2; m->time_now = k->time_now;
                    m->list_capacity = summ[r].list_capacity; m->num_output = summ[r].num_output;
                    m->angle_id = k->angle_id; m->angle_procs = k->angle_procs;
                    m->deg_lo = k->theta_jmin_thread * 180 / M_PI; m->deg_hi = k->theta_jmax_thread * 180 / M_PI;
                    m->mc_dir = k->mc_dir; m->fPtr = k->fPtr;
                }
                pthread_mutex_lock(&wr->mu);
                wr->submitted += 1;
                pthread_cond_broadcast(&wr->cv);
                pthread_mutex_unlock(&wr->mu);
            } else {
                if (cfg->write_checkpoints) {
                    const int per_piece = (1 << 20) / stride > 0 ? (1 << 20) / stride : 1;
                    if (!rec_buf) rec_buf = (mcrat_hip_photon *)malloc(sizeof(mcrat_hip_photon) * (size_t)per_piece * (size_t)stride);
                    if (!rec_buf) { rc = MCRAT_HIP_ENOMEM; break; }
                    for (int r0 = 0; r0 < n_ranks && rc == 0; r0 += per_piece) {
                        const int r1 = r0 + per_piece < n_ranks ? r0 + per_piece : n_ranks;
                        int any = 0;
                        for (int r = r0; r < r1; r++) any |= ranks[r].state == 1;
                        if (!any) continue;
                        if ((rc = mcrat_hip_get_photons_range(pool, r0 * stride, (r1 - r0) * stride, rec_buf))) break;
                        for (int r = r0; r < r1; r++) {
                            mcrat_host_rank *k = &ranks[r];
                            if (k->state != 1) continue;
                            mcrat_hip_photon_list l;
                            memset(&l, 0, sizeof l);
                            l.photons = rec_buf + (size_t)(r - r0) * (size_t)stride;
                            l.list_capacity = summ[r].list_capacity;
                            if (k->fPtr) fprintf(k->fPtr, ">> Proc %d with angles %0.1lf-%0.1lf: Making checkpoint file\n", k->angle_id, RANK_DEG(k));
                            if (mcrat_host_save_checkpoint(k->mc_dir, k->frame, k->frm2, F, k->time_now, NULL, &l, l.list_capacity, cfg->last_frm, k->angle_id,
                                                           k->angle_procs, 0) != 0) {
                                if (k->fPtr) fprintf(k->fPtr, "There is an issue with opening and saving the chkpt file therefore MCRaT is not saving data to the checkpoint or mc_proc files to prevent corruption of those data.\n");
                                rc = 1;
                                break;
                            }
                        }
                    }
                    if (rc) break;
                }
                /* printPhotons (:907): the pool's compacted columns in one transfer (photons with weight != 0 in slot order, i.e. list after list) */
                if (cfg->print_photons) {
                    mcrat_hip_output_columns all;
                    memset(&all, 0, sizeof all);
                    if ((rc = mcrat_hip_get_output(pool, &all))) break;                               /* the count */
                    const size_t cnt = (size_t)all.count;
                    if (cnt > out_cap) {
                        free(out_buf); free(out_type);
                        out_buf = (double *)malloc(sizeof(double) * 17 * (cnt ? cnt : 1));
                        out_type = (char *)malloc(cnt ? cnt : 1);
                        out_cap = cnt;
                        if (!out_buf || !out_type) { rc = MCRAT_HIP_ENOMEM; break; }
                    }
                    double **slot[17] = {&all.p0, &all.p1, &all.p2, &all.p3, &all.comv_p0, &all.comv_p1, &all.comv_p2, &all.comv_p3, &all.r0, &all.r1, &all.r2,
                                         &all.s0, &all.s1, &all.s2, &all.s3, &all.num_scatt, &all.weight};
                    for (int c = 0; c < 17; c++) {
                        const int is_comv = c >= 4 && c < 8, is_stokes = c >= 11 && c < 15;
                        *slot[c] = ((is_comv && !cfg->comv_switch) || (is_stokes && !cfg->stokes_switch)) ? NULL : out_buf + (size_t)c * cnt;
                    }
                    all.type = cfg->save_type ? out_type : NULL;
                    if (cnt && (rc = mcrat_hip_get_output(pool, &all))) break;
                    size_t first = 0;
                    for (int r = 0; r < n_ranks && rc == 0; r++) {
                        mcrat_host_rank *k = &ranks[r];
                        const size_t m = (size_t)summ[r].num_output;
                        if (k->state == 1 && m > 0) {
                            mcrat_hip_output_columns one = all;
                            double **os[17] = {&one.p0, &one.p1, &one.p2, &one.p3, &one.comv_p0, &one.comv_p1, &one.comv_p2, &one.comv_p3, &one.r0, &one.r1, &one.r2,
                                               &one.s0, &one.s1, &one.s2, &one.s3, &one.num_scatt, &one.weight};
                            for (int c = 0; c < 17; c++)
                                if (*os[c]) *os[c] += first;
                            if (one.type) one.type += first;
                            one.count = (int)m;
                            rc = cfg->print_photons(&one, F, k->mc_dir, k->angle_id, k->fPtr);
                        }
                        first += m;
                    }
                    if (rc) break;
                }
            }
            if (captured && (rc = mcrat_hip_pool_select_frame(pool, -1))) break;                 /* back to the live lists (= what frame F + 1 left) */
            second_pending = captured;
            for (int r = 0; r < n_ranks; r++)
                if (ranks[r].state == 1) ranks[r].scatt_frame = F + 1;
            cfg->ms_output += wall_ms() - t_out;
            frames_done += 1;
            if (cfg->max_frames > 0 && frames_done >= cfg->max_frames) stop = 1;
        }
        /* the batch is through its last hydro frame (mcrat.c:920-922): restrt = INITALIZE, freePhotonList, next injection frame */
        for (int r = 0; r < n_ranks && !stop; r++)
            if (ranks[r].state == 1) { ranks[r].state = 2; ranks[r].frame += 1; }
            else if (ranks[r].state == 0 || ranks[r].state == 4) { ranks[r].state = 2; ranks[r].frame += 1; }      /* its injection frame lies beyond last_frm: nothing to scatter in */
    }
    if (wr) {                                              /* the last frames' files, then the writer goes */
        if (wr->started) {
            const double tb = wall_ms();
            const int wrc = writer_drain(wr, 0);
            cfg->ms_output_blocked += wall_ms() - tb;
            cfg->ms_output += wall_ms() - tb;
            if (rc == 0) rc = wrc;
            pthread_mutex_lock(&wr->mu);
            wr->quit = 1;
            pthread_cond_broadcast(&wr->cv);
            pthread_mutex_unlock(&wr->mu);
            pthread_join(wr->thread, NULL);
            cfg->ms_output_writer = wr->ms_busy;
        }
        for (int b = 0; b < 2; b++) { mcrat_hip_outbox_destroy(wr->box[b]); free(wr->job[b].meta); }
        pthread_mutex_destroy(&wr->mu);
        pthread_cond_destroy(&wr->cv);
        free(wr);
    }
    if (rc == 0 && !stop && cfg->write_checkpoints)
        for (int r = 0; r < n_ranks; r++) {                                                       /* the closing saveCheckpoint of :924, list freed */
            mcrat_host_rank *k = &ranks[r];
            (void)mcrat_host_save_checkpoint(k->mc_dir, k->frame, k->frm2, cfg->last_frm + 1, k->time_now, NULL, NULL, 0, cfg->last_frm, k->angle_id,
                                             k->angle_procs, 0);
            if (k->fPtr) { fprintf(k->fPtr, "Process %d has completed the MC calculation.\n", k->angle_id); fflush(k->fPtr); }
        }
    free(summ); free(stats); free(open); free(inj); free(seeds); free(t_now); free(t_rem); free(cs_lists); free(cs_counts);
    free(stats2); free(open2); free(seeds2); free(t_now2); free(t_rem2); free(t_end2);
    free(rec_buf); free(out_buf); free(out_type);
    return rc;
}

/* ------------------------------------------------------------------ shared clock: the host loop (mcrat_hip_host.h) */
int mcrat_host_shared_clock_frame(mcrat_hip_ctx *ctx, int world, int rank, long long slot_base, mcrat_host_allgather_fn exchange, void *user,
                                  void *stream, double *time_now, double remaining_time, uint64_t seed, int rounds_per_poll,
                                  mcrat_hip_frame_stats *stats)
{
    if (!ctx || !time_now || world < 1 || (world > 1 && !exchange)) return MCRAT_HIP_EINVAL;
    if (rounds_per_poll < 1) rounds_per_poll = 32;
    void *send = NULL, *recv = NULL;
    int rc;
    if (mcrat_hip_shared_clock_buffers(ctx, &send, &recv) != 0 &&
        (rc = mcrat_hip_shared_clock_attach(ctx, world, rank, slot_base, NULL, NULL)) != 0)
        return rc;
    if ((rc = mcrat_hip_shared_clock_buffers(ctx, &send, &recv)) != 0) return rc;
    const size_t nb = mcrat_hip_shared_clock_bytes_per_rank();
    if ((rc = mcrat_hip_begin_frame(ctx, seed, *time_now, remaining_time)) != 0) return rc;
    mcrat_hip_frame_stats st;
    int done = 0;
    while (!done) {
        for (int k = 0; k < rounds_per_poll; k++) {
            if ((rc = mcrat_hip_shared_clock_propose(ctx)) != 0) return rc;
            if (exchange && (rc = exchange(user, send, recv, nb, stream)) != 0) return rc;   /* (one rank needs none, but may have one: device exchange) */
            if ((rc = mcrat_hip_shared_clock_resolve(ctx)) != 0) return rc;
        }
        if ((rc = mcrat_hip_shared_clock_poll(ctx, &done, &st)) != 0) return rc;      /* identical on every rank: the loops stay in step */
    }
    if ((rc = mcrat_hip_shared_clock_finish(ctx, &st)) != 0) return rc;
    *time_now = st.time_now;
    if (stats) *stats = st;
    return MCRAT_HIP_OK;
}

int mcrat_host_exchange_device(void *user, const void *send, void *recv, size_t bytes_per_rank, void *stream)
{
    (void)send; (void)recv; (void)bytes_per_rank; (void)stream;    /* the context knows its buffers, peers and stream */
    return mcrat_hip_shared_clock_exchange((mcrat_hip_ctx *)user);
}

/* ------------------------------------------------------------------ A/B shims (mcrat_hip_host.h) */
int mcrat_ab_begin_frame(mcrat_ab_rng *rng, mcrat_hip_ctx *ctx, const mcrat_hip_photon_list *ph, const mcrat_hip_hydro *hydro,
                         uint64_t seed, double time_now, double remaining_time)
{
    int rc;
    if (!rng || !ctx || !ph || !hydro) return MCRAT_HIP_EINVAL;
    rng->ctx = ctx; rng->relocated_seen = 0; rng->scatt_seen = 0;
    if ((rc = mcrat_hip_set_hydro(ctx, hydro)) == 0 && (rc = mcrat_hip_set_photons(ctx, ph)) == 0)
        rc = mcrat_hip_begin_frame(ctx, seed, time_now, remaining_time);
    rng->last_rc = rc;
    return rc;
}

int mcrat_ab_findContainingHydroCell(mcrat_hip_photon_list *ph, const mcrat_hip_hydro *hydro, int find_nearest_block_switch,
                                     mcrat_ab_rng *rng, FILE *fPtr)
{
    mcrat_hip_frame_stats st;
    int rc, n = 0;
    long long k;
    (void)hydro;                                   /* staged by mcrat_ab_begin_frame */
    if (!ph || !rng || !rng->ctx) return 0;
    if ((rc = mcrat_hip_step_locate_sample(rng->ctx, find_nearest_block_switch)) == 0 &&
        (rc = mcrat_hip_frame_statistics(rng->ctx, &st)) == 0 && (rc = mcrat_hip_get_photons(rng->ctx, ph)) == 0) {
        n = (int)(st.num_photons_find_new_element - rng->relocated_seen);       /* 0 on the forced pass, mclib.c:608-611 */
        rng->relocated_seen = st.num_photons_find_new_element;
        if (fPtr)
            for (k = 0; k < st.not_found; ++k)                                   /* mclib.c:583 */
                fprintf(fPtr, "Photon Hydro grid index not found, making sure it doesnt scatter.\n");
    }
    rng->last_rc = rc;
    return n;
}

static const mcrat_hip_photon *ab_sort_key;
static int ab_cmp(const void *a, const void *b)
{
    const int ia = *(const int *)a, ib = *(const int *)b;
    double ta = ab_sort_key[ia].time_to_scatter, tb = ab_sort_key[ib].time_to_scatter;
    if (ta != ta) ta = HUGE_VAL;
    if (tb != tb) tb = HUGE_VAL;
    if (ta < tb) return -1;
    if (ta > tb) return 1;
    return (ia > ib) - (ia < ib);
}

void mcrat_ab_calcMeanFreePath(mcrat_hip_photon_list *ph, const mcrat_hip_hydro *hydro, mcrat_ab_rng *rng, FILE *fPtr)
{
    int i;
    (void)hydro; (void)fPtr;
    if (!ph || !rng || !ph->sorted_indexes) { if (rng) rng->last_rc = MCRAT_HIP_EINVAL; return; }
    /* time_to_scatter came back with mcrat_ab_findContainingHydroCell (one fused kernel); what remains of
     * calcMeanFreePath is its argsort, mclib.c:702-712 */
    for (i = 0; i < ph->list_capacity; ++i) ph->sorted_indexes[i] = i;
    ab_sort_key = ph->photons;
    qsort(ph->sorted_indexes, (size_t)ph->list_capacity, sizeof(int), ab_cmp);
    rng->last_rc = 0;
}

double mcrat_ab_photonEvent(mcrat_hip_photon_list *ph, double dt_max, const mcrat_hip_hydro *hydro, int *scattered_ph_index,
                            int *frame_scatt_cnt, int *frame_abs_cnt, mcrat_ab_rng *rng, FILE *fPtr)
{
    mcrat_hip_frame_stats st;
    int rc;
    (void)hydro; (void)frame_abs_cnt; (void)fPtr;   /* absorption belongs to the cyclo-synchrotron build */
    if (!ph || !rng || !rng->ctx) return 0;
    if ((rc = mcrat_hip_frame_statistics(rng->ctx, &st)) == 0 && st.remaining_time != dt_max)
        rc = MCRAT_HIP_ESTATE;                      /* main() passes remaining_time (mcrat.c:781); the context keeps the same clock */
    if (rc == 0 && (rc = mcrat_hip_step_event(rng->ctx, &st)) == 0 && (rc = mcrat_hip_get_photons(rng->ctx, ph)) == 0) {
        if (scattered_ph_index) *scattered_ph_index = st.last_scattered_index;
        if (frame_scatt_cnt) *frame_scatt_cnt += (int)(st.frame_scatt_cnt - rng->scatt_seen);
        rng->scatt_seen = st.frame_scatt_cnt;
        rng->last_rc = 0;
        return st.last_time_step;
    }
    rng->last_rc = rc;
    return 0;
}

void mcrat_ab_updatePhotonPosition(mcrat_hip_photon_list *ph, double t, mcrat_ab_rng *rng, FILE *fPtr)
{
    int rc;
    (void)fPtr;
    if (!ph || !rng || !rng->ctx) return;
    if ((rc = mcrat_hip_update_photon_position(rng->ctx, t)) == 0) rc = mcrat_hip_get_photons(rng->ctx, ph);
    rng->last_rc = rc;
}
