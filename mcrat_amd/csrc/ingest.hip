// ingest.hip -- getHydroData on the device (Src/mcrat_io.c:1898-1990): from the buffers a reader holds after
// its file reads to the staged hydro frame, without a host pass over the cells.
//   FLASH   readAndDecimate, mclib_flash.c:199-428: leaf blocks (node type 1) expand to 8x8 cells, x fastest, centres at
//           (+-1,3,5,7)/16 of the block size; cells are numbered over the leaf blocks in file order.
//   PLUTO   readPluto, mclib_pluto.c:1130-1456: cell (j,k,l) of the [nz][ny][nx] blocks, x1 fastest.
// Both then keep the cells whose corners reach into the slab of the photons (ph_inj_switch == 0; widened by
// elem_factor c/fps until at least one cell is kept) or whose centre lies beyond 0.95 r_inj (injection frame), in cell
// order.  Here: a "virtual cell" index runs over everything the reader would visit; pass 1 counts the kept cells per
// workgroup (repeated by the host with a larger elem_factor while the total is zero), an exclusive scan turns the
// counts into output positions, pass 2 writes the columns of struct hydro_dataframe in the reference's order.  Then
// fillHydroCoordinateToSpherical (geometry.c:156-174), the SIMULATION_TYPE overwrite (analytic_outflows.c) and the
// staging of the per-cell records the loop kernels read (what mcrat_hip_set_hydro's host loop did).
// All of it streams: 8-byte columns, coalesced over the virtual index; nothing here is worth LDS beyond the counts.
#include <hip/hip_runtime.h>
#include <math.h>
#include "device_types.hpp"
#include "launch.hpp"

namespace mcrat {

namespace {

constexpr int IB = 256;

// geometry.c:66-106
__device__ __forceinline__ void hydro_to_spherical(int dims, int geom, double r0, double r1, double r2, double &r, double &theta)
{
    r = 0; theta = 0;
    if (dims == DIM_TWO || dims == DIM_TWO_POINT_FIVE) {
        if (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL) { r = sqrt(r0 * r0 + r1 * r1); theta = atan2(r0, r1); }
        if (geom == GEOM_SPHERICAL) { r = r0; theta = r1; }
    } else {
        if (geom == GEOM_CARTESIAN) { r = sqrt(r0 * r0 + r1 * r1 + r2 * r2); theta = acos(r2 / r); }
        if (geom == GEOM_SPHERICAL) { r = r0; theta = r1; }
        if (geom == GEOM_POLAR) { r = sqrt(r0 * r0 + r2 * r2); theta = acos(r2 / r); }
    }
}

struct RawCell {
    double x0, x1, x2, s0, s1, s2;
};

// mclib_flash.c:236-268
struct FlashSource {
    FlashDev f;
    __device__ long long count() const { return f.n_blocks * 64; }
    __device__ bool present(long long i) const { return f.node[i >> 6] == 1; }
    __device__ RawCell geom(long long i) const
    {
        const long long b = i >> 6;
        const int j = (int)(i & 63);
        const double bs0 = f.bsize[b * f.bsize_stride], bs1 = f.bsize[b * f.bsize_stride + 1];
        const double c0 = f.coord[b * f.coord_stride], c1 = f.coord[b * f.coord_stride + 1];
        // x1[] of mclib_flash.c:69: (2 k - 7) / 16, exact in binary
        const double o0 = (double)(2 * (j & 7) - 7) / 16.0, o1 = (double)(2 * (j >> 3) - 7) / 16.0;
        RawCell c;
        c.x0 = (c0 + bs0 * o0) * f.L;
        c.x1 = (c1 + bs1 * o1) * f.L;
        c.x2 = 0;
        c.s0 = (bs0 / 8) * f.L;
        c.s1 = (bs1 / 8) * f.L;
        c.s2 = 0;
        return c;
    }
    __device__ void fluid(long long i, double &v0, double &v1, double &v2, double &dens, double &pres) const
    {
        v0 = f.velx[i]; v1 = f.vely[i]; v2 = 0;
        dens = f.dens[i] * f.D;
        pres = f.pres[i] * f.P;
    }
};

// mclib_pluto.c:1137-1215
struct PlutoSource {
    PlutoDev g;
    int scale1, scale2;     // x2 / x3 are lengths in this geometry (mclib_pluto.c:1163-1175)
    int three, v3;
    __device__ long long count() const { return (long long)g.nx * g.ny * g.nz; }
    __device__ bool present(long long) const { return true; }
    __device__ RawCell geom(long long i) const
    {
        const int l = (int)(i % g.nx);
        const long long t = i / g.nx;
        const int k = (int)(t % g.ny), j = (int)(t / g.ny);
        RawCell c;
        c.x0 = g.x1[l] * g.L; c.s0 = g.dx1[l] * g.L;
        c.x1 = g.x2[k]; c.s1 = g.dx2[k];
        if (scale1) { c.x1 *= g.L; c.s1 *= g.L; }
        c.x2 = 0; c.s2 = 0;
        if (three) {
            c.x2 = g.x3[j]; c.s2 = g.dx3[j];
            if (scale2) { c.x2 *= g.L; c.s2 *= g.L; }
        }
        return c;
    }
    __device__ void fluid(long long i, double &v0, double &v1, double &v2, double &dens, double &pres) const
    {
        v0 = g.vx1[i]; v1 = g.vx2[i]; v2 = v3 ? g.vx3[i] : 0.0;
        dens = g.rho[i] * g.D;
        pres = g.prs[i] * g.P;
    }
};

// mclib_pluto.c:520-612 (readPlutoChombo): cells numbered level by level, box by box, x fastest
struct ChomboSource {
    ChomboDev h;
    int scale1, scale2, three, v3;
    __device__ long long count() const { return h.cells; }
    __device__ int box_of(long long i) const
    {
        int lo = 0, hi = h.n_boxes - 1;                  // last box with first_cell <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (h.boxes[mid].first_cell <= i) lo = mid; else hi = mid - 1;
        }
        return lo;
    }
    __device__ bool present(long long i) const { return !h.covered || h.covered[i] == 0; }
    __device__ RawCell geom(long long i) const
    {
        const ChomboBox b = h.boxes[box_of(i)];
        const long long q = i - b.first_cell;
        const int n = (int)(q % b.n[0]);
        const long long t = q / b.n[0];
        const int m = (int)(t % b.n[1]), l = (int)(t / b.n[1]);
        RawCell c;
        c.x0 = h.x[0][b.cb[0] + b.lo[0] + n] * h.L; c.s0 = h.dx[0][b.cb[0] + b.lo[0] + n] * h.L;
        c.x1 = h.x[1][b.cb[1] + b.lo[1] + m]; c.s1 = h.dx[1][b.cb[1] + b.lo[1] + m];
        if (scale1) { c.x1 *= h.L; c.s1 *= h.L; }
        c.x2 = 0; c.s2 = 0;
        if (three) {
            c.x2 = h.x[2][b.cb[2] + b.lo[2] + l]; c.s2 = h.dx[2][b.cb[2] + b.lo[2] + l];
            if (scale2) { c.x2 *= h.L; c.s2 *= h.L; }
        }
        return c;
    }
    __device__ void fluid(long long i, double &v0, double &v1, double &v2, double &dens, double &pres) const
    {
        const ChomboBox b = h.boxes[box_of(i)];
        const long long vs = (long long)b.n[0] * b.n[1] * b.n[2];
        const double *d = h.data + b.data_off + (i - b.first_cell);
        dens = h.kv[0] >= 0 ? d[h.kv[0] * vs] * h.D : 0.0;
        v0 = h.kv[1] >= 0 ? d[h.kv[1] * vs] : 0.0;
        v1 = h.kv[2] >= 0 ? d[h.kv[2] * vs] : 0.0;
        v2 = (v3 && h.kv[3] >= 0) ? d[h.kv[3] * vs] : 0.0;
        pres = h.kv[4] >= 0 ? d[h.kv[4] * vs] * h.P : 0.0;
    }
};

__device__ __forceinline__ int floor_div(int a, int b) { const int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
__device__ __forceinline__ int ceil_div(int a, int b) { return -floor_div(-a, b); }

// good_node_buffer (mclib_pluto.c:206-345): a cell of level i at index g is covered when ref_ratio * g lies inside a box
// of level i+1, i.e. ceil(lo / ref) <= g <= floor(hi / ref) on every axis.  One thread per (coarse box, fine box) pair;
// a pair that overlaps marks its intersection (fine boxes are disjoint, so every covered cell is marked once).
__global__ __launch_bounds__(IB) void chombo_mask_kernel(const ChomboBox *__restrict__ boxes, int c0, int nc, int f0, int nf, int ref, int three,
                                                         unsigned char *__restrict__ covered)
{
    const long long pairs = (long long)nc * nf;
    for (long long p = (long long)blockIdx.x * IB + threadIdx.x; p < pairs; p += (long long)gridDim.x * IB) {
        const ChomboBox c = boxes[c0 + (int)(p / nf)], f = boxes[f0 + (int)(p % nf)];
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        bool any = true;
        for (int a = 0; a < (three ? 3 : 2); ++a) {
            const int flo = ceil_div(f.lo[a], ref), fhi = floor_div(f.lo[a] + f.n[a] - 1, ref);
            lo[a] = max(c.lo[a], flo) - c.lo[a];
            hi[a] = min(c.lo[a] + c.n[a] - 1, fhi) - c.lo[a];
            any = any && lo[a] <= hi[a];
        }
        if (!any) continue;
        for (int l = lo[2]; l <= hi[2]; ++l)
            for (int m = lo[1]; m <= hi[1]; ++m)
                for (int n = lo[0]; n <= hi[0]; ++n)
                    covered[c.first_cell + (long long)l * c.n[0] * c.n[1] + (long long)m * c.n[0] + n] = 1;
    }
}

// mclib_flash.c:288-318 == mclib_pluto.c:1260-1301
__device__ __forceinline__ bool in_slab(const SlabDev &s, const RawCell &c)
{
    double r_in, th_in, r_out, th_out;
    if (s.ph_inj_switch == 0) {
        const double h0 = 0.5 * c.s0, h1 = 0.5 * c.s1, h2 = 0.5 * c.s2;
        if (s.dimensions == DIM_THREE) {
            hydro_to_spherical(s.dimensions, s.geometry, fabs(c.x0) - h0, fabs(c.x1) - h1, fabs(c.x2) - h2, r_in, th_in);
            hydro_to_spherical(s.dimensions, s.geometry, fabs(c.x0) + h0, fabs(c.x1) + h1, fabs(c.x2) + h2, r_out, th_out);
        } else {
            hydro_to_spherical(s.dimensions, s.geometry, c.x0 - h0, c.x1 - h1, 0, r_in, th_in);
            hydro_to_spherical(s.dimensions, s.geometry, c.x0 + h0, c.x1 + h1, 0, r_out, th_out);
        }
        return (s.r_lo <= r_out) && (r_in <= s.r_hi) && (th_out >= s.th_lo) && (th_in <= s.th_hi);
    }
    hydro_to_spherical(s.dimensions, s.geometry, c.x0, c.x1, (s.dimensions == DIM_THREE) ? c.x2 : 0.0, r_in, th_in);
    return r_in > s.r_inj_095;
}

// One count per chunk of IB virtual cells (the unit pass 2 is launched over); a workgroup walks COUNT_CHUNKS chunks so that
// the total costs one atomic per 2048 cells, not per 256 (same-address atomics serialise in L2 at ~10 ns each).
constexpr int COUNT_CHUNKS = 8;
template <class Source>
__global__ __launch_bounds__(IB) void ingest_count_kernel(Source src, SlabDev slab, unsigned *__restrict__ block_count,
                                                          unsigned long long *__restrict__ total)
{
    __shared__ unsigned s_w[COUNT_CHUNKS][IB / 64];
    const long long chunks = (src.count() + IB - 1) / IB;
    for (int c = 0; c < COUNT_CHUNKS; ++c) {
        const long long chunk = (long long)blockIdx.x * COUNT_CHUNKS + c;
        const long long i = chunk * IB + threadIdx.x;
        bool keep = false;
        if (chunk < chunks && i < src.count() && src.present(i)) keep = in_slab(slab, src.geom(i));
        const unsigned long long m = __ballot(keep);
        if ((threadIdx.x & 63) == 0) s_w[c][threadIdx.x >> 6] = (unsigned)__popcll(m);
    }
    __syncthreads();
    unsigned n = 0;
    if (threadIdx.x < COUNT_CHUNKS) {
        const long long chunk = (long long)blockIdx.x * COUNT_CHUNKS + threadIdx.x;
        for (int w = 0; w < IB / 64; ++w) n += s_w[threadIdx.x][w];
        if (chunk < chunks) block_count[chunk] = n;
    }
    for (int off = COUNT_CHUNKS / 2; off > 0; off >>= 1) n += __shfl_down(n, off);
    if (threadIdx.x == 0 && n) atomicAdd(total, (unsigned long long)n);
}

template <class Source, bool FLASH>
__global__ __launch_bounds__(IB) void ingest_write_kernel(Source src, SlabDev slab, const int *__restrict__ block_start, HydroCols out)
{
    __shared__ unsigned s_w[IB / 64];
    const long long i = (long long)blockIdx.x * IB + threadIdx.x;
    bool keep = false;
    RawCell c{};
    if (i < src.count() && src.present(i)) { c = src.geom(i); keep = in_slab(slab, c); }
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_w[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (!keep) return;
    unsigned pos = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    const long long j = (long long)block_start[blockIdx.x] + pos;
    double v0, v1, v2, dens, pres;
    src.fluid(i, v0, v1, v2, dens, pres);
    out.pres[j] = pres;
    out.v0[j] = v0;
    out.v1[j] = v1;
    out.v2[j] = v2;
    out.dens[j] = dens;
    out.r0[j] = c.x0; out.r1[j] = c.x1; out.r2[j] = c.x2;
    out.s0[j] = c.s0; out.s1[j] = c.s1; out.s2[j] = c.s2;
    // mclib_flash.c:362-364 / mclib_pluto.c:1362-1364: gamma from the first two velocity components only
    const double q = sqrt(1.0 - (v0 * v0 + v1 * v1));
    out.gamma[j] = 1 / q;
    out.dens_lab[j] = dens / q;
    out.temp[j] = pow(3 * pres / A_RAD, 1.0 / 4.0);
    // r and theta are overwritten by fillHydroCoordinateToSpherical right after the reader (mcrat_io.c:1962)
    if (FLASH) { out.r[j] = sqrt(c.x0 * c.x0 + c.x1 * c.x1); out.theta[j] = atan2(c.x0, c.x1); }
    else { out.r[j] = c.x0; out.theta[j] = c.x1; }
}

// geometry.c:156-174
__global__ __launch_bounds__(IB) void fill_spherical_kernel(int dims, int geom, HydroCols h, int M)
{
    const int i = blockIdx.x * IB + threadIdx.x;
    if (i >= M) return;
    double r, th;
    hydro_to_spherical(dims, geom, h.r0[i], h.r1[i], (dims == DIM_THREE) ? h.r2[i] : 0.0, r, th);
    h.r[i] = r;
    h.theta[i] = th;
}

// the velocity block the preps share for a radial flow (analytic_outflows.c:97-133 == :185-221)
__device__ __forceinline__ void radial_velocity(int dims, int geom, const HydroCols &h, int i, double vel)
{
    if (dims == DIM_TWO || dims == DIM_TWO_POINT_FIVE) {
        if (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL) {
            const double a = h.r0[i], b = h.r1[i];
            const double r = sqrt(a * a + b * b);
            h.v0[i] = (vel * a) / r;
            h.v1[i] = (vel * b) / r;
        }
        if (geom == GEOM_SPHERICAL) { h.v0[i] = vel; h.v1[i] = 0; }
        if (dims == DIM_TWO_POINT_FIVE) h.v2[i] = 0;
    } else {
        if (geom == GEOM_CARTESIAN) {
            const double a = h.r0[i], b = h.r1[i], c = h.r2[i];
            const double r = sqrt(a * a + b * b + c * c);
            h.v0[i] = (vel * a) / r;
            h.v1[i] = (vel * b) / r;
            h.v2[i] = (vel * c) / r;
        }
        if (geom == GEOM_SPHERICAL) { h.v0[i] = vel; h.v1[i] = 0; h.v2[i] = 0; }
        if (geom == GEOM_POLAR) {
            const double a = h.r0[i], c = h.r2[i];
            const double r = sqrt(a * a + c * c);
            h.v0[i] = (vel * a) / r;
            h.v1[i] = 0;
            h.v2[i] = (vel * c) / r;
        }
    }
}

// analytic_outflows.c: cylindricalPrep :3-61, sphericalPrep :63-136, structuredFireballPrep :138-236
__global__ __launch_bounds__(IB) void outflow_prep_kernel(int dims, int geom, OutflowDev o, HydroCols h, int M)
{
    const int i = blockIdx.x * IB + threadIdx.x;
    if (i >= M) return;
    if (o.simulation_type == 1) {
        const double vel = sqrt(1 - pow(o.gamma_infinity, -2.0)), lab_dens = o.gamma_infinity * o.ddensity;
        h.gamma[i] = o.gamma_infinity;
        h.dens[i] = o.ddensity;
        h.dens_lab[i] = lab_dens;
        h.pres[i] = (A_RAD * pow(o.t_comov, 4.0)) / (3);
        h.temp[i] = o.t_comov;
        if (dims == DIM_TWO || dims == DIM_TWO_POINT_FIVE) {
            if (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL) { h.v0[i] = 0; h.v1[i] = vel; }
            if (geom == GEOM_SPHERICAL) { const double th = h.r1[i]; h.v0[i] = vel * cos(th); h.v1[i] = -vel * sin(th); }
            if (dims == DIM_TWO_POINT_FIVE) h.v2[i] = 0;
        } else {
            if (geom == GEOM_CARTESIAN || geom == GEOM_POLAR) { h.v0[i] = 0; h.v1[i] = 0; h.v2[i] = vel; }
            if (geom == GEOM_SPHERICAL) { const double th = h.r1[i]; h.v0[i] = vel * cos(th); h.v1[i] = -vel * sin(th); h.v2[i] = 0; }
        }
    } else if (o.simulation_type == 2) {
        const double r = h.r[i];
        double gamma, pres;
        if (r >= (o.r00 * o.gamma_infinity)) {
            gamma = o.gamma_infinity;
            pres = (o.lumi * pow(o.r00, 2.0 / 3.0) * pow(r, -8.0 / 3.0)) / (12.0 * M_PI * C_LIGHT * pow(o.gamma_infinity, 4.0 / 3.0));
        } else {
            gamma = r / o.r00;
            pres = (o.lumi * pow(o.r00, 2.0)) / (12.0 * M_PI * C_LIGHT * pow(r, 4.0));
        }
        const double dens = o.lumi / (4 * M_PI * pow(r, 2.0) * pow(C_LIGHT, 3.0) * o.gamma_infinity * gamma);
        h.gamma[i] = gamma;
        h.pres[i] = pres;
        h.dens[i] = dens;
        h.dens_lab[i] = dens * gamma;
        h.temp[i] = pow(3 * pres / A_RAD, 1.0 / 4.0);
        radial_velocity(dims, geom, h, i, sqrt(1 - pow(gamma, -2.0)));
    } else if (o.simulation_type == 3) {
        const double gamma_0 = o.gamma_infinity;
        const double T_0 = pow(o.lumi / (4 * M_PI * o.r00 * o.r00 * A_RAD * C_LIGHT), 1.0 / 4.0);
        const double theta = h.theta[i], r = h.r[i];
        const double theta_ratio = theta / o.theta_j;
        double eta = gamma_0 / sqrt(1 + pow(theta_ratio, 2 * o.p));
        if (theta >= o.theta_j * pow(gamma_0 / 2, 1.0 / o.p)) eta = 2.0;
        const double r_sat = eta * o.r00;
        double gamma, temp;
        if (r >= r_sat) {
            gamma = eta;
            temp = T_0 * pow(r_sat / r, 2.0 / 3.0) / eta;
        } else {
            gamma = r / r_sat;
            temp = T_0;
        }
        const double vel = sqrt(1 - pow(gamma, -2.0));
        const double dens = M_P * o.lumi / (4 * M_PI * M_P * C_LIGHT * C_LIGHT * C_LIGHT * eta * vel * gamma * r * r);
        h.gamma[i] = gamma;
        h.temp[i] = temp;
        h.dens[i] = dens;
        h.dens_lab[i] = dens * gamma;
        h.pres[i] = (A_RAD * pow(temp, 4.0)) / (3);
        radial_velocity(dims, geom, h, i, vel);
    }
}

// The per-cell records the loop kernels read, from the columns (what mcrat_hip_set_hydro's host loop produced): geometry,
// the cell part of hydroVectorToCartesian (geometry.c:189-253; the device adds the photon-azimuth part, physics.hpp
// cell_beta), gamma and dens_lab, the temperature; plus what the cell-lookup grid's plan needs (extents, smallest and
// largest width per axis) and whether any cell is hot enough for the Maxwell-Juttner sampler (electron.c:208).
__global__ __launch_bounds__(IB) void stage_cells_kernel(int dims, int geom, HydroCols h, int M, CellGeom *__restrict__ og, CellGeom2 *__restrict__ og2,
                                                         CellFluid *__restrict__ of, double *__restrict__ ofc, double *__restrict__ otemp,
                                                         double *__restrict__ ogamma, StagePartial *__restrict__ partials)
{
    __shared__ StagePartial s_p[IB / 64];
    StagePartial p;
    for (int k = 0; k < 3; ++k) { p.lo[k] = INFINITY; p.hi[k] = -INFINITY; p.smin[k] = INFINITY; p.smax[k] = 0; }
    p.any_hot = 0; p.pad = 0;
    const bool three = dims == DIM_THREE, two = dims == DIM_TWO;
    for (int i = blockIdx.x * IB + threadIdx.x; i < M; i += gridDim.x * IB) {
        const double c0 = h.r0[i], c1 = h.r1[i], s0 = h.s0[i], s1 = h.s1[i];
        const double c2 = three ? h.r2[i] : 0.0, s2 = three ? h.s2[i] : 0.0;
        CellGeom g; g.c0 = c0; g.c1 = c1; g.s0 = s0; g.s1 = s1;
        og[i] = g;
        if (three) { CellGeom2 g2; g2.c2 = c2; g2.s2 = s2; og2[i] = g2; }
        const double v0 = h.v0[i], v1 = h.v1[i], v2 = two ? 0.0 : h.v2[i];
        ogamma[i] = h.gamma[i];
        double fa, fb, fc = 0;
        if (!three) {
            if (geom == GEOM_SPHERICAL) {
                fa = v0 * sin(c1) + v1 * cos(c1);
                fb = v0 * cos(c1) - v1 * sin(c1);
            } else {
                fa = v0; fb = v1;
            }
            fc = v2;
        } else if (geom == GEOM_CARTESIAN) {
            fa = v0; fb = v1; fc = v2;
        } else if (geom == GEOM_SPHERICAL) {
            fa = v0 * sin(c1) * cos(c2) + v1 * cos(c1) * cos(c2) - v2 * sin(c2);
            fb = v0 * sin(c1) * sin(c2) + v1 * cos(c1) * sin(c2) + v2 * cos(c2);
            fc = v0 * cos(c1) - v1 * sin(c1);
        } else {   // POLAR
            fa = v0 * cos(c1) - v1 * sin(c1);
            fb = v0 * sin(c1) + v1 * cos(c1);
            fc = v2;
        }
        CellFluid f;
        cell_staged_operands(fa, fb, fc, h.gamma[i], h.dens_lab[i], f);
        of[i] = f;
        if (ofc) ofc[i] = fc;
        const double t = h.temp[i];
        otemp[i] = t;
        if (t >= 1e7) p.any_hot = 1;
        const double cc[3] = {c0, c1, c2}, ss[3] = {s0, s1, s2};
        for (int k = 0; k < (three ? 3 : 2); ++k) {
            p.lo[k] = fmin(p.lo[k], cc[k] - 0.5 * ss[k]);
            p.hi[k] = fmax(p.hi[k], cc[k] + 0.5 * ss[k]);
            // NaN or non-positive widths must reach the host's check: fmin would drop a NaN
            p.smin[k] = (ss[k] < p.smin[k] || !(ss[k] == ss[k])) ? ss[k] : p.smin[k];
            p.smax[k] = fmax(p.smax[k], ss[k]);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int k = 0; k < 3; ++k) {
            p.lo[k] = fmin(p.lo[k], __shfl_xor(p.lo[k], off));
            p.hi[k] = fmax(p.hi[k], __shfl_xor(p.hi[k], off));
            const double o = __shfl_xor(p.smin[k], off);
            p.smin[k] = (o < p.smin[k] || !(o == o)) ? o : p.smin[k];
            p.smax[k] = fmax(p.smax[k], __shfl_xor(p.smax[k], off));
        }
        p.any_hot |= __shfl_xor(p.any_hot, off);
    }
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = p;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < IB / 64; ++w) {
            const StagePartial &q = s_p[w];
            for (int k = 0; k < 3; ++k) {
                p.lo[k] = fmin(p.lo[k], q.lo[k]);
                p.hi[k] = fmax(p.hi[k], q.hi[k]);
                p.smin[k] = (q.smin[k] < p.smin[k] || !(q.smin[k] == q.smin[k])) ? q.smin[k] : p.smin[k];
                p.smax[k] = fmax(p.smax[k], q.smax[k]);
            }
            p.any_hot |= q.any_hot;
        }
        partials[blockIdx.x] = p;
    }
}

// every `stride`-th cell's centre and width per axis, for the plan's typical cell width (engine.hip, grid_plan)
__global__ __launch_bounds__(IB) void sample_cells_kernel(HydroCols h, int M, int stride, int nsamp, int naxes, double *__restrict__ out)
{
    const int k = blockIdx.x * IB + threadIdx.x;
    if (k >= nsamp) return;
    const long long i = (long long)k * stride;
    if (i >= M) return;
    const double *c[3] = {h.r0, h.r1, h.r2}, *s[3] = {h.s0, h.s1, h.s2};
    for (int a = 0; a < naxes; ++a) {
        out[(2 * a) * (size_t)nsamp + k] = c[a][i];
        out[(2 * a + 1) * (size_t)nsamp + k] = s[a][i];
    }
}

template <class Source>
hipError_t count_impl(const Source &src, long long n_virtual, const SlabDev &slab, unsigned *block_count, unsigned long long *d_total, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const long long blocks = (ingest_blocks(n_virtual) + COUNT_CHUNKS - 1) / COUNT_CHUNKS;
    hipLaunchKernelGGL(ingest_count_kernel<Source>, dim3((unsigned)blocks), dim3(IB), 0, stream, src, slab, block_count, d_total);
    return hipGetLastError();
}

FlashSource flash_source(const FlashDev &f) { FlashSource s; s.f = f; return s; }

PlutoSource pluto_source(const PlutoDev &g, int dims, int geom)
{
    PlutoSource s;
    s.g = g;
    s.three = dims == DIM_THREE;
    s.v3 = dims != DIM_TWO;
    s.scale1 = (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL);
    s.scale2 = (geom == GEOM_CARTESIAN || geom == GEOM_POLAR);
    if (!s.three) s.g.nz = 1;
    return s;
}

ChomboSource chombo_source(const ChomboDev &h, int dims, int geom)
{
    ChomboSource s;
    s.h = h;
    s.three = dims == DIM_THREE;
    s.v3 = dims != DIM_TWO;
    s.scale1 = (geom == GEOM_CARTESIAN || geom == GEOM_CYLINDRICAL);
    s.scale2 = (geom == GEOM_CARTESIAN || geom == GEOM_POLAR);
    return s;
}

}  // namespace

hipError_t ingest_count_chombo(const ChomboDev &h, const SlabDev &slab, unsigned *block_count, unsigned long long *d_total, hipStream_t stream)
{
    return count_impl(chombo_source(h, slab.dimensions, slab.geometry), h.cells, slab, block_count, d_total, stream);
}

hipError_t ingest_write_chombo(const ChomboDev &h, const SlabDev &slab, const int *block_start, const HydroCols &out, hipStream_t stream)
{
    const long long blocks = ingest_blocks(h.cells);
    hipLaunchKernelGGL((ingest_write_kernel<ChomboSource, false>), dim3((unsigned)blocks), dim3(IB), 0, stream,
                       chombo_source(h, slab.dimensions, slab.geometry), slab, block_start, out);
    return hipGetLastError();
}

hipError_t launch_chombo_mask(const ChomboBox *boxes, int c0, int c1, int f0, int f1, int ref_ratio, int three, unsigned char *covered, hipStream_t stream)
{
    const long long pairs = (long long)(c1 - c0) * (f1 - f0);
    if (pairs <= 0) return hipSuccess;
    long long blocks = (pairs + IB - 1) / IB;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(chombo_mask_kernel, dim3((unsigned)blocks), dim3(IB), 0, stream, boxes, c0, c1 - c0, f0, f1 - f0, ref_ratio, three, covered);
    return hipGetLastError();
}

long long ingest_blocks(long long n_virtual) { return (n_virtual + IB - 1) / IB; }

hipError_t ingest_count_flash(const FlashDev &f, const SlabDev &slab, unsigned *block_count, unsigned long long *d_total, hipStream_t stream)
{
    return count_impl(flash_source(f), f.n_blocks * 64, slab, block_count, d_total, stream);
}

hipError_t ingest_write_flash(const FlashDev &f, const SlabDev &slab, const int *block_start, const HydroCols &out, hipStream_t stream)
{
    const long long blocks = ingest_blocks(f.n_blocks * 64);
    hipLaunchKernelGGL((ingest_write_kernel<FlashSource, true>), dim3((unsigned)blocks), dim3(IB), 0, stream, flash_source(f), slab, block_start, out);
    return hipGetLastError();
}

hipError_t ingest_count_pluto(const PlutoDev &g, const SlabDev &slab, unsigned *block_count, unsigned long long *d_total, hipStream_t stream)
{
    const PlutoSource s = pluto_source(g, slab.dimensions, slab.geometry);
    return count_impl(s, (long long)s.g.nx * s.g.ny * s.g.nz, slab, block_count, d_total, stream);
}

hipError_t ingest_write_pluto(const PlutoDev &g, const SlabDev &slab, const int *block_start, const HydroCols &out, hipStream_t stream)
{
    const PlutoSource s = pluto_source(g, slab.dimensions, slab.geometry);
    const long long blocks = ingest_blocks((long long)s.g.nx * s.g.ny * s.g.nz);
    hipLaunchKernelGGL((ingest_write_kernel<PlutoSource, false>), dim3((unsigned)blocks), dim3(IB), 0, stream, s, slab, block_start, out);
    return hipGetLastError();
}

hipError_t launch_fill_spherical(int dims, int geom, const HydroCols &h, int M, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_spherical_kernel, dim3((M + IB - 1) / IB), dim3(IB), 0, stream, dims, geom, h, M);
    return hipGetLastError();
}

hipError_t launch_outflow_prep(int dims, int geom, const OutflowDev &o, const HydroCols &h, int M, hipStream_t stream)
{
    if (o.simulation_type < 1 || o.simulation_type > 3) return hipSuccess;
    hipLaunchKernelGGL(outflow_prep_kernel, dim3((M + IB - 1) / IB), dim3(IB), 0, stream, dims, geom, o, h, M);
    return hipGetLastError();
}

int stage_cells_blocks(int M) { const int b = (M + IB - 1) / IB; return b < 1 ? 1 : (b > 1024 ? 1024 : b); }

hipError_t launch_stage_cells(int dims, int geom, const HydroCols &h, int M, CellGeom *og, CellGeom2 *og2, CellFluid *of, double *ofc, double *otemp,
                              double *ogamma, StagePartial *partials, double *samples, int stride, int nsamp, hipStream_t stream)
{
    hipLaunchKernelGGL(stage_cells_kernel, dim3(stage_cells_blocks(M)), dim3(IB), 0, stream, dims, geom, h, M, og, og2, of, ofc, otemp, ogamma, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sample_cells_kernel, dim3((nsamp + IB - 1) / IB), dim3(IB), 0, stream, h, M, stride, nsamp, (dims == DIM_THREE) ? 3 : 2, samples);
    return hipGetLastError();
}

}  // namespace mcrat
