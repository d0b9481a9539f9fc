// grid_build.hip -- the cell-lookup grid of a hydro frame, built on the device ("build the device
// cell-lookup structure at load").  Same structure as the host build in engine.hip (kept as the cross-check,
// MCRAT_HIP_HOST_GRID=1): every cell is entered into all buckets its closed extent, widened by 1e-9 relative,
// touches; every bucket list is ascending in cell index (so the first hit of the device's closed-interval test is
// the lowest-index containing cell, what the linear scan of geometry.c:350-391 returns); every entry is a complete
// copy of the cell's records (FatCell); every bucket octant names the one entry whose cell reaches into it, if there
// is exactly one (BucketDir).  Counting sort: count -> exclusive scan -> fill (atomic cursor) -> per-bucket sort +
// records + hints.  A 1 048 576-cell frame (2.6 M buckets, 6.8 M entries, 0.65 GB) takes a few milliseconds here
// against 0.3 s on one host core plus the PCIe copy of the result.
#include <hip/hip_runtime.h>
#include <math.h>
#include "device_types.hpp"
#include "launch.hpp"

namespace mcrat {

namespace {

__device__ __forceinline__ int bucket_of(double x, int logmap, double org, double inv, int dim)
{
    const double u = logmap ? log(x) : x;
    const double f = floor((u - org) * inv);
    if (!(f == f)) return 0;
    if (f < 0.0) return 0;
    if (f > (double)(dim - 1)) return dim - 1;
    return (int)f;
}

struct CellBox {
    double c[3], s[3];
};

__device__ __forceinline__ CellBox load_box(const CellGeom *geom, const CellGeom2 *geom2, int i, int naxes)
{
    const CellGeom g = geom[i];
    CellBox b;
    b.c[0] = g.c0; b.c[1] = g.c1; b.s[0] = g.s0; b.s[1] = g.s1;
    b.c[2] = 0; b.s[2] = 0;
    if (naxes == 3) { const CellGeom2 g2 = geom2[i]; b.c[2] = g2.c2; b.s[2] = g2.s2; }
    return b;
}

// the buckets a cell's widened extent touches, per axis
__device__ __forceinline__ void bucket_range(const GridPlan &p, const CellBox &b, int lo[3], int hi[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = 0; hi[k] = 0;
        if (k < p.naxes) {
            const double m = 1e-9 * (fabs(b.c[k]) + b.s[k]);
            double a = b.c[k] - 0.5 * b.s[k] - m;
            const double e = b.c[k] + 0.5 * b.s[k] + m;
            if (p.logmap[k] && a <= 0) a = 1e-300;
            lo[k] = bucket_of(a, p.logmap[k], p.org[k], p.inv[k], p.dim[k]);
            hi[k] = bucket_of(e, p.logmap[k], p.org[k], p.inv[k], p.dim[k]);
        }
    }
}

__global__ __launch_bounds__(256) void grid_count_kernel(GridPlan p, const CellGeom *__restrict__ geom, const CellGeom2 *__restrict__ geom2, int M,
                                                         unsigned *__restrict__ count, unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long s_sum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long mine = 0;
    if (i < M) {
        const CellBox b = load_box(geom, geom2, i, p.naxes);
        int lo[3], hi[3];
        bucket_range(p, b, lo, hi);
        mine = (unsigned long long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
        if (mine > 4096) mine = 1ull << 40;       // a cell spanning more buckets than that makes the plan too fine: the host sees the total and coarsens
        else
            for (int z = lo[2]; z <= hi[2]; ++z)
                for (int y = lo[1]; y <= hi[1]; ++y)
                    for (int x = lo[0]; x <= hi[0]; ++x) atomicAdd(&count[((size_t)z * p.dim[1] + y) * p.dim[0] + x], 1u);
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}

// exclusive scan of count[0..n) into start[0..n], three small kernels (2048 elements per workgroup)
constexpr int SCAN_PER_BLOCK = 2048;

__global__ __launch_bounds__(256) void scan_blocks_kernel(const unsigned *__restrict__ count, long long n, int *__restrict__ start, int *__restrict__ sums)
{
    __shared__ int s_w[4];
    const long long base = (long long)blockIdx.x * SCAN_PER_BLOCK + (long long)threadIdx.x * 8;
    int v[8], acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long e = base + k;
        const int c = (e < n) ? (int)count[e] : 0;
        v[k] = acc;
        acc += c;
    }
    // exclusive scan of the per-thread totals over the workgroup
    int incl = acc;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int wbase = 0;
    for (int k = 0; k < w; ++k) wbase += s_w[k];
    const int excl = wbase + incl - acc;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long e = base + k;
        if (e < n) start[e] = excl + v[k];
    }
    if (threadIdx.x == 255) sums[blockIdx.x] = wbase + incl;
}

__global__ __launch_bounds__(256) void scan_sums_kernel(int *__restrict__ sums, int nblocks)
{
    __shared__ int s_w[4];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 256) {
        const int e = b0 + threadIdx.x;
        const int c = (e < nblocks) ? sums[e] : 0;
        int incl = c;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off, 64);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_w[w] = incl;
        __syncthreads();
        int wbase = s_carry;
        for (int k = 0; k < w; ++k) wbase += s_w[k];
        if (e < nblocks) sums[e] = wbase + incl - c;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = wbase + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scan_add_kernel(int *__restrict__ start, long long n, const int *__restrict__ sums, int total)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < n) start[e] += sums[e / SCAN_PER_BLOCK];
    if (e == n) start[e] = total;
}

__global__ __launch_bounds__(256) void grid_fill_kernel(GridPlan p, const CellGeom *__restrict__ geom, const CellGeom2 *__restrict__ geom2, int M,
                                                        const int *__restrict__ start, unsigned *__restrict__ cursor, int *__restrict__ entries)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const CellBox b = load_box(geom, geom2, i, p.naxes);
    int lo[3], hi[3];
    bucket_range(p, b, lo, hi);
    for (int z = lo[2]; z <= hi[2]; ++z)
        for (int y = lo[1]; y <= hi[1]; ++y)
            for (int x = lo[0]; x <= hi[0]; ++x) {
                const size_t bk = ((size_t)z * p.dim[1] + y) * p.dim[0] + x;
                const unsigned pos = atomicAdd(&cursor[bk], 1u);
                entries[start[bk] + (int)pos] = i;
            }
}

__device__ __forceinline__ double mapped(double x, int logmap) { return logmap ? log(fmax(x, 1e-300)) : x; }

// one thread per list entry: the entry's complete cell record (device_types.hpp, FatCell).  Neighbouring threads write
// neighbouring 128-B records, so the ~4 M x 128 B of a 10^6-cell frame stream out coalesced (one thread per bucket writing
// its whole list took 650 us per frame, this 130).
__global__ __launch_bounds__(256) void grid_records_kernel(int naxes, long long total, const int *__restrict__ entries,
                                                           const CellGeom *__restrict__ geom, const CellGeom2 *__restrict__ geom2,
                                                           const CellFluid *__restrict__ fluid, const double *__restrict__ fluid_c,
                                                           FatCell *__restrict__ cells)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int ci = entries[e];
    const CellGeom g = geom[ci];
    const CellFluid f = fluid[ci];
    FatCell fc;
    fc.c0 = g.c0; fc.c1 = g.c1; fc.s0 = g.s0; fc.s1 = g.s1;
    fc.a = f.a; fc.b = f.b; fc.c = f.c; fc.w = f.w;
    fc.nsig = f.nsig; fc.gam = f.gam;
    fc.c2 = 0; fc.s2 = 0;
    if (naxes == 3) { const CellGeom2 g2 = geom2[ci]; fc.c2 = g2.c2; fc.s2 = g2.s2; }
    fc.cell = ci; fc.pad = 0; fc.pad2[0] = fc.pad2[1] = fc.pad2[2] = 0.0;
    cells[e] = fc;
}

// one thread per bucket: order its entries by cell index and find the octant hints (from the cells' geometry records)
__global__ __launch_bounds__(256) void grid_finish_kernel(GridPlan p, long long nb, const int *__restrict__ start, int *__restrict__ entries,
                                                          const CellGeom *__restrict__ geom, const CellGeom2 *__restrict__ geom2,
                                                          BucketDir *__restrict__ dir)
{
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const int e0 = start[b], n = start[b + 1] - e0;
    for (int a = 1; a < n; ++a) {                      // insertion sort: the lists hold a handful of entries
        const int key = entries[e0 + a];
        int j = a - 1;
        while (j >= 0 && entries[e0 + j] > key) { entries[e0 + j + 1] = entries[e0 + j]; --j; }
        entries[e0 + j + 1] = key;
    }
    int bi[3];
    bi[0] = (int)(b % p.dim[0]);
    bi[1] = (int)((b / p.dim[0]) % p.dim[1]);
    bi[2] = (int)(b / ((long long)p.dim[0] * p.dim[1]));
    const int nocts = 1 << p.naxes;
    // octant o is the half-bucket [olo, olo + w/2) per axis; a hint names the one entry that reaches into it.  Every entry's
    // mapped extent is computed once and tested against the 2^naxes octants.
    double olo[3][2] = {{0, 0}, {0, 0}, {0, 0}}, half[3] = {0, 0, 0};
    for (int k = 0; k < p.naxes; ++k) {
        const double w = 1.0 / p.inv[k];
        olo[k][0] = p.org[k] + (bi[k] + 0.0) * w;
        olo[k][1] = p.org[k] + (bi[k] + 0.5) * w;
        half[k] = 0.5 * w;
    }
    unsigned char cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, first[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < n; ++e) {
        const int ci = entries[e0 + e];
        const CellGeom g = geom[ci];
        double c2 = 0, s2 = 0;
        if (p.naxes == 3) { const CellGeom2 g2 = geom2[ci]; c2 = g2.c2; s2 = g2.s2; }
        const double cc[3] = {g.c0, g.c1, c2}, ss[3] = {g.s0, g.s1, s2};
        bool r[3][2] = {{true, true}, {true, true}, {true, true}};
        for (int k = 0; k < p.naxes; ++k) {
            const double m = 1e-9 * (fabs(cc[k]) + ss[k]);
            const double clo = mapped(cc[k] - 0.5 * ss[k] + m, p.logmap[k]), chi = mapped(cc[k] + 0.5 * ss[k] - m, p.logmap[k]);
            for (int h = 0; h < 2; ++h) r[k][h] = (clo < olo[k][h] + half[k]) && (chi > olo[k][h]);
        }
        for (int o = 0; o < nocts; ++o) {
            const bool reaches = r[0][o & 1] && r[1][(o >> 1) & 1] && r[2][(o >> 2) & 1];
            if (reaches) {
                if (cnt[o] == 0) first[o] = (unsigned char)(e < (int)GRID_NO_HINT ? e : GRID_NO_HINT);
                if (cnt[o] < 2) cnt[o] += 1;
            }
        }
    }
    unsigned hints = 0;
    for (int o = 0; o < 8; ++o) {
        const unsigned pick = (o < nocts && cnt[o] == 1) ? (unsigned)first[o] : GRID_NO_HINT;
        hints |= pick << (4 * o);
    }
    BucketDir d;
    d.e0 = e0; d.n = n; d.hints = hints; d.pad = 0;
    dir[b] = d;
}

}  // namespace

hipError_t grid_count(const GridPlan &p, const CellGeom *geom, const CellGeom2 *geom2, int M, unsigned *count, long long nb,
                      unsigned long long *d_total, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned) * (size_t)nb, stream);
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(d_total, 0, sizeof(unsigned long long), stream)) != hipSuccess) return e;
    grid_count_kernel<<<dim3((M + 255) / 256), dim3(256), 0, stream>>>(p, geom, geom2, M, count, d_total);
    return hipGetLastError();
}

size_t grid_scan_scratch_ints(long long nb) { return (size_t)((nb + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK) + 1; }

// start[0..n] = exclusive prefix sums of count[0..n), start[n] = total; scratch holds grid_scan_scratch_ints(n) ints
hipError_t launch_exclusive_scan(const unsigned *count, long long n, int *start, int *scratch, long long total, hipStream_t stream)
{
    const int nblocks = (int)((n + SCAN_PER_BLOCK - 1) / SCAN_PER_BLOCK);
    scan_blocks_kernel<<<dim3(nblocks), dim3(256), 0, stream>>>(count, n, start, scratch);
    scan_sums_kernel<<<dim3(1), dim3(256), 0, stream>>>(scratch, nblocks);
    scan_add_kernel<<<dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream>>>(start, n, scratch, (int)total);
    return hipGetLastError();
}

hipError_t grid_build(const GridPlan &p, const CellGeom *geom, const CellGeom2 *geom2, const CellFluid *fluid, const double *fluid_c, int M,
                      unsigned *count, int *start, int *scan_scratch, int *entries, FatCell *cells, BucketDir *dir, long long nb,
                      long long total, hipStream_t stream)
{
    hipError_t e = launch_exclusive_scan(count, nb, start, scan_scratch, total, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(count, 0, sizeof(unsigned) * (size_t)nb, stream);      // now the fill cursors
    if (e != hipSuccess) return e;
    grid_fill_kernel<<<dim3((M + 255) / 256), dim3(256), 0, stream>>>(p, geom, geom2, M, start, count, entries);
    grid_finish_kernel<<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, stream>>>(p, nb, start, entries, geom, geom2, dir);
    if (total > 0)
        grid_records_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(p.naxes, total, entries, geom, geom2, fluid, fluid_c, cells);
    return hipGetLastError();
}

}  // namespace mcrat
