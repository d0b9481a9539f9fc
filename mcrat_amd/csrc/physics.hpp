// physics.hpp -- per-photon physics of the photon loop as HIP device functions (IEEE double,
// compiled with -ffp-contract=off so that sums and products round like the reference's C).
// Each function names the reference lines (under /root/reference/Src) whose arithmetic it follows.
//
// Angles the reference only ever feeds back into sin/cos are never materialised: cos(atan2(y,x)) is
// x/hypot, cos(acos(c)) is c, sin(acos(c)) is sqrt(1-c^2), cos/sin(2 acos(d)) are 2d^2-1 / 2d sqrt(1-d^2).
// These are the same real numbers as the reference's expressions (they differ in the last ulp, inside the
// stated parity tolerance) and take the ~25 f64 libm calls per scattering event, which are pure serial
// latency on the one lane that runs the event, down to two sincos.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "rng.hpp"

namespace mcrat {
namespace phys {

constexpr int REJECTION_CAP = 1 << 22;   // every rejection loop is bounded so that a wave always finishes

// ---------------------------------------------------------------- reciprocal and inverse square root
// The loop's arithmetic is a chain of dependent f64 operations on a wave that has one or two neighbours on its SIMD, so what a
// re-location or a scattering costs is the LENGTH of that chain.  An IEEE division is v_div_scale + v_rcp + five FMAs + v_div_fmas +
// v_div_fixup (eleven dependent instructions), a square root fourteen; the hardware's v_rcp_f64 / v_rsq_f64 with two Newton steps
// give 1/x and 1/sqrt(x) to one or two ulp in five and seven.  Round 3 writes the loop's divisions and roots with these wherever the
// result only feeds further arithmetic (never where it decides a cell or is compared on a face: hydro_coords keeps the IEEE sqrt),
// and folds the divisions the reference repeats (p/|p| thrice in zeroNorm, six by beta^2 in lorentzBoost) into one reciprocal.
// The values are the reference's real numbers rounded differently in the last place or two; the gates are the oracle's
// (integers exact, doubles 1e-9 over trajectories, tests/).
// -DMCRAT_IEEE_ARITH=1 (a variant build, tools/variant_build.py ieee -DMCRAT_IEEE_ARITH=1; not the product): the three helpers below
// become the IEEE division and square root the reference's C compiles to (the logarithm of the free-path draws is the math library's in every build) -- for a maintainer who compares photon for
// photon with MCRaT on a recorded tape (tools/ref_harness) and wants every decision taken on correctly rounded operands.  What stays in either build
// is algebra, not approximation: the boost as p + (kf (b.p) - g p0) b, the optical depth on per-cell operands -- the same real numbers as the
// reference's expressions, rounded in a different order (a few ulp).  tests/test_gpu_tape.py passes on both builds; the product's gate is the oracle's
// 1e-9 over trajectories (north_star: 1e-5), and a comparison decided within a few ulp -- a Klein-Nishina acceptance, two free times, a point on a
// cell face -- can fall the other way than MCRaT's: that photon's trajectory then differs from there on while the statistics do not.
#if defined(MCRAT_IEEE_ARITH) && MCRAT_IEEE_ARITH
__device__ __forceinline__ double rcp_nr(double x) { return 1.0 / x; }
__device__ __forceinline__ double rsqrt_nr(double x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ double sqrt_nr(double x) { return sqrt(x); }
#else
__device__ __forceinline__ double rcp_nr(double x)
{
    const double r0 = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r0, 1.0);
    double r = fma(r0, e, r0);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return (fabs(r0) < INFINITY && r0 != 0.0) ? r : r0;     // 1/0, 1/inf, NaN: the hardware's answer (+-inf, +-0, NaN)
}
__device__ __forceinline__ double rsqrt_nr(double x)
{
    const double y0 = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = fma(-(h * y0), y0, 0.5);
    double y = fma(y0, e, y0);
    e = fma(-(h * y), y, 0.5);
    y = fma(y, e, y);
    return (y0 < INFINITY && y0 > 0.0) ? y : y0;             // x = 0 -> inf, x = inf -> 0, x < 0 or NaN -> NaN
}
// sqrt(x) for x >= 0 as x * rsqrt(x) (0 for x == 0; NaN for negative x, like sqrt)
__device__ __forceinline__ double sqrt_nr(double x)
{
    const double y = x * rsqrt_nr(x);
    return (x == 0.0) ? 0.0 : y;
}
#endif


// ---------------------------------------------------------------- geometry
// geometry.c:15-64
template <int DIMS, int GEOM>
__device__ __forceinline__ void hydro_coords(double x, double y, double z, double &a0, double &a1, double &a2)
{
    a0 = -1; a1 = -1; a2 = -1;
    if constexpr (DIMS == DIM_TWO || DIMS == DIM_TWO_POINT_FIVE) {
        if constexpr (GEOM == GEOM_CARTESIAN || GEOM == GEOM_CYLINDRICAL) {
            a0 = sqrt(x * x + y * y);
            a1 = z;
        } else if constexpr (GEOM == GEOM_SPHERICAL) {
            a0 = sqrt(x * x + y * y + z * z);
            a1 = acos(z / a0);
        }
    } else {
        if constexpr (GEOM == GEOM_CARTESIAN) {
            a0 = x; a1 = y; a2 = z;
        } else if constexpr (GEOM == GEOM_SPHERICAL) {
            a0 = sqrt(x * x + y * y + z * z);
            a1 = acos(z / a0);
            a2 = fmod(atan2(y, x) * 180.0 / M_PI + 360.0, 360.0) * M_PI / 180;
        } else if constexpr (GEOM == GEOM_POLAR) {
            a0 = sqrt(x * x + y * y);
            a1 = fmod(atan2(y, x) * 180.0 / M_PI + 360.0, 360.0) * M_PI / 180;
            a2 = z;
        }
    }
}

// strict domain test of mclib.c:492-504
template <int DIMS>
__device__ __forceinline__ bool in_domain(const HydroDev &h, double a0, double a1, double a2)
{
    bool in = (a1 < h.dom1[1]) & (a1 > h.dom1[0]) & (a0 < h.dom0[1]) & (a0 > h.dom0[0]);      // (`&`: no branch between the comparisons)
    if constexpr (DIMS == DIM_THREE) in = (a2 < h.dom2[1]) & (a2 > h.dom2[0]) & in;
    return in;
}

// geometry.c:394-417, closed intervals
template <int DIMS>
__device__ __forceinline__ bool check_in_block(const HydroDev &h, int cell, double a0, double a1, double a2)
{
    const CellGeom g = h.geom[cell];
    bool in = (2 * fabs(a0 - g.c0) - g.s0 <= 0) & (2 * fabs(a1 - g.c1) - g.s1 <= 0);
    if constexpr (DIMS == DIM_THREE) {
        const CellGeom2 g2 = h.geom2[cell];
        in = in & (2 * fabs(a2 - g2.c2) - g2.s2 <= 0);
    }
    return in;
}

// bucket of the cell-lookup grid that holds a point (engine.hip, build_grid), as a code: bucket index, the octant
// of the bucket the point lies in, and whether it lies far enough (1e-6 bucket widths) from the octant's faces for
// the bucket's hint to be used (device_types.hpp, BucketDir).  -1: the coordinates are NaN.
// Written without a divergent branch (round 4): the tests of an axis are combined with `&`, the clamp is a max / min, a NaN is remembered
// and turns the code into -1 at the end -- as a chain of `&&`, ternaries and an early return this function compiled to a dozen
// exec-mask regions per axis, 190 instructions per point of which 14 were arithmetic (tools/isa_blocks.py on the fused pass of
// rank_loop_kernel).  The logarithmic axes are a wave-uniform branch.  Same values as before, point by point.
template <int NAXES>
__device__ __forceinline__ int grid_bucket_axes(const GridDev &g, double a0, double a1, double a2)
{
    const double a[3] = {a0, a1, a2};
    int b[3] = {0, 0, 0};
    int oct = 0;
    bool ok = true, nan = false;
#pragma unroll
    for (int k = 0; k < NAXES; ++k) {
        double u = a[k];
        if (g.logmap[k]) u = log(a[k]);
        const double x = (u - g.org[k]) * g.inv[k];
        const double f = floor(x);
        const double fr = x - f;
        const double hi = (double)(g.dim[k] - 1);
        nan = nan | !(f == f);
        ok = ok & (f >= 0.0) & (f <= hi) & (fr > 1e-6) & (fabs(fr - 0.5) > 1e-6) & (1.0 - fr > 1e-6);
        oct |= (fr >= 0.5 ? 1 : 0) << k;
        b[k] = (int)fmin(fmax(f, 0.0), hi);                  // (f NaN: 0, and the code is -1 anyway)
    }
    const int code = ((b[2] * g.dim[1] + b[1]) * g.dim[0] + b[0]) | (oct << GRID_CODE_OCT_SHIFT) | (ok ? GRID_CODE_HINT_OK : 0);
    return nan ? -1 : code;
}
__device__ __forceinline__ int grid_bucket(const GridDev &g, double a0, double a1, double a2)
{
    return g.naxes == 3 ? grid_bucket_axes<3>(g, a0, a1, a2) : grid_bucket_axes<2>(g, a0, a1, a2);
}
// ... for callers that know DIMENSIONS at compile time (the grid has three axes in 3-D, two otherwise: engine.hip, stage_hydro)
template <int DIMS>
__device__ __forceinline__ int grid_bucket_of(const GridDev &g, double a0, double a1, double a2)
{
    return grid_bucket_axes<DIMS == DIM_THREE ? 3 : 2>(g, a0, a1, a2);
}

template <int DIMS>
__device__ __forceinline__ bool in_fat_cell(const FatCell &f, double a0, double a1, double a2)
{
    bool in = (2 * fabs(a0 - f.c0) - f.s0 <= 0) && (2 * fabs(a1 - f.c1) - f.s1 <= 0);   // geometry.c:394-417
    if constexpr (DIMS == DIM_THREE) in = in && (2 * fabs(a2 - f.c2) - f.s2 <= 0);
    return in;
}

// inside by a margin of 1e-8 of the cell's size on every axis: then no other cell of a (non-overlapping) mesh holds
// the point, not even through the closed-interval rounding at shared faces
template <int DIMS>
__device__ __forceinline__ bool well_in_fat_cell(const FatCell &f, double a0, double a1, double a2)
{
    bool in = (2 * fabs(a0 - f.c0) - f.s0 < -1e-8 * f.s0) & (2 * fabs(a1 - f.c1) - f.s1 < -1e-8 * f.s1);
    if constexpr (DIMS == DIM_THREE) in = in & (2 * fabs(a2 - f.c2) - f.s2 < -1e-8 * f.s2);
    return in;
}

// the exact walk of a bucket list: lowest-index entry whose closed extent holds the point.  Only the extents are gathered for
// the tests (32 B per entry, 48 B in 3-D), four entries at a time, and the rest of the one entry that holds the point afterwards
// (fewer registers -- step_kernel 168 -> 132 VGPRs -- and a quarter of the gathers on a hint miss; the benchmark frames, where the
// hint nearly always hits, run as before).
template <int DIMS>
__device__ __forceinline__ int walk_bucket(const GridDev &g, int e0, int n, double a0, double a1, double a2, FatCell &hit)
{
    constexpr int BATCH = 4;
    int found = -1;
    for (int b0 = 0; b0 < n && found < 0; b0 += BATCH) {
        bool in[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const FatCell *p = g.cells + e0 + ((b0 + k < n) ? b0 + k : b0);
            const double c0 = p->c0, c1 = p->c1, s0 = p->s0, s1 = p->s1;
            in[k] = (b0 + k < n) && (2 * fabs(a0 - c0) - s0 <= 0) && (2 * fabs(a1 - c1) - s1 <= 0);   // geometry.c:394-417
            if constexpr (DIMS == DIM_THREE) {
                const double c2 = p->c2, s2 = p->s2;
                in[k] = in[k] && (2 * fabs(a2 - c2) - s2 <= 0);
            }
        }
#pragma unroll
        for (int k = BATCH - 1; k >= 0; --k)
            if (in[k]) found = b0 + k;
    }
    if (found < 0) return -1;
    hit = g.cells[e0 + found];
    return hit.cell;
}

// findContainingBlock, geometry.c:350-391: lowest-index cell whose closed extent holds the point, or -1.
// The bucket lists are ascending in cell index and hold every cell whose (slightly widened) extent touches
// the bucket, so the first hit of the list walk equals the reference's linear first match; the hinted entry
// (BucketDir) short-cuts the walk only where it provably gives the same answer.  Two dependent loads: the
// bucket's record, then one entry -- each entry a complete copy of the cell's records.  `hit` receives the entry.
template <int DIMS>
__device__ __forceinline__ int find_in_bucket(const GridDev &g, int code, double a0, double a1, double a2, FatCell &hit)
{
    hit.cell = -1;
    if (code < 0) return -1;
    const BucketDir d = g.dir[code & GRID_CODE_BUCKET_MASK];
    if (d.n <= 0) return -1;
    const unsigned hint = (d.hints >> (4 * ((code >> GRID_CODE_OCT_SHIFT) & 7))) & 15u;
    if ((code & GRID_CODE_HINT_OK) && hint != GRID_NO_HINT) {
        const FatCell f = g.cells[d.e0 + (int)hint];
        if (well_in_fat_cell<DIMS>(f, a0, a1, a2)) { hit = f; return f.cell; }
    }
    return walk_bucket<DIMS>(g, d.e0, d.n, a0, a1, a2, hit);
}

template <int DIMS>
__device__ __forceinline__ int find_containing_block(const HydroDev &h, double a0, double a1, double a2)
{
    FatCell hit;
    return find_in_bucket<DIMS>(h.grid, grid_bucket_of<DIMS>(h.grid, a0, a1, a2), a0, a1, a2, hit);
}

// geometry.c:189-253 on a staged record (see cell_beta below)
template <int DIMS>
__device__ __forceinline__ void beta_from_record(double a, double b, double c, double cphi, double sphi, double out[3])
{
    if constexpr (DIMS == DIM_TWO) {
        out[0] = a * cphi; out[1] = a * sphi; out[2] = b;
    } else if constexpr (DIMS == DIM_TWO_POINT_FIVE) {
        out[0] = a * cphi - c * sphi; out[1] = a * sphi + c * cphi; out[2] = b;
    } else {
        out[0] = a; out[1] = b; out[2] = c;
    }
}

// cos and sin of atan2(y, x) without the angle (atan2(0,0) = 0 -> (1,0))
// (written with selects, not branches, like zero_norm and lorentz_boost below: the values are the same, and a thread that takes several
// photons through these functions in lockstep -- relocate_lockstep, kernels.hip -- gets one straight-line block to interleave)
__device__ __forceinline__ void cos_sin_of_atan2(double y, double x, double &c, double &s)
{
    const double h2 = x * x + y * y;
    const double ih = rsqrt_nr(h2);
    const bool pos = h2 > 0;
    c = pos ? x * ih : ((x < 0 || (x == 0 && signbit(x))) ? -1.0 : 1.0);
    s = pos ? y * ih : 0.0;
}
// the same from a hypotenuse the caller already has (2-D cylindrical / Cartesian: hydro_coords' sqrt(x^2 + y^2) is the photon's azimuth radius)
__device__ __forceinline__ void cos_sin_with_hypot(double y, double x, double h, double &c, double &s)
{
    const double ih = rcp_nr(h);
    const bool pos = h > 0;
    c = pos ? x * ih : ((x < 0 || (x == 0 && signbit(x))) ? -1.0 : 1.0);
    s = pos ? y * ih : 0.0;
}

// the photon's azimuth at a re-location (mclib.c:549-552), given its hydro coordinates: where the first of them IS the azimuth radius
// (2-D / 2.5-D cylindrical and Cartesian: geometry.c:21-24) one reciprocal of it serves, else the reciprocal root of x^2 + y^2.
// (One function for every kernel that re-locates, so that they all produce the same bits.)
template <int DIMS, int GEOM>
__device__ __forceinline__ void relocation_azimuth(double x, double y, double a0, double &c, double &s)
{
    if constexpr (DIMS != DIM_THREE && GEOM != GEOM_SPHERICAL) cos_sin_with_hypot(y, x, a0, c, s);
    else cos_sin_of_atan2(y, x, c, s);
}

// geometry.c:189-253: fluid velocity of a cell (hydro basis) -> Cartesian, for a photon at azimuth
// phi = atan2(r1, r0) given as (cos phi, sin phi).  The per-cell part of the transform is folded into the
// staged record (engine.hip, pack_fluid): axisymmetric runs store (a, b[, c]) with
//   beta = (a cos(phi) - c sin(phi), a sin(phi) + c cos(phi), b)
// (a = v0, b = v1 in CARTESIAN/CYLINDRICAL; a = v0 sin(th)+v1 cos(th), b = v0 cos(th)-v1 sin(th) in SPHERICAL;
// c = v2 in 2.5-D, absent in 2-D) and 3-D runs store the Cartesian vector itself.
template <int DIMS>
__device__ __forceinline__ void cell_beta(const CellFluid &f, double cphi, double sphi, double out[3])
{
    beta_from_record<DIMS>(f.a, f.b, f.c, cphi, sphi, out);
}

// ---------------------------------------------------------------- boosts
// mclib.c:409-434
__device__ __forceinline__ void zero_norm(double p[4])
{
    const double nrm = sqrt(p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    const double q1 = (p[1] / nrm) * p[0], q2 = (p[2] / nrm) * p[0], q3 = (p[3] / nrm) * p[0];
    const bool fix = p[0] != nrm;
    p[1] = fix ? q1 : p[1];
    p[2] = fix ? q2 : p[2];
    p[3] = fix ? q3 : p[3];
}

// mclib.c:302-407
__device__ __forceinline__ void lorentz_boost(const double b[3], const double p[4], double out[4], bool photon)
{
    const double beta = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    double r[4];
    {
        const double gamma = 1.0 / sqrt(1 - beta * beta);
        const double b2 = beta * beta, g1 = gamma - 1;
        const double L01 = -1 * b[0] * gamma, L02 = -1 * b[1] * gamma, L03 = -1 * b[2] * gamma;
        const double L11 = 1 + ((g1 * (b[0] * b[0])) / b2);
        const double L12 = g1 * (b[0] * b[1] / b2);
        const double L13 = g1 * (b[0] * b[2] / b2);
        const double L22 = 1 + ((g1 * (b[1] * b[1])) / b2);
        const double L23 = (g1 * (b[1] * b[2])) / b2;
        const double L33 = 1 + ((g1 * (b[2] * b[2])) / b2);
        const double m0 = ((p[0] * gamma + p[1] * L01) + p[2] * L02) + p[3] * L03;
        const double m1 = ((p[0] * L01 + p[1] * L11) + p[2] * L12) + p[3] * L13;
        const double m2 = ((p[0] * L02 + p[1] * L12) + p[2] * L22) + p[3] * L23;
        const double m3 = ((p[0] * L03 + p[1] * L13) + p[2] * L23) + p[3] * L33;
        const bool moving = beta > 0;                   // a fluid at rest: the identity (mclib.c:313)
        r[0] = moving ? m0 : p[0]; r[1] = moving ? m1 : p[1]; r[2] = moving ? m2 : p[2]; r[3] = moving ? m3 : p[3];
    }
    if (photon) zero_norm(r);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2]; out[3] = r[3];
}

// lorentzBoost + zeroNorm (mclib.c:302-434) in the form the loop uses.  The boost matrix of mclib.c:318-340 is
//   L00 = g, L0i = -g b_i, Lij = delta_ij + (g - 1) b_i b_j / b^2,
// so  p'_0 = g (p_0 - b.p)  and  p'_i = p_i + (kf (b.p) - g p_0) b_i  with kf = (g - 1)/b^2 = g^2/(g + 1): one dot product and four
// multiply-adds once g and kf are known -- per cell for the fluid frame (CellFluid::gam, ::kf), from the electron's energy for the
// electron frame -- instead of a square root and eight divisions per call.  b = 0 needs no special case (g = 1, kf = 1/2).
// zeroNorm rescales the spatial part to the time component with ONE reciprocal root (the reference divides thrice and skips the
// rescaling when the norm already matches: the same vector to an ulp).
__device__ __forceinline__ void zero_norm_lean(double p[4])
{
    const double s = p[0] * rsqrt_nr((p[1] * p[1] + p[2] * p[2]) + p[3] * p[3]);
    p[1] *= s; p[2] *= s; p[3] *= s;
}
template <bool PHOTON>
__device__ __forceinline__ void boost_with(const double b[3], double g, double kf, const double p[4], double out[4])
{
    const double bp = (b[0] * p[1] + b[1] * p[2]) + b[2] * p[3];
    const double f = kf * bp - g * p[0];
    out[0] = g * (p[0] - bp);
    out[1] = p[1] + f * b[0];
    out[2] = p[2] + f * b[1];
    out[3] = p[3] + f * b[2];
    if constexpr (PHOTON) zero_norm_lean(out);
}

// (gam - 1)/v^2 = gam^2/(gam + 1) of a bucket-list entry, which carries gam alone (device_types.hpp, FatCell)
__device__ __forceinline__ double kf_of_gamma(double gam) { return (gam * gam) * rcp_nr(gam + 1.0); }

// calculateOpticalDepth, optical_depth.c:7-59, on the staged operands (CellFluid):  tau' = nsig * sigma_hat * (1 - w (v.p)/|p|)  [1/cm]
__device__ __forceinline__ double optical_depth_staged(const double fluid_beta[3], double w, double nsig, double p1, double p2, double p3,
                                                       double norm_cross_section = 1.0)
{
    const double bp = (fluid_beta[0] * p1 + fluid_beta[1] * p2) + fluid_beta[2] * p3;
    const double ipn = rsqrt_nr((p1 * p1 + p2 * p2) + p3 * p3);
    return (nsig * norm_cross_section) * (1.0 - w * (bp * ipn));
}

// ---------------------------------------------------------------- optical depth (optical_depth_staged above)

// getCrossSection / getThermalCrossSection, optical_depth.c:117-149: 1 in DIRECT; in TABLE
// 10^interp(log10(h nu'/m_e c^2), log10(kT/m_e c^2)) with GSL's bilinear interp2d scheme on the uniform grid of
// hot_x_section.c:461-502 (cell with x_i <= x < x_{i+1}, last cell closed).  Outside the table the reference
// integrates the cross section afresh (interpolateThermalHotCrossSection's GSL_EDOM branch, hot_x_section.c:563-599 ->
// calculateTotalThermalCrossSection, :324-356): table_fallback_* below.
__device__ __forceinline__ int table_cell(double x0, double dx, int n_cells, double x)
{
    int i = (int)floor((x - x0) / dx);
    i = i < 0 ? 0 : (i > n_cells - 1 ? n_cells - 1 : i);
    // settle on the cell GSL's bisection finds: x0 + i dx <= x < x0 + (i+1) dx
    while (i > 0 && x0 + i * dx > x) --i;
    while (i < n_cells - 1 && !(x0 + (i + 1) * dx > x)) ++i;
    return i;
}

// kleinNishinaCrossSection (mcrat_scattering.c:597-623) with the reference's own divisions: 2e6-sized terms cancel to O(1) at the 1e-3 seam, so where the
// value enters an optical depth (below) it is computed in the reference's operation order (kn_cross_section's reciprocals are for the acceptance test)
__device__ __forceinline__ double kn_cross_section_ieee(double e)
{
    if (e >= 1e-3)
        return (3. / 4.) * (2. / (e * e) + (1. / (2. * e) - (1. + e) / (e * e * e)) * log(1. + 2. * e) + (1. + e) / ((1. + 2. * e) * (1. + 2. * e)));
    return (1. - 2. * e);
}

// The look-up proper.  false: `norm` is the cross section (DIRECT: 1; the table's interpolant; or, for plasma colder than the table, 1 / the
// Klein-Nishina cross section of the photon, hot_x_section.c:337-340).  true: (eps, theta) lies outside the table at a temperature the table
// covers or above it -- the reference integrates (table_fallback_* below); eps = h nu'/m_e c^2 and theta = kT/m_e c^2 are handed back.
__device__ __forceinline__ bool thermal_cross_section_lookup(const HydroDev &h, double photon_comv_e, double fluid_temp, double &norm, double &eps, double &theta_out)
{
    norm = 1.0;
    if (!h.hot_table) return false;
    const double normalized_photon_comv_e = photon_comv_e / (M_EL * C_LIGHT);
    const double theta = K_B * fluid_temp / (M_EL * C_LIGHT * C_LIGHT);
    const double x = log10(normalized_photon_comv_e), y = log10(theta);
    const double x_hi = h.hot_e0 + h.hot_n_ph_e * h.hot_de, y_hi = h.hot_t0 + h.hot_n_t * h.hot_dt;
    const bool out = !(x >= h.hot_e0) | (x > x_hi) | !(y >= h.hot_t0) | (y > y_hi);           // gsl_spline2d_eval_e: GSL_EDOM
    if (out) {
        const double theta_min = pow(10.0, h.hot_t0), e_min = pow(10.0, h.hot_e0);               // :337-340
        if (theta < theta_min) { norm = (normalized_photon_comv_e < e_min) ? 1.0 : kn_cross_section_ieee(normalized_photon_comv_e); return false; }
        // :584-588: the integral is taken at 10^x, 10^y (the arguments went through log10 on their way in)
        eps = pow(10.0, x);
        theta_out = pow(10.0, y);
        return true;
    }
    const int xi = table_cell(h.hot_e0, h.hot_de, h.hot_n_ph_e, x);
    const int yi = table_cell(h.hot_t0, h.hot_dt, h.hot_n_t, y);
    const double xmin = h.hot_e0 + xi * h.hot_de, xmax = h.hot_e0 + (xi + 1) * h.hot_de;
    const double ymin = h.hot_t0 + yi * h.hot_dt, ymax = h.hot_t0 + (yi + 1) * h.hot_dt;
    const int ny = h.hot_n_t + 1;
    const double zminmin = h.hot_table[xi * ny + yi], zminmax = h.hot_table[xi * ny + yi + 1];
    const double zmaxmin = h.hot_table[(xi + 1) * ny + yi], zmaxmax = h.hot_table[(xi + 1) * ny + yi + 1];
    const double t = (x - xmin) / (xmax - xmin), u = (y - ymin) / (ymax - ymin);
    const double z = (1. - t) * (1. - u) * zminmin + t * (1. - u) * zmaxmin + (1. - t) * u * zminmax + t * u * zmaxmax;
    norm = pow(10.0, z);
    return false;
}

// ---------------------------------------------------------------- Stokes helpers
// mcrat_scattering.c:41-65
__device__ __forceinline__ void find_xy(const double v[3], const double ref[3], double x[3], double y[3])
{
    y[0] = (v[1] * ref[2] - v[2] * ref[1]);
    y[1] = -1 * (v[0] * ref[2] - v[2] * ref[0]);
    y[2] = (v[0] * ref[1] - v[1] * ref[0]);
    double norm = rsqrt_nr(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]);
    y[0] *= norm; y[1] *= norm; y[2] *= norm;
    x[0] = y[1] * v[2] - y[2] * v[1];
    x[1] = -1 * (y[0] * v[2] - y[2] * v[0]);
    x[2] = y[0] * v[1] - y[1] * v[0];
    norm = rsqrt_nr(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    x[0] *= norm; x[1] *= norm; x[2] *= norm;
}

// findPhi (mcrat_scattering.c:67-101) followed by mullerMatrixRotation (:10-39):
// phi = -sign(x.y') acos(d), d = y.y' (rounded to +-1 if it strays outside [-1,1]); the Mueller matrix needs
// cos(2 phi) = 2 d^2 - 1 and sin(2 phi) = -sign(x.y') 2 d sqrt(1 - d^2).
__device__ __forceinline__ void rotate_stokes_between(const double x_old[3], const double y_old[3], const double y_new[3], double s[4])
{
    double d = (x_old[0] * y_new[0] + x_old[1] * y_new[1]) + x_old[2] * y_new[2];
    const double factor = (d > 0) ? 1.0 : ((d < 0) ? -1.0 : 0.0);
    d = (y_old[0] * y_new[0] + y_old[1] * y_new[1]) + y_old[2] * y_new[2];
    if ((d < -1) || (d > 1)) d = round(d);
    const double c2 = 2 * d * d - 1;
    const double s2 = -1 * factor * 2 * d * sqrt(1 - d * d);
    const double q = s[1] * c2 + s[2] * (-1 * s2);
    const double u = s[1] * s2 + s[2] * c2;
    s[1] = q;
    s[2] = u;
}

// mcrat_scattering.c:103-149
__device__ __forceinline__ void stokes_rotation(const double v[3], const double v_ph[3], const double v_ph_boosted[3], double s[4])
{
    const double z_hat[3] = {0, 0, 1};
    double x[3], y[3], xn[3], yn[3];
    find_xy(v_ph, z_hat, x, y);
    find_xy(v_ph, v, xn, yn);
    rotate_stokes_between(x, y, yn, s);
    find_xy(v_ph_boosted, v, x, y);
    find_xy(v_ph_boosted, z_hat, xn, yn);
    rotate_stokes_between(x, y, yn, s);
}

// ---------------------------------------------------------------- Klein-Nishina
// mcrat_scattering.c:597-623
__device__ __forceinline__ double kn_cross_section(double e)
{
    if (e >= 1e-3) {
        // (3/4) (2/e^2 + (1/(2e) - (1+e)/e^3) log(1+2e) + (1+e)/(1+2e)^2): the reference's terms, its five divisions as two reciprocals
        const double ie = rcp_nr(e), ie2 = ie * ie, i12 = rcp_nr(1. + 2. * e);
        return (3. / 4.) * (2. * ie2 + (0.5 * ie - (1. + e) * (ie2 * ie)) * log(1. + 2. * e) + (1. + e) * (i12 * i12));
    }
    return (1. - 2. * e);
}

// mcrat_scattering.c:509-595 in its two halves: the acceptance test against the Klein-Nishina cross section (:509-523), and -- only for an
// accepted scattering -- the sampling of the polar and azimuthal angles (:525-595; returns cos(theta) and (cos phi, sin phi) of the scattered
// direction).  Nothing after the test can fail, which is what lets rank_pipe_kernel (kernels.hip) start the next pass while the scattering
// is being completed.
template <class RNG>
__device__ __forceinline__ bool kn_accept(double p0, double &energy_ratio, RNG &rng)
{
    energy_ratio = p0 * (1.0 / (M_EL * C_LIGHT));
    const double kn = kn_cross_section(energy_ratio);
    const double rand_num = rng.uniform();
    return rand_num <= kn;
}

template <bool STOKES, class RNG>
__device__ __forceinline__ void kn_angles(double &cos_theta, double &cos_phi, double &sin_phi, double energy_ratio, double q, double u,
                                          RNG &rng)
{
    double cos_theta_dum = 0, f_cos = 0, y_cos = 1;
    for (int it = 0; it < REJECTION_CAP && (y_cos > f_cos); ++it) {
        y_cos = rng.uniform() * 2;
        cos_theta_dum = rng.uniform() * 2 - 1;
        const double ia = rcp_nr(1 + energy_ratio * (1 - cos_theta_dum));
        f_cos = (ia * ia) * (energy_ratio * (1 - cos_theta_dum) + ia + cos_theta_dum * cos_theta_dum);
    }
    cos_theta = cos_theta_dum;
    double phi_dum = 0;
    bool uniform_phi = true;
    if constexpr (STOKES) uniform_phi = (u == 0 && q == 0);
    if (uniform_phi) {
        phi_dum = rng.uniform() * 2 * M_PI;
    } else {
        const double imu = rcp_nr(1 + energy_ratio * (1 - cos_theta_dum));
        const double st = sqrt(1 - cos_theta_dum * cos_theta_dum);
        const double f_theta = (imu + imu * imu * imu - (imu * imu) * st * st) * st;
        // phi_max = |atan2(-u, q)| / 2  ->  cos(2 phi_max) = q / h, sin(2 phi_max) = |u| / h
        const double ih = rsqrt_nr(q * q + u * u);
        const double pol = (imu * imu) * st * st * st;
        const double inorm = rcp_nr(f_theta + pol * (q * (q * ih) - u * (fabs(u) * ih)));
        double y_phi = 1, f_phi = 0;
        for (int it = 0; it < REJECTION_CAP && (y_phi > f_phi); ++it) {
            y_phi = rng.uniform();
            phi_dum = rng.uniform() * 2 * M_PI;
            double s2, c2;
            sincos(2 * phi_dum, &s2, &c2);
            f_phi = (f_theta + pol * (q * c2 - u * s2)) * inorm;
        }
    }
    sincos(phi_dum, &sin_phi, &cos_phi);
}

// ---------------------------------------------------------------- electron
// Marsaglia polar method, one value per call (the published algorithm of gsl_ran_gaussian)
template <class RNG>
__device__ __forceinline__ double gaussian(RNG &rng, double sigma)
{
    double x = 0, y = 0, r2 = 2;
    for (int it = 0; it < REJECTION_CAP && (r2 > 1.0 || r2 == 0.0); ++it) {
        x = -1.0 + 2.0 * rng.uniform_pos();
        y = -1.0 + 2.0 * rng.uniform_pos();
        r2 = x * x + y * y;
    }
    return sigma * y * sqrt(-2.0 * log(r2) / r2);
}

// electron.c:202-237.  k2e = exp(1/Theta) K_2(1/Theta) of the cell (HydroDev::k2e): the reference's
// x^2 beta exp(-x/Theta) / K_2(1/Theta) is evaluated as x^2 beta exp(-(x-1)/Theta) / k2e, the same number
// without the e^-600 intermediate.
// WAVE: the caller is a full wavefront whose 64 lanes hold the same arguments and the same stream (the event walk).  The
// Maxwell-Juettner rejection loop accepts one attempt in 50 to 300 (the envelope is the reference's: x uniform up to 1 + 100 Theta,
// y uniform up to 1/2), and every attempt takes exactly two numbers of the stream, so lane k tries attempt 64 j + k of round j --
// the stream is a counter, any position is one addition away -- and the first accepted attempt in attempt order wins: the same
// gamma and the same stream position as the one-lane loop, in a sixty-fourth of its trips.
template <bool WAVE = false, class RNG>
__device__ __forceinline__ double sample_thermal_electron(double temp, double k2e, RNG &rng)
{
    double gamma = 1;
    if (temp >= 1e7) {
        const double factor = K_B * temp / (M_EL * C_LIGHT * C_LIGHT);
        if constexpr (WAVE) {
            static_assert(REJECTION_CAP % 64 == 0, "whole rounds");
            const int lane = (int)(threadIdx.x & 63);
            const uint64_t base = rng.state;
            double x_dum = 1;
            bool accepted = false;
            for (int it0 = 0; it0 < REJECTION_CAP && !accepted; it0 += 64) {
                EventStream r;
                r.state = base + 0x9E3779B97F4A7C15ull * (uint64_t)(2 * (it0 + lane));
                x_dum = r.uniform_pos() * (1 + 100 * factor);
                const double beta_x_dum = sqrt(1 - (1 / (x_dum * x_dum)));
                const double y_dum = r.uniform() / 2.0;
                const double f_x_dum = x_dum * x_dum * (beta_x_dum / k2e) * exp(-1 * (x_dum - 1.0) / factor);
                const unsigned long long ok = __ballot(!((f_x_dum != f_x_dum) || (y_dum > f_x_dum)));
                if (ok) {
                    const int first = __ffsll((long long)ok) - 1;
                    x_dum = __shfl(x_dum, first);
                    rng.state = base + 0x9E3779B97F4A7C15ull * (uint64_t)(2 * (it0 + first + 1));
                    accepted = true;
                }
            }
            if (!accepted) {                                   // the bounded loop ran out: its last attempt stands
                x_dum = __shfl(x_dum, 63);
                rng.state = base + 0x9E3779B97F4A7C15ull * (uint64_t)(2ull * REJECTION_CAP);
            }
            gamma = x_dum;
        } else {
            double x_dum = 1, y_dum = 1, f_x_dum = 0;
            for (int it = 0; it < REJECTION_CAP && ((f_x_dum != f_x_dum) || (y_dum > f_x_dum)); ++it) {
                x_dum = rng.uniform_pos() * (1 + 100 * factor);
                const double beta_x_dum = sqrt(1 - (1 / (x_dum * x_dum)));
                y_dum = rng.uniform() / 2.0;
                f_x_dum = x_dum * x_dum * (beta_x_dum / k2e) * exp(-1 * (x_dum - 1.0) / factor);
            }
            gamma = x_dum;
        }
    } else {
        const double factor = sqrt(K_B * temp / M_EL);
        double g1, g2, g3;
        if constexpr (WAVE) {
            // The three Gaussians (Marsaglia's polar method, 79 % of its attempts accepted) draw from one sequence of attempts, two
            // numbers each: the first, second and third accepted attempt are the three values.  Lane k evaluates attempt k.
            const int lane = (int)(threadIdx.x & 63);
            double v0 = 0, v1 = 0, v2 = 0;
            int found = 0;
            uint64_t base = rng.state;
            for (int round = 0; round < REJECTION_CAP / 64 && found < 3; ++round) {
                EventStream r;
                r.state = base + 0x9E3779B97F4A7C15ull * (uint64_t)(2 * lane);
                const double x = -1.0 + 2.0 * r.uniform_pos();
                const double y = -1.0 + 2.0 * r.uniform_pos();
                const double r2 = x * x + y * y;
                const double v = factor * y * sqrt(-2.0 * log(r2) / r2);
                unsigned long long ok = __ballot(!(r2 > 1.0 || r2 == 0.0));
                int used = 64;                                     // attempts of this round the serial loops would have made
                while (ok && found < 3) {
                    const int a = __ffsll((long long)ok) - 1;
                    const double picked = __shfl(v, a);
                    if (found == 0) v0 = picked; else if (found == 1) v1 = picked; else v2 = picked;
                    ++found;
                    ok &= ok - 1;
                    if (found == 3) used = a + 1;
                }
                base += 0x9E3779B97F4A7C15ull * (uint64_t)(2 * used);
            }
            rng.state = base;
            g1 = v0 * (1.0 / C_LIGHT); g2 = v1 * (1.0 / C_LIGHT); g3 = v2 * (1.0 / C_LIGHT);
        } else {
            g1 = gaussian(rng, factor) * (1.0 / C_LIGHT);
            g2 = gaussian(rng, factor) * (1.0 / C_LIGHT);
            g3 = gaussian(rng, factor) * (1.0 / C_LIGHT);
        }
        gamma = rsqrt_nr(1 - ((g1 * g1 + g2 * g2) + g3 * g3));
    }
    return gamma;
}

// electron.c:70-94 with sampleElectronTheta (:177-200) and rotateElectron (:126-175) in line
template <bool WAVE = false, class RNG>
__device__ __forceinline__ void single_thermal_electron(double el_p[4], double temp, double k2e, const double ph_p[4], RNG &rng)
{
    const double gamma = sample_thermal_electron<WAVE>(temp, k2e, rng);
    const double ig = rcp_nr(gamma);
    const double beta = sqrt_nr(1 - ig * ig);
    const double phi = rng.uniform() * 2 * M_PI;
    // theta = acos(ct): only cos(theta) = ct and sin(theta) = sqrt(1 - ct^2) are used
    // (clamped: for a uniform of exactly 0 -- or towards 1 -- rounding can leave the cosine an ulp outside [-1, 1], where the reference's acos is NaN)
    const double ct_e = fmin(1.0, fmax(-1.0, (1 - sqrt_nr(1 + beta * beta + 2 * beta - 4 * beta * rng.uniform())) * rcp_nr(beta)));
    const double st_e = sqrt_nr(1 - ct_e * ct_e);
    double sphi, cphi;
    sincos(phi, &sphi, &cphi);
    const double mc = gamma * (M_EL) * (C_LIGHT);
    el_p[0] = mc;
    const double e1 = mc * beta * ct_e;
    const double e2 = mc * beta * st_e * sphi;
    const double e3 = mc * beta * st_e * cphi;

    // ph_phi = atan2(p2, p3), ph_theta = atan2(sqrt(p2^2 + p3^2), p1): with rho^2 = p2^2 + p3^2 and n^2 = rho^2 + p1^2,
    // cos/sin(ph_theta) = p1/n, rho/n and cos/sin(ph_phi) = p3/rho, p2/rho -- two reciprocal roots for the two angles
    const double rho2 = ph_p[2] * ph_p[2] + ph_p[3] * ph_p[3];
    const double irho = rsqrt_nr(rho2), in = rsqrt_nr(rho2 + ph_p[1] * ph_p[1]);
    const bool off_axis = rho2 > 0;
    const double ct = off_axis ? ph_p[1] * in : ((ph_p[1] < 0 || (ph_p[1] == 0 && signbit(ph_p[1]))) ? -1.0 : 1.0);
    const double sth = off_axis ? (rho2 * irho) * in : 0.0;
    const double cp = off_axis ? ph_p[3] * irho : (signbit(ph_p[3]) ? -1.0 : 1.0);
    const double spp = off_axis ? ph_p[2] * irho : 0.0;
    const double sp = -spp;                       // sin(-ph_phi); cos(-ph_phi) = cp
    // R_y: rows (ct, 0, -st), (0,1,0), (st, 0, ct)
    const double w0 = e1 * ct + e3 * (-sth);
    const double w1 = e2;
    const double w2 = e1 * sth + e3 * ct;
    // R_x(-phi): rows (1,0,0), (0, cos(-phi), -sin(-phi)), (0, sin(-phi), cos(-phi))
    el_p[1] = w0;
    el_p[2] = w1 * cp + w2 * (-sp);
    el_p[3] = w1 * sp + w2 * cp;
}

// ---------------------------------------------------------------- the scattering itself
// mcrat_scattering.c:151-485 in two halves.  single_scatter_begin: into the electron's rest frame (:218-225), the two rotations that
// put the photon on the x axis (:244-296), the Klein-Nishina acceptance test (:307, kleinNishinaScatter :509-523); it returns false on a
// rejection (ph_comov and s untouched: s is rotated on a copy).  single_scatter_finish: angles, Compton shift, the rotations undone, Fano's
// matrix, back to the fluid frame (:307-481); it cannot fail.  ScatterMid is what the first half hands to the second.
struct ScatterMid {
    double el_v[3], g_e, kf_e;       // the electron's velocity, its Lorentz factor and (g - 1)/v^2
    double ph_pr[4];                 // the photon in the electron's rest frame (:218), before the scattering
    double c0, s0, c1, s1;           // cos / sin of the two alignment rotations (:244, :269)
    double energy_ratio;             // ph_pr[0] / (m_e c)
    double s[4];                     // Stokes parameters rotated into the electron frame's basis (:225)
};

template <bool STOKES, class RNG>
__device__ __forceinline__ bool single_scatter_begin(const double el_comov[4], const double ph_comov[4], const double s_in[4], ScatterMid &m,
                                                     RNG &rng)
{
    const double ie0 = rcp_nr(el_comov[0]);
    m.el_v[0] = el_comov[1] * ie0; m.el_v[1] = el_comov[2] * ie0; m.el_v[2] = el_comov[3] * ie0;
    // the electron's Lorentz factor from its energy (el_comov[0] = gamma m_e c, electron.c:86) rather than from 1/sqrt(1 - v^2) of the
    // quotient above (mclib.c:316): the same number without the cancellation
    m.g_e = el_comov[0] * (1.0 / (M_EL * C_LIGHT));
    m.kf_e = (m.g_e * m.g_e) * rcp_nr(m.g_e + 1.0);
    boost_with<true>(m.el_v, m.g_e, m.kf_e, ph_comov, m.ph_pr);                   // :218
    m.s[0] = s_in[0]; m.s[1] = s_in[1]; m.s[2] = s_in[2]; m.s[3] = s_in[3];
    if constexpr (STOKES) stokes_rotation(m.el_v, ph_comov + 1, m.ph_pr + 1, m.s);   // :225

    // phi0 = atan2(py, px) (:244): c0 = cos(-phi0), s0 = sin(-phi0)
    double sp0;
    cos_sin_of_atan2(m.ph_pr[2], m.ph_pr[1], m.c0, sp0);
    m.s0 = -sp0;
    // rot0 rows (c0, -s0, 0), (s0, c0, 0), (0,0,1)
    const double r00 = m.ph_pr[1] * m.c0 + m.ph_pr[2] * (-m.s0);
    const double r02 = m.ph_pr[3];
    // phi1 = atan2(r02, r00) (:269): c1 = cos(-phi1), s1 = sin(-phi1)
    double sp1;
    cos_sin_of_atan2(r02, r00, m.c1, sp1);
    m.s1 = -sp1;
    // after the two alignment rotations the photon is (p0, p0, 0, 0) by construction (:294-296)
    return kn_accept(m.ph_pr[0], m.energy_ratio, rng);                            // :307 -> :509-523
}

template <bool STOKES, class RNG>
__device__ __forceinline__ void single_scatter_finish(const ScatterMid &m, double ph_comov[4], double s[4], RNG &rng)
{
    const double z_axis[3] = {0, 0, 1};
    const double *ph_orig = m.ph_pr;
    const double c0 = m.c0, s0 = m.s0, c1 = m.c1, s1 = m.s1;
    s[0] = m.s[0]; s[1] = m.s[1]; s[2] = m.s[2]; s[3] = m.s[3];
    double ct = 0, cphi = 1, sphi = 0;
    kn_angles<STOKES>(ct, cphi, sphi, m.energy_ratio, s[1], s[2], rng);           // :525-595
    const double sth = sqrt_nr(1 - ct * ct);

    double result[4];
    result[0] = ph_orig[0] * rcp_nr(1 + ((ph_orig[0] * (1 - ct)) * (1.0 / (M_EL * C_LIGHT))));     // :322
    result[1] = result[0] * ct;
    result[2] = result[0] * sth * sphi;
    result[3] = result[0] * sth * cphi;

    // undo rot1: rows (c1, 0, s1), (0,1,0), (-s1, 0, c1)                                :360-366
    const double u0 = result[1] * c1 + result[3] * s1;
    const double u1 = result[2];
    const double u2 = result[1] * (-s1) + result[3] * c1;
    // undo rot0: rows (c0, s0, 0), (-s0, c0, 0), (0,0,1)                                :380-386
    double res0[3];
    res0[0] = u0 * c0 + u1 * s0;
    res0[1] = u0 * (-s0) + u1 * c0;
    res0[2] = u2;

    if constexpr (STOKES) {
        double xt[3], yt[3], xn[3], yn[3];
        find_xy(ph_orig + 1, z_axis, xt, yt);                                     // :402
        find_xy(res0, ph_orig + 1, xn, yn);                                       // :403
        rotate_stokes_between(xt, yt, yn, s);                                     // :404-405
        // theta between incoming and scattered photon (:408): only its cosine and sine are used
        const double cth = ((ph_orig[1] * res0[0] + ph_orig[2] * res0[1]) + ph_orig[3] * res0[2]) * rcp_nr(ph_orig[0] * result[0]);
        const double sn2 = 1 - cth * cth;                                         // sin^2
        // Fano's matrix :411-416
        const double de = (ph_orig[0] - result[0]) * (1.0 / (M_EL * C_LIGHT));
        const double t00 = 1.0 + cth * cth + ((1 - cth) * de);
        const double t01 = sn2;
        const double t11 = 1.0 + cth * cth;
        const double t22 = 2.0 * cth;
        const double t33 = 2.0 * cth + ((cth) * (1 - cth) * de);
        const double o0 = s[0] * t00 + s[1] * t01;
        const double o1 = s[0] * t01 + s[1] * t11;
        const double o2 = s[2] * t22;
        const double o3 = s[3] * t33;
        const double io0 = rcp_nr(o0);
        s[0] = (fabs(io0) < INFINITY && io0 != 0.0) ? 1.0 : __builtin_nan("");   // o0 / o0
        s[1] = o1 * io0; s[2] = o2 * io0; s[3] = o3 * io0;                        // :430-433
        find_xy(res0, ph_orig + 1, xt, yt);                                       // :438
        find_xy(res0, z_axis, xn, yn);                                            // :441
        rotate_stokes_between(xt, yt, yn, s);                                     // :444-447
    }

    double ph_out[4] = {result[0], res0[0], res0[1], res0[2]};                    // :452-454
    const double neg_el_v[3] = {-m.el_v[0], -m.el_v[1], -m.el_v[2]};
    double back[4];
    boost_with<true>(neg_el_v, m.g_e, m.kf_e, ph_out, back);                      // :465
    if constexpr (STOKES) stokes_rotation(neg_el_v, ph_out + 1, back + 1, s);     // :473
    ph_comov[0] = back[0]; ph_comov[1] = back[1]; ph_comov[2] = back[2]; ph_comov[3] = back[3];
}

// the two halves in one call.  ph_comov and s are updated only when the scattering happens.
template <bool STOKES, class RNG>
__device__ __forceinline__ bool single_scatter(const double el_comov[4], double ph_comov[4], double s[4], RNG &rng)
{
    ScatterMid m;
    if (!single_scatter_begin<STOKES>(el_comov, ph_comov, s, m, rng)) return false;
    single_scatter_finish<STOKES>(m, ph_comov, s, rng);
    return true;
}

// exp(x) K_2(x) from K_nu(x) = int_0^inf exp(-x cosh t) cosh(nu t) dt (DLMF 10.32.9), trapezoid rule;
// stands for gsl_sf_bessel_Kn(2, x) at electron.c:221 (scaled by e^x).
__device__ __forceinline__ double bessel_k2_scaled(double x)
{
    double h = 0.35 / sqrt(x);
    if (h > 0.125) h = 0.125;
    double sum = 0.5;
    for (int k = 1; k < 100000; ++k) {
        const double t = k * h;
        const double sh = sinh(0.5 * t);
        const double e = -x * 2.0 * sh * sh;
        sum += exp(e) * cosh(2.0 * t);
        if (e + 2.0 * t < -80.0) break;
    }
    return sum * h;
}

// ---------------------------------------------------------------- the hot cross section's integral (hot_x_section.c:324-400)
// 0.5 * int_1^{1+12 theta} dgamma int_-1^1 dmu f_MJ(gamma; theta) sigma_KN(eps gamma (1 - mu beta)) / sigma_T (1 - mu beta) as the reference's plain
// Monte-Carlo rule (gsl_monte_plain_integrate: volume x mean of the integrand at uniformly drawn points).  The samples are dealt to
// HOT_TABLE_SUBSTREAMS = 256 substreams, sample k to substream k % 256, each with its own keyed generator drawing x0 then x1 per sample (the
// reference's draw order); the 256 partial sums are added in substream order.  hot_table.hip builds the table from these pieces (one workgroup
// per entry); the loop uses them where a look-up falls off the table (table_fallback_lane / _wave).

// singleMaxwellJuttner's normalisation, electron.c:538-561 (it depends on theta only)
__device__ __forceinline__ double mj_normalisation(double theta)
{
    return (theta > 1.e-2) ? bessel_k2_scaled(1. / theta) : sqrt(M_PI * theta / 2.);      // K_2(1/theta) e^(1/theta)
}

// the samples first, first + stride, ... < calls of one substream
__device__ __forceinline__ double hot_substream_sum(double ph_comv, double theta, double mj_norm, EventStream rng, long long first, long long calls, int stride)
{
    // hot_x_section.c:334-335: gamma in [1, 1 + 12 theta], mu in [-1, 1]
    const double g_lo = 1, g_w = (1. + 12 * theta) - 1, mu_lo = -1, mu_w = 1 - (-1.);
    double sum = 0;
    for (long long k = first; k < calls; k += stride) {
        const double gamma = g_lo + rng.uniform_pos() * g_w;          // gsl_monte_plain: x = xl + uniform_pos * (xu - xl)
        const double mu = mu_lo + rng.uniform_pos() * mu_w;
        // thermalCrossSectionIntegrand :359-368 = singleMaxwellJuttner * boostedCrossSection :370-400
        const double mj = ((gamma * sqrt(gamma * gamma - 1.) / (theta * mj_norm)) * exp(-(gamma - 1.) / theta));
        const double beta = sqrt(gamma * gamma - 1.) / gamma;
        const double norm_ph_e = ph_comv * gamma * (1. - mu * beta);
        sum += mj * (kn_cross_section(norm_ph_e) * (1. - mu * beta));
    }
    return sum;
}

// volume x mean, then the 0.5 of :355
__device__ __forceinline__ double hot_integral_of_total(double total, double theta, long long calls)
{
    const double g_w = (1. + 12 * theta) - 1, mu_w = 1 - (-1.);
    return 0.5 * ((g_w * mu_w) * (total / (double)calls));
}

// A look-up off the table inside the loop.  The reference takes 2 x 500 000 numbers from the rank's generator at that point; here substream s of
// the integral for (pass, slot) is the keyed stream {iteration = pass | (s + 1) << 48, word2 = slot, purpose = TABLE_FALLBACK, the list's stream}:
// the value depends on the pass, the slot and (eps, theta) only, not on who computes it -- one lane on its own (a slot that missed somewhere in a
// divergent stretch of a kernel: 500 000 integrand evaluations in one lane, a fifth of a second) or a whole wavefront (64 substreams at a time;
// rank_loop_kernel brings its misses to one: a few milliseconds).  What comes back is what getThermalCrossSection returns for it:
// 10^log10(integral) (hot_x_section.c:588, optical_depth.c:143).
constexpr uint32_t RNG_TABLE_FALLBACK = 9u;
constexpr int TABLE_FALLBACK_SUBSTREAMS = 256;

__device__ __forceinline__ EventStream table_fallback_stream(uint64_t seed, uint64_t pass, uint32_t slot, uint32_t stream, int s)
{
    const Philox4 b = keyed_block(seed, pass | ((uint64_t)(s + 1) << 48), slot, RNG_TABLE_FALLBACK, stream);
    EventStream rng;
    rng.state = (uint64_t)b.w[0] | ((uint64_t)b.w[1] << 32);
    return rng;
}

__device__ __noinline__ inline double table_fallback_lane(double eps, double theta, uint64_t seed, uint64_t pass, uint32_t slot, uint32_t stream, int calls)
{
    const double mj_norm = mj_normalisation(theta);
    double total = 0;
    for (int s = 0; s < TABLE_FALLBACK_SUBSTREAMS && s < calls; ++s)
        total += hot_substream_sum(eps, theta, mj_norm, table_fallback_stream(seed, pass, slot, stream, s), s, calls, TABLE_FALLBACK_SUBSTREAMS);
    return pow(10.0, log10(hot_integral_of_total(total, theta, calls)));
}

// the same number from a full wavefront: every lane passes the same arguments, lane l takes the substreams l, l + 64, l + 128, l + 192
__device__ __noinline__ inline double table_fallback_wave(double eps, double theta, uint64_t seed, uint64_t pass, uint32_t slot, uint32_t stream, int calls)
{
    const int lane = (int)(threadIdx.x & 63u);
    const double mj_norm = mj_normalisation(theta);
    double part[TABLE_FALLBACK_SUBSTREAMS / 64];
#pragma unroll
    for (int g = 0; g < TABLE_FALLBACK_SUBSTREAMS / 64; ++g) {
        const int s = 64 * g + lane;
        part[g] = s < calls ? hot_substream_sum(eps, theta, mj_norm, table_fallback_stream(seed, pass, slot, stream, s), s, calls, TABLE_FALLBACK_SUBSTREAMS) : 0.0;
    }
    double total = 0;
#pragma unroll
    for (int g = 0; g < TABLE_FALLBACK_SUBSTREAMS / 64; ++g)
        for (int l = 0; l < 64; ++l) total += __shfl(part[g], l, 64);      // substream order
    return pow(10.0, log10(hot_integral_of_total(total, theta, calls)));
}

}  // namespace phys
}  // namespace mcrat
