// fetch_calibration.hip -- what rocprofv3's FETCH_SIZE reports for the access widths of the photon loop, against KNOWN byte counts.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calibration.hip -o /tmp/fetch_calibration
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calibration        (tools/fetch_calibration.sh does both)
// On gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reports half the bytes of a wide coalesced stream (128-B requests tallied at 64 B;
// MI355X_MICROARCH.md): the factor to apply depends on the request size the access pattern produces, so it is measured here for
//   stream8    one double (8 B) per lane, coalesced: the photon columns as rank_loop_kernel and step_kernel read them
//   stream16   one double2 (16 B) per lane, coalesced: step_kernel's slot pairs
//   gather16   16 B from a record of its own 128-B line per lane (random records): the bucket directory
//   gather128  a whole 128-B record per lane in 16-B pieces (random records): the cell-lookup entries (FatCell)
// Every buffer is far larger than the 256-MB Infinity Cache and read once.  Prints the bytes every kernel requests (its lanes' loads) and the bytes of
// whole 128-B lines it touches.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stream8(const double *__restrict__ a, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void stream16(const double2 *__restrict__ a, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
struct alignas(128) Rec { double d[16]; };
__global__ void gather16(const Rec *__restrict__ a, const uint32_t *__restrict__ idx, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = *reinterpret_cast<const double2 *>(&a[idx[i]].d[0]);
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}
__global__ void gather128(const Rec *__restrict__ a, const uint32_t *__restrict__ idx, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 *p = reinterpret_cast<const double2 *>(&a[idx[i]].d[0]);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double2 v = p[k]; s += v.x + v.y; }
    }
    if (s == 12345.678) out[0] = s;
}

int main()
{
    const size_t bytes = 2ull << 30;                       // 2 GiB per streamed buffer
    const size_t n_rec = 16ull << 20;                      // 16 Mi records of 128 B = 2 GiB; every record visited once, in a random order
    double *a = nullptr, *out = nullptr;
    uint32_t *idx = nullptr;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&idx, n_rec * sizeof(uint32_t)));
    CK(hipMemset(a, 0, bytes));
    std::vector<uint32_t> h(n_rec);
    for (size_t i = 0; i < n_rec; ++i) h[i] = (uint32_t)i;
    uint64_t st = 88172645463325252ull;
    for (size_t i = n_rec - 1; i > 0; --i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; const size_t j = st % (i + 1); const uint32_t t = h[i]; h[i] = h[j]; h[j] = t; }
    CK(hipMemcpy(idx, h.data(), n_rec * sizeof(uint32_t), hipMemcpyHostToDevice));
    const dim3 grid(256 * 8), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        stream8<<<grid, block>>>(a, bytes / 8, out);
        stream16<<<grid, block>>>(reinterpret_cast<const double2 *>(a), bytes / 16, out);
        gather16<<<grid, block>>>(reinterpret_cast<const Rec *>(a), idx, n_rec, out);
        gather128<<<grid, block>>>(reinterpret_cast<const Rec *>(a), idx, n_rec, out);
    }
    CK(hipDeviceSynchronize());
    printf("known_bytes stream8 requested=%zu lines=%zu\n", bytes, bytes);
    printf("known_bytes stream16 requested=%zu lines=%zu\n", bytes, bytes);
    printf("known_bytes gather16 requested=%zu lines=%zu (+ %zu B of indices, streamed 4 B per lane)\n", n_rec * 16, n_rec * 128, n_rec * 4);
    printf("known_bytes gather128 requested=%zu lines=%zu (+ %zu B of indices)\n", n_rec * 128, n_rec * 128, n_rec * 4);
    return 0;
}
