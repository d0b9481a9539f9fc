// gather_bench.hip -- what a per-lane record gather costs on gfx950 (run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gb && /tmp/gb)
// Every lane reads one 128-B record (a FatCell, device_types.hpp) at a random place of a table, as a re-location does.
//   own      the lane reads its record itself: 8 (or 6, 5) global_load_dwordx4, every lane on a cache line of its own
//   coop     8 lanes read one record together (one 16-B piece each): a wave-instruction touches 8 lines instead of 64; the pieces go through
//            LDS to the lane that owns the record
// Prints records per microsecond per CU for 1, 2 and 4 workgroups of 256 threads per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct alignas(16) Rec { double v[16]; };

__device__ __forceinline__ unsigned next_index(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }

template <int PIECES>
__global__ __launch_bounds__(256) void own_kernel(const Rec *__restrict__ table, unsigned mask, int iters, double *out)
{
    unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        x = next_index(x);
        const double2 *p = reinterpret_cast<const double2 *>(table + (x & mask));
        double2 q[PIECES];
#pragma unroll
        for (int k = 0; k < PIECES; ++k) q[k] = p[k];
#pragma unroll
        for (int k = 0; k < PIECES; ++k) acc += q[k].x * 1.0000001 + q[k].y;
        x += (unsigned)(acc != 0.5);     // the next address depends on the data: one gather in flight per lane, as in the loop
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void coop_kernel(const Rec *__restrict__ table, unsigned mask, int iters, double *out)
{
    __shared__ double2 stage[4][64 * 8 + 8];       // per wave: 64 records x 8 pieces (+ padding)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        x = next_index(x);
        const unsigned mine = x & mask;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int owner = 8 * k + (lane >> 3);
            const unsigned rec = (unsigned)__shfl((int)mine, owner, 64);
            const double2 piece = reinterpret_cast<const double2 *>(table + rec)[lane & 7];
            stage[w][owner * 8 + ((lane & 7) ^ (owner & 7))] = piece;      // xor swizzle: the read-back below is conflict-free
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        double2 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = stage[w][lane * 8 + (k ^ (lane & 7))];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += q[k].x * 1.0000001 + q[k].y;
        x += (unsigned)(acc != 0.5);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    for (size_t kb : {16, 2048, 32768}) {      // L1-resident, L2-resident, Infinity-Cache-resident
        const size_t mb = kb, nrec = kb * 1024 / sizeof(Rec);
        Rec *table; double *out;
        CK(hipMalloc(&table, nrec * sizeof(Rec)));
        CK(hipMemset(table, 0, nrec * sizeof(Rec)));
        CK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
        const unsigned mask = (unsigned)nrec - 1;
        const int iters = 400;
        for (int per_cu : {1, 2, 4}) {
            const int blocks = cus * per_cu;
            auto time = [&](auto launch) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                launch(); hipDeviceSynchronize();
                hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
                float ms = 0; hipEventElapsedTime(&ms, a, b);
                return (double)blocks * 256 * iters / (ms * 1e3) / cus;      // records per us per CU
            };
            const double r8 = time([&] { own_kernel<8><<<blocks, 256>>>(table, mask, iters, out); });
            const double r6 = time([&] { own_kernel<6><<<blocks, 256>>>(table, mask, iters, out); });
            const double r5 = time([&] { own_kernel<5><<<blocks, 256>>>(table, mask, iters, out); });
            const double r2 = time([&] { own_kernel<2><<<blocks, 256>>>(table, mask, iters, out); });
            const double rc = time([&] { coop_kernel<<<blocks, 256>>>(table, mask, iters, out); });
            printf("table %6zu KB, %d workgroups of 256 per CU: records/us/CU  own 8 pieces %.1f | 6 pieces %.1f | 5 pieces %.1f | 2 pieces %.1f | cooperative (8 lanes per record, through LDS) %.1f\n",
                   mb, per_cu, r8, r6, r5, r2, rc);
        }
        CK(hipFree(table)); CK(hipFree(out));
    }
    return 0;
}
