#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CH>
__global__ void chain(double *out, long long *ticks, int n, double a, double b)
{
    double x[CH];
    for (int k = 0; k < CH; ++k) x[k] = threadIdx.x * 1e-3 + k;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < CH; ++k) x[k] = __builtin_fma(x[k], a, b);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < CH; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int CH>
__global__ void chain_rcp(double *out, long long *ticks, int n, double a)
{
    double x[CH];
    for (int k = 0; k < CH; ++k) x[k] = 1.5 + threadIdx.x * 1e-3 + k;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < CH; ++k) x[k] = __builtin_amdgcn_rcp(x[k]) + a;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int k = 0; k < CH; ++k) s += x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main()
{
    double *out; long long *ticks;
    hipMalloc(&out, 8 * 1024); hipMalloc(&ticks, 8 * 16);
    long long h;
    const int n = 1000;
#define RUN(K, CH, W) do { K<CH><<<1, 64 * W>>>(out, ticks, n, 1.0000001, 1e-9); hipDeviceSynchronize(); K<CH><<<1, 64 * W>>>(out, ticks, n, 1.0000001, 1e-9); hipDeviceSynchronize(); \
    hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost); printf("%-10s chains %d waves %d: %.2f ticks per instruction per wave (%.2f per chain step)\n", #K, CH, W, (double)h / (n * 16.0 * CH), (double)h / (n * 16.0)); } while (0)
    RUN(chain, 1, 1); RUN(chain, 2, 1); RUN(chain, 4, 1); RUN(chain, 8, 1);
    RUN(chain, 1, 4); RUN(chain, 2, 4); RUN(chain, 1, 8); RUN(chain, 2, 8); RUN(chain, 4, 8);
#define RUN2(CH, W) do { chain_rcp<CH><<<1, 64 * W>>>(out, ticks, n, 1e-9); hipDeviceSynchronize(); chain_rcp<CH><<<1, 64 * W>>>(out, ticks, n, 1e-9); hipDeviceSynchronize(); \
    hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost); printf("rcp+add    chains %d waves %d: %.2f ticks per pair per wave (%.2f per chain step)\n", CH, W, (double)h / (n * 16.0 * CH), (double)h / (n * 16.0)); } while (0)
    RUN2(1, 1); RUN2(2, 1); RUN2(4, 1); RUN2(1, 8);
    return 0;
}
