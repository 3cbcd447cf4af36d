// Attainable fp64 VALU rate on this GPU: every lane runs 8 independent FMA chains with scalar
// multiplicands, 4 waves per SIMD, no memory traffic.  Prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void fma_kernel(double *out, double a, double b, int iters)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x0 = __builtin_fma(x0, a, b);
            x1 = __builtin_fma(x1, a, b);
            x2 = __builtin_fma(x2, a, b);
            x3 = __builtin_fma(x3, a, b);
            x4 = __builtin_fma(x4, a, b);
            x5 = __builtin_fma(x5, a, b);
            x6 = __builtin_fma(x6, a, b);
            x7 = __builtin_fma(x7, a, b);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    const int blocks = 256 * 4, iters = 16384;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fma_kernel, dim3(blocks), dim3(256), 0, 0, d, 0.999999, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 64.0 * iters * blocks * 256;
        printf("fp64 FMA: %.3f ms, %.1f TFLOP/s\n", ms, flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
