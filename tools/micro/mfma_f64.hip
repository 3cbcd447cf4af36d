// fp64 matrix instruction v_mfma_f64_16x16x4_f64 on this GPU: (1) its rate alone, (2) its rate with independent
// fp64 vector FMAs interleaved (do the two pipes run together?), (3) whether its result equals the chain
// fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c)))) bit for bit (k in increasing order) or some other order.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int VALU_PER_MFMA, bool USE_MFMA>
__global__ __launch_bounds__(256) void mix_kernel(double *out, double a, double b, int iters)
{
    double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const double av = threadIdx.x * 1e-3, bv = 1.0 + threadIdx.x * 1e-4;
    double x[8];
    for (int u = 0; u < 8; ++u) x[u] = threadIdx.x + u;
    for (int i = 0; i < iters; ++i) {
        if (USE_MFMA) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < VALU_PER_MFMA; ++u) x[u & 7] = __builtin_fma(x[u & 7], a, b);
        if (USE_MFMA) {
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < VALU_PER_MFMA; ++u) x[(u + 4) & 7] = __builtin_fma(x[(u + 4) & 7], a, b);
    }
    double s = 0;
    for (int u = 0; u < 8; ++u) s += x[u];
    out[blockIdx.x * 256 + threadIdx.x] = s + acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

// one MFMA: A 16x4, B 4x16 from memory, C given; D written out (row-major 16x16)
__global__ __launch_bounds__(64) void one_mfma(const double *A, const double *B, const double *C, double *D)
{
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];   // A[row = l & 15][k = l >> 4]
    const double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l >> 4][col = l & 15]
    double4_t c;
    for (int r = 0; r < 4; ++r) c[r] = C[((l >> 4) + 4 * r) * 16 + (l & 15)];
    const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
}

template <int V, bool M>
double run(double *d, int blocks, int iters, const char *label)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mix_kernel<V, M>), dim3(blocks), dim3(256), 0, 0, d, 0.999999, 1e-9, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
    }
    const double waves = blocks * 4.0;
    const double mfma_flop = M ? 2.0 * 2048.0 * iters * waves : 0.0;
    const double valu_flop = 2.0 * V * 2.0 * 64.0 * iters * waves;
    printf("%-34s %.3f ms   matrix %.1f TFLOP/s + vector %.1f TFLOP/s = %.1f\n", label, best, mfma_flop / (best * 1e-3) / 1e12,
           valu_flop / (best * 1e-3) / 1e12, (mfma_flop + valu_flop) / (best * 1e-3) / 1e12);
    return best;
}

int main()
{
    const int blocks = 256 * 4, iters = 4096;
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    run<0, true>(d, blocks, iters, "MFMA f64 alone");
    run<16, false>(d, blocks, iters, "16 vector FMAs alone");
    run<4, true>(d, blocks, iters, "MFMA + 4 vector FMAs");
    run<8, true>(d, blocks, iters, "MFMA + 8 vector FMAs");
    run<12, true>(d, blocks, iters, "MFMA + 12 vector FMAs");
    run<16, true>(d, blocks, iters, "MFMA + 16 vector FMAs");
    run<24, true>(d, blocks, iters, "MFMA + 24 vector FMAs");
    // accumulation order
    double hA[64], hB[64], hC[256], hD[256];
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 * 2000.0 - 1000.0; };
    for (double &v : hA) v = rnd();
    for (double &v : hB) v = rnd();
    for (double &v : hC) v = rnd();
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof(hC), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    int inc = 0, dec = 0, exact = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double up = hC[r * 16 + c], down = hC[r * 16 + c];
            for (int k = 0; k < 4; ++k) up = std::fma(hA[r * 4 + k], hB[k * 16 + c], up);
            for (int k = 3; k >= 0; --k) down = std::fma(hA[r * 4 + k], hB[k * 16 + c], down);
            long double ex = hC[r * 16 + c];
            for (int k = 0; k < 4; ++k) ex += (long double)hA[r * 4 + k] * hB[k * 16 + c];
            inc += hD[r * 16 + c] == up;
            dec += hD[r * 16 + c] == down;
            exact += hD[r * 16 + c] == (double)ex;
        }
    printf("of 256 results: %d equal the fma chain k = 0..3, %d the chain k = 3..0, %d the correctly rounded sum\n", inc, dec, exact);
    return 0;
}
