// What a dependent instruction costs ONE wavefront that has its SIMD to itself (the state of the tracks
// that decide how long a tracker launch lasts): cycles per instruction of dependent and independent
// chains of the instruction kinds the tracker's iteration is made of.  One workgroup of 64 threads;
// shader-clock ticks from clock64(), the constant 100 MHz counter beside it gives the clock rate.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 512;   // instructions per chain kind

template <int KIND>
__global__ __launch_bounds__(64) void chain_kernel(double *out, long long *ticks, float a, float b, double da, double db)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    double d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    if (KIND == 0) {   // dependent fp32 FMA
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
    } else if (KIND == 1) {   // 4 independent fp32 FMA chains
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(a), "v"(b));
        }
    } else if (KIND == 2) {   // dependent fp64 FMA
#pragma unroll
        for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(da), "v"(db));
    } else if (KIND == 3) {   // 4 independent fp64 FMA chains
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d1) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d2) : "v"(da), "v"(db));
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d3) : "v"(da), "v"(db));
        }
    } else if (KIND == 4) {   // dependent fp32 adds through DPP (the wavefront reductions)
#pragma unroll
        for (int i = 0; i < N; ++i)
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x0));
    } else if (KIND == 5) {   // dependent fp64 add -> floor -> add (the estimate's integer / fractional split)
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(da));
            asm volatile("v_floor_f64 %0, %0" : "+v"(d0));
        }
    } else if (KIND == 6) {   // dependent fp64 -> fp32 -> fp64 conversions
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x0) : "v"(d0));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d0) : "v"(x0));
        }
    } else if (KIND == 7) {   // a vector compare feeding a scalar branch, in a chain with an fp64 add
#pragma unroll 1
        for (int i = 0; i < N / 2; ++i) {
            d0 += da;
            if (__builtin_amdgcn_ballot_w64(d0 > 1e300) != 0) break;
        }
    } else if (KIND == 8) {   // a loop of one dependent fp64 add whose only branch is the scalar loop counter's
#pragma unroll 1
        for (int i = 0; i < N / 2; ++i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(da));
    } else if (KIND == 9) {   // ... plus a vector compare whose result nothing branches on (accumulated in a mask)
        unsigned long long m = 0;
#pragma unroll 1
        for (int i = 0; i < N / 2; ++i) {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(da));
            m |= __builtin_amdgcn_ballot_w64(d0 > 1e300);
        }
        d1 += (double)m;
    } else if (KIND == 10) {   // ... two compares, two conditional exits
#pragma unroll 1
        for (int i = 0; i < N / 2; ++i) {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(da));
            if (__builtin_amdgcn_ballot_w64(d0 > 1e300) != 0) break;
            if (__builtin_amdgcn_ballot_w64(d0 < -1e300) != 0) break;
        }
    } else if (KIND == 11) {   // ... two compares, ONE conditional exit on their union
#pragma unroll 1
        for (int i = 0; i < N / 2; ++i) {
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(da));
            if ((__builtin_amdgcn_ballot_w64(d0 > 1e300) | __builtin_amdgcn_ballot_w64(d0 < -1e300)) != 0) break;
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        ticks[0] = t1 - t0;
        ticks[1] = w1 - w0;
    }
    out[threadIdx.x] = x0 + x1 + x2 + x3 + d0 + d1 + d2 + d3;
}

template <int KIND>
void run(const char *name, double *d_out, long long *d_ticks, int n)
{
    long long best[2] = {1ll << 60, 1ll << 60};
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(chain_kernel<KIND>, dim3(1), dim3(64), 0, 0, d_out, d_ticks, 0.999f, 1e-3f, 0.999, 1e-3);
        long long t[2];
        hipMemcpy(t, d_ticks, sizeof(t), hipMemcpyDeviceToHost);
        if (t[0] < best[0]) best[0] = t[0], best[1] = t[1];
    }
    printf("%-44s %6.2f ticks per instruction (%lld ticks, %lld x 10 ns)\n", name, (double)best[0] / n, best[0], best[1]);
}

int main()
{
    double *d_out;
    long long *d_ticks;
    hipMalloc(&d_out, sizeof(double) * 64);
    hipMalloc(&d_ticks, sizeof(long long) * 2);
    run<0>("fp32 FMA, dependent", d_out, d_ticks, N);
    run<1>("fp32 FMA, 4 independent chains", d_out, d_ticks, N);
    run<2>("fp64 FMA, dependent", d_out, d_ticks, N);
    run<3>("fp64 FMA, 4 independent chains", d_out, d_ticks, N);
    run<4>("fp32 add by DPP (+ s_nop 1), dependent", d_out, d_ticks, N);
    run<5>("fp64 add -> floor, dependent", d_out, d_ticks, N);
    run<6>("fp64 -> fp32 -> fp64 conversions, dependent", d_out, d_ticks, N);
    run<7>("fp64 add + compare + ballot + branch (per pair)", d_out, d_ticks, N / 2);
    run<8>("loop: fp64 add, counter branch only", d_out, d_ticks, N / 2);
    run<9>("loop: + compare into a mask, no exit", d_out, d_ticks, N / 2);
    run<10>("loop: + two compares, two exits", d_out, d_ticks, N / 2);
    run<11>("loop: + two compares, one exit on the union", d_out, d_ticks, N / 2);
    return 0;
}
