// What the memory system sustains for the access patterns of the batched pyramid kernels on gfx950: a pure
// write stream (8 bytes per lane, as level 0 is written), a pure read stream, a copy, and "1 byte in, 4 bytes
// out" (uint8 image -> float level, the shape of pyr_first_kernel's traffic).
// hipcc -O3 --offload-arch=gfx950 -o /tmp/stream_bw tools/micro/stream_bw.hip && /tmp/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void write_kernel(float2 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = make_float2((float)i, 1.0f);
}
__global__ __launch_bounds__(256) void read_kernel(const float2 *src, size_t n, float *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float acc = 0.f;
    if (i < n) { const float2 v = src[i]; acc = v.x + v.y; }
    if (acc == 123.456f) *sink = acc;
}
__global__ __launch_bounds__(256) void copy_kernel(const float2 *src, float2 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void widen_kernel(const uint16_t *src, float2 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { const uint16_t v = src[i]; dst[i] = make_float2((float)(v & 0xff), (float)(v >> 8)); }
}

__global__ __launch_bounds__(256) void widen_nt_kernel(const uint16_t *src, float2 *dst, size_t n)
{
    typedef float pair __attribute__((ext_vector_type(2)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const uint16_t v = src[i];
        __builtin_nontemporal_store((pair){(float)(v & 0xff), (float)(v >> 8)}, reinterpret_cast<pair *>(dst) + i);
    }
}
__global__ __launch_bounds__(256) void widen4_nt_kernel(const uint32_t *src, float4 *dst, size_t n)
{
    typedef float quad __attribute__((ext_vector_type(4)));
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const uint32_t v = src[i];
        __builtin_nontemporal_store((quad){(float)(v & 0xff), (float)((v >> 8) & 0xff), (float)((v >> 16) & 0xff), (float)(v >> 24)},
                                    reinterpret_cast<quad *>(dst) + i);
    }
}

int main()
{
    const size_t n = (size_t)60 << 20;   // float2 elements: 480 MB written, as one batch of 256 level-0 images
    float2 *a, *b; uint16_t *u; float *sink;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMalloc(&u, n * 2)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(a, 1, n * 8)); CK(hipMemset(u, 1, n * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    auto time = [&](auto launch, double bytes, const char *name) {
        float best = 1e30f;
        for (int r = 0; r < 6; ++r) {
            (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        std::printf("%-34s %8.3f ms  %7.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    time([&] { hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(256), 0, 0, b, n); }, n * 8.0, "write 8 B per lane");
    time([&] { hipLaunchKernelGGL(read_kernel, dim3(blocks), dim3(256), 0, 0, a, n, sink); }, n * 8.0, "read 8 B per lane");
    time([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(256), 0, 0, a, b, n); }, n * 16.0, "copy (bytes read + written)");
    time([&] { hipLaunchKernelGGL(widen_kernel, dim3(blocks), dim3(256), 0, 0, u, b, n); }, n * 10.0, "2 B in, 8 B out per lane");
    time([&] { hipLaunchKernelGGL(widen_nt_kernel, dim3(blocks), dim3(256), 0, 0, u, b, n); }, n * 10.0, "2 B in, 8 B out, non-temporal");
    const unsigned blocks4 = (unsigned)((n / 2 + 255) / 256);
    time([&] { hipLaunchKernelGGL(widen4_nt_kernel, dim3(blocks4), dim3(256), 0, 0, (const uint32_t *)u, (float4 *)b, n / 2); }, n * 10.0,
         "4 B in, 16 B out, non-temporal");
    CK(hipDeviceSynchronize());
    return 0;
}
