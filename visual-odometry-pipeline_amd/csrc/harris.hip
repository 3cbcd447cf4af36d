// Harris / minimum-eigenvalue corner detection for gfx950.
//
// Stands in for detectHarrisFeatures / detectMinEigenFeatures followed by the border drop of
// extractFeatures, as genKeypoints.m:16-18,31-32 calls them (options 'MinQuality', 'FilterSize').
// The toolbox internals are not public; the detector is defined as:
//   I = single(image)/255 extended by replication; gx, gy by the [-1 0 1] filter;
//   A, B, C = separable Gaussian (FilterSize taps, sigma = FilterSize/3) of gx^2, gy^2, gx gy;
//   Harris R = (AB - C^2) - 0.04 (A+B)^2, MinEigen R = ((A+B) - sqrt((A-B)^2 + 4C^2))/2;
//   corner <=> R > MinQuality*max(R), strict maximum of its 8-neighbourhood, >= border px from
//   every edge; sub-pixel offset per axis from the parabola through 3 responses (+-0.5 max).
// Kernels:
//   corner_response_kernel  fused tile kernel: uint8 tile + 3 px halo -> LDS, gradient products
//                           in LDS, separable Gaussian through LDS, response to global memory,
//                           tile maximum -> one atomicMax per workgroup.  fp32, evaluated in a
//                           fixed order without FMA contraction (bit-reproducible).
//   corner_nms_kernel       threshold + 3x3 strict maximum -> one bit per pixel; every wavefront
//                           owns one 64-bit bitmap word (ballot), so there are no atomics.
//   corner_emit_kernel      popcount prefix + block scan over the bitmap = ordered, deterministic
//                           compaction; sub-pixel refinement; output.
#include "vohip_internal.h"

#include <cmath>

namespace {

constexpr int TA = 64, TB = 16, FS = 5, HALF = FS / 2;
constexpr int IA = TA + 2 * HALF + 2, IB = TB + 2 * HALF + 2;  // staged image tile
constexpr int PA = TA + 2 * HALF, PB = TB + 2 * HALF;          // gradient-product tile

struct taps5 {
    float w[FS];
};

__device__ __forceinline__ unsigned ordered_bits(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void corner_response_kernel(const uint8_t *__restrict__ img, int A, int B,
                                                              taps5 G, int method, float *__restrict__ R,
                                                              unsigned *__restrict__ max_bits)
{
    __shared__ float sI[IA * IB];
    __shared__ float sP[3][PA * PB];
    __shared__ float sH[3][TA * PB];
    __shared__ unsigned sMax;
    const int tid = threadIdx.x;
    const int a0 = blockIdx.x * TA, b0 = blockIdx.y * TB;
    if (tid == 0) sMax = 0u;
    const float s255 = 1.0f / 255.0f;
    for (int q = tid; q < IA * IB; q += 256) {
        const int rb = q / IA, ra = q - rb * IA;
        const int a = min(max(a0 - HALF - 1 + ra, 0), A - 1);
        const int b = min(max(b0 - HALF - 1 + rb, 0), B - 1);
        sI[q] = (float)img[(size_t)b * A + a] * s255;
    }
    __syncthreads();
    for (int q = tid; q < PA * PB; q += 256) {
        const int rb = q / PA, ra = q - rb * PA;
        const float *c = sI + (rb + 1) * IA + (ra + 1);
        const float gx = c[1] - c[-1];
        const float gy = c[IA] - c[-IA];
        sP[0][q] = gx * gx;
        sP[1][q] = gy * gy;
        sP[2][q] = gx * gy;
    }
    __syncthreads();
    for (int q = tid; q < TA * PB; q += 256) {
        const int rb = q / TA, ra = q - rb * TA;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float *p = sP[ch] + rb * PA + ra;
            float s = G.w[0] * p[0];
#pragma unroll
            for (int i = 1; i < FS; ++i) s = s + G.w[i] * p[i];
            sH[ch][q] = s;
        }
    }
    __syncthreads();
    float best = -INFINITY;
    for (int q = tid; q < TA * TB; q += 256) {
        const int rb = q / TA, ra = q - rb * TA;
        float v[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float *p = sH[ch] + rb * TA + ra;
            float s = G.w[0] * p[0];
#pragma unroll
            for (int i = 1; i < FS; ++i) s = s + G.w[i] * p[i * TA];
            v[ch] = s;
        }
        float r;
        if (method == 0) {
            const float det = v[0] * v[1] - v[2] * v[2];
            const float tr = v[0] + v[1];
            r = det - 0.04f * (tr * tr);
        } else {
            const float d = v[0] - v[1];
            r = ((v[0] + v[1]) - sqrtf(d * d + 4.0f * (v[2] * v[2]))) * 0.5f;
        }
        const int a = a0 + ra, b = b0 + rb;
        if (a < A && b < B) {
            R[(size_t)b * A + a] = r;
            best = fmaxf(best, r);
        }
    }
    // tile maximum -> one atomic per workgroup
    unsigned kb = ordered_bits(best);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) kb = max(kb, (unsigned)__shfl_xor((int)kb, m, 64));
    if ((tid & 63) == 0) atomicMax(&sMax, kb);
    __syncthreads();
    if (tid == 0) atomicMax(max_bits, sMax);
}

// Threshold + strict 3x3 maximum.  Threads run over the linear pixel index, so each wavefront
// covers 64 consecutive pixels and owns exactly one 64-bit word of the corner bitmap (written by
// lane 0, no atomics, every word written on every call).
__global__ __launch_bounds__(256) void corner_nms_kernel(const float *__restrict__ R, int A, int B, int lo,
                                                         float min_quality, const unsigned *__restrict__ max_bits,
                                                         unsigned long long *__restrict__ bitmap)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int b = idx / A, a = idx - b * A;
    bool corner = false;
    if (idx < A * B && a >= lo && a < A - lo && b >= lo && b < B - lo) {
        const float thr = min_quality * from_ordered_bits(*max_bits);
        const float *c = R + idx;
        const float r = c[0];
        corner = r > thr && r > c[-1] && r > c[1] && r > c[-A] && r > c[A] && r > c[-A - 1] && r > c[-A + 1] &&
                 r > c[A - 1] && r > c[A + 1];
    }
    const unsigned long long m = __ballot(corner);
    if ((threadIdx.x & 63) == 0 && (idx & ~63) < A * B) bitmap[idx >> 6] = m;
}

// Ordered, deterministic compaction of the corner bitmap.  Thread t of workgroup g owns bitmap word
// 256 g + t.  Every workgroup first sums the popcounts of all words before its own (a redundant
// but cheap read of at most the whole bitmap from L2; no inter-workgroup hand-off), then scans its
// own 256 counts, and every thread writes the corners of its word at their ranks: sub-pixel
// position (x, y 1-based; swap_xy as in the tracker) and metric.  The last workgroup stores the
// total.
__global__ __launch_bounds__(256) void corner_emit_kernel(const float *__restrict__ R, int A, int B,
                                                          const unsigned long long *__restrict__ bitmap, int cap,
                                                          int swap_xy, float *__restrict__ xy_out,
                                                          float *__restrict__ metric_out, int *__restrict__ count)
{
    __shared__ int sPre[4], sWave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwords = (A * B + 63) / 64;
    const int first = blockIdx.x * 256;
    int before = 0;
    for (int w = tid; w < first; w += 256) before += __popcll(bitmap[w]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) before += __shfl_xor(before, m, 64);
    const int w = first + tid;
    unsigned long long m = w < nwords ? bitmap[w] : 0ull;
    const int c = __popcll(m);
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    if (lane == 0) sPre[wave] = before;
    __syncthreads();
    int pos = sPre[0] + sPre[1] + sPre[2] + sPre[3] + incl - c;
    for (int q = 0; q < wave; ++q) pos += sWave[q];
    if (blockIdx.x == gridDim.x - 1 && tid == 255) *count = pos + c;
    while (m) {
        const int bit = __ffsll((long long)m) - 1;
        m &= m - 1;
        if (pos < cap) {
            const int key = w * 64 + bit;
            const int b = key / A, a = key - b * A;
            const float *cc = R + key;
            const float r = cc[0];
            const float da = cc[1] - cc[-1], dda = (cc[1] - 2.0f * r) + cc[-1];
            const float db = cc[A] - cc[-A], ddb = (cc[A] - 2.0f * r) + cc[-A];
            float oa = dda < 0.0f ? -0.5f * da / dda : 0.0f;
            float ob = ddb < 0.0f ? -0.5f * db / ddb : 0.0f;
            oa = fminf(fmaxf(oa, -0.5f), 0.5f);
            ob = fminf(fmaxf(ob, -0.5f), 0.5f);
            const float pa = (float)a + oa + 1.0f, pb = (float)b + ob + 1.0f;
            xy_out[pos] = swap_xy ? pb : pa;
            xy_out[cap + pos] = swap_xy ? pa : pb;
            metric_out[pos] = r;
        }
        ++pos;
    }
}

#pragma clang fp contract(fast)

void gaussian_taps(int filter_size, float *w)
{
    const double sigma = filter_size / 3.0;
    double t[32], sum = 0.0;
    const int half = filter_size / 2;
    for (int i = 0; i < filter_size; ++i) {
        const double d = i - half;
        t[i] = std::exp(-d * d / (2.0 * sigma * sigma));
        sum += t[i];
    }
    for (int i = 0; i < filter_size; ++i) w[i] = (float)(t[i] / sum);
}

}  // namespace

extern "C" {

// d_xy: cap x 2 floats (x column then y column), d_metric: cap floats, d_count: 1 int (total
// found, may exceed cap).  Asynchronous.
int vohip_detect_corners_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor,
                             double min_quality, int filter_size, int border, int method, float *d_xy,
                             float *d_metric, int cap, int *d_count)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_img && d_xy && d_metric && d_count, "NULL argument");
    VOHIP_REQUIRE(ctx, rows >= 8 && cols >= 8, "image too small");
    VOHIP_REQUIRE(ctx, filter_size == FS, "only FilterSize 5 (the toolbox default) is built");
    VOHIP_REQUIRE(ctx, method == 0 || method == 1, "method: 0 = Harris, 1 = MinEigen");
    VOHIP_REQUIRE(ctx, cap >= 1 && cap <= 16384, "cap must be 1..16384");
    VOHIP_REQUIRE(ctx, border >= 0, "negative border");
    const int A = colmajor ? rows : cols, B = colmajor ? cols : rows;
    void *d_R, *d_bitmap, *d_max;
    const int nwords = (A * B + 63) / 64;
    VOHIP_TRY(vohip_upload(ctx, B_HARRIS_R, nullptr, sizeof(float) * (size_t)A * B, &d_R));
    VOHIP_TRY(vohip_upload(ctx, B_HARRIS_KEYS, nullptr, sizeof(unsigned long long) * (size_t)nwords, &d_bitmap));
    VOHIP_TRY(vohip_upload(ctx, B_HARRIS_MAX, nullptr, 256, &d_max));
    taps5 G;
    gaussian_taps(filter_size, G.w);
    hipStream_t st = ctx->stream;
    VOHIP_HIP(ctx, hipMemsetAsync(d_max, 0, 256, st));
    hipLaunchKernelGGL(corner_response_kernel, dim3((A + TA - 1) / TA, (B + TB - 1) / TB), dim3(256), 0, st, d_img, A,
                       B, G, method, (float *)d_R, (unsigned *)d_max);
    const int lo = border < 1 ? 1 : border;
    hipLaunchKernelGGL(corner_nms_kernel, dim3((A * B + 255) / 256), dim3(256), 0, st, (const float *)d_R, A, B, lo,
                       (float)min_quality, (const unsigned *)d_max, (unsigned long long *)d_bitmap);
    hipLaunchKernelGGL(corner_emit_kernel, dim3((nwords + 255) / 256), dim3(256), 0, st, (const float *)d_R, A, B,
                       (const unsigned long long *)d_bitmap, cap, colmajor ? 1 : 0, d_xy, d_metric, d_count);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int vohip_detect_corners(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double min_quality,
                         int filter_size, int border, int method, float *xy_out, float *metric_out, int cap,
                         int *n_out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, img && xy_out && metric_out && n_out, "NULL argument");
    VOHIP_REQUIRE(ctx, rows >= 1 && cols >= 1 && cap >= 1, "bad size");
    void *d_img, *d_xy, *d_metric, *d_count;
    VOHIP_TRY(vohip_upload(ctx, B_IMG_STAGE, img, (size_t)rows * cols, &d_img));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, nullptr, sizeof(float) * 2 * (size_t)cap, &d_xy));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, nullptr, sizeof(float) * (size_t)cap, &d_metric));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, 256, &d_count));
    VOHIP_TRY(vohip_detect_corners_dev(ctx, (const uint8_t *)d_img, rows, cols, colmajor, min_quality, filter_size,
                                       border, method, (float *)d_xy, (float *)d_metric, cap, (int *)d_count));
    int total = 0;
    VOHIP_TRY(vohip_download(ctx, &total, d_count, sizeof(int)));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n = total < cap ? total : cap;
    // compact the two coordinate columns to an n x 2 column-major matrix
    VOHIP_TRY(vohip_download(ctx, xy_out, d_xy, sizeof(float) * n));
    VOHIP_TRY(vohip_download(ctx, xy_out + n, (const float *)d_xy + cap, sizeof(float) * n));
    VOHIP_TRY(vohip_download(ctx, metric_out, d_metric, sizeof(float) * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    if (total > cap)
        return vohip_fail(ctx, VOHIP_CAPACITY_EXCEEDED, "%d corners found, capacity %d: raise cap or MinQuality", total,
                          cap);
    return VOHIP_OK;
}

}  // extern "C"
