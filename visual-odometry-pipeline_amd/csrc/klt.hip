// Image pyramids and the pyramidal Lucas-Kanade tracker for gfx950.
//
// Stands in for vision.PointTracker as the reference uses it (constructed at
// main_demo.m:342-343,348-349 with BlockSize [21 21], NumPyramidLevels 5, MaxIterations 20,
// MaxBidirectionalError 6; initialised and stepped at main_demo.m:344,380,417-424 and
// update_landmarks.m:10).  The toolbox's internals are not public; the algorithm implemented
// here is defined in full below and in the comments of the kernels.
//
// Layout: an image is img[b * A + a] with a the fast axis.  A MATLAB column-major HxW image is
// exactly that with a = row, b = column; the tracker is symmetric in the two axes, so images
// are never transposed -- only the point coordinates are mapped ((x, y) 1-based <-> (a, b)
// 0-based).
//
// klt_track_kernel: ONE WAVEFRONT PER TRACK, forward then backward pass in the same launch.
//   per pyramid level (coarse to fine):
//     * the (W+3)^2 neighbourhood of the point in the previous image is staged in LDS; the window's
//       rows are split into 64 / W bands and a lane owns one window column within one band (7
//       vertically consecutive pixels for W = 21), so the taps and the row-wise interpolations that
//       neighbouring pixels share stay in registers; it keeps the template values and the
//       central-difference gradients of its pixels in registers;
//     * G = sum [gx gx, gx gy; gy gy] (fp64) and, per iteration, b = sum e [gx; gy] (fp32) are
//       accumulated per lane and reduced across the wavefront with DPP row operations + readlane
//       (every lane ends up with the same bits, so the Gauss-Newton step is wave-uniform);
//     * a (W+8)^2 region of the current image around the running estimate is staged in LDS and
//       re-used by successive iterations while the estimate stays within +-3 px of its centre; the lane's
//       taps of it stay in registers while the integer part of the estimate does not change.
//   Pixel arithmetic is fp32 with explicit fmaf (bit-identical to a scalar evaluation of the
//   same formulas).  G is summed in fp64; the right-hand side b is summed per lane in fp32 down the
//   lane's column and across lanes by a fixed fp32 tree.
#include "vohip_internal.h"
#include "seq_index.h"
#include "wave_reduce.h"

#include <cmath>
#include <cstring>

namespace {

constexpr int MAX_LEVELS = 8;
constexpr double KLT_EPS_STEP = 0.01;  // px; iteration stops after a shorter step
constexpr double KLT_MIN_EIG = 1e-3;   // lambda_min(G) / W^2 below this: untrackable

struct pyr_desc {
    const float *lvl[MAX_LEVELS];
    int A[MAX_LEVELS], B[MAX_LEVELS];
    int levels;
};

#pragma clang fp contract(off)

__device__ __forceinline__ float tap5(float t0, float t1, float t2, float t3, float t4)
{
    return ((t0 + t4) + 4.0f * (t1 + t3) + 6.0f * t2) * 0.0625f;
}

// (blockIdx.y: image of a batch; the strides are 0 for a single image)
__global__ __launch_bounds__(256) void u8_to_float_kernel(const uint8_t *__restrict__ src, int n,
                                                          float *__restrict__ dst, size_t src_stride = 0,
                                                          size_t dst_stride = 0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[blockIdx.y * dst_stride + i] = (float)src[blockIdx.y * src_stride + i];
}

// [1 4 6 4 1]/16 along a then along b, replicated border, even samples kept.  A thread makes PD_ROWS
// vertically adjacent samples: their windows share three of five source rows (2 PD_ROWS + 3 horizontal
// passes instead of 5 PD_ROWS; each sample's arithmetic is what a thread of its own would do).  One sample
// per thread for a single image (its small levels are a matter of latency: 12.4 us per pyramid against 18.0
// with four), four for a batch (throughput: 0.52 -> 0.47 ms per 256 pyramids).
// (blockIdx.z: image of a batch, `stride` floats apart in src and dst; 0 for a single image)
template <int PD_ROWS>
__global__ __launch_bounds__(256) void pyr_down_kernel(const float *__restrict__ src, int A, int B,
                                                       float *__restrict__ dst, int A2, int B2, size_t stride)
{
    const int a = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b_first = (blockIdx.y * 4 + (threadIdx.x >> 6)) * PD_ROWS;
    if (a >= A2 || b_first >= B2) return;
    src += blockIdx.z * stride;
    dst += blockIdx.z * stride;
    const int a0 = max(2 * a - 2, 0), a1 = max(2 * a - 1, 0), a2 = 2 * a, a3 = min(2 * a + 1, A - 1),
              a4 = min(2 * a + 2, A - 1);
    float h[2 * PD_ROWS + 3];
    if constexpr (PD_ROWS > 1) {
        // (a batch: every row's five loads are issued before the first value is used -- the loop below, left to the
        // compiler, waits for memory after every row or two)
        float t[2 * PD_ROWS + 3][5];
        // away from the left and right edges a row's five values are neighbours in memory: a 16-byte load (at the
        // values' own address: 4-byte aligned) and a 4-byte one instead of five
        const bool inner = 2 * a - 2 >= 0 && 2 * a + 2 <= A - 1;
        if (__builtin_amdgcn_ballot_w64(!inner) == 0) {
            typedef float floats4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int j = 0; j < 2 * PD_ROWS + 3; ++j) {
                const int bb = min(max(2 * b_first + j - 2, 0), B - 1);
                const float *row = src + (size_t)bb * A + 2 * a - 2;
                const floats4 q = *reinterpret_cast<const floats4 *>(row);
                t[j][0] = q.x;
                t[j][1] = q.y;
                t[j][2] = q.z;
                t[j][3] = q.w;
                t[j][4] = row[4];
            }
        } else {
            asm volatile("" ::: "memory");   // (keeps these loads on this side of the branch: hoisted above it, both sides' loads would run)
#pragma unroll
            for (int j = 0; j < 2 * PD_ROWS + 3; ++j) {
                const int bb = min(max(2 * b_first + j - 2, 0), B - 1);
                const float *row = src + (size_t)bb * A;
                t[j][0] = row[a0];
                t[j][1] = row[a1];
                t[j][2] = row[a2];
                t[j][3] = row[a3];
                t[j][4] = row[a4];
            }
        }
#pragma unroll
        for (int j = 0; j < 2 * PD_ROWS + 3; ++j) asm volatile("" : "+v"(t[j][0]), "+v"(t[j][1]), "+v"(t[j][2]), "+v"(t[j][3]), "+v"(t[j][4]));
#pragma unroll
        for (int j = 0; j < 2 * PD_ROWS + 3; ++j) h[j] = tap5(t[j][0], t[j][1], t[j][2], t[j][3], t[j][4]);
    } else {
#pragma unroll
        for (int j = 0; j < 2 * PD_ROWS + 3; ++j) {
            const int bb = min(max(2 * b_first + j - 2, 0), B - 1);
            const float *row = src + (size_t)bb * A;
            h[j] = tap5(row[a0], row[a1], row[a2], row[a3], row[a4]);
        }
    }
#pragma unroll
    for (int r = 0; r < PD_ROWS; ++r) {
        const int b = b_first + r;
        if (b < B2) dst[(size_t)b * A2 + a] = tap5(h[2 * r], h[2 * r + 1], h[2 * r + 2], h[2 * r + 3], h[2 * r + 4]);
    }
}

// Levels 0 and 1 from the uint8 image in one pass: a level-1 sample comes from the 5 x 5 pixels around
// (2a, 2b) -- converted as they are read, the same taps in the same order as pyr_down_kernel applies to the
// float level -- and the 2 x 2 block of level-0 floats at (2a, 2b) lies inside that window, so the float
// image is written here and never read back: level 0 costs its write only.  A thread makes PF_ROWS
// vertically adjacent samples: their windows share three of five image rows, so it reads 2 PF_ROWS + 3 rows
// of five pixels instead of 5 PF_ROWS.
// This kernel serves a single image, whose pyramid is a matter of latency (byte loads row by row: the fewest
// instructions); batches take pyr_first_pairs_kernel below.  (blockIdx.z: image of a batch.)
constexpr int PF_ROWS = 4;
__global__ __launch_bounds__(256) void pyr_first_kernel(const uint8_t *__restrict__ src, int A, int B,
                                                        float *__restrict__ lvl0, float *__restrict__ lvl1, int A2, int B2,
                                                        size_t src_stride, size_t dst_stride)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int b_first = blockIdx.y * PF_ROWS;
    if (a >= A2) return;
    src += blockIdx.z * src_stride;
    lvl0 += blockIdx.z * dst_stride;
    lvl1 += blockIdx.z * dst_stride;
    const int a0 = max(2 * a - 2, 0), a1 = max(2 * a - 1, 0), a2 = 2 * a, a3 = min(2 * a + 1, A - 1),
              a4 = min(2 * a + 2, A - 1);
    const bool right = 2 * a + 1 < A;
    // image rows 2 b_first - 2 .. 2 b_first + 2 PF_ROWS: the horizontal pass of each, once
    float h[2 * PF_ROWS + 3];
#pragma unroll
    for (int j = 0; j < 2 * PF_ROWS + 3; ++j) {
        const int y = 2 * b_first + j - 2;
        const int bb = min(max(y, 0), B - 1);
        const uint8_t *row = src + (size_t)bb * A;
        const float t0 = (float)row[a0], t1 = (float)row[a1], t2 = (float)row[a2], t3 = (float)row[a3], t4 = (float)row[a4];
        h[j] = tap5(t0, t1, t2, t3, t4);
        if (j >= 2 && j < 2 * PF_ROWS + 2 && y < B) {   // (y >= 0 here) an image row of this thread's blocks: its two pixels
            float *o = lvl0 + (size_t)y * A + 2 * a;
            o[0] = t2;
            if (right) o[1] = t3;
        }
    }
#pragma unroll
    for (int r = 0; r < PF_ROWS; ++r) {
        const int b = b_first + r;
        if (b < B2) lvl1[(size_t)b * A2 + a] = tap5(h[2 * r], h[2 * r + 1], h[2 * r + 2], h[2 * r + 3], h[2 * r + 4]);
    }
}

// The same two levels for a batch of images (A >= 8), where the bytes moved count: a thread makes the level-1
// samples 2p and 2p + 1 of PF_ROWS rows.  Their two 5-pixel windows lie in the seven pixels 4p - 2 .. 4p + 4,
// read as ONE 8-byte load per image row at the pixels' own address -- the window is moved inside the row at the
// left and right edges and the seven (clamped) pixels are picked out of it by byte index (v_perm_b32), so every
// thread runs the same instructions and all eleven loads are in flight before the first is used -- and the row's
// four level-0 floats leave as one non-temporal 16-byte store (level 0 is 2/3 of the bytes this kernel moves and
// is not read again before the trackers: written through the caches it held the kernel at 3.9 TB/s where the
// same pattern reaches 7 with non-temporal 16-byte stores).  Same pixels, same taps, same order as pyr_first_kernel:
// the same bits.  Workgroup: 64 sample pairs x 4 groups of PF_ROWS rows.
__global__ __launch_bounds__(256) void pyr_first_pairs_kernel(const uint8_t *__restrict__ src, int A, int B,
                                                              float *__restrict__ lvl0, float *__restrict__ lvl1, int A2,
                                                              int B2, size_t src_stride, size_t dst_stride)
{
    typedef float quad __attribute__((ext_vector_type(4), aligned(4)));
    typedef float pair __attribute__((ext_vector_type(2), aligned(4)));
    struct __attribute__((packed, aligned(1))) bytes8 { uint32_t lo, hi; };
    const int p = blockIdx.x * 64 + (threadIdx.x & 63);
    const int b_first = (blockIdx.y * 4 + (threadIdx.x >> 6)) * PF_ROWS;
    if (2 * p >= A2 || b_first >= B2) return;
    src += blockIdx.z * src_stride;
    lvl0 += blockIdx.z * dst_stride;
    lvl1 += blockIdx.z * dst_stride;
    const int c0 = 4 * p - 2;                       // column of the first of the seven pixels
    const int win = min(max(c0, 0), A - 8);         // first column of the 8-byte window
    // selectors: byte k of the result = byte (column of pixel k, clamped to the row, - win) of the window
    uint32_t sel_lo = 0, sel_hi = 0x0c000000u;      // (0x0c: a zero byte)
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const uint32_t byte = (uint32_t)(min(max(c0 + k, 0), A - 1) - win);
        if (k < 4) sel_lo |= byte << (8 * k);
        else sel_hi |= byte << (8 * (k - 4));
    }
    const int n0 = min(4, A - 4 * p);               // level-0 floats of a row that exist (columns 4p .. 4p + 3)
    const bool second = 2 * p + 1 < A2;             // the thread's second level-1 sample exists
    bytes8 w[2 * PF_ROWS + 3];
#pragma unroll
    for (int j = 0; j < 2 * PF_ROWS + 3; ++j) {
        const int bb = min(max(2 * b_first + j - 2, 0), B - 1);
        w[j] = *reinterpret_cast<const bytes8 *>(src + (size_t)bb * A + win);
    }
    float hA[2 * PF_ROWS + 3], hB[2 * PF_ROWS + 3];
#pragma unroll
    for (int j = 0; j < 2 * PF_ROWS + 3; ++j) {
        const int y = 2 * b_first + j - 2;
        const uint32_t q0 = __builtin_amdgcn_perm(w[j].hi, w[j].lo, sel_lo), q1 = __builtin_amdgcn_perm(w[j].hi, w[j].lo, sel_hi);
        const float p0 = (float)(q0 & 0xffu), p1 = (float)((q0 >> 8) & 0xffu), p2 = (float)((q0 >> 16) & 0xffu),
                    p3 = (float)(q0 >> 24), p4 = (float)(q1 & 0xffu), p5 = (float)((q1 >> 8) & 0xffu),
                    p6 = (float)((q1 >> 16) & 0xffu);
        hA[j] = tap5(p0, p1, p2, p3, p4);
        hB[j] = tap5(p2, p3, p4, p5, p6);
        if (j >= 2 && j < 2 * PF_ROWS + 2 && y < B) {   // an image row of this thread's blocks: pixels 4p .. 4p + 3
            float *o = lvl0 + (size_t)y * A + 4 * p;
            if (n0 == 4) {
                __builtin_nontemporal_store((quad){p2, p3, p4, p5}, reinterpret_cast<quad *>(o));
            } else {
                o[0] = p2;
                if (n0 > 1) o[1] = p3;
                if (n0 > 2) o[2] = p4;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PF_ROWS; ++r) {
        const int b = b_first + r;
        if (b < B2) {
            const float sA = tap5(hA[2 * r], hA[2 * r + 1], hA[2 * r + 2], hA[2 * r + 3], hA[2 * r + 4]);
            const float sB = tap5(hB[2 * r], hB[2 * r + 1], hB[2 * r + 2], hB[2 * r + 3], hB[2 * r + 4]);
            float *o = lvl1 + (size_t)b * A2 + 2 * p;
            if (second) *reinterpret_cast<pair *>(o) = (pair){sA, sB};
            else o[0] = sA;
        }
    }
}

#pragma clang fp contract(fast)

// (int)min(max(t, -1e6), 1e6) for a finite integral t: the conversion instruction saturates, so the
// bounds can be applied to the integer.
__device__ __forceinline__ int bounded_int(double t)
{
    int j;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(j) : "v"(t));
    return min(max(j, -1000000), 1000000);
}

// Wavefronts per SIMD the 21 x 21 kernel is compiled for (register budget), in two builds.  A launch of a few
// thousand tracks lasts as long as its slowest track: the build with the most registers is the fastest (3 per
// SIMD).  A launch that fills the GPU several times over (a group of loops: 16 x 4 000 tracks) is bound by
// the sum of its wavefronts' stalls -- two round trips to memory and some 900 mostly dependent instructions per
// level and direction -- and gains from more of them in flight: 3 / 4 / 5 / 6 per SIMD give 13 640 / 14 680 /
// 15 000 / 14 340 frames/s for a group of 16 (6: 80 registers, 52 bytes of spills per lane), and cost a single loop 1 - 5 %.
#ifndef KLT_WAVES_21
#define KLT_WAVES_21 3
#endif
#ifndef KLT_WAVES_21_FULL
#define KLT_WAVES_21_FULL 5
#endif
constexpr long long KLT_FULL_FROM = 8192;   // tracks per launch from which the second build is used

template <int W>
struct klt_cfg {
    static constexpr int HALF = W / 2;
    static constexpr int NPIX = W * W;
    // window pixel (wb, wa) belongs to lane (wb / RPL) * W + wa: a lane owns RPL vertically
    // consecutive pixels of one window column, so neighbouring pixels share their taps in registers
    static constexpr int GROUPS = 64 / W;                     // row groups (3 for W = 21, 2 for W = 31)
    static constexpr int RPL = (W + GROUPS - 1) / GROUPS;     // rows per lane (7 and 16)
    static constexpr int SI = W + 3;        // template neighbourhood side
    static constexpr int SI_ROWS = GROUPS * RPL + 3;          // rows of LDS behind it (the last ones only feed masked pixels)
    // px the estimate may drift before the region is re-staged (measured: 3 beats 1..11; wider regions
    // cost every track more staging than they save the few wandering ones)
    static constexpr int MARG = 3;
    static constexpr int SJ = W + 2 * MARG + 2;
};

// Stages the S x S patch of a level image (A x B, replicated border) whose first pixel is (a0, b0) in
// LDS, row-major with row length S.  All loads are issued before the first LDS store so that their
// latencies overlap.  For S <= 32 a load instruction takes two patch rows (32 lanes each): the
// clamped column is computed once, a row costs a clamp and a multiply-add; wider patches fall back
// to a linear walk over the patch.
template <int S>
__device__ __forceinline__ void stage_patch(const float *__restrict__ img, int A, int B, int a0, int b0, float *dst)
{
    const int lane = threadIdx.x;
    if constexpr (S <= 32) {
        constexpr int NLD = (S + 1) / 2;
        const int col = lane & 31, half = lane >> 5;
        const int a = min(max(a0 + col, 0), A - 1);
        float tmp[NLD];
        // (a level image has fewer than 2^31 pixels and fewer than 2^24 rows and columns: 32-bit offsets,
        // 24-bit products)
        const int b0u = __builtin_amdgcn_readfirstlane(b0);   // (the same in every lane: a scalar branch)
        if (b0u >= 0 && b0u + 2 * NLD - 1 <= B - 1) {
            // every row of the patch lies inside the image (nearly always): the rows are a constant stride apart
            // (row pair k starts at a wave-uniform address, which the scalar unit advances; the lane's own part
            // of the address, in bytes, is the same for every row pair: one vector instruction per patch instead
            // of two per load)
            const char *rowp = reinterpret_cast<const char *>(img + __mul24(b0u, A));
            const unsigned lane_bytes = 4u * (unsigned)(__mul24(half, A) + a);
            const unsigned stride = 8u * (unsigned)A;
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                tmp[k] = *reinterpret_cast<const float *>(rowp + lane_bytes);
                rowp += stride;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int b = min(max(b0 + 2 * k + half, 0), B - 1);
                tmp[k] = img[(unsigned)(__mul24(b, A) + a)];   // columns / rows past the patch are clamped reads, never stored
            }
        }
        if (col < S) {   // (one branch: only the last row pair can reach past the patch)
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int rb = 2 * k + half;
                if (2 * k + 1 < S || rb < S) dst[rb * S + col] = tmp[k];
            }
        }
    } else {
        constexpr int NLD = (S * S + 63) / 64;
        float tmp[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int q = lane + 64 * k;
            const int rb = q / S, ra = q - rb * S;
            const int a = min(max(a0 + ra, 0), A - 1);
            const int b = min(max(b0 + rb, 0), B - 1);
            tmp[k] = img[(size_t)b * A + a];
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int q = lane + 64 * k;
            if (q < S * S) dst[q] = tmp[k];
        }
    }
}

#ifdef KLT_PHASE_PROBE
// (a measuring build only: shader clock ticks per phase, summed over all tracks of all launches)
// (a track's ticks are collected in its workgroup's LDS and added to one of 256 rows at the end of the track: the probe
// must not serialise the wavefronts it measures)
__device__ unsigned long long g_klt_probe[256][8];
#define KLT_PROBE_NOW() clock64()
#define KLT_PROBE_ADD(slot, ticks) do { if (threadIdx.x == 0) s_klt_probe[slot] += (unsigned long long)(ticks); } while (0)
#else
#define KLT_PROBE_NOW() 0ll
#define KLT_PROBE_ADD(slot, ticks) do { } while (0)
#endif

// Tracks (pa, pb) from pyramid P to pyramid Q.  sI / sJ: this wavefront's LDS.
template <int W>
__device__ bool track_dir(const pyr_desc &P, const pyr_desc &Q, size_t seq_floats, int max_iter, double min_lmin, double pa,
                          double pb, double &qa, double &qb, float *sI, float *sJ, int &iters)
{
#ifdef KLT_PHASE_PROBE
    unsigned long long *s_klt_probe = reinterpret_cast<unsigned long long *>(sJ + klt_cfg<W>::SJ * klt_cfg<W>::SJ + 3);
#endif
    using C = klt_cfg<W>;
    const int lane = threadIdx.x;
    double da = 0.0, db = 0.0;
    bool ok0 = true;
    // this lane's pixels: window column wa, rows wb0 .. wb0 + RPL - 1 (those past the window, and the
    // lanes past the last column group, are masked: they read valid LDS and contribute zeros)
    const bool lane_on = lane < C::GROUPS * W;
    const int grp = lane_on ? lane / W : 0, wa = lane_on ? lane - grp * W : 0, wb0 = grp * C::RPL;
    bool live[C::RPL];
#pragma unroll
    for (int k = 0; k < C::RPL; ++k) live[k] = lane_on && wb0 + k < W;
    const int offI = (wb0 + 1) * C::SI + (wa + 1);   // top-left tap of the lane's first pixel
    const int offJ = wb0 * C::SJ + wa;
    for (int l = P.levels - 1; l >= 0; --l) {
        const int A = P.A[l], B = P.B[l];
        const float *__restrict__ Pl = P.lvl[l] + seq_floats;   // (the serving sequence's pyramids, see vohip_seqs)
        const float *__restrict__ Ql = Q.lvl[l] + seq_floats;
        const double scale = 1.0 / (double)(1 << l);
        const double ca = pa * scale, cb = pb * scale;
        const double fca = floor(ca), fcb = floor(cb);
        const int ia = (int)fca, ib = (int)fcb;
        const float fa = (float)(ca - fca), fb = (float)(cb - fcb);

        // stage the template neighbourhood: raw pixels at offsets -(HALF+1) .. HALF+2
        [[maybe_unused]] const long long probe0 = KLT_PROBE_NOW();
        __syncthreads();
        stage_patch<C::SI>(Pl, A, B, ia - C::HALF - 1, ib - C::HALF - 1, sI);
        __syncthreads();
        [[maybe_unused]] const long long probe1 = KLT_PROBE_NOW();
        [[maybe_unused]] long long probe_stage = 0;
        KLT_PROBE_ADD(0, probe1 - probe0);

        float GX[C::RPL], GY[C::RPL];
        double gxx = 0.0, gxy = 0.0, gyy = 0.0;
        // sum (T - J) G = sum T G - sum J G: the first sum does not change during the iterations, so the
        // lane keeps its part of it (fp32, rows in increasing order) and an iteration only forms sum J G
        float ctx_ = 0.f, cty_ = 0.f;
        {
            // S(q, j): the previous image sampled at window row wb0 + q, column wa - 1 + j.  Tap rows are
            // interpolated along a first (h; tap row r serves the samples of rows r - 1 and r), so
            // S(r - 1, .) is complete at tap row r: pixel r - 1 gets its value and its a-gradient there,
            // pixel r - 2 its b-gradient (S(r - 1, centre) is the sample below it).
            const float *c = sI + offI;
            float h_prev[3], s_prev = 0.f, s_cur = 0.f, sm = 0.f, sp = 0.f;
#pragma unroll
            for (int r = -1; r <= C::RPL + 1; ++r) {
                const float *row = c + r * C::SI;
                const float t0 = row[-1], t1 = row[0], t2 = row[1], t3 = row[2];
                float h[3];
                h[0] = __builtin_fmaf(fa, t1 - t0, t0);
                h[1] = __builtin_fmaf(fa, t2 - t1, t1);
                h[2] = __builtin_fmaf(fa, t3 - t2, t2);
                if (r >= 0) {
                    // samples of row r - 1 (every column) are complete now
                    const float s_next = __builtin_fmaf(fb, h[1] - h_prev[1], h_prev[1]);   // S(r - 1, centre)
                    sm = __builtin_fmaf(fb, h[0] - h_prev[0], h_prev[0]);
                    sp = __builtin_fmaf(fb, h[2] - h_prev[2], h_prev[2]);
                    if (r >= 2) {
                        const int k = r - 2;
                        const float gy = live[k] ? (s_next - s_prev) * 0.5f : 0.f;
                        GY[k] = gy;
                        gyy += (double)gy * (double)gy;
                        gxy += (double)GX[k] * (double)gy;
                        cty_ = __builtin_fmaf(s_cur, gy, cty_);   // (s_cur: pixel k's own value, the sample of row r - 2)
                    }
                    if (r >= 1 && r <= C::RPL) {
                        const int k = r - 1;
                        const float gx = live[k] ? (sp - sm) * 0.5f : 0.f;
                        GX[k] = gx;   // (a masked pixel's gradients are zero: its template value never counts)
                        gxx += (double)gx * (double)gx;
                        ctx_ = __builtin_fmaf(s_next, gx, ctx_);
                    }
                    s_prev = s_cur;
                    s_cur = s_next;
                }
                h_prev[0] = h[0];
                h_prev[1] = h[1];
                h_prev[2] = h[2];
            }
        }
        vohip_dev::wave_sum3_f64(gxx, gxy, gyy);
        const double det = gxx * gyy - gxy * gxy;
        const double lmin = 0.5 * ((gxx + gyy) - sqrt((gxx - gyy) * (gxx - gyy) + 4.0 * gxy * gxy));
        // lambda_min / W^2 >= KLT_MIN_EIG, without the division: min_lmin is the smallest double whose
        // quotient by W^2 rounds to at least KLT_MIN_EIG (the rounded quotient is monotone in lambda_min), so
        // the comparison below decides exactly as the quotient would
        const bool good = lmin >= min_lmin && det > 0.0;
        const double inv_det = 1.0 / det;
        // the inverse of the gradient matrix, formed once per level: a step is then two products and two
        // fused multiply-adds
        const double iaa = gyy * inv_det, iab = gxy * inv_det, ibb = gxx * inv_det;
        if (l == 0) ok0 = good;
        [[maybe_unused]] const long long probe2 = KLT_PROBE_NOW();
        KLT_PROBE_ADD(1, probe2 - probe1);
        if (good) {
            int cja = 0, cjb = 0;
            bool staged = false;
            double pa_prev = 0.0, pb_prev = 0.0;
            // The lane's taps of the current image stay in registers while the integer part of the
            // estimate does not change (most steps are shorter than a pixel): t0 = left tap, dt = right
            // minus left, per tap row.
            double fta_taps = __builtin_nan(""), ftb_taps = __builtin_nan("");
            float t0[C::RPL + 1], dt[C::RPL + 1];
            for (int it = 0; it < max_iter; ++it) {
                const double ta = ca + da, tb = cb + db;
                const double fta = floor(ta), ftb = floor(tb);
                const float ga = (float)(ta - fta), gb = (float)(tb - ftb);
                if (__builtin_amdgcn_ballot_w64(fta == fta_taps && ftb == ftb_taps) == 0) {   // (wave-uniform)
                    fta_taps = fta;
                    ftb_taps = ftb;
                    // positions far outside the image cannot come back; keep integer math bounded
                    const int ja = bounded_int(fta), jb = bounded_int(ftb);
                    if (!staged || abs(ja - cja) > C::MARG || abs(jb - cjb) > C::MARG) {
                        // an estimate that walked out of the region is likely to walk on: the new region is
                        // centred ahead of it, in the direction it left by (where a region lies does not
                        // change what is read from it)
                        const int la = staged ? min(max(ja - cja, -C::MARG), C::MARG) : 0;
                        const int lb = staged ? min(max(jb - cjb, -C::MARG), C::MARG) : 0;
                        cja = ja + la;
                        cjb = jb + lb;
                        staged = true;
                        [[maybe_unused]] const long long ps = KLT_PROBE_NOW();
                        __syncthreads();
                        stage_patch<C::SJ>(Ql, A, B, cja - C::HALF - C::MARG, cjb - C::HALF - C::MARG, sJ);
                        __syncthreads();
                        probe_stage += KLT_PROBE_NOW() - ps;
                    }
                    const float *org = sJ + __mul24(jb - cjb + C::MARG, C::SJ) + (ja - cja + C::MARG) + offJ;
                    float t1[C::RPL + 1];
#pragma unroll
                    for (int r = 0; r <= C::RPL; ++r) {   // all LDS reads first, so that their latencies overlap
                        t0[r] = org[r * C::SJ];
                        t1[r] = org[r * C::SJ + 1];
                    }
                    // (one subtraction each: left to itself the compiler pairs them up for the packed
                    // instruction and spends three moves per pair on lining the operands up)
#pragma unroll
                    for (int r = 0; r <= C::RPL; ++r) asm("v_sub_f32 %0, %1, %2" : "=v"(dt[r]) : "v"(t1[r]), "v"(t0[r]));
                }
                // per-lane partial sums in fp32 (RPL fused multiply-adds, rows in increasing order), then
                // a fixed fp32 tree across the wavefront
                float h[C::RPL + 1], sjx = 0.f, sjy = 0.f;
#pragma unroll
                for (int r = 0; r <= C::RPL; ++r) h[r] = __builtin_fmaf(ga, dt[r], t0[r]);
#pragma unroll
                for (int k = 0; k < C::RPL; ++k) {
                    const float j = __builtin_fmaf(gb, h[k + 1] - h[k], h[k]);
                    sjx = __builtin_fmaf(j, GX[k], sjx);
                    sjy = __builtin_fmaf(j, GY[k], sjy);
                }
                float baf = ctx_ - sjx, bbf = cty_ - sjy;
                vohip_dev::wave_sum2_f32(baf, bbf);
                const double ba = (double)baf, bb = (double)bbf;
                const double sa = __builtin_fma(-iab, bb, iaa * ba);
                const double sb = __builtin_fma(-iab, ba, ibb * bb);
                da += sa;
                db += sb;
                ++iters;
                if (sa * sa + sb * sb < KLT_EPS_STEP * KLT_EPS_STEP) break;
                // a step that undoes the previous one: the estimate oscillates around the minimum;
                // settle half-way and stop
                if (it > 0 && (sa + pa_prev) * (sa + pa_prev) + (sb + pb_prev) * (sb + pb_prev) <
                                  KLT_EPS_STEP * KLT_EPS_STEP) {
                    da -= 0.5 * sa;
                    db -= 0.5 * sb;
                    break;
                }
                pa_prev = sa;
                pb_prev = sb;
            }
        }
        {
            [[maybe_unused]] const long long probe3 = KLT_PROBE_NOW();
            KLT_PROBE_ADD(2, probe_stage);
            KLT_PROBE_ADD(3, probe3 - probe2 - probe_stage);
            KLT_PROBE_ADD(4, 1);
            (void)probe3;
        }
        if (l > 0) {
            da *= 2.0;
            db *= 2.0;
        }
    }
    qa = pa + da;
    qb = pb + db;
    return ok0;
}

// pts_xy / out_xy: x column then y column with leading dimension ld, MATLAB's (x, y), 1-based.
// swap_xy: the image is a column-major MATLAB image, so a = y - 1, b = x - 1; otherwise a = x - 1,
// b = y - 1.  A launch carries up to three independent point sets between the same two images
// (the frame pipeline tracks landmarks, candidates and fresh detections in one launch): workgroups
// past one set serve the next.
struct klt_sets {
    vohip_klt_set s[3];
};

template <int W, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES)))
void klt_track_kernel(pyr_desc P, pyr_desc Q, klt_sets S, int swap_xy, int max_iter, double max_bidir, double min_lmin,
                      unsigned long long *__restrict__ stats, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    const size_t seq_floats = (size_t)(SQ.stride_prep * (unsigned long long)seq__) / sizeof(float);
    using C = klt_cfg<W>;
    __shared__ float sI[C::SI_ROWS * C::SI + 8];
#ifdef KLT_PHASE_PROBE
    __shared__ __align__(8) float sJ[C::SJ * C::SJ + 3 + 16];   // (+ the probe's five 64-bit sums behind the region)
    {
        unsigned long long *pr = reinterpret_cast<unsigned long long *>(sJ + C::SJ * C::SJ + 3);
        if (threadIdx.x < 8) pr[threadIdx.x] = 0;
        __syncthreads();
    }
#else
    __shared__ float sJ[C::SJ * C::SJ];
#endif
    int i = blockIdx.x, k = 0;
    if (i >= S.s[0].n) {
        i -= S.s[0].n;
        k = 1;
        if (i >= S.s[1].n) {
            i -= S.s[1].n;
            k = 2;
        }
    }
    // (selected field by field: the sets are kernel arguments, an indexed copy would go through scratch)
#define KLT_PICK(field) (k == 0 ? S.s[0].field : k == 1 ? S.s[1].field : S.s[2].field)
    const double *__restrict__ pts_xy = KLT_PICK(pts_xy);
    double *__restrict__ out_xy = KLT_PICK(out_xy);
    uint8_t *__restrict__ valid = KLT_PICK(valid);
    double *__restrict__ fb_err = KLT_PICK(fb_err);
    const int32_t *__restrict__ rows = KLT_PICK(rows);
    const uint8_t *__restrict__ live = KLT_PICK(live);
    const int *__restrict__ n_dev = KLT_PICK(n_dev);
    const int *__restrict__ first_dev = KLT_PICK(first_dev);
    const int ld = KLT_PICK(ld), base = KLT_PICK(base), out_base = KLT_PICK(out_base);
    int n = KLT_PICK(n);
#undef KLT_PICK
    VOHIP_SEQ_STATE(pts_xy);
    VOHIP_SEQ_STATE(out_xy);
    VOHIP_SEQ_STATE(valid);
    VOHIP_SEQ_STATE_OPT(fb_err);
    VOHIP_SEQ_STATE_OPT(rows);
    VOHIP_SEQ_STATE_OPT(live);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE_OPT(first_dev);
    if (first_dev) {   // tracks [*first_dev, *n_dev) of the list; n only sizes the grid
        i += *first_dev;
        n = *n_dev;
    } else if (n_dev) {
        n = min(n, *n_dev);
    }
    if (i >= n) return;
    const long long t_start = stats ? clock64() : 0;
    const int r = rows ? rows[i] : base + i;
    const int o = out_base < 0 ? r : out_base + i;
    if (live && !live[r]) {
        if (threadIdx.x == 0) valid[o] = 0;
        return;
    }
    const double x = pts_xy[r] - 1.0, y = pts_xy[ld + r] - 1.0;
    const double pa = swap_xy ? y : x, pb = swap_xy ? x : y;
    double qa, qb, ra, rb;
    int iters = 0;
    const bool ok_f = track_dir<W>(P, Q, seq_floats, max_iter, min_lmin, pa, pb, qa, qb, sI, sJ, iters);
    const bool inside = qa >= 0.0 && qa <= (double)(P.A[0] - 1) && qb >= 0.0 && qb <= (double)(P.B[0] - 1);
    // A track that is already invalid after the forward pass (untrackable at level 0, or outside the image)
    // is not tracked back: its forward-backward distance is reported as infinite.  (qa, qb are the same in
    // every lane, so the whole wavefront takes the same side.)
    bool ok_b = false;
    double err = __builtin_inf();
    if (ok_f && inside) {
        ok_b = track_dir<W>(Q, P, seq_floats, max_iter, min_lmin, qa, qb, ra, rb, sI, sJ, iters);
        const double ea = ra - pa, eb = rb - pb;
        err = sqrt(ea * ea + eb * eb);
    }
    if (threadIdx.x == 0) {
        out_xy[o] = (swap_xy ? qb : qa) + 1.0;
        out_xy[ld + o] = (swap_xy ? qa : qb) + 1.0;
        valid[o] = (ok_f && ok_b && inside && err <= max_bidir) ? 1 : 0;
        if (fb_err) fb_err[o] = err;
#ifdef KLT_PHASE_PROBE
        {
            const unsigned long long *pr = reinterpret_cast<const unsigned long long *>(sJ + C::SJ * C::SJ + 3);
            for (int q = 0; q < 5; ++q) atomicAdd(&g_klt_probe[blockIdx.x & 255][q], pr[q]);
        }
#endif
        if (stats) {   // work counters: tracks, Gauss-Newton iterations (both directions, all levels)
            atomicAdd(stats, 1ull);
            atomicAdd(stats + 1, (unsigned long long)iters);
            atomicMax(stats + 2, (unsigned long long)(clock64() - t_start));   // longest track, shader clock ticks
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// The smallest double x with x / W^2 >= KLT_MIN_EIG as the division rounds it.
static double lmin_threshold(int W)
{
    const double d = (double)(W * W);
    double c = KLT_MIN_EIG * d;
    while (c / d >= KLT_MIN_EIG) c = std::nextafter(c, 0.0);
    while (!(c / d >= KLT_MIN_EIG)) c = std::nextafter(c, INFINITY);
    return c;
}

int vohip_pyramid_desc(vohip_ctx *ctx, int slot, pyr_desc &d)
{
    VOHIP_REQUIRE(ctx, slot >= 0 && slot < VOHIP_PYR_SLOTS, "pyramid slot out of range");
    const vohip_pyramid &p = ctx->pyr[slot];
    VOHIP_REQUIRE(ctx, p.levels > 0, "pyramid slot is empty");
    d.levels = p.levels;
    const float *base = ctx->pyr_ext[slot] ? ctx->pyr_ext[slot] : (const float *)ctx->bufs[B_PYR0 + slot].ptr;
    for (int l = 0; l < MAX_LEVELS; ++l) {
        d.lvl[l] = l < p.levels ? base + p.offset[l] : nullptr;
        d.A[l] = l < p.levels ? p.A[l] : 0;
        d.B[l] = l < p.levels ? p.B[l] : 0;
    }
    return VOHIP_OK;
}

// Level l of a pyramid starts `offset[l]` floats into it (levels back to back, 64-float aligned).
static size_t pyramid_layout(int rows, int cols, int colmajor, int levels, int *As, int *Bs, size_t *offset)
{
    size_t total = 0;
    int A = colmajor ? rows : cols, B = colmajor ? cols : rows;
    for (int l = 0; l < levels; ++l) {
        if (As) As[l] = A;
        if (Bs) Bs[l] = B;
        if (offset) offset[l] = total;
        total += ((size_t)A * B + 63) / 64 * 64;
        A = (A + 1) / 2;
        B = (B + 1) / 2;
    }
    return total;
}

int vohip_pyramid_attach(vohip_ctx *ctx, int slot, const float *d_pyr, int rows, int cols, int colmajor, int levels)
{
    VOHIP_REQUIRE(ctx, slot >= 0 && slot < VOHIP_PYR_SLOTS, "pyramid slot out of range");
    VOHIP_REQUIRE(ctx, d_pyr && rows >= 1 && cols >= 1 && levels >= 1 && levels <= MAX_LEVELS, "bad pyramid");
    vohip_pyramid &p = ctx->pyr[slot];
    p.rows = rows;
    p.cols = cols;
    p.colmajor = colmajor ? 1 : 0;
    p.levels = levels;
    pyramid_layout(rows, cols, colmajor, levels, p.A, p.B, p.offset);
    ctx->pyr_ext[slot] = d_pyr;
    return VOHIP_OK;
}

extern "C" {

int vohip_pyramid_build_dev(vohip_ctx *ctx, int slot, const uint8_t *d_img, int rows, int cols, int colmajor,
                            int levels)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, slot >= 0 && slot < VOHIP_PYR_SLOTS, "pyramid slot out of range");
    VOHIP_REQUIRE(ctx, d_img && rows >= 1 && cols >= 1, "bad image");
    VOHIP_REQUIRE(ctx, levels >= 1 && levels <= MAX_LEVELS, "levels must be 1..8");
    vohip_pyramid &p = ctx->pyr[slot];
    p.rows = rows;
    p.cols = cols;
    p.colmajor = colmajor ? 1 : 0;
    p.levels = levels;
    size_t total = 0;
    int A = colmajor ? rows : cols, B = colmajor ? cols : rows;
    for (int l = 0; l < levels; ++l) {
        p.A[l] = A;
        p.B[l] = B;
        p.offset[l] = total;
        total += ((size_t)A * B + 63) / 64 * 64;
        A = (A + 1) / 2;
        B = (B + 1) / 2;
    }
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_PYR0 + slot], total * sizeof(float)));
    ctx->pyr_ext[slot] = nullptr;
    float *base = (float *)ctx->bufs[B_PYR0 + slot].ptr;
    const int n0 = p.A[0] * p.B[0];
    if (levels == 1)
        hipLaunchKernelGGL(u8_to_float_kernel, dim3((n0 + 255) / 256), dim3(256), 0, ctx->stream, d_img, n0, base, (size_t)0,
                           (size_t)0);
    else   // levels 0 and 1 in one pass over the image
        hipLaunchKernelGGL(pyr_first_kernel, dim3((p.A[1] + 255) / 256, (p.B[1] + PF_ROWS - 1) / PF_ROWS), dim3(256), 0, ctx->stream, d_img,
                           p.A[0], p.B[0], base, base + p.offset[1], p.A[1], p.B[1], (size_t)0, (size_t)0);
    for (int l = 2; l < levels; ++l)
        hipLaunchKernelGGL(pyr_down_kernel<1>, dim3((p.A[l] + 63) / 64, (p.B[l] + 3) / 4), dim3(256), 0, ctx->stream,
                           base + p.offset[l - 1], p.A[l - 1], p.B[l - 1], base + p.offset[l], p.A[l], p.B[l], (size_t)0);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int vohip_pyramid_build(vohip_ctx *ctx, int slot, const uint8_t *img, int rows, int cols, int colmajor, int levels)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, img && rows >= 1 && cols >= 1, "bad image");
    void *d_img;
    VOHIP_TRY(vohip_upload(ctx, B_IMG_STAGE, img, (size_t)rows * cols, &d_img));
    return vohip_pyramid_build_dev(ctx, slot, (const uint8_t *)d_img, rows, cols, colmajor, levels);
}

size_t vohip_pyramid_floats(int rows, int cols, int colmajor, int levels)
{
    if (rows < 1 || cols < 1 || levels < 1 || levels > MAX_LEVELS) return 0;
    return pyramid_layout(rows, cols, colmajor, levels, nullptr, nullptr, nullptr);
}

int vohip_pyramid_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor, int levels,
                            float *d_pyramids)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_imgs && d_pyramids && rows >= 1 && cols >= 1, "bad image");
    VOHIP_REQUIRE(ctx, levels >= 1 && levels <= MAX_LEVELS, "levels must be 1..8");
    VOHIP_REQUIRE(ctx, n >= 1 && n <= 65535, "1..65535 images per call");
    int A[MAX_LEVELS], B[MAX_LEVELS];
    size_t off[MAX_LEVELS];
    const size_t total = pyramid_layout(rows, cols, colmajor, levels, A, B, off);
    const int n0 = A[0] * B[0];
    if (levels == 1)
        hipLaunchKernelGGL(u8_to_float_kernel, dim3((n0 + 255) / 256, n), dim3(256), 0, ctx->stream, d_imgs, n0, d_pyramids,
                           (size_t)n0, total);
    else if (A[0] >= 8)
        hipLaunchKernelGGL(pyr_first_pairs_kernel, dim3(((A[1] + 1) / 2 + 63) / 64, (B[1] + 4 * PF_ROWS - 1) / (4 * PF_ROWS), n), dim3(256), 0,
                           ctx->stream, d_imgs, A[0], B[0], d_pyramids, d_pyramids + off[1], A[1], B[1], (size_t)n0, total);
    else
        hipLaunchKernelGGL(pyr_first_kernel, dim3((A[1] + 255) / 256, (B[1] + PF_ROWS - 1) / PF_ROWS, n), dim3(256), 0, ctx->stream, d_imgs,
                           A[0], B[0], d_pyramids, d_pyramids + off[1], A[1], B[1], (size_t)n0, total);
    for (int l = 2; l < levels; ++l)
        hipLaunchKernelGGL(pyr_down_kernel<4>, dim3((A[l] + 63) / 64, (B[l] + 15) / 16, n), dim3(256), 0, ctx->stream,
                           (const float *)(d_pyramids + off[l - 1]), A[l - 1], B[l - 1], d_pyramids + off[l], A[l], B[l], total);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int vohip_pyramid_level(vohip_ctx *ctx, int slot, int level, float *out, int *A, int *B)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    pyr_desc d;
    VOHIP_TRY(vohip_pyramid_desc(ctx, slot, d));
    VOHIP_REQUIRE(ctx, level >= 0 && level < d.levels && A && B, "bad level");
    *A = d.A[level];
    *B = d.B[level];
    if (out) {
        VOHIP_TRY(vohip_download(ctx, out, d.lvl[level], sizeof(float) * (size_t)d.A[level] * d.B[level]));
        VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return VOHIP_OK;
}

int vohip_klt_track_dev(vohip_ctx *ctx, int slot_prev, int slot_cur, const double *d_pts_xy, int n, int block_size,
                        int max_iter, double max_bidir_error, double *d_out_xy, uint8_t *d_valid, double *d_fb_err)
{
    return vohip_klt_enqueue(ctx, slot_prev, slot_cur, d_pts_xy, n, n, nullptr, block_size, max_iter, max_bidir_error,
                             d_out_xy, d_valid, d_fb_err);
}

}  // extern "C"

int vohip_klt_enqueue(vohip_ctx *ctx, int slot_prev, int slot_cur, const double *d_pts_xy, int ld, int n,
                      const int *d_n, int block_size, int max_iter, double max_bidir_error, double *d_out_xy,
                      uint8_t *d_valid, double *d_fb_err)
{
    const vohip_klt_set set = {d_pts_xy, d_out_xy, d_valid, d_fb_err, nullptr, nullptr, d_n, ld, n, 0, -1, nullptr};
    return vohip_klt_enqueue_sets(ctx, slot_prev, slot_cur, &set, 1, block_size, max_iter, max_bidir_error);
}

int vohip_klt_enqueue_sets(vohip_ctx *ctx, int slot_prev, int slot_cur, const vohip_klt_set *sets, int nsets,
                           int block_size, int max_iter, double max_bidir_error, const vohip_seqs *seqs)
{
    const vohip_seqs SQ = seqs ? *seqs : vohip_seqs();
    const int ny = vohip_seq_count(SQ);
    if (ny < 1) return VOHIP_OK;
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, sets && nsets >= 1 && nsets <= 3 && max_iter >= 1, "bad argument");
    klt_sets S;
    std::memset(&S, 0, sizeof(S));
    int total = 0;
    for (int q = 0; q < nsets; ++q) {
        const vohip_klt_set &t = sets[q];
        VOHIP_REQUIRE(ctx, t.n >= 0 && t.base >= 0 && (t.rows || t.ld >= t.base + t.n), "bad point set");
        VOHIP_REQUIRE(ctx, t.n == 0 || (t.pts_xy && t.out_xy && t.valid), "NULL array");
        VOHIP_REQUIRE(ctx, !t.first_dev || (t.n_dev && t.rows), "a device-side start needs a device-side end and a row list");
        S.s[q] = t;
        total += t.n;
    }
    if (total == 0) return VOHIP_OK;
    pyr_desc P, Q;
    VOHIP_TRY(vohip_pyramid_desc(ctx, slot_prev, P));
    VOHIP_TRY(vohip_pyramid_desc(ctx, slot_cur, Q));
    const vohip_pyramid &pp = ctx->pyr[slot_prev], &pq = ctx->pyr[slot_cur];
    VOHIP_REQUIRE(ctx, pp.rows == pq.rows && pp.cols == pq.cols && pp.colmajor == pq.colmajor && pp.levels == pq.levels,
                  "previous and current pyramids differ in shape");
    const int swap_xy = pp.colmajor;
    // the work counters cost two same-address atomics per track: only on request
    if (ctx->opt_klt_stats && !ctx->bufs[B_KLT_STATS].ptr) {
        VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_KLT_STATS], 64));
        VOHIP_HIP(ctx, hipMemsetAsync(ctx->bufs[B_KLT_STATS].ptr, 0, 64, ctx->stream));
        VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    unsigned long long *d_stats = ctx->opt_klt_stats ? (unsigned long long *)ctx->bufs[B_KLT_STATS].ptr : nullptr;
    // (measurement only: an event pair around the launch, on the stream it runs on)
    const bool timed = ctx->opt_klt_time && ctx->klt_ev && ctx->klt_ev_used < vohip_ctx::KLT_TIME_PAIRS;
    if (timed) VOHIP_HIP(ctx, hipEventRecord(ctx->klt_ev[2 * ctx->klt_ev_used], ctx->stream));
    // (sets whose counts live on the device are sized by an upper bound: how many tracks a single loop's launch
    // really carries -- a few thousand -- is not known here, a group's launches carry thousands per sequence)
    bool bound_only = false;
    for (int q = 0; q < nsets; ++q) bound_only = bound_only || sets[q].n_dev != nullptr;
    const bool full = bound_only ? ny >= 2 : (long long)total * ny >= KLT_FULL_FROM;
    if (block_size == 21 && full)
        hipLaunchKernelGGL((klt_track_kernel<21, KLT_WAVES_21_FULL>), dim3(total, ny), dim3(64), 0, ctx->stream, P, Q, S, swap_xy,
                           max_iter, max_bidir_error, lmin_threshold(21), d_stats, SQ);
    else if (block_size == 21)
        hipLaunchKernelGGL((klt_track_kernel<21, KLT_WAVES_21>), dim3(total, ny), dim3(64), 0, ctx->stream, P, Q, S, swap_xy,
                           max_iter, max_bidir_error, lmin_threshold(21), d_stats, SQ);
    else if (block_size == 31)
        hipLaunchKernelGGL((klt_track_kernel<31, 2>), dim3(total, ny), dim3(64), 0, ctx->stream, P, Q, S, swap_xy, max_iter,
                           max_bidir_error, lmin_threshold(31), d_stats, SQ);
    else
        return vohip_fail(ctx, VOHIP_ERR_INVALID_ARGUMENT, "BlockSize %d not built (21 and 31 are)", block_size);
    VOHIP_HIP(ctx, hipGetLastError());
    if (timed) {
        VOHIP_HIP(ctx, hipEventRecord(ctx->klt_ev[2 * ctx->klt_ev_used + 1], ctx->stream));
        ctx->klt_ev_used += 1;
    }
    return VOHIP_OK;
}

extern "C" {

int vohip_klt_stats(vohip_ctx *ctx, long long *tracks, long long *iterations, int reset)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    unsigned long long h[3] = {0, 0, 0};
    if (ctx->bufs[B_KLT_STATS].ptr) {
        VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        VOHIP_HIP(ctx, hipDeviceSynchronize());   // trackers may run on other streams of this context's owner
        VOHIP_HIP(ctx, hipMemcpy(h, ctx->bufs[B_KLT_STATS].ptr, sizeof(h), hipMemcpyDeviceToHost));
        if (reset) {   // the fill may run asynchronously: finish it before a tracker can be launched
            VOHIP_HIP(ctx, hipMemset(ctx->bufs[B_KLT_STATS].ptr, 0, 64));
            VOHIP_HIP(ctx, hipDeviceSynchronize());
        }
    }
    if (std::getenv("VOHIP_KLT_CYCLES")) std::fprintf(stderr, "vohip klt: longest track %llu ticks\n", h[2]);
#ifdef KLT_PHASE_PROBE
    {
        static unsigned long long rows[256][8];
        unsigned long long pr[8] = {0};
        (void)hipMemcpyFromSymbol(rows, HIP_SYMBOL(g_klt_probe), sizeof(rows));
        for (int r = 0; r < 256; ++r)
            for (int q = 0; q < 8; ++q) pr[q] += rows[r][q];
        const double n = pr[4] ? (double)pr[4] : 1.0;
        std::fprintf(stderr, "vohip klt probe: %llu passes; ticks per pass: template staging %.0f, template + G %.0f, "
                     "region staging %.0f, iterations %.0f\n", pr[4], pr[0] / n, pr[1] / n, pr[2] / n, pr[3] / n);
        if (reset) {
            std::memset(rows, 0, sizeof(rows));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_klt_probe), rows, sizeof(rows));
        }
    }
#endif
    if (tracks) *tracks = (long long)h[0];
    if (iterations) *iterations = (long long)h[1];
    return VOHIP_OK;
}

int vohip_klt_time(vohip_ctx *ctx, double *total_ms, double *longest_ms, int *launches, int reset)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    VOHIP_HIP(ctx, hipDeviceSynchronize());   // (tracker launches run on several streams of a frame loop)
    double sum = 0.0, longest = 0.0;
    for (int k = 0; k < ctx->klt_ev_used; ++k) {
        float ms = 0.f;
        VOHIP_HIP(ctx, hipEventElapsedTime(&ms, ctx->klt_ev[2 * k], ctx->klt_ev[2 * k + 1]));
        sum += ms;
        longest = ms > longest ? ms : longest;
    }
    if (total_ms) *total_ms = sum;
    if (longest_ms) *longest_ms = longest;
    if (launches) *launches = ctx->klt_ev_used;
    if (reset) ctx->klt_ev_used = 0;
    return VOHIP_OK;
}

int vohip_klt_track(vohip_ctx *ctx, int slot_prev, int slot_cur, const double *pts_xy, int n, int block_size,
                    int max_iter, double max_bidir_error, double *out_xy, uint8_t *valid, double *fb_err)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, n >= 0, "bad argument");
    if (n == 0) return VOHIP_OK;
    VOHIP_REQUIRE(ctx, pts_xy && out_xy && valid, "NULL array");
    void *d_in, *d_out, *d_valid, *d_fb;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, pts_xy, sizeof(double) * 2 * n, &d_in));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, nullptr, sizeof(double) * 2 * n, &d_out));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, (size_t)n, &d_valid));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, sizeof(double) * n, &d_fb));
    VOHIP_TRY(vohip_klt_track_dev(ctx, slot_prev, slot_cur, (const double *)d_in, n, block_size, max_iter,
                                  max_bidir_error, (double *)d_out, (uint8_t *)d_valid, (double *)d_fb));
    VOHIP_TRY(vohip_download(ctx, out_xy, d_out, sizeof(double) * 2 * n));
    VOHIP_TRY(vohip_download(ctx, valid, d_valid, (size_t)n));
    if (fb_err) VOHIP_TRY(vohip_download(ctx, fb_err, d_fb, sizeof(double) * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

}  // extern "C"
