// BRISK keypoints for gfx950.
//
// Stands in for detectBRISKFeatures as genKeypoints.m:20,31 calls it (one of main_eval.m:19's feature variants,
// options {'MinQuality', 2e-3}: main_eval.m:170-171).  The toolbox internals are not public; the detector is
// defined after section 3.1 of Leutenegger, Chli, Siegwart, "BRISK: Binary Robust Invariant Scalable Keypoints"
// (ICCV 2011):
//   layers k = 0 .. 2 NumOctaves - 1 in the order of their scale sigma_k = 1, 1.5, 2, 3, 4, 6, 8, 12: the octaves
//        c_0 = image, c_i = half(c_i-1) (even k) and the intra-octaves d_0 = twothirds(c_0), d_i = half(d_i-1) (odd k);
//        half: (2 x 2 block summed + 2) / 4; twothirds: every 3 x 3 block -> 2 x 2 pixels, each the area-weighted
//        mean of the 1.5 x 1.5 pixels it covers, (4 corner + 2 edge + 2 edge + centre + 4) / 9;
//        pixel (x, y) of layer k lies at ((x + .5) sigma_k - .5, (y + .5) sigma_k - .5) of the image;
//   S_k = the FAST 9-16 score of layer k (0 below t = round(255 MinContrast)), S_-1 = the FAST 5-8 score of layer 0;
//   keypoint <=> s = S_k(x, y) > 0 strictly greater than its 8 neighbours in S_k and than every S_k'(x', y') of the
//        layers k' = k -+ 1 over the pixels whose centres lie within the 3 x 3 patch of layer k around (x, y),
//        s > MinQuality * max over all S_k, the pixel's centre in the image at least `border` pixels from every edge;
//   position: per axis the vertex of the parabola through three scores (at most half a pixel away), mapped to the
//        image; Scale: the vertex of the parabola through the scores of the three layers over their scales (layer 0:
//        (2/3, S_-1) below; the top layer: its own scale); Metric = s;  output order: layer, then pixel order.
// Exact integer arithmetic up to the refinement, which is fp32 / fp64 in a fixed order without contraction.
// Kernels:
//   brisk_layers_kernel    all layers above the image in one launch, a thread per output pixel: the nested means
//                          straight from the image (integer functions: the values of a layer by layer construction);
//   brisk_sample_kernel    for a batch of images: one layer from the one it is made of, seven launches
//   brisk_score_kernel     all layers in one launch: 64 x 16 tiles with a 3 px halo staged in LDS as integers; every
//                          pixel takes the four-lookup compass test a passing pixel cannot fail, the few that pass
//                          queue up in LDS and get the full segment-test score, the others score 0 -> one byte per
//                          pixel; layer 0's 5-8 scores from the same tile; the maximum of all scores by one atomic
//                          per workgroup
//   brisk_extrema_kernel   all layers in one launch, a wavefront per span of 64 pixels of a row (it ends at once when
//                          the score pass has marked the span as holding no score above 0): the tests above from
//                          the score maps (L2-resident bytes), the hits become records in pixel order (ballot ranks),
//                          counted per span and, by an integer atomic, per (layer, row) cell
//   brisk_emit_kernel      a wavefront per non-empty span ranks it behind the cells and spans before it and copies
//                          its records: ordered and deterministic
// Per image: the uint8 image read twice, ~1 byte per pixel of layers written and read, 3.2 bytes of scores per
// pixel written once and read ~once, 16 bytes per keypoint written twice.
#include "vohip_internal.h"

#include <cmath>

#include "fast_score.h"

namespace {

constexpr int BK_LAYERS = 8, BK_SUB_CAP = 32;   // strict maxima are never neighbours: at most 32 per span of 64
constexpr int BK_TA = 64, BK_TB = 16, BK_FA = BK_TA + 6, BK_FB = BK_TB + 6;   // score tiles; staged with the circle's reach
constexpr int BK_BATCH_BYTES_MAX = 1 << 30;

struct brisk_point {   // 16 bytes
    float pa, pb, metric, scale;
};

struct brisk_geom {
    int nl;
    int A[BK_LAYERS], B[BK_LAYERS];
    int sigma2[BK_LAYERS];              // twice the scale of layer k
    unsigned px_off[BK_LAYERS];         // layer k's pixels within an image's pyramid block (layer 0: the image itself)
    unsigned sc_off[BK_LAYERS + 1];     // layer k's scores within an image's score block; [8]: layer 0's 5-8 scores
    int tiles_a[BK_LAYERS], spans[BK_LAYERS];
    int tile_first[BK_LAYERS + 1];      // workgroups of the score launch before layer k
    int xt_first[BK_LAYERS + 1];        // workgroups of the extrema launch before layer k (a span of a row per wavefront)
    int cell_first[BK_LAYERS + 1];      // (layer, row) cells before layer k
    int sub_first[BK_LAYERS + 1];       // (layer, row, span) sub-cells before layer k
    // per image of a batch
    size_t img_stride, px_stride, sc_stride, cell_stride, sub_stride, rec_stride;
    int xy_stride, metric_stride, count_stride;
};

brisk_geom make_geom(int A, int B, int num_octaves)
{
    brisk_geom g{};
    g.nl = num_octaves < 1 ? 1 : (num_octaves > 4 ? BK_LAYERS : 2 * num_octaves);
    static const int s2[BK_LAYERS] = {2, 3, 4, 6, 8, 12, 16, 24};
    unsigned px = 0, sc = 0;
    int tiles = 0, xt = 0, cells = 0, subs = 0;
    auto r64 = [](size_t v) { return (unsigned)((v + 63) / 64 * 64); };
    for (int k = 0; k < BK_LAYERS; ++k) {
        g.sigma2[k] = s2[k];
        if (k == 0) {
            g.A[k] = A;
            g.B[k] = B;
        } else if (k == 1) {
            g.A[k] = 2 * (A / 3);
            g.B[k] = 2 * (B / 3);
        } else {
            g.A[k] = g.A[k - 2] / 2;
            g.B[k] = g.B[k - 2] / 2;
        }
        if (k >= g.nl) g.A[k] = g.B[k] = 0;
        g.px_off[k] = px;
        if (k > 0) px += r64((size_t)g.A[k] * g.B[k]);
        g.sc_off[k] = sc;
        sc += r64((size_t)g.A[k] * g.B[k]);
        g.tiles_a[k] = (g.A[k] + BK_TA - 1) / BK_TA;
        g.spans[k] = (g.A[k] + 63) / 64;
        g.tile_first[k] = tiles;
        tiles += g.tiles_a[k] * ((g.B[k] + BK_TB - 1) / BK_TB);
        g.xt_first[k] = xt;
        xt += g.spans[k] * ((g.B[k] + 3) / 4);
        g.cell_first[k] = cells;
        cells += g.B[k];
        g.sub_first[k] = subs;
        subs += g.B[k] * g.spans[k];
    }
    g.sc_off[BK_LAYERS] = sc;
    sc += r64((size_t)A * B);
    g.tile_first[BK_LAYERS] = tiles;
    g.xt_first[BK_LAYERS] = xt;
    g.cell_first[BK_LAYERS] = cells;
    g.sub_first[BK_LAYERS] = subs;
    auto r256 = [](size_t v) { return (v + 255) / 256 * 256; };
    g.img_stride = (size_t)A * B;
    g.px_stride = r256(px);
    g.sc_stride = r256(sc);
    g.cell_stride = r256(sizeof(int) * (size_t)(cells + 1)) / sizeof(int);   // (one more: the image's maximum score)
    g.sub_stride = r256(sizeof(int) * (size_t)subs) / sizeof(int);
    g.rec_stride = (size_t)subs * BK_SUB_CAP;
    return g;
}

size_t image_scratch_bytes(const brisk_geom &g)
{
    return g.px_stride + g.sc_stride + sizeof(int) * (g.cell_stride + g.sub_stride) + sizeof(brisk_point) * g.rec_stride;
}

// A pixel of an octave (L halvings of the image) or of an intra-octave (the 2/3 resampling, then L halvings) straight
// from the image: the nested means are integer functions of the pixels below them, so the value is the one the layer
// by layer construction gives, and all seven layers come out of one launch (an image's 0.47 MB stay in the caches).
template <int L>
__device__ __forceinline__ int octave_px(const uint8_t *__restrict__ img, int A, int x, int y)
{
    if constexpr (L == 0) {
        return (int)img[(size_t)y * A + x];
    } else {
        return (octave_px<L - 1>(img, A, 2 * x, 2 * y) + octave_px<L - 1>(img, A, 2 * x + 1, 2 * y) +
                octave_px<L - 1>(img, A, 2 * x, 2 * y + 1) + octave_px<L - 1>(img, A, 2 * x + 1, 2 * y + 1) + 2) / 4;
    }
}
template <int L>
__device__ __forceinline__ int intra_px(const uint8_t *__restrict__ img, int A, int x, int y)
{
    if constexpr (L == 0) {
        const int a0 = 3 * (x / 2) + 2 * (x % 2), b0 = 3 * (y / 2) + 2 * (y % 2);   // the pixel covered in full
        const int am = 3 * (x / 2) + 1, bm = 3 * (y / 2) + 1;                       // the middle of the 3 x 3 block
        return (4 * (int)img[(size_t)b0 * A + a0] + 2 * (int)img[(size_t)b0 * A + am] + 2 * (int)img[(size_t)bm * A + a0] +
                (int)img[(size_t)bm * A + am] + 4) / 9;
    } else {
        return (intra_px<L - 1>(img, A, 2 * x, 2 * y) + intra_px<L - 1>(img, A, 2 * x + 1, 2 * y) +
                intra_px<L - 1>(img, A, 2 * x, 2 * y + 1) + intra_px<L - 1>(img, A, 2 * x + 1, 2 * y + 1) + 2) / 4;
    }
}

// All layers above the image in one launch: a thread per output pixel, workgroups of 64 x 4 pixels dealt to the layers.
__global__ __launch_bounds__(256) void brisk_layers_kernel(const uint8_t *__restrict__ px0, uint8_t *__restrict__ px, brisk_geom g)
{
    px0 += blockIdx.y * g.img_stride;
    px += blockIdx.y * g.px_stride;
    int k = 1;
#pragma unroll
    for (int q = 2; q < BK_LAYERS; ++q) k += (int)blockIdx.x >= g.xt_first[q] - g.xt_first[1] ? 1 : 0;
    const int t = (int)blockIdx.x - (g.xt_first[k] - g.xt_first[1]);
    const int a = (t % g.spans[k]) * 64 + (threadIdx.x & 63), b = (t / g.spans[k]) * 4 + (threadIdx.x >> 6);
    if (a >= g.A[k] || b >= g.B[k]) return;
    const int A0 = g.A[0];
    int v;
    switch (k) {   // (the same in every lane of the workgroup)
        case 1: v = intra_px<0>(px0, A0, a, b); break;
        case 2: v = octave_px<1>(px0, A0, a, b); break;
        case 3: v = intra_px<1>(px0, A0, a, b); break;
        case 4: v = octave_px<2>(px0, A0, a, b); break;
        case 5: v = intra_px<2>(px0, A0, a, b); break;
        case 6: v = octave_px<3>(px0, A0, a, b); break;
        default: v = intra_px<3>(px0, A0, a, b); break;
    }
    px[g.px_off[k] + (size_t)b * g.A[k] + a] = (uint8_t)v;
}

// One layer from the one it is made of, a thread per output pixel (mode 0: half; 1: twothirds): the streaming form
// of the construction, which a batch of images uses (seven launches serve them all; the nested form reads more).
__global__ __launch_bounds__(256) void brisk_sample_kernel(const uint8_t *__restrict__ src, size_t src_stride, int As,
                                                           uint8_t *__restrict__ dst, size_t dst_stride, int Ad, int Bd, int mode)
{
    src += blockIdx.z * src_stride;
    dst += blockIdx.z * dst_stride;
    const int a = blockIdx.x * 64 + (threadIdx.x & 63), b = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (a >= Ad || b >= Bd) return;
    int v;
    if (mode == 0) {
        const uint8_t *p = src + (size_t)(2 * b) * As + 2 * a;
        v = ((int)p[0] + (int)p[1] + (int)p[As] + (int)p[As + 1] + 2) / 4;
    } else {
        const int a0 = 3 * (a / 2) + 2 * (a % 2), b0 = 3 * (b / 2) + 2 * (b % 2);   // the pixel covered in full
        const int am = 3 * (a / 2) + 1, bm = 3 * (b / 2) + 1;                       // the middle of the 3 x 3 block
        v = (4 * (int)src[(size_t)b0 * As + a0] + 2 * (int)src[(size_t)b0 * As + am] + 2 * (int)src[(size_t)bm * As + a0] +
             (int)src[(size_t)bm * As + am] + 4) / 9;
    }
    dst[(size_t)b * Ad + a] = (uint8_t)v;
}

// px0: the images (layer 0); px: the pyramid blocks; sc: the score blocks; smax: an image's maximum score.
__global__ __launch_bounds__(256) void brisk_score_kernel(const uint8_t *__restrict__ px0, const uint8_t *__restrict__ px,
                                                          uint8_t *__restrict__ sc, int *__restrict__ smax,
                                                          int *__restrict__ subs, brisk_geom g, int threshold)
{
    px0 += blockIdx.y * g.img_stride;
    px += blockIdx.y * g.px_stride;
    sc += blockIdx.y * g.sc_stride;
    smax += blockIdx.y * g.cell_stride;
    subs += blockIdx.y * g.sub_stride;
    __shared__ int sI[BK_FA * BK_FB];
    __shared__ int sMax, sQn;
    __shared__ unsigned sRows;   // bit r: row r of the tile holds a score above 0
    __shared__ unsigned short sQ[BK_TA * BK_TB];
    const int tid = threadIdx.x, lane = tid & 63;
    int k = 0;
#pragma unroll
    for (int q = 1; q < BK_LAYERS; ++q) k += (int)blockIdx.x >= g.tile_first[q] ? 1 : 0;
    const int A = g.A[k], B = g.B[k];
    const uint8_t *img = k == 0 ? px0 : px + g.px_off[k];
    const int t = (int)blockIdx.x - g.tile_first[k];
    const int a0 = (t % g.tiles_a[k]) * BK_TA, b0 = (t / g.tiles_a[k]) * BK_TB;
    if (tid == 0) {
        sMax = 0;
        sQn = 0;
        sRows = 0u;
    }
    // (every load of the thread issued before the first one is waited for)
    constexpr int N = BK_FA * BK_FB, ROUNDS = (N + 255) / 256;
    uint8_t v[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int q = min(tid + 256 * r, N - 1);
        const int rb = q / BK_FA, ra = q - rb * BK_FA;
        const int a = min(max(a0 - 3 + ra, 0), A - 1), b = min(max(b0 - 3 + rb, 0), B - 1);
        v[r] = img[(size_t)b * A + a];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int q = tid + 256 * r;
        if (q < N) sI[q] = (int)v[r];
    }
    __syncthreads();
    // Two rounds.  An arc of 9 of the 16 circle pixels holds two neighbouring compass points (top, right, bottom,
    // left), so a pixel whose score reaches the threshold has two neighbouring compass points both brighter than the
    // centre by more than the threshold, or both darker: every pixel takes that test of four lookups, the few that
    // pass (a few per cent at the toolbox's MinContrast) queue up in LDS and the full segment test then runs on the
    // queue with all lanes busy.  The others score 0, as the full test would have said.
    const int T = threshold + 1;
    static_assert(BK_TA * BK_TB % 256 == 0, "every thread makes the same number of rounds (ballots)");
    for (int q = tid; q < BK_TA * BK_TB; q += 256) {
        const int rb = q / BK_TA, ra = q - rb * BK_TA;
        const int a = a0 + ra, b = b0 + rb;
        const int *c = sI + (rb + 3) * BK_FA + (ra + 3);
        bool cand = false;
        if (a < A && b < B) {
            if (a >= 3 && a < A - 3 && b >= 3 && b < B - 3) {
                const int p = c[0];
                const int d0 = c[-3 * BK_FA] - p, d4 = c[3] - p, d8 = c[3 * BK_FA] - p, d12 = c[-3] - p;
                const bool b0_ = d0 >= T, b4 = d4 >= T, b8 = d8 >= T, b12 = d12 >= T;
                const bool k0 = d0 <= -T, k4 = d4 <= -T, k8 = d8 <= -T, k12 = d12 <= -T;
                cand = ((b0_ | b8) & (b4 | b12)) | ((k0 | k8) & (k4 | k12));
            }
            if (!cand) sc[g.sc_off[k] + (size_t)b * A + a] = (uint8_t)0;
            if (k == 0) {
                const int s58 = a >= 1 && a < A - 1 && b >= 1 && b < B - 1 ? vohip_dev::fast_score_5_8(c, BK_FA) : 0;
                sc[g.sc_off[BK_LAYERS] + (size_t)b * A + a] = (uint8_t)s58;
            }
        }
        const unsigned long long bal = __ballot(cand);
        if (bal) {   // (wave-uniform)
            int base = 0;
            if (lane == 0) base = atomicAdd(&sQn, __popcll(bal));
            base = __shfl(base, 0, 64);
            if (cand) sQ[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)q;
        }
    }
    __syncthreads();
    int best = 0;
    for (int i = tid, n = sQn; i < n; i += 256) {
        const int q = sQ[i];
        const int rb = q / BK_TA, ra = q - rb * BK_TA;
        const int s = vohip_dev::fast_score_9_16(sI + (rb + 3) * BK_FA + (ra + 3), BK_FA);
        const int score = s >= threshold ? s : 0;
        sc[g.sc_off[k] + (size_t)(b0 + rb) * A + (a0 + ra)] = (uint8_t)score;
        best = max(best, score);
        if (score > 0) atomicOr(&sRows, 1u << rb);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) best = max(best, __shfl_xor(best, m, 64));
    if ((tid & 63) == 0 && best > 0) atomicMax(&sMax, best);
    __syncthreads();
    if (tid == 0 && sMax > 0) atomicMax(smax, sMax);
    // a tile is as wide as the spans of the extrema pass: tell that pass which spans of a row hold no score at all
    // (most of them), so that it does not read them; it replaces the mark by the span's number of keypoints
    static_assert(BK_TA == 64, "a tile row = a span of the extrema pass");
    if (tid < BK_TB && b0 + tid < B)
        subs[g.sub_first[k] + (b0 + tid) * g.spans[k] + a0 / BK_TA] = (sRows >> tid) & 1u ? -1 : 0;
}

__device__ __forceinline__ int floor_div(int n, int d) { return n >= 0 ? n / d : -((-n + d - 1) / d); }

// the largest score of a layer (A2 x B2, scores S2) over the pixels whose centres lie within the 3 x 3 patch of the
// pixel (a, b) of a layer whose scale is p / q times that layer's
__device__ __forceinline__ int patch_max(const uint8_t *__restrict__ S2, int A2, int B2, int a, int b, int p, int q)
{
    const int a_lo = max(-floor_div(-(((2 * a - 1) * p) - q), 2 * q), 0), a_hi = min(floor_div((2 * a + 3) * p - q, 2 * q), A2 - 1);
    const int b_lo = max(-floor_div(-(((2 * b - 1) * p) - q), 2 * q), 0), b_hi = min(floor_div((2 * b + 3) * p - q, 2 * q), B2 - 1);
    int m = 0;
    for (int y = b_lo; y <= b_hi; ++y)
        for (int x = a_lo; x <= a_hi; ++x) m = max(m, (int)S2[(size_t)y * A2 + x]);
    return m;
}

#pragma clang fp contract(off)

__global__ __launch_bounds__(256) void brisk_extrema_kernel(const uint8_t *__restrict__ sc, int *__restrict__ cells,
                                                            int *__restrict__ subs, brisk_point *__restrict__ recs,
                                                            brisk_geom g, float min_quality, int lo)
{
    sc += blockIdx.y * g.sc_stride;
    cells += blockIdx.y * g.cell_stride;
    subs += blockIdx.y * g.sub_stride;
    recs += blockIdx.y * g.rec_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int k = 0;
#pragma unroll
    for (int q = 1; q < BK_LAYERS; ++q) k += (int)blockIdx.x >= g.xt_first[q] ? 1 : 0;
    const int A = g.A[k], B = g.B[k], s2 = g.sigma2[k];
    const int t = (int)blockIdx.x - g.xt_first[k];
    const int span = t % g.spans[k], b = (t / g.spans[k]) * 4 + wave, a = span * 64 + lane;
    if (b >= B) return;   // (a whole wavefront)
    const int sub = g.sub_first[k] + b * g.spans[k] + span;
    if (subs[sub] == 0) return;   // (a whole wavefront: the score pass found nothing above 0 in this span)
    const float thr = min_quality * (float)cells[g.cell_first[BK_LAYERS]];   // (the slot behind the cells: the maximum score)
    const uint8_t *S = sc + g.sc_off[k];
    bool hit = false;
    brisk_point pt;
    if (a >= 1 && a < A - 1 && b >= 1 && b < B - 1) {
        const uint8_t *c = S + (size_t)b * A + a;
        const int s = c[0];
        if (s > 0 && (float)s > thr) {
            const int l = c[-1], r = c[1], u = c[-A], d = c[A];
            bool ok = s > l && s > r && s > u && s > d && s > (int)c[-A - 1] && s > (int)c[-A + 1] && s > (int)c[A - 1] &&
                      s > (int)c[A + 1];
            const int Xa = ((2 * a + 1) * s2) / 4, Xb = ((2 * b + 1) * s2) / 4;   // the pixel's centre in the image
            ok = ok && Xa >= lo && Xa < g.A[0] - lo && Xb >= lo && Xb < g.B[0] - lo;
            int m_below = 0, m_above = 0;
            if (ok && k > 0) {
                m_below = patch_max(sc + g.sc_off[k - 1], g.A[k - 1], g.B[k - 1], a, b, s2, g.sigma2[k - 1]);
                ok = s > m_below;
            }
            if (ok && k + 1 < g.nl) {
                m_above = patch_max(sc + g.sc_off[k + 1], g.A[k + 1], g.B[k + 1], a, b, s2, g.sigma2[k + 1]);
                ok = s > m_above;
            }
            if (ok) {
                const float da = (float)(r - l), dda = (float)((r - 2 * s) + l);
                const float db = (float)(d - u), ddb = (float)((d - 2 * s) + u);
                float oa = dda < 0.0f ? -0.5f * da / dda : 0.0f;
                float ob = ddb < 0.0f ? -0.5f * db / ddb : 0.0f;
                oa = fminf(fmaxf(oa, -0.5f), 0.5f);
                ob = fminf(fmaxf(ob, -0.5f), 0.5f);
                const double sg = 0.5 * (double)s2;
                pt.pa = (float)((((double)a + (double)oa) + 0.5) * sg - 0.5);
                pt.pb = (float)((((double)b + (double)ob) + 0.5) * sg - 0.5);
                pt.metric = (float)s;
                double scale = sg;
                if (k + 1 < g.nl) {
                    double x0, y0;
                    bool have = true;
                    if (k > 0) {
                        x0 = 0.5 * (double)g.sigma2[k - 1];
                        y0 = (double)m_below;
                    } else {
                        x0 = 2.0 / 3.0;
                        y0 = (double)sc[g.sc_off[BK_LAYERS] + (size_t)b * A + a];
                        have = (double)s > y0;
                    }
                    if (have) {
                        const double x1 = sg, y1 = (double)s, x2 = 0.5 * (double)g.sigma2[k + 1], y2 = (double)m_above;
                        const double num = ((x1 * x1 - x2 * x2) * y0 + (x2 * x2 - x0 * x0) * y1) + (x0 * x0 - x1 * x1) * y2;
                        const double den = ((x1 - x2) * y0 + (x2 - x0) * y1) + (x0 - x1) * y2;
                        scale = 0.5 * num / den;
                    }
                }
                pt.scale = (float)scale;
                hit = true;
            }
        }
    }
    const unsigned long long bal = __ballot(hit);
    const int n = __popcll(bal);
    if (hit) recs[(size_t)sub * BK_SUB_CAP + __popcll(bal & ((1ull << lane) - 1ull))] = pt;
    if (lane == 0) {
        subs[sub] = n;
        if (n) atomicAdd(&cells[g.cell_first[k] + b], n);
    }
}

#pragma clang fp contract(fast)

// One wavefront per sub-cell, in the order of the output: layer, row, span.
__global__ __launch_bounds__(256) void brisk_emit_kernel(const int *__restrict__ cells, const int *__restrict__ subs,
                                                         const brisk_point *__restrict__ recs, brisk_geom g, int cap,
                                                         int swap_xy, float *__restrict__ xy_out,
                                                         float *__restrict__ metric_out, float *__restrict__ scale_out,
                                                         int *__restrict__ count)
{
    cells += blockIdx.y * g.cell_stride;
    subs += blockIdx.y * g.sub_stride;
    recs += blockIdx.y * g.rec_stride;
    xy_out += blockIdx.y * (size_t)g.xy_stride;
    metric_out += blockIdx.y * (size_t)g.metric_stride;
    if (scale_out) scale_out += blockIdx.y * (size_t)g.metric_stride;
    count += blockIdx.y * (size_t)g.count_stride;
    const int lane = threadIdx.x & 63, sub = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nsubs = g.sub_first[BK_LAYERS];
    if (sub >= nsubs) return;
    const int n = subs[sub];
    if (n == 0 && sub != nsubs - 1) return;
    int k = 0;
#pragma unroll
    for (int q = 1; q < BK_LAYERS; ++q) k += sub >= g.sub_first[q] && g.sub_first[q] < nsubs ? 1 : 0;
    const int spans = g.spans[k];
    const int row = (sub - g.sub_first[k]) / spans, span = (sub - g.sub_first[k]) - row * spans;
    const int cell = g.cell_first[k] + row;
    int base = 0;   // keypoints in the cells before this one and in the spans of this cell before this one
    for (int q = lane; q < cell; q += 64) base += cells[q];
    for (int q = lane; q < span; q += 64) base += subs[sub - span + q];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) base += __shfl_xor(base, m, 64);
    if (sub == nsubs - 1 && lane == 0) *count = base + n;   // (the last sub-cell is the last span of the last cell)
    const int pos = base + lane;
    if (lane < n && pos < cap) {
        const brisk_point pt = recs[(size_t)sub * BK_SUB_CAP + lane];
        const float pa = pt.pa + 1.0f, pb = pt.pb + 1.0f;
        xy_out[pos] = swap_xy ? pb : pa;
        xy_out[cap + pos] = swap_xy ? pa : pb;
        metric_out[pos] = pt.metric;
        if (scale_out) scale_out[pos] = pt.scale;
    }
}

// The launches over n images whose scratch lies image after image at `scratch`.
int enqueue_images(vohip_ctx *ctx, hipStream_t st, void *scratch, brisk_geom g, const uint8_t *d_imgs, int n, int colmajor,
                   double min_contrast, double min_quality, int border, float *d_xy, float *d_metric, float *d_scale, int cap,
                   int *d_counts)
{
    char *p = (char *)scratch;
    int *cells = (int *)p;   // (first: cleared with the maxima behind them by one fill)
    p += sizeof(int) * g.cell_stride * n;
    int *subs = (int *)p;
    p += sizeof(int) * g.sub_stride * n;
    uint8_t *px = (uint8_t *)p;
    p += g.px_stride * n;
    uint8_t *sc = (uint8_t *)p;
    p += g.sc_stride * n;
    brisk_point *recs = (brisk_point *)p;
    g.xy_stride = 2 * cap;
    g.metric_stride = cap;
    g.count_stride = 1;
    VOHIP_HIP(ctx, hipMemsetAsync(cells, 0, sizeof(int) * g.cell_stride * n, st));
    if (n == 1) {   // (one launch: a single image's layers are a matter of launch latencies)
        if (g.xt_first[BK_LAYERS] > g.xt_first[1])
            hipLaunchKernelGGL(brisk_layers_kernel, dim3(g.xt_first[BK_LAYERS] - g.xt_first[1], n), dim3(256), 0, st, d_imgs, px, g);
    } else {
        for (int k = 1; k < g.nl; ++k) {
            if (g.A[k] < 1 || g.B[k] < 1) continue;
            const int from = k == 1 ? 0 : k - 2;
            const uint8_t *src = from == 0 ? d_imgs : px + g.px_off[from];
            hipLaunchKernelGGL(brisk_sample_kernel, dim3((g.A[k] + 63) / 64, (g.B[k] + 3) / 4, n), dim3(256), 0, st, src,
                               from == 0 ? g.img_stride : g.px_stride, g.A[from], px + g.px_off[k], g.px_stride, g.A[k], g.B[k],
                               k == 1 ? 1 : 0);
        }
    }
    const int lo = border < 1 ? 1 : border, nsubs = g.sub_first[BK_LAYERS];
    int *smax = cells + g.cell_first[BK_LAYERS];
    hipLaunchKernelGGL(brisk_score_kernel, dim3(g.tile_first[BK_LAYERS], n), dim3(256), 0, st, d_imgs, (const uint8_t *)px, sc,
                       smax, subs, g, (int)std::floor(min_contrast * 255.0 + 0.5));
    hipLaunchKernelGGL(brisk_extrema_kernel, dim3(g.xt_first[BK_LAYERS], n), dim3(256), 0, st, (const uint8_t *)sc, cells, subs,
                       recs, g, (float)min_quality, lo);
    hipLaunchKernelGGL(brisk_emit_kernel, dim3((nsubs + 3) / 4, n), dim3(256), 0, st, (const int *)cells, (const int *)subs,
                       (const brisk_point *)recs, g, cap, colmajor ? 1 : 0, d_xy, d_metric, d_scale, d_counts);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int check_args(vohip_ctx *ctx, int rows, int cols, double min_contrast, double min_quality, int num_octaves, int border,
               int cap)
{
    VOHIP_REQUIRE(ctx, rows >= 8 && cols >= 8, "image too small");
    VOHIP_REQUIRE(ctx, (size_t)rows * cols <= ((size_t)1 << 26), "BRISK: at most 2^26 pixels");
    VOHIP_REQUIRE(ctx, min_contrast > 0.0 && min_contrast < 1.0, "MinContrast must lie in (0, 1)");
    VOHIP_REQUIRE(ctx, min_quality >= 0.0 && min_quality <= 1.0, "MinQuality must lie in [0, 1]");
    VOHIP_REQUIRE(ctx, num_octaves >= 0 && num_octaves <= 4, "NumOctaves must be 0..4");
    VOHIP_REQUIRE(ctx, cap >= 1 && cap <= 16384, "cap must be 1..16384");
    VOHIP_REQUIRE(ctx, border >= 0, "negative border");
    return VOHIP_OK;
}

}  // namespace

size_t vohip_brisk_scratch_bytes(int rows, int cols)
{
    const size_t a = image_scratch_bytes(make_geom(rows, cols, 4)), b = image_scratch_bytes(make_geom(cols, rows, 4));
    return a > b ? a : b;
}

// BRISK detection on `stream` with caller-owned scratch of vohip_brisk_scratch_bytes(rows, cols) bytes.
int vohip_detect_brisk_enqueue(vohip_ctx *ctx, hipStream_t st, void *scratch, const uint8_t *d_img, int rows, int cols,
                               int colmajor, double min_contrast, double min_quality, int num_octaves, int border,
                               float *d_xy, float *d_metric, float *d_scale, int cap, int *d_count)
{
    VOHIP_REQUIRE(ctx, d_img && d_xy && d_metric && d_count && scratch, "NULL argument");
    VOHIP_TRY(check_args(ctx, rows, cols, min_contrast, min_quality, num_octaves, border, cap));
    const brisk_geom g = make_geom(colmajor ? rows : cols, colmajor ? cols : rows, num_octaves);
    return enqueue_images(ctx, st, scratch, g, d_img, 1, colmajor, min_contrast, min_quality, border, d_xy, d_metric, d_scale,
                          cap, d_count);
}

// n images, image after image in device memory, on the context's stream: the lists of n single calls, bit for bit.
// The scratch of a pass is kept under 1 GiB: a long batch runs as several passes over the same memory.
int vohip_detect_brisk_batch_enqueue(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                     double min_contrast, double min_quality, int num_octaves, int border, float *d_xy,
                                     float *d_metric, int cap, int *d_counts)
{
    VOHIP_REQUIRE(ctx, d_imgs && d_xy && d_metric && d_counts, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && n <= 65535, "1..65535 images per call");
    VOHIP_TRY(check_args(ctx, rows, cols, min_contrast, min_quality, num_octaves, border, cap));
    const brisk_geom g = make_geom(colmajor ? rows : cols, colmajor ? cols : rows, num_octaves);
    const size_t per = image_scratch_bytes(g);
    int pass = (int)((size_t)BK_BATCH_BYTES_MAX / per);
    pass = pass < 1 ? 1 : (pass > n ? n : pass);
    vohip_buf &buf = ctx->bufs[B_BRISK_BATCH];
    VOHIP_TRY(vohip_reserve(ctx, buf, per * pass));
    for (int c = 0; c < n; c += pass) {
        const int m = n - c < pass ? n - c : pass;
        VOHIP_TRY(enqueue_images(ctx, ctx->stream, buf.ptr, g, d_imgs + g.img_stride * c, m, colmajor, min_contrast, min_quality,
                                 border, d_xy + 2 * (size_t)cap * c, d_metric + (size_t)cap * c, nullptr, cap, d_counts + c));
    }
    return VOHIP_OK;
}

namespace {

int brisk_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double min_contrast, double min_quality,
              int num_octaves, int border, float *d_xy, float *d_metric, float *d_scale, int cap, int *d_count)
{
    VOHIP_TRY(check_args(ctx, rows, cols, min_contrast, min_quality, num_octaves, border, cap));
    vohip_buf &b = ctx->bufs[B_BRISK_SCRATCH];
    VOHIP_TRY(vohip_reserve(ctx, b, vohip_brisk_scratch_bytes(rows, cols)));
    return vohip_detect_brisk_enqueue(ctx, ctx->stream, b.ptr, d_img, rows, cols, colmajor, min_contrast, min_quality,
                                      num_octaves, border, d_xy, d_metric, d_scale, cap, d_count);
}

}  // namespace

extern "C" {

int vohip_detect_brisk_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double min_contrast,
                           double min_quality, int num_octaves, int border, float *d_xy, float *d_metric, float *d_scale,
                           int cap, int *d_count)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return brisk_dev(ctx, d_img, rows, cols, colmajor, min_contrast, min_quality, num_octaves, border, d_xy, d_metric, d_scale,
                     cap, d_count);
}

int vohip_detect_brisk_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                 double min_contrast, double min_quality, int num_octaves, int border, float *d_xy,
                                 float *d_metric, int cap, int *d_counts)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return vohip_detect_brisk_batch_enqueue(ctx, d_imgs, n, rows, cols, colmajor, min_contrast, min_quality, num_octaves, border,
                                            d_xy, d_metric, cap, d_counts);
}

int vohip_detect_brisk(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double min_contrast,
                       double min_quality, int num_octaves, int border, float *xy_out, float *metric_out, float *scale_out,
                       int cap, int *n_out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, img && xy_out && metric_out && n_out, "NULL argument");
    VOHIP_REQUIRE(ctx, rows >= 1 && cols >= 1 && cap >= 1, "bad size");
    void *d_img, *d_xy, *d_metric, *d_scale, *d_count;
    VOHIP_TRY(vohip_upload(ctx, B_IMG_STAGE, img, (size_t)rows * cols, &d_img));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, nullptr, sizeof(float) * 2 * (size_t)cap, &d_xy));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, nullptr, sizeof(float) * (size_t)cap, &d_metric));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, 256, &d_count));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, sizeof(float) * (size_t)cap, &d_scale));
    VOHIP_TRY(brisk_dev(ctx, (const uint8_t *)d_img, rows, cols, colmajor, min_contrast, min_quality, num_octaves, border,
                        (float *)d_xy, (float *)d_metric, (float *)d_scale, cap, (int *)d_count));
    int total = 0;
    VOHIP_TRY(vohip_download(ctx, &total, d_count, sizeof(int)));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int n = total < cap ? total : cap;
    VOHIP_TRY(vohip_download(ctx, xy_out, d_xy, sizeof(float) * n));
    VOHIP_TRY(vohip_download(ctx, xy_out + n, (const float *)d_xy + cap, sizeof(float) * n));
    VOHIP_TRY(vohip_download(ctx, metric_out, d_metric, sizeof(float) * n));
    if (scale_out) VOHIP_TRY(vohip_download(ctx, scale_out, d_scale, sizeof(float) * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    if (total > cap)
        return vohip_fail(ctx, VOHIP_CAPACITY_EXCEEDED, "%d keypoints found, capacity %d: raise cap, MinContrast or MinQuality",
                          total, cap);
    return VOHIP_OK;
}

}  // extern "C"
