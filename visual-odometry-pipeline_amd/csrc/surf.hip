// SURF interest points for gfx950.
//
// Stands in for detectSURFFeatures as genKeypoints.m:16,31 calls it (the reference's default detector,
// main_demo.m:25; option 'MetricThreshold', main_eval.m:151,167).  The toolbox internals are not public; the
// detector is defined after Bay, Ess, Tuytelaars, Van Gool, "Speeded-Up Robust Features" (CVIU 2008) with
// the filter sizes of the toolbox documentation (NumOctaves 3, NumScaleLevels 4):
//   octave o = 0, 1, 2 samples the image every s = 2^o pixels with box filters of side
//        L = 9 15 21 27 | 15 27 39 51 | 27 51 75 99,   lobe l = L / 3;
//   at a sample centred on pixel (a, b) whose filter lies inside the image (0 elsewhere)
//        Dxx = (sum over [a-(L-1)/2, a+(L-1)/2] x [b-l+1, b+l-1] - 3 * its middle third along a) / (l (2l-1)),
//        Dyy the same with the axes exchanged,
//        Dxy = (the l x l boxes at the (-,-) and (+,+) corners, one pixel off the centre row and column,
//               minus those at (+,-) and (-,+)) / l^2,
//        det = Dxx Dyy - 0.81 Dxy^2   (intensities 0..255; fp64 in a fixed order, stored as fp32);
//   interest point <=> middle level m = 1, 2 of an octave, det > MetricThreshold, det strictly greater than
//        its 26 neighbours in the 3 x 3 x 3 block of samples (levels m-1, m, m+1), the filter of level m+1
//        inside the image at every sample of the block, the sample's pixel >= border px from every edge;
//   refinement x = -H^-1 g of the quadratic through the block; dropped when singular or an |x_i| > 1;
//        position (sample + x) s, Scale = (1.2 / 9)(L_m + x_level (L_m - L_m-1)), Metric = det;
//   output order: octave, level, b, a ascending.
// Kernels:
//   surf_integral_rows_kernel      one wavefront per image row: running sums along the fast axis (four pixels per
//                                  lane, a wavefront scan of the lanes' sums, a carry from one 256-pixel span to
//                                  the next) into the integral image's rows (uint32, a zero first row and column
//                                  so that box sums need no edge cases).
//   surf_integral_colsum_kernel /  the slow axis in chunks of 32 rows: per-chunk column sums, then every
//   surf_integral_colscan_kernel   chunk adds the chunks before it and runs down its own rows (all 32 loads
//                                  of a thread in flight together); consecutive threads = consecutive columns.
//   surf_response_kernel           one launch, two kinds of workgroup.  Octave 0 (three quarters of the responses) in
//                                  one pass: an LDS tile of the integral image (64 x 8 samples + ring + filter reach,
//                                  13.8 KB) -> the four levels of the tile and its ring in LDS (10.6 KB) -> both middle
//                                  levels tested there; its maps never reach memory.  The 8 maps of octaves 1 and 2:
//                                  a thread per sample through the caches (the integral image stays in L2: 1.9 MB for
//                                  1241 x 376).  32 lookups per response (exact integer box sums, fp64 determinant).
//   surf_extrema_kernel            octaves 1 and 2: a workgroup per tile of 64 x 4 samples, whose four levels
//                                  and a one-sample ring go to LDS once (6.3 KB); every thread tests its sample
//                                  on both middle levels from there and refines the hits; a wavefront = one row
//                                  of 64 samples = one (octave, level, row, span) sub-cell, whose points become
//                                  records in sample order (ballot ranks) and whose count is added to its
//                                  (octave, level, row) cell (integer atomics: the sums do not depend on order).
//   surf_emit_kernel               one wavefront per sub-cell that holds points: ranks it behind the cells before
//                                  its own and the spans before it in its cell, and writes its records there.
//                                  Ordered and deterministic.
// Per image: the uint8 image read once, the integral image written twice and read ~3 times, the 0.58 M
// responses of octaves 1 and 2 (2.3 MB for 1241 x 376) written once and read once, 16 bytes per point written twice.
#include "vohip_internal.h"

#include <cmath>

namespace {

constexpr int SURF_OCT = 3, SURF_LEV = 4, SURF_MAPS = SURF_OCT * SURF_LEV, SURF_MID = 2 * SURF_OCT;
constexpr int SURF_CH = 32;   // rows of the integral image per chunk of the slow-axis pass
constexpr int SURF_BATCH_BYTES_MAX = 1 << 30;   // scratch of one pass over a batch of images
constexpr int SURF_SUB_CAP = 32;   // points of a span of 64 samples of one row: strict maxima are never neighbours

struct surf_point {   // 16 bytes
    float pa, pb, metric, scale;
};

constexpr int RT_A = 64, RT_B = 8, RT_H = 13;   // octave 0's tile of samples and the reach of its largest filter

struct surf_geom {
    int A, B, A1;                        // image axes; row length of the integral image
    int GA[SURF_OCT], GB[SURF_OCT];      // samples per octave
    int L[SURF_MAPS];                    // filter side of map 4 o + level
    unsigned map_off[SURF_MAPS];         // of a map of octaves 1, 2 within an image's responses (floats)
    int tile_first[SURF_MAPS + 1];       // workgroups of the response launch before map k
    int tiles_a[SURF_OCT];
    int cell_first[SURF_MID + 1];        // cells before (octave, middle level) pair k = 2 o + (m - 1)
    int sub_first[SURF_MID + 1];         // sub-cells (a cell's spans of 64 samples) before pair k
    int otile_first[SURF_OCT + 1];       // workgroups of the extrema launch before octave o
    int resp0_tiles;                     // workgroups of the response launch that make octave 0 (64 x 8 samples, all levels)
    int nch;                             // chunks of the slow-axis pass
    // per image of a batch
    size_t img_stride, t_stride, p_stride, r_stride, cell_stride, sub_stride, rec_stride;
    int xy_stride, metric_stride, count_stride;
};

constexpr int SIZES[SURF_OCT][SURF_LEV] = {{9, 15, 21, 27}, {15, 27, 39, 51}, {27, 51, 75, 99}};

surf_geom make_geom(int A, int B)
{
    surf_geom g{};
    g.A = A;
    g.B = B;
    g.A1 = A + 1;
    unsigned off = 0;
    int tiles = 0, cells = 0, subs = 0, otiles = 0;
    for (int o = 0; o < SURF_OCT; ++o) {
        const int s = 1 << o;
        g.GA[o] = (A + s - 1) / s;
        g.GB[o] = (B + s - 1) / s;
        g.tiles_a[o] = (g.GA[o] + 63) / 64;
        for (int i = 0; i < SURF_LEV; ++i) {
            g.L[4 * o + i] = SIZES[o][i];
            g.map_off[4 * o + i] = off;
            if (o > 0) off += (unsigned)((size_t)g.GA[o] * g.GB[o] + 63) / 64 * 64;   // (octave 0 is never stored)
            g.tile_first[4 * o + i] = tiles;
            tiles += g.tiles_a[o] * ((g.GB[o] + 3) / 4);
        }
        for (int m = 0; m < 2; ++m) {
            g.cell_first[2 * o + m] = cells;
            cells += g.GB[o];
            g.sub_first[2 * o + m] = subs;
            subs += g.GB[o] * g.tiles_a[o];
        }
        g.otile_first[o] = otiles;
        otiles += g.tiles_a[o] * ((g.GB[o] + 3) / 4);
    }
    g.tile_first[SURF_MAPS] = tiles;
    g.cell_first[SURF_MID] = cells;
    g.sub_first[SURF_MID] = subs;
    g.otile_first[SURF_OCT] = otiles;
    g.resp0_tiles = g.tiles_a[0] * ((B + RT_B - 1) / RT_B);
    g.nch = (B + SURF_CH - 1) / SURF_CH;
    auto r256 = [](size_t v) { return (v + 255) / 256 * 256; };
    g.img_stride = (size_t)A * B;
    g.t_stride = r256(sizeof(uint32_t) * (size_t)g.A1 * (B + 1)) / sizeof(uint32_t);
    g.p_stride = r256(sizeof(uint32_t) * (size_t)g.nch * A) / sizeof(uint32_t);
    g.r_stride = r256(sizeof(float) * (size_t)off) / sizeof(float);
    g.cell_stride = r256(sizeof(int) * (size_t)cells) / sizeof(int);
    g.sub_stride = r256(sizeof(int) * (size_t)subs) / sizeof(int);
    g.rec_stride = (size_t)subs * SURF_SUB_CAP;
    return g;
}

size_t image_scratch_bytes(const surf_geom &g)
{
    return sizeof(uint32_t) * (g.t_stride + g.p_stride) + sizeof(float) * g.r_stride +
           sizeof(int) * (g.cell_stride + g.sub_stride) + sizeof(surf_point) * g.rec_stride;
}

// (also clears the per-cell point counts the extrema kernel adds to)
__global__ __launch_bounds__(256) void surf_integral_rows_kernel(const uint8_t *__restrict__ img, uint32_t *__restrict__ T,
                                                                 int *__restrict__ cells, surf_geom g)
{
    img += blockIdx.y * g.img_stride;
    T += blockIdx.y * g.t_stride;
    cells += blockIdx.y * g.cell_stride;
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0) {
        for (int a = threadIdx.x; a < g.A1; a += 256) T[a] = 0u;
        for (int q = threadIdx.x; q < g.cell_first[SURF_MID]; q += 256) cells[q] = 0;
    }
    if (b >= g.B) return;
    uint32_t carry = 0u;
    uint32_t *row = T + (size_t)(b + 1) * g.A1;
    const uint8_t *src = img + (size_t)b * g.A;
    if (lane == 0) row[0] = 0u;
    for (int c = 0; c < g.A; c += 256) {
        const int a = c + 4 * lane;
        uint32_t x[4];
        if (a + 3 < g.A) {
            uint32_t w;
            __builtin_memcpy(&w, src + a, 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = (w >> (8 * k)) & 0xffu;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = a + k < g.A ? (uint32_t)src[a + k] : 0u;
        }
        x[1] += x[0];
        x[2] += x[1];
        x[3] += x[2];
        uint32_t v = x[3];   // inclusive scan of the lanes' sums
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
            if (lane >= d) v += t;
        }
        const uint32_t before = carry + (v - x[3]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (a + k < g.A) row[a + k + 1] = before + x[k];
        carry += (uint32_t)__shfl((int)v, 63, 64);
    }
}

// P[chunk][a] = the sum of the row sums of column a + 1 over the chunk's rows
__global__ __launch_bounds__(256) void surf_integral_colsum_kernel(const uint32_t *__restrict__ T, uint32_t *__restrict__ P,
                                                                   surf_geom g)
{
    T += blockIdx.z * g.t_stride;
    P += blockIdx.z * g.p_stride;
    const int a = blockIdx.x * 64 + threadIdx.x, ch = blockIdx.y * 4 + threadIdx.y;
    if (a >= g.A || ch >= g.nch) return;
    const uint32_t *col = T + (size_t)(ch * SURF_CH + 1) * g.A1 + (a + 1);
    const int rows = min(SURF_CH, g.B - ch * SURF_CH);
    uint32_t v[SURF_CH];
#pragma unroll
    for (int r = 0; r < SURF_CH; ++r) v[r] = r < rows ? col[(size_t)r * g.A1] : 0u;
    uint32_t s = 0u;
#pragma unroll
    for (int r = 0; r < SURF_CH; ++r) s += v[r];
    P[(size_t)ch * g.A + a] = s;
}

__global__ __launch_bounds__(256) void surf_integral_colscan_kernel(uint32_t *__restrict__ T, const uint32_t *__restrict__ P,
                                                                    surf_geom g)
{
    T += blockIdx.z * g.t_stride;
    P += blockIdx.z * g.p_stride;
    const int a = blockIdx.x * 64 + threadIdx.x, ch = blockIdx.y * 4 + threadIdx.y;
    if (a >= g.A || ch >= g.nch) return;
    uint32_t *col = T + (size_t)(ch * SURF_CH + 1) * g.A1 + (a + 1);
    const int rows = min(SURF_CH, g.B - ch * SURF_CH);
    uint32_t v[SURF_CH];
#pragma unroll
    for (int r = 0; r < SURF_CH; ++r) v[r] = r < rows ? col[(size_t)r * g.A1] : 0u;
    uint32_t s = 0u;
    for (int c = 0; c < ch; ++c) s += P[(size_t)c * g.A + a];
#pragma unroll
    for (int r = 0; r < SURF_CH; ++r) {
        s += v[r];
        if (r < rows) col[(size_t)r * g.A1] = s;
    }
}

#pragma clang fp contract(off)

// sum of the image over [a0, a1) x [b0, b1) (differences of uint32 sums: exact while a box holds < 2^31)
__device__ __forceinline__ int box_sum(const uint32_t *__restrict__ T, int A1, int a0, int b0, int a1, int b1)
{
    const uint32_t *r0 = T + (size_t)b0 * A1, *r1 = T + (size_t)b1 * A1;
    return (int)((r1[a1] - r0[a1]) - (r1[a0] - r0[a0]));
}

struct surf_norms {
    double inv_lobe[SURF_MAPS], inv_sq[SURF_MAPS];   // 1 / (l (2l - 1)), 1 / l^2, divided on the host
};

// The determinant of one sample from box sums: `sum(a0, b0, a1, b1)` = the image summed over [a0, a1) x [b0, b1).
template <class BOX>
__device__ __forceinline__ float surf_det(BOX sum, int a, int b, int L, double inv_lobe, double inv_sq)
{
    const int l = L / 3, h = (L - 1) / 2;
    const int nxx = sum(a - h, b - l + 1, a + h + 1, b + l) - 3 * sum(a - h + l, b - l + 1, a - h + 2 * l, b + l);
    const int nyy = sum(a - l + 1, b - h, a + l, b + h + 1) - 3 * sum(a - l + 1, b - h + l, a + l, b - h + 2 * l);
    const int nxy = sum(a - l, b - l, a, b) + sum(a + 1, b + 1, a + 1 + l, b + 1 + l) - sum(a + 1, b - l, a + 1 + l, b) -
                    sum(a - l, b + 1, a, b + 1 + l);
    const double dxx = (double)nxx * inv_lobe, dyy = (double)nyy * inv_lobe, dxy = (double)nxy * inv_sq;
    const double t1 = dxx * dyy, t2 = dxy * dxy, t3 = 0.81 * t2;
    return (float)(t1 - t3);
}

// The maps of octaves 1 and 2 (a quarter of all responses), whose filters reach up to 50 pixels: a thread per sample
// reads the integral image through the caches.  Octave 0 never becomes a map: surf_octave0_block below.
// block = the workgroup's number among those of the maps 4..11.
__device__ __forceinline__ void surf_maps_block(const uint32_t *__restrict__ T, float *__restrict__ R, const surf_geom &g,
                                                const surf_norms &nm, int block)
{
    const int A1 = g.A1;
    const int bx = block + g.tile_first[SURF_LEV];
    int map = SURF_LEV;
#pragma unroll
    for (int k = SURF_LEV + 1; k < SURF_MAPS; ++k) map += bx >= g.tile_first[k] ? 1 : 0;
    const int o = map >> 2, s = 1 << o;
    const int t = bx - g.tile_first[map];
    const int ga = (t % g.tiles_a[o]) * 64 + (threadIdx.x & 63), gb = (t / g.tiles_a[o]) * 4 + (threadIdx.x >> 6);
    if (ga >= g.GA[o] || gb >= g.GB[o]) return;
    const int L = g.L[map], h = (L - 1) / 2;
    const int a = ga * s, b = gb * s;
    float r = 0.0f;
    if (a - h >= 0 && a + h <= g.A - 1 && b - h >= 0 && b + h <= g.B - 1) {
        auto sum = [&](int x0, int y0, int x1, int y1) { return box_sum(T, A1, x0, y0, x1, y1); };
        r = surf_det(sum, a, b, L, nm.inv_lobe[map], nm.inv_sq[map]);
    }
    R[g.map_off[map] + (size_t)gb * g.GA[o] + ga] = r;
}

// The sample (ga, gb) of a middle level: an interest point?  p0, p1, p2 point at the sample in the level below, its
// own and the level above; consecutive rows of samples lie GA floats apart.
__device__ __forceinline__ bool surf_test(const float *p0, const float *p1, const float *p2, int GA, int ga, int gb, float thr,
                                          int s, int Lm, int dL, surf_point *out)
{
    const float v = p1[0];
    if (!(v > thr)) return false;
    const float *lv[3] = {p0, p1, p2};
    bool is_max = true;
#pragma unroll
    for (int di = 0; di < 3; ++di)
#pragma unroll
        for (int db = -1; db <= 1; ++db)
#pragma unroll
            for (int da = -1; da <= 1; ++da)
                if (!(di == 1 && db == 0 && da == 0)) is_max = is_max && v > lv[di][db * GA + da];
    if (!is_max) return false;
#define NB(di, db, da) ((double)lv[(di) + 1][(db) * GA + (da)])
    const double vd = (double)v;
    const double g_a = (NB(0, 0, 1) - NB(0, 0, -1)) * 0.5;
    const double g_b = (NB(0, 1, 0) - NB(0, -1, 0)) * 0.5;
    const double g_s = (NB(1, 0, 0) - NB(-1, 0, 0)) * 0.5;
    const double h_aa = (NB(0, 0, 1) + NB(0, 0, -1)) - 2.0 * vd;
    const double h_bb = (NB(0, 1, 0) + NB(0, -1, 0)) - 2.0 * vd;
    const double h_ss = (NB(1, 0, 0) + NB(-1, 0, 0)) - 2.0 * vd;
    const double h_ab = ((NB(0, 1, 1) - NB(0, 1, -1)) - (NB(0, -1, 1) - NB(0, -1, -1))) * 0.25;
    const double h_as = ((NB(1, 0, 1) - NB(1, 0, -1)) - (NB(-1, 0, 1) - NB(-1, 0, -1))) * 0.25;
    const double h_bs = ((NB(1, 1, 0) - NB(1, -1, 0)) - (NB(-1, 1, 0) - NB(-1, -1, 0))) * 0.25;
#undef NB
    // cofactors of the symmetric matrix [aa ab as; ab bb bs; as bs ss]
    const double c_aa = h_bb * h_ss - h_bs * h_bs;
    const double c_ab = h_as * h_bs - h_ab * h_ss;
    const double c_as = h_ab * h_bs - h_as * h_bb;
    const double c_bb = h_aa * h_ss - h_as * h_as;
    const double c_bs = h_ab * h_as - h_aa * h_bs;
    const double c_ss = h_aa * h_bb - h_ab * h_ab;
    const double det3 = (h_aa * c_aa + h_ab * c_ab) + h_as * c_as;
    if (det3 == 0.0) return false;
    const double x_a = -(((c_aa * g_a + c_ab * g_b) + c_as * g_s) / det3);
    const double x_b = -(((c_ab * g_a + c_bb * g_b) + c_bs * g_s) / det3);
    const double x_s = -(((c_as * g_a + c_bs * g_b) + c_ss * g_s) / det3);
    if (!(fabs(x_a) <= 1.0 && fabs(x_b) <= 1.0 && fabs(x_s) <= 1.0)) return false;
    out->pa = (float)(((double)ga + x_a) * (double)s);
    out->pb = (float)(((double)gb + x_b) * (double)s);
    out->metric = v;
    out->scale = (float)((1.2 / 9.0) * ((double)Lm + x_s * (double)dL));
    return true;
}

// The samples of (octave o, middle level m), pair k = 2 o + (m - 1), that may hold a point: columns
// first_a..last_a of rows first_b..last_b (empty when a first exceeds its last).  Those at which the filter of level
// m + 1 lies inside the image, less one sample for the block; then the border.  Made on the host.
struct surf_ranges {
    int first_a[SURF_MID], last_a[SURF_MID], first_b[SURF_MID], last_b[SURF_MID];
};

constexpr int XT_A = 64, XT_B = 4, XS_A = XT_A + 2, XS_B = XT_B + 2;   // tile of samples; staged with a ring of one

__global__ __launch_bounds__(256) void surf_extrema_kernel(const float *__restrict__ R, int *__restrict__ cells,
                                                           int *__restrict__ subs, surf_point *__restrict__ recs, surf_geom g,
                                                           float thr, surf_ranges rg)
{
    R += blockIdx.y * g.r_stride;
    cells += blockIdx.y * g.cell_stride;
    subs += blockIdx.y * g.sub_stride;
    recs += blockIdx.y * g.rec_stride;
    __shared__ float sR[SURF_LEV][XS_B * XS_A];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx = (int)blockIdx.x + g.otile_first[1];   // (octave 0 is surf_response_kernel's)
    int o = 1;
#pragma unroll
    for (int k = 2; k < SURF_OCT; ++k) o += bx >= g.otile_first[k] ? 1 : 0;
    const int t = bx - g.otile_first[o];
    const int ta = t % g.tiles_a[o], tb = t / g.tiles_a[o];
    const int ga0 = ta * XT_A, gb0 = tb * XT_B, GA = g.GA[o], GB = g.GB[o];
    // (every load of the thread issued before the first one is waited for)
    constexpr int N = SURF_LEV * XS_B * XS_A, ROUNDS = (N + 255) / 256;
    float v[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int q = tid + 256 * k;
        const int lev = q / (XS_B * XS_A), r = q - lev * (XS_B * XS_A);
        const int rb = r / XS_A, ra = r - rb * XS_A;
        const int ga = ga0 - 1 + ra, gb = gb0 - 1 + rb;
        v[k] = q < N && ga >= 0 && ga < GA && gb >= 0 && gb < GB ? R[g.map_off[4 * o + lev] + (size_t)gb * GA + ga] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int q = tid + 256 * k;
        if (q < N) (&sR[0][0])[q] = v[k];
    }
    __syncthreads();
    const int ga = ga0 + lane, gb = gb0 + wave;
    if (gb >= GB) return;   // (a whole wavefront)
    const int c = (wave + 1) * XS_A + (lane + 1);
#pragma unroll
    for (int m = 1; m <= 2; ++m) {
        const int pair = 2 * o + (m - 1);
        surf_point pt;
        const bool hit = ga >= rg.first_a[pair] && ga <= rg.last_a[pair] && gb >= rg.first_b[pair] && gb <= rg.last_b[pair] &&
                         surf_test(&sR[m - 1][c], &sR[m][c], &sR[m + 1][c], XS_A, ga, gb, thr, 1 << o, g.L[4 * o + m],
                                   g.L[4 * o + m] - g.L[4 * o + m - 1], &pt);
        const unsigned long long bal = __ballot(hit);
        const int n = __popcll(bal);
        const int sub = g.sub_first[pair] + gb * g.tiles_a[o] + ta;
        if (hit) recs[(size_t)sub * SURF_SUB_CAP + __popcll(bal & ((1ull << lane) - 1ull))] = pt;
        if (lane == 0) {
            subs[sub] = n;
            if (n) atomicAdd(&cells[g.cell_first[pair] + gb], n);
        }
    }
}

// Octave 0 (three quarters of all responses) without its maps ever reaching memory: a workgroup stages the integral
// image under a tile of 64 x 8 samples, a ring of one sample and the reach of the octave's largest filter (13 before,
// 14 after: 93 x 37 words, 13.8 KB), makes the four levels of the tile and its ring from there into LDS (66 x 10 x 4
// responses, 10.6 KB; consecutive lanes read consecutive words of the integral tile) and tests both middle levels
// right away, as surf_extrema_kernel does for the other octaves: records per span of a row, counts per span and cell.
constexpr int RS_A = RT_A + 2 + 2 * RT_H + 1, RS_B = RT_B + 2 + 2 * RT_H + 1;   // staged integral tile
constexpr int RR_A = RT_A + 2, RR_B = RT_B + 2;                                // responses of the tile and its ring

__device__ __forceinline__ void surf_octave0_block(const uint32_t *__restrict__ T, int *__restrict__ cells,
                                                   int *__restrict__ subs, surf_point *__restrict__ recs, const surf_geom &g,
                                                   const surf_norms &nm, float thr, const surf_ranges &rg, int block)
{
    static_assert(SIZES[0][SURF_LEV - 1] == 2 * RT_H + 1, "the staged reach is that of octave 0's largest filter");
    __shared__ uint32_t sT[RS_B * RS_A];
    __shared__ float sR[SURF_LEV][RR_B * RR_A];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ta = block % g.tiles_a[0], tb = block / g.tiles_a[0];
    const int a0 = ta * RT_A, b0 = tb * RT_B, A1 = g.A1;
    const int ia0 = a0 - 1 - RT_H, ib0 = b0 - 1 - RT_H;   // the first staged entry of the integral image
    {
        constexpr int N = RS_A * RS_B, ROUNDS = (N + 255) / 256;
        uint32_t v[ROUNDS];   // (all loads of the thread in flight together)
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int q = min(tid + 256 * k, N - 1);
            const int rb = q / RS_A, ra = q - rb * RS_A;
            // (entries outside the integral image are only reached from samples whose filter does not fit)
            const int ia = min(max(ia0 + ra, 0), g.A), ib = min(max(ib0 + rb, 0), g.B);
            v[k] = T[(size_t)ib * A1 + ia];
        }
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int q = tid + 256 * k;
            if (q < N) sT[q] = v[k];
        }
    }
    __syncthreads();
    auto sum = [&](int x0, int y0, int x1, int y1) {
        const int r0 = (y0 - ib0) * RS_A - ia0, r1 = (y1 - ib0) * RS_A - ia0;
        return (int)((sT[r1 + x1] - sT[r0 + x1]) - (sT[r1 + x0] - sT[r0 + x0]));
    };
    for (int q = tid; q < RR_A * RR_B; q += 256) {
        const int rb = q / RR_A, ra = q - rb * RR_A;
        const int a = a0 - 1 + ra, b = b0 - 1 + rb;
#pragma unroll
        for (int i = 0; i < SURF_LEV; ++i) {
            constexpr int Ls[SURF_LEV] = {SIZES[0][0], SIZES[0][1], SIZES[0][2], SIZES[0][3]};
            const int h = (Ls[i] - 1) / 2;
            float r = 0.0f;
            if (a - h >= 0 && a + h <= g.A - 1 && b - h >= 0 && b + h <= g.B - 1)
                r = surf_det(sum, a, b, Ls[i], nm.inv_lobe[i], nm.inv_sq[i]);
            sR[i][q] = r;
        }
    }
    __syncthreads();
    const int ga = a0 + lane;
#pragma unroll
    for (int half = 0; half < RT_B / 4; ++half) {
        const int gb = b0 + wave + 4 * half;
        if (gb >= g.GB[0]) break;   // (a whole wavefront)
        const int c = (wave + 4 * half + 1) * RR_A + (lane + 1);
#pragma unroll
        for (int m = 1; m <= 2; ++m) {
            const int pair = m - 1;
            surf_point pt;
            const bool hit = ga >= rg.first_a[pair] && ga <= rg.last_a[pair] && gb >= rg.first_b[pair] && gb <= rg.last_b[pair] &&
                             surf_test(&sR[m - 1][c], &sR[m][c], &sR[m + 1][c], RR_A, ga, gb, thr, 1, g.L[m], g.L[m] - g.L[m - 1],
                                       &pt);
            const unsigned long long bal = __ballot(hit);
            const int n = __popcll(bal);
            const int sub = g.sub_first[pair] + gb * g.tiles_a[0] + ta;
            if (hit) recs[(size_t)sub * SURF_SUB_CAP + __popcll(bal & ((1ull << lane) - 1ull))] = pt;
            if (lane == 0) {
                subs[sub] = n;
                if (n) atomicAdd(&cells[g.cell_first[pair] + gb], n);
            }
        }
    }
}

// One launch for both: the first resp0_tiles workgroups take octave 0 through LDS, the others make the maps of
// octaves 1 and 2 through the caches, so that for a single image the two kinds of work share the chip.
__global__ __launch_bounds__(256) void surf_response_kernel(const uint32_t *__restrict__ T, float *__restrict__ R,
                                                            int *__restrict__ cells, int *__restrict__ subs,
                                                            surf_point *__restrict__ recs, surf_geom g, surf_norms nm, float thr,
                                                            surf_ranges rg)
{
    T += blockIdx.y * g.t_stride;
    if ((int)blockIdx.x < g.resp0_tiles)
        surf_octave0_block(T, cells + blockIdx.y * g.cell_stride, subs + blockIdx.y * g.sub_stride,
                           recs + blockIdx.y * g.rec_stride, g, nm, thr, rg, (int)blockIdx.x);
    else
        surf_maps_block(T, R + blockIdx.y * g.r_stride, g, nm, (int)blockIdx.x - g.resp0_tiles);
}

// One wavefront per sub-cell, in the order of the output: (octave, level) pair, row of samples, span.
__global__ __launch_bounds__(256) void surf_emit_kernel(const int *__restrict__ cells, const int *__restrict__ subs,
                                                        const surf_point *__restrict__ recs, surf_geom g, int cap, int swap_xy,
                                                        float *__restrict__ xy_out, float *__restrict__ metric_out,
                                                        float *__restrict__ scale_out, int *__restrict__ count)
{
    cells += blockIdx.y * g.cell_stride;
    subs += blockIdx.y * g.sub_stride;
    recs += blockIdx.y * g.rec_stride;
    xy_out += blockIdx.y * (size_t)g.xy_stride;
    metric_out += blockIdx.y * (size_t)g.metric_stride;
    if (scale_out) scale_out += blockIdx.y * (size_t)g.metric_stride;
    count += blockIdx.y * (size_t)g.count_stride;
    const int lane = threadIdx.x & 63, sub = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nsubs = g.sub_first[SURF_MID];
    if (sub >= nsubs) return;
    const int n = subs[sub];
    if (n == 0 && sub != nsubs - 1) return;
    int pair = 0;
#pragma unroll
    for (int k = 1; k < SURF_MID; ++k) pair += sub >= g.sub_first[k] ? 1 : 0;
    const int o = pair >> 1, spans = g.tiles_a[o];
    const int row = (sub - g.sub_first[pair]) / spans, span = (sub - g.sub_first[pair]) - row * spans;
    const int cell = g.cell_first[pair] + row;
    int base = 0;   // points in the cells before this one and in the spans of this cell before this one
    for (int q = lane; q < cell; q += 64) base += cells[q];
    for (int q = lane; q < span; q += 64) base += subs[sub - span + q];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) base += __shfl_xor(base, m, 64);
    if (sub == nsubs - 1 && lane == 0) *count = base + n;   // (the last sub-cell is the last span of the last cell)
    const int pos = base + lane;
    if (lane < n && pos < cap) {
        const surf_point pt = recs[(size_t)sub * SURF_SUB_CAP + lane];
        const float pa = pt.pa + 1.0f, pb = pt.pb + 1.0f;
        xy_out[pos] = swap_xy ? pb : pa;
        xy_out[cap + pos] = swap_xy ? pa : pb;
        metric_out[pos] = pt.metric;
        if (scale_out) scale_out[pos] = pt.scale;
    }
}

#pragma clang fp contract(fast)

surf_ranges make_ranges(const surf_geom &g, int lo)
{
    surf_ranges r;
    for (int o = 0; o < SURF_OCT; ++o)
        for (int m = 1; m <= 2; ++m) {
            const int s = 1 << o, k = 2 * o + (m - 1), htop = (g.L[4 * o + m + 1] - 1) / 2;
            const int lo_g = (htop + s - 1) / s + 1, lo_s = (lo + s - 1) / s;
            auto last_of = [&](int len) {   // the last sample whose block lies inside and whose pixel keeps the border
                const int fit = len - 1 - htop >= 0 ? (len - 1 - htop) / s - 1 : -1;
                const int keep = len - lo - 1 >= 0 ? (len - lo - 1) / s : -1;
                return fit < keep ? fit : keep;
            };
            r.first_a[k] = r.first_b[k] = lo_g > lo_s ? lo_g : lo_s;
            r.last_a[k] = last_of(g.A);
            r.last_b[k] = last_of(g.B);
        }
    return r;
}

surf_norms make_norms(const surf_geom &g)
{
    surf_norms nm;
    for (int k = 0; k < SURF_MAPS; ++k) {
        const int l = g.L[k] / 3;
        nm.inv_lobe[k] = 1.0 / (double)(l * (2 * l - 1));
        nm.inv_sq[k] = 1.0 / (double)(l * l);
    }
    return nm;
}

// The launches over n images whose scratch lies image after image at `scratch`.
int enqueue_images(vohip_ctx *ctx, hipStream_t st, void *scratch, surf_geom g, const uint8_t *d_imgs, int n, int colmajor,
                   double metric_threshold, int border, float *d_xy, float *d_metric, float *d_scale, int cap, int *d_counts)
{
    char *p = (char *)scratch;
    uint32_t *T = (uint32_t *)p;
    p += sizeof(uint32_t) * g.t_stride * n;
    uint32_t *P = (uint32_t *)p;
    p += sizeof(uint32_t) * g.p_stride * n;
    float *R = (float *)p;
    p += sizeof(float) * g.r_stride * n;
    int *cells = (int *)p;
    p += sizeof(int) * g.cell_stride * n;
    int *subs = (int *)p;
    p += sizeof(int) * g.sub_stride * n;
    surf_point *recs = (surf_point *)p;
    g.xy_stride = 2 * cap;
    g.metric_stride = cap;
    g.count_stride = 1;
    const int lo = border < 1 ? 1 : border, nsubs = g.sub_first[SURF_MID];
    hipLaunchKernelGGL(surf_integral_rows_kernel, dim3((g.B + 3) / 4, n), dim3(256), 0, st, d_imgs, T, cells, g);
    const dim3 cgrid((g.A + 63) / 64, (g.nch + 3) / 4, n);
    hipLaunchKernelGGL(surf_integral_colsum_kernel, cgrid, dim3(64, 4), 0, st, (const uint32_t *)T, P, g);
    hipLaunchKernelGGL(surf_integral_colscan_kernel, cgrid, dim3(64, 4), 0, st, T, (const uint32_t *)P, g);
    const surf_norms nm = make_norms(g);
    const surf_ranges rg = make_ranges(g, lo);
    hipLaunchKernelGGL(surf_response_kernel, dim3(g.resp0_tiles + g.tile_first[SURF_MAPS] - g.tile_first[SURF_LEV], n),
                       dim3(256), 0, st, (const uint32_t *)T, R, cells, subs, recs, g, nm, (float)metric_threshold, rg);
    hipLaunchKernelGGL(surf_extrema_kernel, dim3(g.otile_first[SURF_OCT] - g.otile_first[1], n), dim3(256), 0, st,
                       (const float *)R, cells, subs, recs, g, (float)metric_threshold, rg);
    hipLaunchKernelGGL(surf_emit_kernel, dim3((nsubs + 3) / 4, n), dim3(256), 0, st, (const int *)cells, (const int *)subs,
                       (const surf_point *)recs, g, cap, colmajor ? 1 : 0, d_xy, d_metric, d_scale, d_counts);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int check_args(vohip_ctx *ctx, int rows, int cols, double metric_threshold, int border, int cap)
{
    VOHIP_REQUIRE(ctx, rows >= 8 && cols >= 8, "image too small");
    VOHIP_REQUIRE(ctx, (size_t)rows * cols <= ((size_t)1 << 24), "SURF: at most 2^24 pixels (32-bit integral image)");
    VOHIP_REQUIRE(ctx, metric_threshold >= 0.0 && std::isfinite(metric_threshold), "MetricThreshold must be >= 0");
    VOHIP_REQUIRE(ctx, cap >= 1 && cap <= 16384, "cap must be 1..16384");
    VOHIP_REQUIRE(ctx, border >= 0, "negative border");
    return VOHIP_OK;
}

}  // namespace

size_t vohip_surf_scratch_bytes(int rows, int cols)
{
    // (the same for either storage order up to rounding: take the larger)
    const size_t a = image_scratch_bytes(make_geom(rows, cols)), b = image_scratch_bytes(make_geom(cols, rows));
    return a > b ? a : b;
}

// SURF detection on `stream` with caller-owned scratch of vohip_surf_scratch_bytes(rows, cols) bytes.
int vohip_detect_surf_enqueue(vohip_ctx *ctx, hipStream_t st, void *scratch, const uint8_t *d_img, int rows, int cols,
                              int colmajor, double metric_threshold, int border, float *d_xy, float *d_metric,
                              float *d_scale, int cap, int *d_count)
{
    VOHIP_REQUIRE(ctx, d_img && d_xy && d_metric && d_count && scratch, "NULL argument");
    VOHIP_TRY(check_args(ctx, rows, cols, metric_threshold, border, cap));
    const surf_geom g = make_geom(colmajor ? rows : cols, colmajor ? cols : rows);
    return enqueue_images(ctx, st, scratch, g, d_img, 1, colmajor, metric_threshold, border, d_xy, d_metric, d_scale, cap,
                          d_count);
}

// n images, image after image in device memory, on the context's stream: the lists of n single calls, bit for
// bit (every image has its own integral image, responses, cells and output lists).  The scratch of a pass is
// kept under 1 GiB: a long batch runs as several passes over the same memory.
int vohip_detect_surf_batch_enqueue(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                    double metric_threshold, int border, float *d_xy, float *d_metric, int cap,
                                    int *d_counts)
{
    VOHIP_REQUIRE(ctx, d_imgs && d_xy && d_metric && d_counts, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && n <= 65535, "1..65535 images per call");
    VOHIP_TRY(check_args(ctx, rows, cols, metric_threshold, border, cap));
    const surf_geom g = make_geom(colmajor ? rows : cols, colmajor ? cols : rows);
    const size_t per = image_scratch_bytes(g);
    int pass = (int)((size_t)SURF_BATCH_BYTES_MAX / per);
    pass = pass < 1 ? 1 : (pass > n ? n : pass);
    vohip_buf &buf = ctx->bufs[B_SURF_BATCH];
    VOHIP_TRY(vohip_reserve(ctx, buf, per * pass));
    for (int c = 0; c < n; c += pass) {
        const int m = n - c < pass ? n - c : pass;
        VOHIP_TRY(enqueue_images(ctx, ctx->stream, buf.ptr, g, d_imgs + g.img_stride * c, m, colmajor, metric_threshold,
                                 border, d_xy + 2 * (size_t)cap * c, d_metric + (size_t)cap * c, nullptr, cap, d_counts + c));
    }
    return VOHIP_OK;
}

namespace {

int surf_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double metric_threshold, int border,
             float *d_xy, float *d_metric, float *d_scale, int cap, int *d_count)
{
    VOHIP_TRY(check_args(ctx, rows, cols, metric_threshold, border, cap));
    vohip_buf &b = ctx->bufs[B_SURF_SCRATCH];
    VOHIP_TRY(vohip_reserve(ctx, b, vohip_surf_scratch_bytes(rows, cols)));
    return vohip_detect_surf_enqueue(ctx, ctx->stream, b.ptr, d_img, rows, cols, colmajor, metric_threshold, border, d_xy,
                                     d_metric, d_scale, cap, d_count);
}

}  // namespace

extern "C" {

int vohip_detect_surf_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double metric_threshold,
                          int border, float *d_xy, float *d_metric, float *d_scale, int cap, int *d_count)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return surf_dev(ctx, d_img, rows, cols, colmajor, metric_threshold, border, d_xy, d_metric, d_scale, cap, d_count);
}

int vohip_detect_surf_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                double metric_threshold, int border, float *d_xy, float *d_metric, int cap, int *d_counts)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return vohip_detect_surf_batch_enqueue(ctx, d_imgs, n, rows, cols, colmajor, metric_threshold, border, d_xy, d_metric,
                                           cap, d_counts);
}

int vohip_detect_surf(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double metric_threshold,
                      int border, float *xy_out, float *metric_out, float *scale_out, int cap, int *n_out)
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
    VOHIP_TRY(surf_dev(ctx, (const uint8_t *)d_img, rows, cols, colmajor, metric_threshold, border, (float *)d_xy,
                       (float *)d_metric, (float *)d_scale, cap, (int *)d_count));
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
        return vohip_fail(ctx, VOHIP_CAPACITY_EXCEEDED, "%d points found, capacity %d: raise cap or MetricThreshold", total,
                          cap);
    return VOHIP_OK;
}

}  // extern "C"

// ---- SURF descriptors and their matching ----------------------------------------------------------------
//
// Stand-ins for extractFeatures(I, SURFPoints) (genKeypoints.m:32) and matchFeatures(feat0, feat1) with its
// defaults (matching_init.m:26-29, the 'Des_match' bootstrap branch); closed toolbox functions, defined here
// after section 4 of Bay et al.  The image is addressed as (x, y) whatever its storage; box sums are clipped
// to the image;
//   haarX(x, y, z) = sum over [x, x + z/2) x [y - z/2, y - z/2 + z) - sum over [x - z/2, x) x (same rows),
//   haarY the same with the axes exchanged;  S = round(Scale), (X, Y) = (round(x), round(y)).
//   Orientation: the 109 offsets (i, j), i^2 + j^2 < 36: r = exp(-(i^2 + j^2)/8) (haarX, haarY)(X + i S, Y + j S, 4 S);
//     42 windows between the directions 0.15 k and 0.15 k + pi/3 sum the r inside them (two cross products
//     >= 0); the window with the largest squared sum (the first of equals) gives (co, si) = sum / |sum|.
//   Descriptor: samples (i, j), i, j = -10..9, at ((i + .5) Scale, (j + .5) Scale) in the rotated frame, rounded
//     to pixels; (haarX, haarY)(., ., 2 S) weighted by exp(-((i + .5)^2 + (j + .5)^2) / (2 3.3^2)) and rotated
//     back; each of the 4 x 4 subregions of 5 x 5 samples sums dx, dy, |dx|, |dy|; the 64 numbers divided by their
//     Euclidean norm.  fp64 in a fixed order (sums in sample order), stored as fp32; every transcendental comes
//     from tables the host makes, so the kernel's decisions (which window, which pixel) are those of a CPU
//     statement of the same rule, bit for bit.
//   Matching: d(i, j) = sum_k (f0[i][k] - f1[j][k])^2 in fp32, k ascending; for every i the nearest j (the
//     lowest of equals) and the second nearest distance; kept when d1 <= 4 MatchThreshold / 100 and
//     d1 <= MaxRatio^2 d2 (the ratio test on Euclidean distances); pairs in the order of i.
// Kernels: surf_descriptor_kernel, a wavefront per point (109 + 400 Haar responses through LDS, the windows and
// the subregions one lane each); match_nearest_kernel, a wavefront per feature of the first set (lanes stride
// over the second set, two-smallest merge by shuffles); match_compact_kernel, one workgroup, ordered.
namespace {

struct surf_tables {
    double w_ori[36];        // exp(-r2 / 8), r2 = i^2 + j^2
    double w_desc[100];      // exp(-((i + .5)^2 + (j + .5)^2) / (2 3.3^2)), index 10 |i + .5| + |j + .5| (floors)
    double win[4][42];       // window k: first edge (cos, sin), second edge (cos, sin)
    int off_i[112], off_j[112];   // the 109 offsets, j outer, i inner
};

surf_tables make_tables()
{
    surf_tables t{};
    for (int r2 = 0; r2 < 36; ++r2) t.w_ori[r2] = std::exp(-(double)r2 / 8.0);
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j)
            t.w_desc[10 * i + j] = std::exp(-((i + 0.5) * (i + 0.5) + (j + 0.5) * (j + 0.5)) / (2.0 * 3.3 * 3.3));
    for (int k = 0; k < 42; ++k) {
        t.win[0][k] = std::cos(0.15 * k);
        t.win[1][k] = std::sin(0.15 * k);
        t.win[2][k] = std::cos(0.15 * k + 3.14159265358979323846 / 3.0);
        t.win[3][k] = std::sin(0.15 * k + 3.14159265358979323846 / 3.0);
    }
    int m = 0;
    for (int j = -6; j <= 6; ++j)
        for (int i = -6; i <= 6; ++i)
            if (i * i + j * j < 36) {
                t.off_i[m] = i;
                t.off_j[m] = j;
                ++m;
            }
    return t;
}

const surf_tables &host_tables()
{
    static const surf_tables t = make_tables();   // (made once, by whichever thread comes first)
    return t;
}

#pragma clang fp contract(off)

struct px_image {   // the integral image addressed as (x, y)
    const uint32_t *T;
    int A1, W, H, swap;
    __device__ __forceinline__ int box(int x0, int y0, int x1, int y1) const
    {
        x0 = max(x0, 0);
        y0 = max(y0, 0);
        x1 = min(x1, W);
        y1 = min(y1, H);
        if (x1 <= x0 || y1 <= y0) return 0;
        return swap ? box_sum(T, A1, y0, x0, y1, x1) : box_sum(T, A1, x0, y0, x1, y1);
    }
    __device__ __forceinline__ double haar_x(int x, int y, int z) const
    {
        return (double)(box(x, y - z / 2, x + z / 2, y - z / 2 + z) - box(x - z / 2, y - z / 2, x, y - z / 2 + z));
    }
    __device__ __forceinline__ double haar_y(int x, int y, int z) const
    {
        return (double)(box(x - z / 2, y, x - z / 2 + z, y + z / 2) - box(x - z / 2, y - z / 2, x - z / 2 + z, y));
    }
};

constexpr int DESC_WAVES = 4;

// xy: x column then y column (leading dimension ld), 1-based as the toolbox objects hold them
__global__ __launch_bounds__(64 * DESC_WAVES) void surf_descriptor_kernel(px_image im, const surf_tables *__restrict__ tab,
                                                                          const float *__restrict__ xy, int ld,
                                                                          const float *__restrict__ scale, int n, int upright,
                                                                          float *__restrict__ features,
                                                                          float *__restrict__ orientation)
{
    __shared__ double sX[DESC_WAVES][400], sY[DESC_WAVES][400];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = min((int)blockIdx.x * DESC_WAVES + wave, n - 1);   // (a spare wavefront repeats the last point)
    double *rx = sX[wave], *ry = sY[wave];
    const double x = (double)xy[p] - 1.0, y = (double)xy[ld + p] - 1.0, sc = (double)scale[p];
    const int S = (int)floor(sc + 0.5), X = (int)floor(x + 0.5), Y = (int)floor(y + 0.5);
    double co = 1.0, si = 0.0;
    if (!upright) {
        for (int m = lane; m < 109; m += 64) {
            const int i = tab->off_i[m], j = tab->off_j[m];
            const double w = tab->w_ori[i * i + j * j];
            rx[m] = w * im.haar_x(X + i * S, Y + j * S, 4 * S);
            ry[m] = w * im.haar_y(X + i * S, Y + j * S, 4 * S);
        }
        __syncthreads();
        double nn = -1.0, sx = 0.0, sy = 0.0;
        if (lane < 42) {
            const double wu = tab->win[0][lane], wv = tab->win[1][lane], eu = tab->win[2][lane], ev = tab->win[3][lane];
            for (int q = 0; q < 109; ++q) {
                const double c1 = wu * ry[q] - wv * rx[q], c2 = rx[q] * ev - ry[q] * eu;
                if (c1 >= 0.0 && c2 >= 0.0) {
                    sx += rx[q];
                    sy += ry[q];
                }
            }
            nn = sx * sx + sy * sy;
        }
        int k = lane;   // the largest squared sum, the first window of equals
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const double n2 = __shfl_xor(nn, m, 64), x2 = __shfl_xor(sx, m, 64), y2 = __shfl_xor(sy, m, 64);
            const int k2 = __shfl_xor(k, m, 64);
            if (n2 > nn || (n2 == nn && k2 < k)) {
                nn = n2;
                sx = x2;
                sy = y2;
                k = k2;
            }
        }
        if (nn > 0.0) {
            const double nrm = sqrt(nn);
            co = sx / nrm;
            si = sy / nrm;
        }
        __syncthreads();   // (the responses are dead: their memory is used again below)
    }
    for (int t = lane; t < 400; t += 64) {
        const int j = t / 20 - 10, i = t - (t / 20) * 20 - 10;
        const double u = ((double)i + 0.5) * sc, v = ((double)j + 0.5) * sc;
        const int xs = (int)floor((x + (co * u - si * v)) + 0.5), ys = (int)floor((y + (si * u + co * v)) + 0.5);
        const int ai = i >= 0 ? i : -1 - i, aj = j >= 0 ? j : -1 - j;   // floor(|i + .5|)
        const double w = tab->w_desc[10 * ai + aj];
        const double hx = w * im.haar_x(xs, ys, 2 * S), hy = w * im.haar_y(xs, ys, 2 * S);
        rx[t] = co * hx + si * hy;
        ry[t] = co * hy - si * hx;
    }
    __syncthreads();
    double c[4] = {0.0, 0.0, 0.0, 0.0};
    if (lane < 16) {
        const int q = lane >> 2, pp = lane & 3;
        for (int jj = 0; jj < 5; ++jj)
            for (int ii = 0; ii < 5; ++ii) {
                const int t = (5 * q + jj) * 20 + (5 * pp + ii);
                const double dx = rx[t], dy = ry[t];
                c[0] += dx;
                c[1] += dy;
                c[2] += fabs(dx);
                c[3] += fabs(dy);
            }
    }
    __syncthreads();
    if (lane < 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) rx[4 * lane + k] = c[k];
    }
    __syncthreads();
    double ss = 0.0;
    for (int k = 0; k < 64; ++k) ss += rx[k] * rx[k];
    const double nrm = sqrt(ss);
    if ((int)blockIdx.x * DESC_WAVES + wave < n) {
        features[(size_t)p * 64 + lane] = (float)(nrm > 0.0 ? rx[lane] / nrm : 0.0);
        if (orientation && lane == 0) orientation[p] = (float)atan2(si, co);
    }
}

// Nearest and second nearest feature of the second set for feature i of the first: one wavefront per i.
__global__ __launch_bounds__(256) void match_nearest_kernel(const float *__restrict__ f0, int n0, const float *__restrict__ f1,
                                                            int n1, int dim, float thr, float r2, int *__restrict__ match_j,
                                                            float *__restrict__ match_d)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n0) return;
    const float *a = f0 + (size_t)i * dim;
    float d1 = INFINITY, d2 = INFINITY;
    int j1 = 0x7fffffff;
    for (int j = lane; j < n1; j += 64) {
        const float *b = f1 + (size_t)j * dim;
        float d = 0.0f;
        for (int k = 0; k < dim; ++k) {
            const float e = a[k] - b[k];
            d += e * e;
        }
        if (d < d1) {   // (a lane sees its j in ascending order: the first of equals stays)
            d2 = d1;
            d1 = d;
            j1 = j;
        } else if (d < d2) {
            d2 = d;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float e1 = __shfl_xor(d1, m, 64), e2 = __shfl_xor(d2, m, 64);
        const int k1 = __shfl_xor(j1, m, 64);
        // the two smallest of {d1, d2, e1, e2}; of equal nearest distances the lower index
        const bool theirs = e1 < d1 || (e1 == d1 && k1 < j1);
        const float lo = theirs ? e1 : d1, other = theirs ? d1 : e1, second = theirs ? e2 : d2;
        j1 = theirs ? k1 : j1;
        d1 = lo;
        d2 = fminf(other, second);
    }
    if (lane == 0) {
        bool keep = j1 != 0x7fffffff && d1 <= thr;
        if (keep && n1 > 1) keep = d1 <= r2 * d2;
        match_j[i] = keep ? j1 : -1;
        match_d[i] = d1;
    }
}

// The same for features of at most 64 numbers (SURF's): a workgroup takes 16 features of the first set (4 per
// wavefront) and walks the second set in tiles of 64 staged in LDS (rows 65 floats apart: a lane reads its own row
// without bank conflicts); a lane keeps one feature of the tile against its wavefront's four, so every number of the
// tile is read once per four distances and the first set's numbers come as 16-byte broadcasts.  Each distance is
// still the sum over k ascending in fp32: the results are those of match_nearest_kernel, bit for bit.
constexpr int MT_I = 16, MT_J = 64, MT_D = 64;
__global__ __launch_bounds__(256) void match_nearest_tiled_kernel(const float *__restrict__ f0, int n0,
                                                                  const float *__restrict__ f1, int n1, int dim, float thr,
                                                                  float r2, int *__restrict__ match_j,
                                                                  float *__restrict__ match_d)
{
    __shared__ __align__(16) float sA[MT_I][MT_D];
    __shared__ float sB[MT_J][MT_D + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * MT_I;
    for (int q = tid; q < MT_I * MT_D; q += 256) {
        const int r = q / MT_D, k = q - r * MT_D;
        sA[r][k] = i0 + r < n0 && k < dim ? f0[(size_t)(i0 + r) * dim + k] : 0.0f;
    }
    float d1[4], d2[4];
    int j1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d1[r] = d2[r] = INFINITY;
        j1[r] = 0x7fffffff;
    }
    const int kmax = (dim + 3) & ~3;   // (numbers past dim are zero on both sides: (0 - 0)^2 adds nothing)
    for (int j0 = 0; j0 < n1; j0 += MT_J) {
        __syncthreads();
        for (int q = tid; q < MT_J * MT_D; q += 256) {
            const int r = q / MT_D, k = q - r * MT_D;
            sB[r][k] = j0 + r < n1 && k < dim ? f1[(size_t)(j0 + r) * dim + k] : 0.0f;
        }
        __syncthreads();
        float d[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = 0; k < kmax; k += 4) {
            float b[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = sB[lane][k + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 a = *reinterpret_cast<const float4 *>(&sA[4 * wave + r][k]);
                const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float e = av[c] - b[c];
                    d[r] += e * e;
                }
            }
        }
        const int j = j0 + lane;
        if (j < n1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (d[r] < d1[r]) {
                    d2[r] = d1[r];
                    d1[r] = d[r];
                    j1[r] = j;
                } else if (d[r] < d2[r]) {
                    d2[r] = d[r];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a1 = d1[r], a2 = d2[r];
        int k1 = j1[r];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float e1 = __shfl_xor(a1, m, 64), e2 = __shfl_xor(a2, m, 64);
            const int q1 = __shfl_xor(k1, m, 64);
            const bool theirs = e1 < a1 || (e1 == a1 && q1 < k1);
            const float lo = theirs ? e1 : a1, other = theirs ? a1 : e1, second = theirs ? e2 : a2;
            k1 = theirs ? q1 : k1;
            a1 = lo;
            a2 = fminf(other, second);
        }
        const int i = i0 + 4 * wave + r;
        if (lane == 0 && i < n0) {
            bool keep = k1 != 0x7fffffff && a1 <= thr;
            if (keep && n1 > 1) keep = a1 <= r2 * a2;
            match_j[i] = keep ? k1 : -1;
            match_d[i] = a1;
        }
    }
}

// The kept pairs in the order of i: one workgroup of 1024 threads walks the first set in spans of 1024.
__global__ __launch_bounds__(1024) void match_compact_kernel(const int *__restrict__ match_j, const float *__restrict__ match_d,
                                                             int n0, int *__restrict__ pairs, float *__restrict__ metric,
                                                             int *__restrict__ count)
{
    __shared__ int sWave[16];
    __shared__ int sBase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) sBase = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n0; i0 += 1024) {
        const int i = i0 + tid;
        const int j = i < n0 ? match_j[i] : -1;
        const unsigned long long bal = __ballot(j >= 0);
        if (lane == 0) sWave[wave] = __popcll(bal);
        __syncthreads();
        int before = sBase;
        for (int w = 0; w < wave; ++w) before += sWave[w];
        if (j >= 0) {
            const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
            pairs[2 * pos] = i;
            pairs[2 * pos + 1] = j;
            metric[pos] = match_d[i];
        }
        __syncthreads();
        if (tid == 0) {
            int total = 0;
            for (int w = 0; w < 16; ++w) total += sWave[w];
            sBase += total;
        }
        __syncthreads();
    }
    if (tid == 0) *count = sBase;
}

#pragma clang fp contract(fast)

int features_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, const float *d_xy, int ld,
                 const float *d_scale, int n, int upright, float *d_features, float *d_orientation)
{
    VOHIP_REQUIRE(ctx, d_img && d_xy && d_scale && d_features, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && ld >= n, "n >= 1 points, ld >= n");
    VOHIP_TRY(check_args(ctx, rows, cols, 0.0, 0, 1));
    const surf_geom g = make_geom(colmajor ? rows : cols, colmajor ? cols : rows);
    vohip_buf &b = ctx->bufs[B_SURF_SCRATCH];
    VOHIP_TRY(vohip_reserve(ctx, b, vohip_surf_scratch_bytes(rows, cols)));
    void *d_tab;
    VOHIP_TRY(vohip_upload(ctx, B_SURF_TABLES, &host_tables(), sizeof(surf_tables), &d_tab));
    hipStream_t st = ctx->stream;
    char *p = (char *)b.ptr;
    uint32_t *T = (uint32_t *)p;
    p += sizeof(uint32_t) * g.t_stride;
    uint32_t *P = (uint32_t *)p;
    p += sizeof(uint32_t) * g.p_stride + sizeof(float) * g.r_stride;
    int *cells = (int *)p;
    hipLaunchKernelGGL(surf_integral_rows_kernel, dim3((g.B + 3) / 4, 1), dim3(256), 0, st, d_img, T, cells, g);
    const dim3 cgrid((g.A + 63) / 64, (g.nch + 3) / 4, 1);
    hipLaunchKernelGGL(surf_integral_colsum_kernel, cgrid, dim3(64, 4), 0, st, (const uint32_t *)T, P, g);
    hipLaunchKernelGGL(surf_integral_colscan_kernel, cgrid, dim3(64, 4), 0, st, T, (const uint32_t *)P, g);
    px_image im;
    im.T = T;
    im.A1 = g.A1;
    im.W = cols;
    im.H = rows;
    im.swap = colmajor ? 1 : 0;
    hipLaunchKernelGGL(surf_descriptor_kernel, dim3((n + DESC_WAVES - 1) / DESC_WAVES), dim3(64 * DESC_WAVES), 0, st, im,
                       (const surf_tables *)d_tab, d_xy, ld, d_scale, n, upright ? 1 : 0, d_features, d_orientation);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int match_dev(vohip_ctx *ctx, const float *d_f0, int n0, const float *d_f1, int n1, int dim, double match_threshold,
              double max_ratio, int *d_pairs, float *d_metric, int *d_count)
{
    VOHIP_REQUIRE(ctx, d_f0 && d_f1 && d_pairs && d_metric && d_count, "NULL argument");
    VOHIP_REQUIRE(ctx, n0 >= 1 && n1 >= 1 && dim >= 1, "empty feature set");
    VOHIP_REQUIRE(ctx, match_threshold > 0.0 && match_threshold <= 100.0, "MatchThreshold must lie in (0, 100]");
    VOHIP_REQUIRE(ctx, max_ratio > 0.0 && max_ratio <= 1.0, "MaxRatio must lie in (0, 1]");
    vohip_buf &b = ctx->bufs[B_SURF_MATCH];
    VOHIP_TRY(vohip_reserve(ctx, b, (sizeof(int) + sizeof(float)) * (size_t)n0));
    int *match_j = (int *)b.ptr;
    float *match_d = (float *)(match_j + n0);
    hipStream_t st = ctx->stream;
    if (dim <= MT_D)
        hipLaunchKernelGGL(match_nearest_tiled_kernel, dim3((n0 + MT_I - 1) / MT_I), dim3(256), 0, st, d_f0, n0, d_f1, n1, dim,
                           (float)(4.0 * match_threshold / 100.0), (float)(max_ratio * max_ratio), match_j, match_d);
    else
        hipLaunchKernelGGL(match_nearest_kernel, dim3((n0 + 3) / 4), dim3(256), 0, st, d_f0, n0, d_f1, n1, dim,
                           (float)(4.0 * match_threshold / 100.0), (float)(max_ratio * max_ratio), match_j, match_d);
    hipLaunchKernelGGL(match_compact_kernel, dim3(1), dim3(1024), 0, st, (const int *)match_j, (const float *)match_d, n0,
                       d_pairs, d_metric, d_count);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

}  // namespace

extern "C" {

int vohip_extract_surf_features_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, const float *d_xy,
                                    int ld, const float *d_scale, int n, int upright, float *d_features,
                                    float *d_orientation)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return features_dev(ctx, d_img, rows, cols, colmajor, d_xy, ld, d_scale, n, upright, d_features, d_orientation);
}

int vohip_extract_surf_features(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, const float *xy,
                                const float *scale, int n, int upright, float *features_out, float *orientation_out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, img && xy && scale && features_out, "NULL argument");
    VOHIP_REQUIRE(ctx, rows >= 1 && cols >= 1 && n >= 1, "bad size");
    for (int i = 0; i < n; ++i)   // (the sampling pattern is 20 Scale wide: integer pixel arithmetic stays far from its limits)
        VOHIP_REQUIRE(ctx, scale[i] > 0.0f && scale[i] <= 1.0e4f && std::fabs(xy[i]) <= 1.0e6f && std::fabs(xy[n + i]) <= 1.0e6f,
                      "points must be finite, within 1e6 pixels of the image, with 0 < Scale <= 1e4");
    void *d_img, *d_xy, *d_scale, *d_feat, *d_ori;
    VOHIP_TRY(vohip_upload(ctx, B_IMG_STAGE, img, (size_t)rows * cols, &d_img));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, xy, sizeof(float) * 2 * (size_t)n, &d_xy));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, scale, sizeof(float) * (size_t)n, &d_scale));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, sizeof(float) * 64 * (size_t)n, &d_feat));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, sizeof(float) * (size_t)n, &d_ori));
    VOHIP_TRY(features_dev(ctx, (const uint8_t *)d_img, rows, cols, colmajor, (const float *)d_xy, n, (const float *)d_scale, n,
                           upright, (float *)d_feat, (float *)d_ori));
    VOHIP_TRY(vohip_download(ctx, features_out, d_feat, sizeof(float) * 64 * (size_t)n));
    if (orientation_out) VOHIP_TRY(vohip_download(ctx, orientation_out, d_ori, sizeof(float) * (size_t)n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_match_features_dev(vohip_ctx *ctx, const float *d_f0, int n0, const float *d_f1, int n1, int dim,
                             double match_threshold, double max_ratio, int *d_pairs, float *d_metric, int *d_count)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return match_dev(ctx, d_f0, n0, d_f1, n1, dim, match_threshold, max_ratio, d_pairs, d_metric, d_count);
}

int vohip_match_features(vohip_ctx *ctx, const float *f0, int n0, const float *f1, int n1, int dim, double match_threshold,
                         double max_ratio, int *pairs_out, float *metric_out, int *n_pairs_out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, f0 && f1 && pairs_out && metric_out && n_pairs_out, "NULL argument");
    VOHIP_REQUIRE(ctx, n0 >= 1 && n1 >= 1 && dim >= 1, "empty feature set");
    void *d_f0, *d_f1, *d_pairs, *d_metric, *d_count;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, f0, sizeof(float) * (size_t)n0 * dim, &d_f0));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, f1, sizeof(float) * (size_t)n1 * dim, &d_f1));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, sizeof(int) * 2 * (size_t)n0, &d_pairs));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, sizeof(float) * (size_t)n0, &d_metric));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE4, nullptr, 256, &d_count));
    VOHIP_TRY(match_dev(ctx, (const float *)d_f0, n0, (const float *)d_f1, n1, dim, match_threshold, max_ratio, (int *)d_pairs,
                        (float *)d_metric, (int *)d_count));
    int m = 0;
    VOHIP_TRY(vohip_download(ctx, &m, d_count, sizeof(int)));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    VOHIP_TRY(vohip_download(ctx, pairs_out, d_pairs, sizeof(int) * 2 * (size_t)m));
    VOHIP_TRY(vohip_download(ctx, metric_out, d_metric, sizeof(float) * (size_t)m));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_pairs_out = m;
    return VOHIP_OK;
}

}  // extern "C"
