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
//   surf_integral_rows_kernel      one wavefront per image row: running sums along the fast axis (wavefront
//                                  scans with a carry) into the integral image's rows (uint32, a zero first row
//                                  and column so that box sums need no edge cases).
//   surf_integral_colsum_kernel /  the slow axis in chunks of 32 rows: per-chunk column sums, then every
//   surf_integral_colscan_kernel   chunk adds the chunks before it and runs down its own rows (all 32 loads
//                                  of a thread in flight together); consecutive threads = consecutive columns.
//   surf_response_kernel           all 12 (octave, level) maps in one launch, a thread per sample: 32 lookups
//                                  of the integral image (which stays in L2: 1.9 MB for 1241 x 376), exact
//                                  integer box sums, fp64 determinant.
//   surf_count_kernel /            one wavefront per (octave, level, sample row) cell: tests its samples,
//   surf_emit_kernel               counts the points; the second pass ranks every cell behind the cells
//                                  before it and writes the points in order.  No atomics; deterministic.
// Per image: the uint8 image read once, the integral image written twice and read ~3 times, 2.45 M
// responses (9.8 MB for 1241 x 376) written once and read ~twice.
#include "vohip_internal.h"

#include <cmath>

namespace {

constexpr int SURF_OCT = 3, SURF_LEV = 4, SURF_MAPS = SURF_OCT * SURF_LEV, SURF_MID = 2 * SURF_OCT;
constexpr int SURF_CH = 32;   // rows of the integral image per chunk of the slow-axis pass
constexpr int SURF_BATCH_BYTES_MAX = 1 << 30;   // scratch of one pass over a batch of images

struct surf_geom {
    int A, B, A1;                        // image axes; row length of the integral image
    int GA[SURF_OCT], GB[SURF_OCT];      // samples per octave
    int L[SURF_MAPS];                    // filter side of map 4 o + level
    unsigned map_off[SURF_MAPS];         // of a map within an image's responses (floats)
    int tile_first[SURF_MAPS + 1];       // workgroups of the response launch before map k
    int tiles_a[SURF_OCT];
    int cell_first[SURF_MID + 1];        // cells before (octave, middle level) pair k = 2 o + (m - 1)
    int nch;                             // chunks of the slow-axis pass
    // per image of a batch
    size_t img_stride, t_stride, p_stride, r_stride, cell_stride;
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
    int tiles = 0, cells = 0;
    for (int o = 0; o < SURF_OCT; ++o) {
        const int s = 1 << o;
        g.GA[o] = (A + s - 1) / s;
        g.GB[o] = (B + s - 1) / s;
        g.tiles_a[o] = (g.GA[o] + 63) / 64;
        for (int i = 0; i < SURF_LEV; ++i) {
            g.L[4 * o + i] = SIZES[o][i];
            g.map_off[4 * o + i] = off;
            off += (unsigned)((size_t)g.GA[o] * g.GB[o] + 63) / 64 * 64;
            g.tile_first[4 * o + i] = tiles;
            tiles += g.tiles_a[o] * ((g.GB[o] + 3) / 4);
        }
        for (int m = 0; m < 2; ++m) {
            g.cell_first[2 * o + m] = cells;
            cells += g.GB[o];
        }
    }
    g.tile_first[SURF_MAPS] = tiles;
    g.cell_first[SURF_MID] = cells;
    g.nch = (B + SURF_CH - 1) / SURF_CH;
    auto r256 = [](size_t v) { return (v + 255) / 256 * 256; };
    g.img_stride = (size_t)A * B;
    g.t_stride = r256(sizeof(uint32_t) * (size_t)g.A1 * (B + 1)) / sizeof(uint32_t);
    g.p_stride = r256(sizeof(uint32_t) * (size_t)g.nch * A) / sizeof(uint32_t);
    g.r_stride = r256(sizeof(float) * (size_t)off) / sizeof(float);
    g.cell_stride = r256(sizeof(int) * (size_t)cells) / sizeof(int);
    return g;
}

size_t image_scratch_bytes(const surf_geom &g)
{
    return sizeof(uint32_t) * (g.t_stride + g.p_stride) + sizeof(float) * g.r_stride + sizeof(int) * g.cell_stride;
}

__global__ __launch_bounds__(256) void surf_integral_rows_kernel(const uint8_t *__restrict__ img, uint32_t *__restrict__ T,
                                                                 surf_geom g)
{
    img += blockIdx.y * g.img_stride;
    T += blockIdx.y * g.t_stride;
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0)
        for (int a = threadIdx.x; a < g.A1; a += 256) T[a] = 0u;
    if (b >= g.B) return;
    uint32_t carry = 0u;
    uint32_t *row = T + (size_t)(b + 1) * g.A1;
    if (lane == 0) row[0] = 0u;
    for (int c = 0; c < g.A; c += 64) {
        const int a = c + lane;
        uint32_t v = a < g.A ? (uint32_t)img[(size_t)b * g.A + a] : 0u;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64);
            if (lane >= d) v += t;
        }
        if (a < g.A) row[a + 1] = v + carry;
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

__global__ __launch_bounds__(256) void surf_response_kernel(const uint32_t *__restrict__ T, float *__restrict__ R, surf_geom g,
                                                            surf_norms nm)
{
    T += blockIdx.y * g.t_stride;
    R += blockIdx.y * g.r_stride;
    int map = 0;
#pragma unroll
    for (int k = 1; k < SURF_MAPS; ++k) map += (int)blockIdx.x >= g.tile_first[k] ? 1 : 0;
    const int o = map >> 2, s = 1 << o;
    const int t = (int)blockIdx.x - g.tile_first[map];
    const int ga = (t % g.tiles_a[o]) * 64 + (threadIdx.x & 63), gb = (t / g.tiles_a[o]) * 4 + (threadIdx.x >> 6);
    if (ga >= g.GA[o] || gb >= g.GB[o]) return;
    const int L = g.L[map], l = L / 3, h = (L - 1) / 2;
    const int a = ga * s, b = gb * s, A1 = g.A1;
    float r = 0.0f;
    if (a - h >= 0 && a + h <= g.A - 1 && b - h >= 0 && b + h <= g.B - 1) {
        const int nxx = box_sum(T, A1, a - h, b - l + 1, a + h + 1, b + l) -
                        3 * box_sum(T, A1, a - h + l, b - l + 1, a - h + 2 * l, b + l);
        const int nyy = box_sum(T, A1, a - l + 1, b - h, a + l, b + h + 1) -
                        3 * box_sum(T, A1, a - l + 1, b - h + l, a + l, b - h + 2 * l);
        const int nxy = box_sum(T, A1, a - l, b - l, a, b) + box_sum(T, A1, a + 1, b + 1, a + 1 + l, b + 1 + l) -
                        box_sum(T, A1, a + 1, b - l, a + 1 + l, b) - box_sum(T, A1, a - l, b + 1, a, b + 1 + l);
        const double dxx = (double)nxx * nm.inv_lobe[map], dyy = (double)nyy * nm.inv_lobe[map];
        const double dxy = (double)nxy * nm.inv_sq[map];
        const double t1 = dxx * dyy, t2 = dxy * dxy, t3 = 0.81 * t2;
        r = (float)(t1 - t3);
    }
    R[g.map_off[map] + (size_t)gb * g.GA[o] + ga] = r;
}

struct surf_point {
    float pa, pb, metric, scale;
};

// The sample (ga, gb) of the middle level whose map is R1 (R0 below, R2 above): an interest point?
__device__ __forceinline__ bool surf_test(const float *__restrict__ R0, const float *__restrict__ R1,
                                          const float *__restrict__ R2, int GA, int ga, int gb, float thr, int s, int Lm,
                                          int dL, surf_point *out)
{
    const size_t c = (size_t)gb * GA + ga;
    const float v = R1[c];
    if (!(v > thr)) return false;
    const float *lv[3] = {R0 + c, R1 + c, R2 + c};
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

// The cell of wavefront `cell`: its maps, its row of samples and the range of samples that may hold a point.
struct surf_cell {
    const float *R0, *R1, *R2;
    int GA, gb, s, Lm, dL, first, last;   // samples first..last (empty when first > last)
};

__device__ __forceinline__ surf_cell cell_of(const surf_geom &g, const float *__restrict__ R, int cell, int lo)
{
    int pair = 0;
#pragma unroll
    for (int k = 1; k < SURF_MID; ++k) pair += cell >= g.cell_first[k] ? 1 : 0;
    const int o = pair >> 1, m = 1 + (pair & 1), s = 1 << o;
    surf_cell c;
    c.GA = g.GA[o];
    c.gb = cell - g.cell_first[pair];
    c.s = s;
    c.Lm = g.L[4 * o + m];
    c.dL = c.Lm - g.L[4 * o + m - 1];
    c.R0 = R + g.map_off[4 * o + m - 1];
    c.R1 = R + g.map_off[4 * o + m];
    c.R2 = R + g.map_off[4 * o + m + 1];
    // samples at which the filter of level m + 1 lies inside the image, less one sample for the block; then the border
    const int htop = (g.L[4 * o + m + 1] - 1) / 2;
    const int lo_g = (htop + s - 1) / s + 1;
    const int hi_a = (g.A - 1 - htop >= 0 ? (g.A - 1 - htop) / s : -1) - 1;
    const int hi_b = (g.B - 1 - htop >= 0 ? (g.B - 1 - htop) / s : -1) - 1;
    const int b = c.gb * s;
    const bool row_ok = c.gb >= lo_g && c.gb <= hi_b && b >= lo && b < g.B - lo;
    c.first = max(lo_g, (lo + s - 1) / s);
    c.last = min(hi_a, (g.A - lo - 1) / s);   // a = ga s <= A - lo - 1
    if (!row_ok) c.last = c.first - 1;
    return c;
}

__global__ __launch_bounds__(256) void surf_count_kernel(const float *__restrict__ R, int *__restrict__ cells, surf_geom g,
                                                         float thr, int lo)
{
    R += blockIdx.y * g.r_stride;
    cells += blockIdx.y * g.cell_stride;
    const int lane = threadIdx.x & 63, cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= g.cell_first[SURF_MID]) return;
    const surf_cell c = cell_of(g, R, cell, lo);
    int n = 0;
    for (int ga0 = c.first; ga0 <= c.last; ga0 += 64) {
        const int ga = ga0 + lane;
        surf_point pt;
        const bool hit = ga <= c.last && surf_test(c.R0, c.R1, c.R2, c.GA, ga, c.gb, thr, c.s, c.Lm, c.dL, &pt);
        n += __popcll(__ballot(hit));
    }
    if (lane == 0) cells[cell] = n;
}

__global__ __launch_bounds__(256) void surf_emit_kernel(const float *__restrict__ R, const int *__restrict__ cells, surf_geom g,
                                                        float thr, int lo, int cap, int swap_xy, float *__restrict__ xy_out,
                                                        float *__restrict__ metric_out, float *__restrict__ scale_out,
                                                        int *__restrict__ count)
{
    R += blockIdx.y * g.r_stride;
    cells += blockIdx.y * g.cell_stride;
    xy_out += blockIdx.y * (size_t)g.xy_stride;
    metric_out += blockIdx.y * (size_t)g.metric_stride;
    if (scale_out) scale_out += blockIdx.y * (size_t)g.metric_stride;
    count += blockIdx.y * (size_t)g.count_stride;
    const int lane = threadIdx.x & 63, cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ncells = g.cell_first[SURF_MID];
    if (cell >= ncells) return;
    int base = 0;   // points in the cells before this one
    for (int q = lane; q < cell; q += 64) base += cells[q];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) base += __shfl_xor(base, m, 64);
    const int mine = cells[cell];
    if (cell == ncells - 1 && lane == 0) *count = base + mine;
    if (mine == 0) return;
    const surf_cell c = cell_of(g, R, cell, lo);
    int run = 0;
    for (int ga0 = c.first; ga0 <= c.last; ga0 += 64) {
        const int ga = ga0 + lane;
        surf_point pt;
        const bool hit = ga <= c.last && surf_test(c.R0, c.R1, c.R2, c.GA, ga, c.gb, thr, c.s, c.Lm, c.dL, &pt);
        const unsigned long long bal = __ballot(hit);
        const int pos = base + run + __popcll(bal & ((1ull << lane) - 1ull));
        run += __popcll(bal);
        if (hit && pos < cap) {
            const float pa = pt.pa + 1.0f, pb = pt.pb + 1.0f;
            xy_out[pos] = swap_xy ? pb : pa;
            xy_out[cap + pos] = swap_xy ? pa : pb;
            metric_out[pos] = pt.metric;
            if (scale_out) scale_out[pos] = pt.scale;
        }
    }
}

#pragma clang fp contract(fast)

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

// The six launches over n images whose scratch lies image after image at `scratch`.
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
    g.xy_stride = 2 * cap;
    g.metric_stride = cap;
    g.count_stride = 1;
    const int lo = border < 1 ? 1 : border, ncells = g.cell_first[SURF_MID];
    hipLaunchKernelGGL(surf_integral_rows_kernel, dim3((g.B + 3) / 4, n), dim3(256), 0, st, d_imgs, T, g);
    const dim3 cgrid((g.A + 63) / 64, (g.nch + 3) / 4, n);
    hipLaunchKernelGGL(surf_integral_colsum_kernel, cgrid, dim3(64, 4), 0, st, (const uint32_t *)T, P, g);
    hipLaunchKernelGGL(surf_integral_colscan_kernel, cgrid, dim3(64, 4), 0, st, T, (const uint32_t *)P, g);
    hipLaunchKernelGGL(surf_response_kernel, dim3(g.tile_first[SURF_MAPS], n), dim3(256), 0, st, (const uint32_t *)T, R, g,
                       make_norms(g));
    hipLaunchKernelGGL(surf_count_kernel, dim3((ncells + 3) / 4, n), dim3(256), 0, st, (const float *)R, cells, g,
                       (float)metric_threshold, lo);
    hipLaunchKernelGGL(surf_emit_kernel, dim3((ncells + 3) / 4, n), dim3(256), 0, st, (const float *)R, (const int *)cells, g,
                       (float)metric_threshold, lo, cap, colmajor ? 1 : 0, d_xy, d_metric, d_scale, d_counts);
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
