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
// Kernels (the response map never goes to memory):
//   corner_tile_kernel   one fused tile kernel: uint8 tile + 4 px halo (1 gradient + 2 Gaussian + 1
//                        non-maximum suppression) -> LDS (all loads of a thread in flight together), the
//                        row pass of the Gaussian over gradient products made on the way, the column pass
//                        and the response of the tile and a 1 px ring around it (two outputs per thread,
//                        packed fp32), 24 KB of LDS per workgroup; every strict 3x3
//                        maximum of the tile inside the border margin becomes a 16-byte record (pixel
//                        index, response, sub-pixel position) in the tile's own list, in pixel order
//                        (ballot compaction, no atomics); tile maximum -> one atomicMax per workgroup.
//                        fp32, evaluated in a fixed order without FMA contraction (bit-reproducible).
//   corner_count_kernel  one wavefront per tile: applies R > MinQuality * max(R) to the tile's records and
//                        counts the survivors per (tile, image row) cell.
//   corner_emit_kernel   one wavefront per tile: prefix over the cells in pixel order, then the survivors
//                        go to their ranks: ordered and deterministic, no atomics.
// Per image: the uint8 image is read once (plus the halos), ~16 bytes are written and read per local
// maximum and 12 bytes written per corner.
//
// FAST (detectFASTFeatures as genKeypoints.m:19 calls it, options 'MinContrast' / 'MinQuality',
// main_eval.m:172-173) shares everything but the response: fast_tile_kernel scores the segment test of
// Rosten & Drummond on the 16-pixel circle of radius 3 (9 contiguous pixels all brighter than the
// centre + t or all darker than the centre - t, t = round(255 MinContrast)):
//   score = max over the 16 arcs of 9 of max(min(I_q - I_p), min(I_p - I_q)) - 1,
// the largest t for which the pixel still passes; R = score where score >= t, else 0; pixels closer than
// 3 px to an edge score 0; then the same strict 3x3 maxima, R > MinQuality * max(R), border margin and
// ordered emit as above, at integer positions.  Exact integer arithmetic.
#include "vohip_internal.h"

#include <cmath>

#include "fast_score.h"

namespace {

constexpr int TA = 64, TB = 16, FS = 5, HALF = FS / 2;
constexpr int RA = TA + 2, RB = TB + 2;                        // response tile: the tile and a 1 px ring
constexpr int PA = RA + 2 * HALF, PB = RB + 2 * HALF;          // gradient-product tile
constexpr int IA = PA + 2, IB = PB + 2;                        // staged image tile
constexpr int TILE_CAP = TA * TB / 4;                          // strict 3x3 maxima cannot touch: at most one per 2x2

typedef float v2f __attribute__((ext_vector_type(2)));

struct taps5 {
    float w[FS];
};

struct corner_rec {   // 16 bytes
    int key;          // pixel index b * A + a
    float r, pa, pb;  // response, sub-pixel position (0-based)
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

// The second half of a tile kernel: sR holds the responses of the tile and a 1 px ring, `best` the
// thread's maximum over the tile's own pixels.  Image maximum by one atomic per workgroup, then the
// strict 3x3 maxima of the tile inside the border margin become records in pixel order.
template <bool SUBPIXEL>
__device__ __forceinline__ void tile_maxima(const float *sR, float best, unsigned *sMaxp, int *sCnt, int A, int B, int a0,
                                            int b0, int lo, int tile, corner_rec *__restrict__ recs,
                                            int *__restrict__ tile_count, unsigned *__restrict__ max_now)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned kb = ordered_bits(best);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) kb = max(kb, (unsigned)__shfl_xor((int)kb, m, 64));
    if (lane == 0) atomicMax(sMaxp, kb);
    __syncthreads();
    // (a maximum only grows: a tile whose own maximum does not exceed the value it reads has nothing to add, and
    // most tiles do not -- the atomics of the tiles of an image all go to one address and would queue there)
    if (tid == 0 && *sMaxp > __hip_atomic_load(max_now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_now, *sMaxp);
    // strict 3x3 maxima of the tile, in pixel order: pixel q = 256 k + tid of the tile is (row q / 64,
    // column q % 64), so wavefront `wave` of round k holds one tile row and the order is (k, wave, lane)
    bool is_max[4];
    unsigned long long bal[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rb = 4 * k + wave, ra = lane;
        const int a = a0 + ra, b = b0 + rb;
        bool m = false;
        if (a >= lo && a < A - lo && b >= lo && b < B - lo) {
            const float *c = sR + (rb + 1) * RA + (ra + 1);
            const float r = c[0];
            m = r > c[-1] && r > c[1] && r > c[-RA] && r > c[RA] && r > c[-RA - 1] && r > c[-RA + 1] && r > c[RA - 1] &&
                r > c[RA + 1];
        }
        is_max[k] = m;
        bal[k] = __ballot(m);
        if (lane == 0) sCnt[4 * k + wave] = __popcll(bal[k]);
    }
    __syncthreads();
    // Exclusive prefix of the 16 (round, wavefront) counts, by every wavefront for itself; the maxima's tile
    // pixels then go to a list in pixel order and ONE thread per maximum makes its record: the sub-pixel
    // arithmetic runs once per maximum, not once per wavefront and round that holds a maximum.
    static_assert(TILE_CAP <= 256 && TA == 64 && TB == 16, "one thread per record; 4 rounds of 4 tile rows");
    __shared__ unsigned short sList[TILE_CAP];
    const int mine = lane < 16 ? sCnt[lane] : 0;
    int inc = mine;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if (lane >= d) inc += t;
    }
    const int exc = inc - mine;
    const int total = __shfl(inc, 15, 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int before = __shfl(exc, 4 * k + wave, 64);
        if (is_max[k])
            sList[before + __popcll(bal[k] & ((1ull << lane) - 1ull))] = (unsigned short)((4 * k + wave) * 64 + lane);
    }
    __syncthreads();
    if (tid < total) {
        const int q = sList[tid], rb = q >> 6, ra = q & 63;
        const int a = a0 + ra, b = b0 + rb;
        const float *cc = sR + (rb + 1) * RA + (ra + 1);
        const float r = cc[0];
        const float da = cc[1] - cc[-1], dda = (cc[1] - 2.0f * r) + cc[-1];
        const float db = cc[RA] - cc[-RA], ddb = (cc[RA] - 2.0f * r) + cc[-RA];
        float oa = dda < 0.0f ? -0.5f * da / dda : 0.0f;
        float ob = ddb < 0.0f ? -0.5f * db / ddb : 0.0f;
        oa = SUBPIXEL ? fminf(fmaxf(oa, -0.5f), 0.5f) : 0.0f;
        ob = SUBPIXEL ? fminf(fmaxf(ob, -0.5f), 0.5f) : 0.0f;
        corner_rec rec;
        rec.key = b * A + a;
        rec.r = r;
        rec.pa = (float)a + oa;
        rec.pb = (float)b + ob;
        recs[(size_t)tile * TILE_CAP + tid] = rec;
    }
    if (tid == 0) tile_count[tile] = total;
}

// Strides of image blockIdx.y of a batch (all zero for a single image): every image has its own records,
// counts, maximum slots and output lists.
struct corner_batch {
    size_t img = 0, recs = 0;       // pixels per image; records per image
    int tiles = 0;                  // tiles per image (tile_count, totals; cells: tiles * TB)
    int max_slots = 0;              // maximum slots per image
    int xy = 0, metric = 0;         // floats per image in xy_out / metric_out
    int count = 0;                  // ints per image in count
};

// max_bits[0 / 1]: the maximum of this call / cleared here for the next call (the two alternate).
__global__ __launch_bounds__(256) void corner_tile_kernel(const uint8_t *__restrict__ img, int A, int B, int tiles_a,
                                                          int tiles_b, taps5 G, int method, int lo,
                                                          corner_rec *__restrict__ recs,
                                                          int *__restrict__ tile_count,
                                                          unsigned *__restrict__ max_now,
                                                          unsigned *__restrict__ max_next, corner_batch bt)
{
    img += blockIdx.y * bt.img;
    recs += blockIdx.y * bt.recs;
    tile_count += blockIdx.y * (size_t)bt.tiles;
    max_now += blockIdx.y * (size_t)bt.max_slots;
    max_next += blockIdx.y * (size_t)bt.max_slots;
    // the staged image is dead when the responses are written: the two share their memory (24 KB of LDS per
    // workgroup in all, six workgroups per compute unit)
    constexpr int N_IR = IA * IB > RA * RB ? IA * IB : RA * RB;
    __shared__ __align__(8) float sIR[N_IR];
    __shared__ __align__(8) float sH[3][RA * PB];
    float *const sI = sIR, *const sR = sIR;
    __shared__ unsigned sMax;
    __shared__ int sCnt[16];
    const int tid = threadIdx.x;
    // Workgroups are dealt to the 8 XCDs in turn and each XCD has its own L2: workgroup w takes tile
    // (w % 8) * per_xcd + w / 8, so that an XCD works on a contiguous band of the image and the halos its
    // tiles share are fetched into its L2 once.
    const int ntiles = tiles_a * tiles_b, per_xcd = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || tile >= ntiles) return;
    const int a0 = (tile % tiles_a) * TA, b0 = (tile / tiles_a) * TB;
    if (tid == 0) {
        sMax = 0u;
        if (tile == 0) *max_next = 0u;
    }
    const float s255 = 1.0f / 255.0f;
    // every load of the thread is issued before the first one is waited for (one round trip to memory for the
    // tile instead of one per round)
    constexpr int ROUNDS = (IA * IB + 255) / 256;
    uint8_t px[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int q = min(tid + 256 * k, IA * IB - 1);
        const int rb = q / IA, ra = q - rb * IA;
        const int a = min(max(a0 - HALF - 2 + ra, 0), A - 1);
        const int b = min(max(b0 - HALF - 2 + rb, 0), B - 1);
        px[k] = img[(size_t)b * A + a];
    }
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int q = tid + 256 * k;
        if (q < IA * IB) sI[q] = (float)px[k] * s255;
    }
    __syncthreads();
    // Row pass of the 5-tap filter over the three gradient products, which are made on the way from the staged
    // image (central differences; no pass and no memory of their own).  Both filter passes make two outputs
    // per thread with packed fp32 instructions (two IEEE products or sums per instruction, every output's own
    // operations in the same order as one at a time): the row pass for a vertical pair of rows, the column pass
    // for a horizontal pair of columns.
    static_assert(PB % 2 == 0 && RA % 2 == 0, "pairs of rows / columns");
    for (int q = tid; q < RA * (PB / 2); q += 256) {
        const int k = q / RA, ra = q - k * RA;
        // product (rb, ra + i) is centred on staged pixel (rb + 1, ra + i + 1); rows rb = 2k and 2k + 1
        const float *rowM = sI + (2 * k) * IA + ra, *row0 = rowM + IA, *row1 = row0 + IA, *rowP = row1 + IA;
        float c0[FS + 2], c1[FS + 2];
#pragma unroll
        for (int i = 0; i < FS + 2; ++i) {
            c0[i] = row0[i];
            c1[i] = row1[i];
        }
        v2f s[3];
#pragma unroll
        for (int i = 0; i < FS; ++i) {
            const v2f gx = (v2f){c0[i + 2], c1[i + 2]} - (v2f){c0[i], c1[i]};
            const v2f gy = (v2f){c1[i + 1], rowP[i + 1]} - (v2f){rowM[i + 1], c0[i + 1]};
            const v2f pr[3] = {gx * gx, gy * gy, gx * gy};
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) s[ch] = i == 0 ? G.w[0] * pr[ch] : s[ch] + G.w[i] * pr[ch];
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            sH[ch][(2 * k) * RA + ra] = s[ch].x;
            sH[ch][(2 * k + 1) * RA + ra] = s[ch].y;
        }
    }
    __syncthreads();
    float best = -INFINITY;
    for (int q = tid; q < (RA / 2) * RB; q += 256) {
        const int rb = q / (RA / 2), ra = 2 * (q - rb * (RA / 2));
        v2f v[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const v2f *p = reinterpret_cast<const v2f *>(sH[ch] + rb * RA + ra);
            v2f s = G.w[0] * p[0];
#pragma unroll
            for (int i = 1; i < FS; ++i) s = s + G.w[i] * p[i * (RA / 2)];
            v[ch] = s;
        }
        v2f r;
        if (method == 0) {
            const v2f det = v[0] * v[1] - v[2] * v[2];
            const v2f tr = v[0] + v[1];
            r = det - 0.04f * (tr * tr);
        } else {
            const v2f d = v[0] - v[1];
            const v2f e = d * d + 4.0f * (v[2] * v[2]);
            r = ((v[0] + v[1]) - (v2f){sqrtf(e.x), sqrtf(e.y)}) * 0.5f;
        }
        *reinterpret_cast<v2f *>(sR + rb * RA + ra) = r;
        // the image maximum is taken over the pixels of the tile itself (every pixel belongs to one tile)
        const int b = b0 - 1 + rb;
        if (rb >= 1 && rb <= TB && b < B) {
            const int a = a0 - 1 + ra;
            if (ra >= 1 && ra <= TA && a < A) best = fmaxf(best, r.x);
            if (ra + 1 <= TA && a + 1 < A) best = fmaxf(best, r.y);
        }
    }
    tile_maxima<true>(sR, best, &sMax, sCnt, A, B, a0, b0, lo, tile, recs, tile_count, max_now);
}

// FAST: the segment-test score of every pixel of the tile and its 1 px ring from the uint8 tile with a
// 4 px halo (3 for the circle + 1 for the ring).
__global__ __launch_bounds__(256) void fast_tile_kernel(const uint8_t *__restrict__ img, int A, int B, int tiles_a,
                                                        int tiles_b, int threshold, int lo,
                                                        corner_rec *__restrict__ recs, int *__restrict__ tile_count,
                                                        unsigned *__restrict__ max_now,
                                                        unsigned *__restrict__ max_next, corner_batch bt)
{
    img += blockIdx.y * bt.img;
    recs += blockIdx.y * bt.recs;
    tile_count += blockIdx.y * (size_t)bt.tiles;
    max_now += blockIdx.y * (size_t)bt.max_slots;
    max_next += blockIdx.y * (size_t)bt.max_slots;
    constexpr int FA = TA + 8, FB = TB + 8;
    __shared__ int sI[FA * FB];
    __shared__ float sR[RA * RB];
    __shared__ unsigned sMax;
    __shared__ int sCnt[16];
    const int tid = threadIdx.x;
    const int ntiles = tiles_a * tiles_b, per_xcd = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // (an XCD takes a contiguous band, as above)
    if ((int)(blockIdx.x >> 3) >= per_xcd || tile >= ntiles) return;
    const int a0 = (tile % tiles_a) * TA, b0 = (tile / tiles_a) * TB;
    if (tid == 0) {
        sMax = 0u;
        if (tile == 0) *max_next = 0u;
    }
    // (every load of the thread issued before the first one is waited for)
    constexpr int ROUNDS = (FA * FB + 255) / 256;
    uint8_t px[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int q = min(tid + 256 * k, FA * FB - 1);
        const int rb = q / FA, ra = q - rb * FA;
        const int a = min(max(a0 - 4 + ra, 0), A - 1);
        const int b = min(max(b0 - 4 + rb, 0), B - 1);
        px[k] = img[(size_t)b * A + a];
    }
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int q = tid + 256 * k;
        if (q < FA * FB) sI[q] = (int)px[k];
    }
    __syncthreads();
    float best = -INFINITY;
    for (int q = tid; q < RA * RB; q += 256) {
        const int rb = q / RA, ra = q - rb * RA;
        const int a = a0 - 1 + ra, b = b0 - 1 + rb;
        int score = 0;
        if (a >= 3 && a < A - 3 && b >= 3 && b < B - 3) {
            const int sc = vohip_dev::fast_score_9_16(sI + (rb + 3) * FA + (ra + 3), FA);
            score = sc >= threshold ? sc : 0;
        }
        const float r = (float)score;
        sR[q] = r;
        if (ra >= 1 && ra <= TA && rb >= 1 && rb <= TB && a < A && b < B) best = fmaxf(best, r);
    }
    tile_maxima<false>(sR, best, &sMax, sCnt, A, B, a0, b0, lo, tile, recs, tile_count, max_now);
}

// Survivors of the quality threshold per cell (tile, image row): one wavefront per tile.  cells: 16 ints
// per tile, totals: one int per tile.
__global__ __launch_bounds__(256) void corner_count_kernel(const corner_rec *__restrict__ recs,
                                                           const int *__restrict__ tile_count, int ntiles, int tiles_a,
                                                           int A, float min_quality,
                                                           const unsigned *__restrict__ max_bits,
                                                           int *__restrict__ cells, int *__restrict__ totals,
                                                           corner_batch bt)
{
    recs += blockIdx.y * bt.recs;
    tile_count += blockIdx.y * (size_t)bt.tiles;
    max_bits += blockIdx.y * (size_t)bt.max_slots;
    cells += blockIdx.y * (size_t)bt.tiles * TB;
    totals += blockIdx.y * (size_t)bt.tiles;
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const float thr = min_quality * from_ordered_bits(*max_bits);
    const int b0 = (t / tiles_a) * TB;
    const int n = tile_count[t];
    int mine = 0, total = 0;   // lane k < 16 collects the count of image row k
    for (int c = 0; c < n; c += 64) {
        bool pass = false;
        int rb = -1;
        if (c + lane < n) {
            const corner_rec r = recs[(size_t)t * TILE_CAP + c + lane];
            pass = r.r > thr;
            rb = r.key / A - b0;
        }
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const int cnt = __popcll(__ballot(pass && rb == k));
            if (lane == k) mine += cnt;
            total += cnt;
        }
    }
    if (lane < TB) cells[(size_t)t * TB + lane] = mine;
    if (lane == 0) totals[t] = total;
}

// One wavefront per tile writes the tile's survivors at their ranks in ascending pixel order: [survivors
// in earlier rows of tiles] + [survivors in the cells (image row, tile) before the record's cell in this
// row of tiles] + [survivors before it in its own cell] -- within a tile the records are in pixel order,
// so those of one image row are consecutive.  Ordered and deterministic; no atomics.
__global__ __launch_bounds__(256) void corner_emit_kernel(const corner_rec *__restrict__ recs,
                                                          const int *__restrict__ tile_count,
                                                          const int *__restrict__ cells,
                                                          const int *__restrict__ totals, int ntiles, int tiles_a,
                                                          int A, float min_quality,
                                                          const unsigned *__restrict__ max_bits, int cap, int swap_xy,
                                                          float *__restrict__ xy_out, float *__restrict__ metric_out,
                                                          int *__restrict__ count, corner_batch bt)
{
    recs += blockIdx.y * bt.recs;
    tile_count += blockIdx.y * (size_t)bt.tiles;
    max_bits += blockIdx.y * (size_t)bt.max_slots;
    cells += blockIdx.y * (size_t)bt.tiles * TB;
    totals += blockIdx.y * (size_t)bt.tiles;
    xy_out += blockIdx.y * (size_t)bt.xy;
    metric_out += blockIdx.y * (size_t)bt.metric;
    count += blockIdx.y * (size_t)bt.count;
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const float thr = min_quality * from_ordered_bits(*max_bits);
    const int ty = t / tiles_a, tx = t - ty * tiles_a;
    const int b0 = ty * TB;
    // survivors in the rows of tiles before this one
    int base = 0;
    for (int q = lane; q < ty * tiles_a; q += 64) base += totals[q];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) base += __shfl_xor(base, m, 64);
    // this row of tiles: lanes 4 rb .. 4 rb + 3 share image row rb; `full` = survivors of the whole image
    // row, `part` = those in the tiles before this one
    const int rb_l = lane >> 2, sub = lane & 3;
    int full = 0, part = 0;
    for (int x = sub; x < tiles_a; x += 4) {
        const int v = cells[((size_t)ty * tiles_a + x) * TB + rb_l];
        full += v;
        part += x < tx ? v : 0;
    }
    full += __shfl_xor(full, 1, 64);
    full += __shfl_xor(full, 2, 64);
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    int incl = full;   // inclusive prefix over the image rows (lanes with the same `sub` hold the same values)
#pragma unroll
    for (int d = 4; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d, 64);
        if (lane >= d) incl += v;
    }
    const int offset = base + (incl - full) + part;   // rank of the first survivor of cell (rb_l, tx)
    if (t == ntiles - 1 && lane == 63) *count = base + incl;
    const int n = tile_count[t];
    int run = 0;   // lane k < 16: survivors of image row k in the chunks done so far
    for (int c = 0; c < n; c += 64) {
        corner_rec r;
        bool pass = false;
        int rb = -1;
        if (c + lane < n) {
            r = recs[(size_t)t * TILE_CAP + c + lane];
            pass = r.r > thr;
            rb = r.key / A - b0;
        }
        unsigned long long same = 0ull;
        int run_next = run;
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const unsigned long long b = __ballot(pass && rb == k);
            if (rb == k) same = b;
            if (lane == k) run_next += __popcll(b);
        }
        const int rbc = rb < 0 ? 0 : rb;
        const int pos = __shfl(offset, 4 * rbc, 64) + __shfl(run, rbc, 64) + __popcll(same & ((1ull << lane) - 1ull));
        run = run_next;
        if (pass && pos < cap) {
            const float pa = r.pa + 1.0f, pb = r.pb + 1.0f;
            xy_out[pos] = swap_xy ? pb : pa;
            xy_out[cap + pos] = swap_xy ? pa : pb;
            metric_out[pos] = r.r;
        }
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

size_t vohip_corner_scratch_bytes(int rows, int cols)
{
    // (the same for either storage order: the tile grid is transposed with the image)
    const int ta = (rows + TA - 1) / TA, tb = (cols + TB - 1) / TB, ta2 = (cols + TA - 1) / TA, tb2 = (rows + TB - 1) / TB;
    const size_t tiles = (size_t)ta * tb > (size_t)ta2 * tb2 ? (size_t)ta * tb : (size_t)ta2 * tb2;
    return 1024 + tiles * ((2 + TB) * sizeof(int) + TILE_CAP * sizeof(corner_rec));
}

// Corner detection on `stream` with caller-owned scratch memory: `scratch` holds
// vohip_corner_scratch_bytes(rows, cols) bytes whose first 256 were zero before its first use; *calls
// counts the uses of this scratch (the two maximum slots alternate: a call clears the other one).
int vohip_detect_corners_enqueue(vohip_ctx *ctx, hipStream_t st, void *scratch, unsigned *calls, const uint8_t *d_img,
                                 int rows, int cols, int colmajor, double min_quality, int filter_size, int border,
                                 int method, double min_contrast, float *d_xy, float *d_metric, int cap, int *d_count,
                                 int brisk_octaves)
{
    VOHIP_REQUIRE(ctx, d_img && d_xy && d_metric && d_count && scratch && calls, "NULL argument");
    if (method == 3)   // SURF: `min_contrast` carries 'MetricThreshold', the scratch is vohip_surf_scratch_bytes(rows, cols) long
        return vohip_detect_surf_enqueue(ctx, st, scratch, d_img, rows, cols, colmajor, min_contrast, border, d_xy, d_metric,
                                         nullptr, cap, d_count);
    if (method == 4)   // BRISK: the scratch is vohip_brisk_scratch_bytes(rows, cols) long
        return vohip_detect_brisk_enqueue(ctx, st, scratch, d_img, rows, cols, colmajor, min_contrast, min_quality, brisk_octaves,
                                          border, d_xy, d_metric, nullptr, cap, d_count);
    VOHIP_REQUIRE(ctx, rows >= 8 && cols >= 8, "image too small");
    VOHIP_REQUIRE(ctx, method >= 0 && method <= 2, "method: 0 = Harris, 1 = MinEigen, 2 = FAST, 3 = SURF, 4 = BRISK");
    VOHIP_REQUIRE(ctx, method == 2 || filter_size == FS, "only FilterSize 5 (the toolbox default) is built");
    VOHIP_REQUIRE(ctx, method != 2 || (min_contrast > 0.0 && min_contrast < 1.0), "MinContrast must lie in (0, 1)");
    VOHIP_REQUIRE(ctx, cap >= 1 && cap <= 16384, "cap must be 1..16384");
    VOHIP_REQUIRE(ctx, border >= 0, "negative border");
    const int A = colmajor ? rows : cols, B = colmajor ? cols : rows;
    const int tiles_a = (A + TA - 1) / TA, tiles_b = (B + TB - 1) / TB;
    unsigned *max2 = (unsigned *)scratch;
    const int ntiles = tiles_a * tiles_b;
    auto round256 = [](size_t v) { return (v + 255) / 256 * 256; };
    char *p = (char *)scratch + 256;
    int *tile_count = (int *)p;
    p += round256(sizeof(int) * (size_t)ntiles);
    int *totals = (int *)p;
    p += round256(sizeof(int) * (size_t)ntiles);
    int *cells = (int *)p;
    p += round256(sizeof(int) * (size_t)ntiles * TB);
    corner_rec *recs = (corner_rec *)p;
    const unsigned slot = (*calls)++ & 1u;
    taps5 G;
    gaussian_taps(filter_size, G.w);
    const int lo = border < 1 ? 1 : border;
    if (method == 2)   // the contrast threshold on the uint8 scale, rounded as MATLAB's uint8() rounds
        hipLaunchKernelGGL(fast_tile_kernel, dim3(8 * ((ntiles + 7) / 8)), dim3(256), 0, st, d_img, A, B, tiles_a, tiles_b,
                           (int)std::floor(min_contrast * 255.0 + 0.5), lo, recs, tile_count, max2 + slot,
                           max2 + (slot ^ 1u), corner_batch());
    else
        hipLaunchKernelGGL(corner_tile_kernel, dim3(8 * ((ntiles + 7) / 8)), dim3(256), 0, st, d_img, A, B, tiles_a,
                           tiles_b, G, method, lo, recs, tile_count, max2 + slot, max2 + (slot ^ 1u), corner_batch());
    hipLaunchKernelGGL(corner_count_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, (const corner_rec *)recs,
                       (const int *)tile_count, ntiles, tiles_a, A, (float)min_quality, (const unsigned *)(max2 + slot),
                       cells, totals, corner_batch());
    hipLaunchKernelGGL(corner_emit_kernel, dim3((ntiles + 3) / 4), dim3(256), 0, st, (const corner_rec *)recs,
                       (const int *)tile_count, (const int *)cells, (const int *)totals, ntiles, tiles_a, A,
                       (float)min_quality, (const unsigned *)(max2 + slot), cap, colmajor ? 1 : 0, d_xy, d_metric, d_count,
                       corner_batch());
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

// The same three kernels over n images at once (blockIdx.y = image): every image keeps its own records,
// counts, maximum and output lists, so the lists are those of n single calls, bit for bit.
int vohip_detect_corners_batch_enqueue(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                       double min_quality, int filter_size, int border, int method, float *d_xy,
                                       float *d_metric, int cap, int *d_counts, double min_contrast, int brisk_octaves)
{
    VOHIP_REQUIRE(ctx, d_imgs && d_xy && d_metric && d_counts, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && n <= 65535, "1..65535 images per call");
    if (method == 3)   // SURF: `min_contrast` carries 'MetricThreshold'
        return vohip_detect_surf_batch_enqueue(ctx, d_imgs, n, rows, cols, colmajor, min_contrast, border, d_xy, d_metric, cap,
                                               d_counts);
    if (method == 4)
        return vohip_detect_brisk_batch_enqueue(ctx, d_imgs, n, rows, cols, colmajor, min_contrast, min_quality, brisk_octaves,
                                                border, d_xy, d_metric, cap, d_counts);
    VOHIP_REQUIRE(ctx, rows >= 8 && cols >= 8, "image too small");
    VOHIP_REQUIRE(ctx, method >= 0 && method <= 2, "method: 0 = Harris, 1 = MinEigen, 2 = FAST, 3 = SURF, 4 = BRISK");
    VOHIP_REQUIRE(ctx, method == 2 || filter_size == FS, "only FilterSize 5 (the toolbox default) is built");
    VOHIP_REQUIRE(ctx, method != 2 || (min_contrast > 0.0 && min_contrast < 1.0), "MinContrast must lie in (0, 1)");
    VOHIP_REQUIRE(ctx, cap >= 1 && cap <= 16384, "cap must be 1..16384");
    VOHIP_REQUIRE(ctx, border >= 0, "negative border");
    const int A = colmajor ? rows : cols, B = colmajor ? cols : rows;
    const int tiles_a = (A + TA - 1) / TA, tiles_b = (B + TB - 1) / TB, ntiles = tiles_a * tiles_b;
    auto round256 = [](size_t v) { return (v + 255) / 256 * 256; };
    // scratch of the whole batch: [maxima | tile counts | totals | cells | records | spare maxima], image after image in each
    const size_t b_max = round256(sizeof(unsigned) * (size_t)n), b_cnt = round256(sizeof(int) * (size_t)n * ntiles),
                 b_cells = round256(sizeof(int) * (size_t)n * ntiles * TB),
                 b_recs = sizeof(corner_rec) * (size_t)n * ntiles * TILE_CAP;
    vohip_buf &buf = ctx->bufs[B_HARRIS_BATCH];
    VOHIP_TRY(vohip_reserve(ctx, buf, 2 * b_max + 2 * b_cnt + b_cells + b_recs));
    char *p = (char *)buf.ptr;
    unsigned *maxima = (unsigned *)p;
    p += b_max;
    int *tile_count = (int *)p;
    p += b_cnt;
    int *totals = (int *)p;
    p += b_cnt;
    int *cells = (int *)p;
    p += b_cells;
    corner_rec *recs = (corner_rec *)p;
    unsigned *spare = (unsigned *)(p + b_recs);   // (n words: the slot an image's first tile clears "for the next call")
    hipStream_t st = ctx->stream;
    VOHIP_HIP(ctx, hipMemsetAsync(maxima, 0, b_max, st));
    corner_batch bt;
    bt.img = (size_t)A * B;
    bt.recs = (size_t)ntiles * TILE_CAP;
    bt.tiles = ntiles;
    bt.max_slots = 1;
    bt.xy = 2 * cap;
    bt.metric = cap;
    bt.count = 1;
    taps5 G;
    gaussian_taps(filter_size, G.w);
    const int lo = border < 1 ? 1 : border;
    if (method == 2)
        hipLaunchKernelGGL(fast_tile_kernel, dim3(8 * ((ntiles + 7) / 8), n), dim3(256), 0, st, d_imgs, A, B, tiles_a, tiles_b,
                           (int)std::floor(min_contrast * 255.0 + 0.5), lo, recs, tile_count, maxima, spare, bt);
    else
        hipLaunchKernelGGL(corner_tile_kernel, dim3(8 * ((ntiles + 7) / 8), n), dim3(256), 0, st, d_imgs, A, B, tiles_a, tiles_b,
                           G, method, lo, recs, tile_count, maxima, spare, bt);
    hipLaunchKernelGGL(corner_count_kernel, dim3((ntiles + 3) / 4, n), dim3(256), 0, st, (const corner_rec *)recs,
                       (const int *)tile_count, ntiles, tiles_a, A, (float)min_quality, (const unsigned *)maxima, cells, totals,
                       bt);
    hipLaunchKernelGGL(corner_emit_kernel, dim3((ntiles + 3) / 4, n), dim3(256), 0, st, (const corner_rec *)recs,
                       (const int *)tile_count, (const int *)cells, (const int *)totals, ntiles, tiles_a, A,
                       (float)min_quality, (const unsigned *)maxima, cap, colmajor ? 1 : 0, d_xy, d_metric, d_counts, bt);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

namespace {

int detect_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double min_quality,
               int filter_size, int border, int method, double min_contrast, float *d_xy, float *d_metric, int cap,
               int *d_count)
{
    VOHIP_REQUIRE(ctx, rows >= 8 && cols >= 8, "image too small");
    const size_t bytes = vohip_corner_scratch_bytes(rows, cols) + 256;
    vohip_buf &b = ctx->bufs[B_HARRIS_SCRATCH];
    if (!b.ptr || b.cap < bytes) {   // (a new allocation: its maximum slots start at zero)
        VOHIP_TRY(vohip_reserve(ctx, b, bytes));
        VOHIP_HIP(ctx, hipMemsetAsync(b.ptr, 0, 256, ctx->stream));
    }
    return vohip_detect_corners_enqueue(ctx, ctx->stream, b.ptr, &ctx->harris_calls, d_img, rows, cols, colmajor,
                                        min_quality, filter_size, border, method, min_contrast, d_xy, d_metric, cap,
                                        d_count);
}

int detect_host(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double min_quality,
                int filter_size, int border, int method, double min_contrast, float *xy_out, float *metric_out, int cap,
                int *n_out)
{
    VOHIP_REQUIRE(ctx, img && xy_out && metric_out && n_out, "NULL argument");
    VOHIP_REQUIRE(ctx, rows >= 1 && cols >= 1 && cap >= 1, "bad size");
    void *d_img, *d_xy, *d_metric, *d_count;
    VOHIP_TRY(vohip_upload(ctx, B_IMG_STAGE, img, (size_t)rows * cols, &d_img));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, nullptr, sizeof(float) * 2 * (size_t)cap, &d_xy));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, nullptr, sizeof(float) * (size_t)cap, &d_metric));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, 256, &d_count));
    VOHIP_TRY(detect_dev(ctx, (const uint8_t *)d_img, rows, cols, colmajor, min_quality, filter_size, border, method,
                         min_contrast, (float *)d_xy, (float *)d_metric, cap, (int *)d_count));
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

}  // namespace

extern "C" {

// d_xy: cap x 2 floats (x column then y column), d_metric: cap floats, d_count: 1 int (total
// found, may exceed cap).  Asynchronous.
int vohip_detect_corners_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor,
                             double min_quality, int filter_size, int border, int method, float *d_xy,
                             float *d_metric, int cap, int *d_count)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, method == 0 || method == 1, "method: 0 = Harris, 1 = MinEigen");
    return detect_dev(ctx, d_img, rows, cols, colmajor, min_quality, filter_size, border, method, 0.0, d_xy, d_metric,
                      cap, d_count);
}

int vohip_detect_corners_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                   double min_quality, int filter_size, int border, int method, float *d_xy,
                                   float *d_metric, int cap, int *d_counts)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, method == 0 || method == 1, "method: 0 = Harris, 1 = MinEigen");
    return vohip_detect_corners_batch_enqueue(ctx, d_imgs, n, rows, cols, colmajor, min_quality, filter_size, border, method,
                                              d_xy, d_metric, cap, d_counts);
}

int vohip_detect_corners(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double min_quality,
                         int filter_size, int border, int method, float *xy_out, float *metric_out, int cap,
                         int *n_out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, method == 0 || method == 1, "method: 0 = Harris, 1 = MinEigen");
    return detect_host(ctx, img, rows, cols, colmajor, min_quality, filter_size, border, method, 0.0, xy_out,
                       metric_out, cap, n_out);
}

int vohip_detect_fast_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double min_contrast,
                          double min_quality, int border, float *d_xy, float *d_metric, int cap, int *d_count)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return detect_dev(ctx, d_img, rows, cols, colmajor, min_quality, FS, border, 2, min_contrast, d_xy, d_metric, cap,
                      d_count);
}

int vohip_detect_fast_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                double min_contrast, double min_quality, int border, float *d_xy, float *d_metric, int cap,
                                int *d_counts)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return vohip_detect_corners_batch_enqueue(ctx, d_imgs, n, rows, cols, colmajor, min_quality, FS, border, 2, d_xy, d_metric,
                                              cap, d_counts, min_contrast);
}

int vohip_detect_fast(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double min_contrast,
                      double min_quality, int border, float *xy_out, float *metric_out, int cap, int *n_out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    return detect_host(ctx, img, rows, cols, colmajor, min_quality, FS, border, 2, min_contrast, xy_out, metric_out, cap,
                       n_out);
}

}  // extern "C"
