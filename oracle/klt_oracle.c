/*
 * klt_oracle.c -- CPU statement of the pyramidal Lucas-Kanade tracker that stands in for the
 * reference's vision.PointTracker (constructed at main_demo.m:342-343,348-349, stepped at
 * main_demo.m:380 and update_landmarks.m:10).
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
 *
 * PARITY UNPINNED: vision.PointTracker is closed source (Computer Vision Toolbox) and the
 * reference holds no tracked-point fixtures, so the algorithm below is this project's own
 * definition of "pyramidal KLT with BlockSize, NumPyramidLevels, MaxIterations and
 * MaxBidirectionalError"; it is pinned by analytic tests (known translations / affine-free
 * warps of band-limited textures) and the HIP kernel is held to it within 0.05 px.
 *
 * Definition
 *  - Images are addressed as img[b * A + a], a = fast axis (length A), b = slow axis (length B).
 *    A MATLAB column-major HxW image is exactly this with a = row, b = column; the tracker is
 *    symmetric in the two axes, so no transpose is ever made.  Coordinates here are 0-based.
 *  - Pyramid: level 0 = float(image); level l+1 = 5-tap binomial [1 4 6 4 1]/16 along a, then
 *    along b (replicated border), keeping even samples: size (A+1)/2 x (B+1)/2.
 *    Every tap sum is evaluated as ((t0 + t4) + 4 (t1 + t3) + 6 t2) / 16 in float.
 *  - Bilinear sampling with replicated border, in float:
 *      top = fma(fa, I01 - I00, I00), bot = fma(fa, I11 - I10, I10), v = fma(fb, bot - top, top).
 *  - Per level, coarse to fine, window W x W centred on the point: template T = prev sampled
 *    at p + w, gradients by central differences of the sampled prev image
 *    (gx = (prev(p + w + ea) - prev(p + w - ea)) / 2), G = sum [gx gx, gx gy; gx gy, gy gy],
 *    then at most max_iter Gauss-Newton steps d += G^-1 sum (T - cur(p + d + w)) [gx; gy]
 *    (G^-1 = adj(G) * (1 / det G)),
 *    stopping after a step shorter than 0.01 px, or when a step cancels the previous one to
 *    within 0.01 px (oscillation: half of the last step is taken back).  G is summed in double; the right-hand side is
 *    summed in float within each of 64 pixel groups (the window's rows are split into 64 / W bands;
 *    a group is one window column within one band, summed down the column with fmaf; W <= 64) and
 *    across the groups by a fixed float tree: fours ((g0+g1)+(g2+g3)), pairs of fours,
 *    pairs of eights, pairs of sixteens ((r0+r1)+(r2+r3)).
 *    The displacement doubles between levels.
 *  - A point is valid iff at level 0 G is well conditioned in both directions
 *    (lambda_min(G) / W^2 >= 1e-3), the forward result lies inside the image, and the
 *    forward-backward distance is <= max_bidir_error.  A point that is already invalid after the
 *    forward pass is not tracked back; its forward-backward distance is reported as infinite.
 */
#include "vo_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

#define KLT_EPS_STEP 0.01
#define KLT_MIN_EIG 1e-3

/* Sum of 64 floats in the fixed order of the tracker's wavefront reduction. */
static float tree64(const float *g)
{
    float r[4];
    for (int n = 0; n < 4; ++n) {
        float h[2];
        for (int m = 0; m < 2; ++m) {
            const float *p = g + 16 * n + 8 * m;
            const float q0 = (p[0] + p[1]) + (p[2] + p[3]);
            const float q1 = (p[4] + p[5]) + (p[6] + p[7]);
            h[m] = q0 + q1;
        }
        r[n] = h[0] + h[1];
    }
    return (r[0] + r[1]) + (r[2] + r[3]);
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static float tap5(float t0, float t1, float t2, float t3, float t4)
{
    return ((t0 + t4) + 4.0f * (t1 + t3) + 6.0f * t2) * 0.0625f;
}

/* one pyramid reduction: src A x B -> dst (A+1)/2 x (B+1)/2 */
void vo_pyr_down(const float *src, int A, int B, float *dst)
{
    int A2 = (A + 1) / 2, B2 = (B + 1) / 2;
    for (int b = 0; b < B2; ++b)
        for (int a = 0; a < A2; ++a) {
            float col[5];
            for (int j = 0; j < 5; ++j) {
                int bb = clampi(2 * b + j - 2, 0, B - 1);
                const float *row = src + (size_t)bb * A;
                col[j] = tap5(row[clampi(2 * a - 2, 0, A - 1)], row[clampi(2 * a - 1, 0, A - 1)], row[2 * a],
                              row[clampi(2 * a + 1, 0, A - 1)], row[clampi(2 * a + 2, 0, A - 1)]);
            }
            dst[(size_t)b * A2 + a] = tap5(col[0], col[1], col[2], col[3], col[4]);
        }
}

/* Builds `levels` levels; sizes written to As/Bs; returns malloc'ed array of level pointers. */
float **vo_pyramid_build(const uint8_t *img, int A, int B, int levels, int *As, int *Bs)
{
    float **pyr = (float **)malloc(sizeof(float *) * (size_t)levels);
    pyr[0] = (float *)malloc(sizeof(float) * (size_t)A * B);
    for (size_t i = 0; i < (size_t)A * B; ++i) pyr[0][i] = (float)img[i];
    As[0] = A;
    Bs[0] = B;
    for (int l = 1; l < levels; ++l) {
        As[l] = (As[l - 1] + 1) / 2;
        Bs[l] = (Bs[l - 1] + 1) / 2;
        pyr[l] = (float *)malloc(sizeof(float) * (size_t)As[l] * Bs[l]);
        vo_pyr_down(pyr[l - 1], As[l - 1], Bs[l - 1], pyr[l]);
    }
    return pyr;
}

void vo_pyramid_free(float **pyr, int levels)
{
    for (int l = 0; l < levels; ++l) free(pyr[l]);
    free(pyr);
}

/* level l of the pyramid of an image, for tests: out must hold As[l]*Bs[l] floats */
void vo_pyramid_level(const uint8_t *img, int A, int B, int level, float *out, int *Aout, int *Bout)
{
    int As[16], Bs[16];
    float **pyr = vo_pyramid_build(img, A, B, level + 1, As, Bs);
    memcpy(out, pyr[level], sizeof(float) * (size_t)As[level] * Bs[level]);
    *Aout = As[level];
    *Bout = Bs[level];
    vo_pyramid_free(pyr, level + 1);
}

static float bilinear(const float *img, int A, int B, int ia, int ib, float fa, float fb)
{
    int a0 = clampi(ia, 0, A - 1), a1 = clampi(ia + 1, 0, A - 1);
    int b0 = clampi(ib, 0, B - 1), b1 = clampi(ib + 1, 0, B - 1);
    float i00 = img[(size_t)b0 * A + a0], i01 = img[(size_t)b0 * A + a1];
    float i10 = img[(size_t)b1 * A + a0], i11 = img[(size_t)b1 * A + a1];
    float top = fmaf(fa, i01 - i00, i00);
    float bot = fmaf(fa, i11 - i10, i10);
    return fmaf(fb, bot - top, top);
}

/* One direction: tracks (pa, pb) [0-based, level-0 coordinates] from pyramid P to pyramid Q.
 * Returns 1 when the level-0 normal matrix is well conditioned. */
static int track_one(float **P, float **Q, const int *As, const int *Bs, int levels, int W, int max_iter,
                     double pa, double pb, double *qa, double *qb)
{
    const int half = W / 2;
    double da = 0.0, db = 0.0;
    int ok0 = 1;
    float *T = (float *)malloc(sizeof(float) * 3 * (size_t)W * W);
    float *GX = T + (size_t)W * W, *GY = GX + (size_t)W * W;
    for (int l = levels - 1; l >= 0; --l) {
        const int A = As[l], B = Bs[l];
        const double scale = 1.0 / (double)(1 << l);
        const double ca = pa * scale, cb = pb * scale;
        const int ia = (int)floor(ca), ib = (int)floor(cb);
        const float fa = (float)(ca - ia), fb = (float)(cb - ib);
        double gxx = 0, gxy = 0, gyy = 0;
        for (int wb = -half; wb <= half; ++wb)
            for (int wa = -half; wa <= half; ++wa) {
                int k = (wb + half) * W + (wa + half);
                float c = bilinear(P[l], A, B, ia + wa, ib + wb, fa, fb);
                float ap = bilinear(P[l], A, B, ia + wa + 1, ib + wb, fa, fb);
                float am = bilinear(P[l], A, B, ia + wa - 1, ib + wb, fa, fb);
                float bp = bilinear(P[l], A, B, ia + wa, ib + wb + 1, fa, fb);
                float bm = bilinear(P[l], A, B, ia + wa, ib + wb - 1, fa, fb);
                float gx = (ap - am) * 0.5f, gy = (bp - bm) * 0.5f;
                T[k] = c;
                GX[k] = gx;
                GY[k] = gy;
                gxx += (double)gx * (double)gx;
                gxy += (double)gx * (double)gy;
                gyy += (double)gy * (double)gy;
            }
        const double det = gxx * gyy - gxy * gxy;
        const double tr = gxx + gyy;
        const double lmin = 0.5 * (tr - sqrt((gxx - gyy) * (gxx - gyy) + 4.0 * gxy * gxy));
        const int good = (lmin / (double)(W * W) >= KLT_MIN_EIG) && det > 0.0;
        if (l == 0) ok0 = good;
        const double inv_det = 1.0 / det;
        const double iaa = gyy * inv_det, iab = gxy * inv_det, ibb = gxx * inv_det;
        if (good) {
            double psa = 0.0, psb = 0.0;
            for (int it = 0; it < max_iter; ++it) {
                const double ta = ca + da, tb = cb + db;
                const int ja = (int)floor(ta), jb = (int)floor(tb);
                const float ga = (float)(ta - ja), gb = (float)(tb - jb);
                /* right-hand side: the window's rows are split into 64 / W bands of rpl rows; band g of
                 * window column c is group g W + c (the remaining groups are empty); each group is
                 * summed in float with fmaf down its column, the 64 group sums are added in float by
                 * the fixed tree of tree64() */
                const int bands = 64 / W, rpl = (W + bands - 1) / bands;
                float pa_[64], pb_[64];
                for (int grp = 0; grp < 64; ++grp) {
                    /* sum (T - J) G formed as sum T G - sum J G within the group (both sums in float with
                     * fmaf down the column): the first one does not depend on the iteration */
                    float ctx = 0.f, cty = 0.f, sjx = 0.f, sjy = 0.f;
                    const int band = grp / W, col = grp % W;
                    for (int row = band * rpl; grp < bands * W && row < (band + 1) * rpl && row < W; ++row) {
                        int k = row * W + col;
                        int wb = k / W - half, wa = k % W - half;
                        float j = bilinear(Q[l], A, B, ja + wa, jb + wb, ga, gb);
                        ctx = fmaf(T[k], GX[k], ctx);
                        cty = fmaf(T[k], GY[k], cty);
                        sjx = fmaf(j, GX[k], sjx);
                        sjy = fmaf(j, GY[k], sjy);
                    }
                    pa_[grp] = ctx - sjx;
                    pb_[grp] = cty - sjy;
                }
                const double ba = (double)tree64(pa_), bb = (double)tree64(pb_);
                /* the inverse of the gradient matrix is formed once per level; a - b c as fma(-b, c, a) */
                const double sa = fma(-iab, bb, iaa * ba);
                const double sb = fma(-iab, ba, ibb * bb);
                da += sa;
                db += sb;
                if (sa * sa + sb * sb < KLT_EPS_STEP * KLT_EPS_STEP) break;
                if (it > 0 && (sa + psa) * (sa + psa) + (sb + psb) * (sb + psb) < KLT_EPS_STEP * KLT_EPS_STEP) {
                    da -= 0.5 * sa; /* oscillation around the minimum: settle half-way */
                    db -= 0.5 * sb;
                    break;
                }
                psa = sa;
                psb = sb;
            }
        }
        if (l > 0) {
            da *= 2.0;
            db *= 2.0;
        }
    }
    free(T);
    *qa = pa + da;
    *qb = pb + db;
    return ok0;
}

/* Threads the tracker's loop over the points may use from now on (n < 1: as many as OpenMP finds); returns the
 * number in force.  For the benchmark's CPU baseline, which is quoted single-threaded and on all cores. */
int vo_set_threads(int n)
{
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
    else omp_set_num_threads(omp_get_num_procs());
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* pts / out: n x 2 column-major (first n values = a coordinate, next n = b coordinate), 0-based.
 * fb_err (may be NULL) receives the forward-backward distance. */
void vo_klt_track(const uint8_t *prev, const uint8_t *cur, int A, int B, const double *pts, int n, int W,
                  int levels, int max_iter, double max_bidir_error, double *out, uint8_t *valid, double *fb_err)
{
    int As[16], Bs[16];
    if (levels > 16) levels = 16;
    float **P = vo_pyramid_build(prev, A, B, levels, As, Bs);
    float **Q = vo_pyramid_build(cur, A, B, levels, As, Bs);
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < n; ++i) {
        double qa, qb, ra, rb;
        int ok_f = track_one(P, Q, As, Bs, levels, W, max_iter, pts[i], pts[n + i], &qa, &qb);
        int inside = qa >= 0.0 && qa <= (double)(A - 1) && qb >= 0.0 && qb <= (double)(B - 1);
        /* a track that is invalid after the forward pass is not tracked back (distance: infinite) */
        int ok_b = 0;
        double err = INFINITY;
        if (ok_f && inside) {
            ok_b = track_one(Q, P, As, Bs, levels, W, max_iter, qa, qb, &ra, &rb);
            double ea = ra - pts[i], eb = rb - pts[n + i];
            err = sqrt(ea * ea + eb * eb);
        }
        out[i] = qa;
        out[n + i] = qb;
        valid[i] = (uint8_t)(ok_f && ok_b && inside && err <= max_bidir_error);
        if (fb_err) fb_err[i] = err;
    }
    vo_pyramid_free(P, levels);
    vo_pyramid_free(Q, levels);
}
