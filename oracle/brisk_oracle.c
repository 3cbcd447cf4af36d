/*
 * TEST INFRASTRUCTURE ONLY: the CPU statement the GPU detector is checked against.
 *
 * BRISK keypoints, the stand-in for detectBRISKFeatures as genKeypoints.m:20,31 calls it (one of main_eval.m:19's
 * feature variants, options {'MinQuality', 2e-3}: main_eval.m:170-171).  PARITY UNPINNED: a closed-source toolbox
 * function and no keypoint fixtures in the reference.  This project's definition, after section 3.1 of
 * Leutenegger, Chli, Siegwart, "BRISK: Binary Robust Invariant Scalable Keypoints" (ICCV 2011):
 *
 *   layers k = 0 .. 2 NumOctaves - 1 in the order of their scale sigma_k = 1, 1.5, 2, 3, 4, 6, 8, 12: the octaves
 *        c_0 = image, c_i = half(c_i-1) (layers 0, 2, 4, 6) and the intra-octaves d_0 = twothirds(c_0),
 *        d_i = half(d_i-1) (layers 1, 3, 5, 7);  NumOctaves 0: layer 0 alone;
 *        half:      out(x, y) = (the 2 x 2 block at (2x, 2y) summed + 2) / 4, floor(W/2) x floor(H/2);
 *        twothirds: every 3 x 3 block -> 2 x 2 pixels, each the area-weighted mean of the 1.5 x 1.5 pixels it covers:
 *                   (4 corner + 2 edge + 2 edge + centre + 4) / 9, 2 floor(W/3) x 2 floor(H/3);
 *        pixel (x, y) of layer k lies at ((x + .5) sigma_k - .5, (y + .5) sigma_k - .5) of the image;
 *   S_k = the FAST 9-16 score of layer k (vo_fast_score: the largest threshold the segment test still passes, 0 where
 *        that is below t = round(255 MinContrast)); S_-1 = the FAST 5-8 score of layer 0 (the 8 neighbours, 5
 *        contiguous, never thresholded, at least 0);
 *   keypoint <=> s = S_k(x, y) > 0 is strictly greater than its 8 neighbours in S_k, strictly greater than every
 *        S_k'(x', y'), k' = k - 1 and k + 1 where they exist, over the pixels of layer k' whose centres lie within the
 *        3 x 3 pixel patch of layer k around (x, y) [x' from ceil(((2x-1) p - q) / (2q)) to floor(((2x+3) p - q) / (2q)),
 *        p / q = sigma_k / sigma_k'], s > MinQuality * max over all S_k, and the pixel's centre in the image, rounded
 *        down, at least `border` pixels from every edge;
 *   position: per axis the vertex of the parabola through the three scores, at most half a pixel away, in the layer,
 *        then mapped to the image; Scale: the vertex of the parabola through (sigma_k-1, m_below), (sigma_k, s),
 *        (sigma_k+1, m_above), m = the maxima of the test above (layer 0: (2/3, S_-1(x, y)), used when s exceeds it;
 *        the top layer: sigma_k itself); Metric = s;
 *   output order: layer, then b, a ascending.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BRISK_MAX_LAYERS 8
static const int SIGMA2[BRISK_MAX_LAYERS] = {2, 3, 4, 6, 8, 12, 16, 24};   /* twice the scale of layer k */

void vo_brisk_half(const uint8_t *src, int A, int B, uint8_t *dst)
{
    const int A2 = A / 2, B2 = B / 2;
    for (int b = 0; b < B2; ++b)
        for (int a = 0; a < A2; ++a) {
            const uint8_t *p = src + (size_t)(2 * b) * A + 2 * a;
            dst[(size_t)b * A2 + a] = (uint8_t)((p[0] + p[1] + p[A] + p[A + 1] + 2) / 4);
        }
}

void vo_brisk_twothirds(const uint8_t *src, int A, int B, uint8_t *dst)
{
    const int A2 = 2 * (A / 3), B2 = 2 * (B / 3);
    for (int b = 0; b < B2; ++b)
        for (int a = 0; a < A2; ++a) {
            /* output (a, b) covers [1.5 a, 1.5 a + 1.5): its corner pixel of the 3 x 3 block and half of the middle one */
            const int a0 = 3 * (a / 2) + 2 * (a % 2), b0 = 3 * (b / 2) + 2 * (b % 2);   /* the fully covered pixel */
            const int am = 3 * (a / 2) + 1, bm = 3 * (b / 2) + 1;                       /* the block's middle */
            const int corner = src[(size_t)b0 * A + a0], ea = src[(size_t)b0 * A + am], eb = src[(size_t)bm * A + a0],
                      centre = src[(size_t)bm * A + am];
            dst[(size_t)b * A2 + a] = (uint8_t)((4 * corner + 2 * ea + 2 * eb + centre + 4) / 9);
        }
}

/* FAST 5-8 score of every pixel (0 on the outermost ring): the 8 neighbours clockwise from the top. */
void vo_fast58_score(const uint8_t *img, int A, int B, int *S)
{
    static const int ca[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    static const int cb[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
    memset(S, 0, sizeof(int) * (size_t)A * B);
    for (int b = 1; b < B - 1; ++b)
        for (int a = 1; a < A - 1; ++a) {
            const int p = img[(size_t)b * A + a];
            int best = -256;
            for (int s = 0; s < 8; ++s) {
                int mn_bright = 256, mn_dark = 256;
                for (int j = 0; j < 5; ++j) {
                    const int k = (s + j) % 8;
                    const int d = (int)img[(size_t)(b + cb[k]) * A + (a + ca[k])] - p;
                    if (d < mn_bright) mn_bright = d;
                    if (-d < mn_dark) mn_dark = -d;
                }
                if (mn_bright > best) best = mn_bright;
                if (mn_dark > best) best = mn_dark;
            }
            S[(size_t)b * A + a] = best - 1 > 0 ? best - 1 : 0;
        }
}

static int floor_div(int n, int d) { return n >= 0 ? n / d : -((-n + d - 1) / d); }

/* the largest score of layer k2 (A2 x B2) over the pixels whose centres lie within the 3 x 3 patch of the pixel
 * (a, b) of a layer whose scale is p / q times that of layer k2's */
static int patch_max(const int *S2, int A2, int B2, int a, int b, int p, int q)
{
    int a_lo = -floor_div(-(((2 * a - 1) * p) - q), 2 * q), a_hi = floor_div((2 * a + 3) * p - q, 2 * q);
    int b_lo = -floor_div(-(((2 * b - 1) * p) - q), 2 * q), b_hi = floor_div((2 * b + 3) * p - q, 2 * q);
    if (a_lo < 0) a_lo = 0;
    if (b_lo < 0) b_lo = 0;
    if (a_hi > A2 - 1) a_hi = A2 - 1;
    if (b_hi > B2 - 1) b_hi = B2 - 1;
    int m = 0;
    for (int y = b_lo; y <= b_hi; ++y)
        for (int x = a_lo; x <= a_hi; ++x)
            if (S2[(size_t)y * A2 + x] > m) m = S2[(size_t)y * A2 + x];
    return m;
}

/* Layer k of the pyramid (0 .. 7): its size and, when out != NULL, its pixels (A_k * B_k bytes). */
void vo_brisk_layer(const uint8_t *img, int A, int B, int k, uint8_t *out, int *Ak, int *Bk)
{
    uint8_t *cur = (uint8_t *)malloc((size_t)A * B + 1), *nxt = (uint8_t *)malloc((size_t)A * B + 1);
    memcpy(cur, img, (size_t)A * B);
    int a = A, b = B;
    if (k & 1) {
        vo_brisk_twothirds(cur, a, b, nxt);
        a = 2 * (a / 3);
        b = 2 * (b / 3);
        uint8_t *t = cur;
        cur = nxt;
        nxt = t;
    }
    for (int i = 0; i < k / 2; ++i) {
        vo_brisk_half(cur, a, b, nxt);
        a /= 2;
        b /= 2;
        uint8_t *t = cur;
        cur = nxt;
        nxt = t;
    }
    if (out) memcpy(out, cur, (size_t)a * b);
    *Ak = a;
    *Bk = b;
    free(cur);
    free(nxt);
}

/* Returns the number of keypoints found (may exceed cap; only the first cap are written).
 * ab_out: cap x 2 column-major 0-based (a..., b...) positions in the image; metric_out, scale_out: cap. */
int vo_detect_brisk(const uint8_t *img, int A, int B, double min_contrast, double min_quality, int num_octaves, int border,
                    float *ab_out, float *metric_out, float *scale_out, int cap)
{
    const int nl = num_octaves < 1 ? 1 : (num_octaves > 4 ? 8 : 2 * num_octaves);
    const int t = (int)floor(min_contrast * 255.0 + 0.5);
    const int lo = border < 1 ? 1 : border;
    int *S[BRISK_MAX_LAYERS], Ak[BRISK_MAX_LAYERS], Bk[BRISK_MAX_LAYERS];
    int smax = 0;
    for (int k = 0; k < nl; ++k) {
        vo_brisk_layer(img, A, B, k, NULL, &Ak[k], &Bk[k]);
        uint8_t *L = (uint8_t *)malloc((size_t)Ak[k] * Bk[k] + 1);
        vo_brisk_layer(img, A, B, k, L, &Ak[k], &Bk[k]);
        S[k] = (int *)calloc((size_t)Ak[k] * Bk[k] + 1, sizeof(int));
        if (Ak[k] >= 7 && Bk[k] >= 7) vo_fast_score(L, Ak[k], Bk[k], t, S[k]);
        for (size_t i = 0; i < (size_t)Ak[k] * Bk[k]; ++i)
            if (S[k][i] > smax) smax = S[k][i];
        free(L);
    }
    int *S58 = (int *)calloc((size_t)A * B, sizeof(int));
    vo_fast58_score(img, A, B, S58);
    const float thr = (float)min_quality * (float)smax;
    int n = 0;
    for (int k = 0; k < nl; ++k) {
        const int Aw = Ak[k], Bw = Bk[k], s2 = SIGMA2[k];
        for (int b = 1; b < Bw - 1; ++b)
            for (int a = 1; a < Aw - 1; ++a) {
                const int *c = S[k] + (size_t)b * Aw + a;
                const int s = c[0];
                if (s <= 0 || !((float)s > thr)) continue;
                if (!(s > c[-1] && s > c[1] && s > c[-Aw] && s > c[Aw] && s > c[-Aw - 1] && s > c[-Aw + 1] &&
                      s > c[Aw - 1] && s > c[Aw + 1]))
                    continue;
                const int Xa = ((2 * a + 1) * s2) / 4, Xb = ((2 * b + 1) * s2) / 4;   /* the pixel's centre in the image */
                if (Xa < lo || Xa >= A - lo || Xb < lo || Xb >= B - lo) continue;
                int m_below = 0, m_above = 0;
                if (k > 0) {
                    m_below = patch_max(S[k - 1], Ak[k - 1], Bk[k - 1], a, b, s2, SIGMA2[k - 1]);
                    if (!(s > m_below)) continue;
                }
                if (k + 1 < nl) {
                    m_above = patch_max(S[k + 1], Ak[k + 1], Bk[k + 1], a, b, s2, SIGMA2[k + 1]);
                    if (!(s > m_above)) continue;
                }
                if (n < cap) {
                    const float r = (float)s;
                    const float da = (float)(c[1] - c[-1]), dda = (float)((c[1] - 2 * s) + c[-1]);
                    const float db = (float)(c[Aw] - c[-Aw]), ddb = (float)((c[Aw] - 2 * s) + c[-Aw]);
                    float oa = dda < 0.0f ? -0.5f * da / dda : 0.0f;
                    float ob = ddb < 0.0f ? -0.5f * db / ddb : 0.0f;
                    oa = fminf(fmaxf(oa, -0.5f), 0.5f);
                    ob = fminf(fmaxf(ob, -0.5f), 0.5f);
                    const double sg = 0.5 * (double)s2;
                    ab_out[n] = (float)((((double)a + (double)oa) + 0.5) * sg - 0.5);
                    ab_out[cap + n] = (float)((((double)b + (double)ob) + 0.5) * sg - 0.5);
                    metric_out[n] = r;
                    double scale = sg;
                    if (k + 1 < nl) {
                        double x0, y0;
                        int have = 1;
                        if (k > 0) {
                            x0 = 0.5 * (double)SIGMA2[k - 1];
                            y0 = (double)m_below;
                        } else {
                            x0 = 2.0 / 3.0;
                            y0 = (double)S58[(size_t)b * A + a];
                            have = (double)s > y0;
                        }
                        if (have) {
                            const double x1 = sg, y1 = (double)s, x2 = 0.5 * (double)SIGMA2[k + 1], y2 = (double)m_above;
                            const double num = ((x1 * x1 - x2 * x2) * y0 + (x2 * x2 - x0 * x0) * y1) + (x0 * x0 - x1 * x1) * y2;
                            const double den = ((x1 - x2) * y0 + (x2 - x0) * y1) + (x0 - x1) * y2;
                            scale = 0.5 * num / den;
                        }
                    }
                    if (scale_out) scale_out[n] = (float)scale;
                }
                ++n;
            }
    }
    for (int k = 0; k < nl; ++k) free(S[k]);
    free(S58);
    return n;
}
