/*
 * harris_oracle.c -- CPU statement of the corner detector that stands in for
 * detectHarrisFeatures / detectMinEigenFeatures + the border drop of extractFeatures as
 * genKeypoints.m:16-18,31-32 calls them.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
 *
 * PARITY UNPINNED: the detectors are closed-source Computer Vision Toolbox functions and the
 * reference holds no keypoint fixtures.  The definition below is this project's own:
 *   I      = single(image) / 255, extended beyond the border by replication
 *   gx, gy = I(a+1) - I(a-1), I(b+1) - I(b-1)                       ([-1 0 1] filter)
 *   A,B,C  = gaussian(gx gx), gaussian(gy gy), gaussian(gx gy)      separable, FilterSize taps,
 *                                                                   sigma = FilterSize / 3
 *   Harris   R = (A B - C C) - 0.04 (A + B)^2
 *   MinEigen R = ((A + B) - sqrt((A - B)^2 + 4 C C)) / 2
 *   corner  <=> R > MinQuality * max(R), R strictly greater than its 8 neighbours, and at least
 *               `border` pixels away from every image edge
 *   sub-pixel offset per axis from the parabola through the 3 responses, clamped to +-0.5
 * Output order: ascending linear index b * A + a (for a column-major MATLAB image that is the
 * order of find() on the corner mask).  All arithmetic single precision, evaluated left to
 * right without contraction.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void vo_gaussian_taps(int filter_size, float *w)
{
    double sigma = filter_size / 3.0, sum = 0.0, t[32];
    int half = filter_size / 2;
    for (int i = 0; i < filter_size; ++i) {
        double d = i - half;
        t[i] = exp(-d * d / (2.0 * sigma * sigma));
        sum += t[i];
    }
    for (int i = 0; i < filter_size; ++i) w[i] = (float)(t[i] / sum);
}

/* response map R (A x B floats, img[b*A+a] layout); method 0 = Harris, 1 = MinEigen */
void vo_corner_response(const uint8_t *img, int A, int B, int filter_size, int method, float *R)
{
    const int half = filter_size / 2;
    float w[32];
    vo_gaussian_taps(filter_size, w);
    const float s255 = 1.0f / 255.0f;
    /* products on the image extended by `half` in every direction */
    const int EA = A + 2 * half, EB = B + 2 * half;
    float *pxx = (float *)malloc(sizeof(float) * 3 * (size_t)EA * EB);
    float *pyy = pxx + (size_t)EA * EB, *pxy = pyy + (size_t)EA * EB;
#define IMG(a, b) ((float)img[(size_t)clampi(b, 0, B - 1) * A + clampi(a, 0, A - 1)] * s255)
    for (int eb = 0; eb < EB; ++eb)
        for (int ea = 0; ea < EA; ++ea) {
            int a = ea - half, b = eb - half;
            float gx = IMG(a + 1, b) - IMG(a - 1, b);
            float gy = IMG(a, b + 1) - IMG(a, b - 1);
            size_t k = (size_t)eb * EA + ea;
            pxx[k] = gx * gx;
            pyy[k] = gy * gy;
            pxy[k] = gx * gy;
        }
#undef IMG
    /* separable gaussian: along a, then along b */
    float *hx = (float *)malloc(sizeof(float) * 3 * (size_t)A * EB);
    float *hy = hx + (size_t)A * EB, *hxy = hy + (size_t)A * EB;
    for (int eb = 0; eb < EB; ++eb)
        for (int a = 0; a < A; ++a) {
            float sx = 0, sy = 0, sxy = 0;
            for (int i = 0; i < filter_size; ++i) {
                size_t k = (size_t)eb * EA + a + i;
                if (i == 0) { sx = w[0] * pxx[k]; sy = w[0] * pyy[k]; sxy = w[0] * pxy[k]; }
                else { sx = sx + w[i] * pxx[k]; sy = sy + w[i] * pyy[k]; sxy = sxy + w[i] * pxy[k]; }
            }
            size_t o = (size_t)eb * A + a;
            hx[o] = sx; hy[o] = sy; hxy[o] = sxy;
        }
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            float sx = 0, sy = 0, sxy = 0;
            for (int i = 0; i < filter_size; ++i) {
                size_t k = (size_t)(b + i) * A + a;
                if (i == 0) { sx = w[0] * hx[k]; sy = w[0] * hy[k]; sxy = w[0] * hxy[k]; }
                else { sx = sx + w[i] * hx[k]; sy = sy + w[i] * hy[k]; sxy = sxy + w[i] * hxy[k]; }
            }
            float r;
            if (method == 0) {
                float det = sx * sy - sxy * sxy;
                float tr = sx + sy;
                r = det - 0.04f * (tr * tr);
            } else {
                float d = sx - sy;
                r = ((sx + sy) - sqrtf(d * d + 4.0f * (sxy * sxy))) * 0.5f;
            }
            R[(size_t)b * A + a] = r;
        }
    free(pxx);
    free(hx);
}

/* Returns the number of corners found (may exceed cap; only the first cap are written).
 * ab_out: cap x 2 column-major 0-based (a..., b...) sub-pixel positions; metric_out: cap. */
int vo_detect_corners(const uint8_t *img, int A, int B, double min_quality, int filter_size, int border,
                      int method, float *ab_out, float *metric_out, int cap)
{
    float *R = (float *)malloc(sizeof(float) * (size_t)A * B);
    vo_corner_response(img, A, B, filter_size, method, R);
    float rmax = R[0];
    for (size_t i = 1; i < (size_t)A * B; ++i) if (R[i] > rmax) rmax = R[i];
    const float thr = (float)min_quality * rmax;
    const int lo = border < 1 ? 1 : border;
    int n = 0;
    for (int b = lo; b < B - lo; ++b)
        for (int a = lo; a < A - lo; ++a) {
            const float *c = R + (size_t)b * A + a;
            float r = c[0];
            if (!(r > thr)) continue;
            if (!(r > c[-1] && r > c[1] && r > c[-A] && r > c[A] && r > c[-A - 1] && r > c[-A + 1] &&
                  r > c[A - 1] && r > c[A + 1]))
                continue;
            if (n < cap) {
                float da = c[1] - c[-1], dda = (c[1] - 2.0f * r) + c[-1];
                float db = c[A] - c[-A], ddb = (c[A] - 2.0f * r) + c[-A];
                float oa = dda < 0.0f ? -0.5f * da / dda : 0.0f;
                float ob = ddb < 0.0f ? -0.5f * db / ddb : 0.0f;
                oa = fminf(fmaxf(oa, -0.5f), 0.5f);
                ob = fminf(fmaxf(ob, -0.5f), 0.5f);
                ab_out[n] = (float)a + oa;
                ab_out[cap + n] = (float)b + ob;
                metric_out[n] = r;
            }
            ++n;
        }
    free(R);
    return n;
}

/*
 * FAST, the stand-in for detectFASTFeatures + extractFeatures' border drop as genKeypoints.m:19,31-32
 * calls them (main_eval.m:172-173: 'MinContrast' 0.11).  PARITY UNPINNED (closed-source toolbox
 * function, no keypoint fixtures in the reference).  This project's definition, after Rosten &
 * Drummond's segment test:
 *   circle  = the 16 pixels of the Bresenham circle of radius 3 around p
 *   corner <=> 9 contiguous circle pixels are all > I_p + t or all < I_p - t, t = round(255 MinContrast)
 *   Metric  = the largest t for which that still holds
 *           = max over the 16 arcs of max(min_q (I_q - I_p), min_q (I_p - I_q)) - 1
 *   pixels closer than 3 px to an image edge are never corners
 *   kept   <=> Metric strictly greater than the Metric of the 8 neighbours (non-corners count 0),
 *              Metric > MinQuality * max(Metric), at least `border` px from every edge
 * Integer positions, output order as vo_detect_corners.  Written as the plain double loop over arcs.
 */
void vo_fast_score(const uint8_t *img, int A, int B, int threshold, int *S)
{
    static const int ca[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    static const int cb[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
    memset(S, 0, sizeof(int) * (size_t)A * B);
    for (int b = 3; b < B - 3; ++b)
        for (int a = 3; a < A - 3; ++a) {
            const int p = img[(size_t)b * A + a];
            int best = -256;
            for (int s = 0; s < 16; ++s) {
                int mn_bright = 256, mn_dark = 256;
                for (int j = 0; j < 9; ++j) {
                    const int k = (s + j) % 16;
                    const int d = (int)img[(size_t)(b + cb[k]) * A + (a + ca[k])] - p;
                    if (d < mn_bright) mn_bright = d;
                    if (-d < mn_dark) mn_dark = -d;
                }
                if (mn_bright > best) best = mn_bright;
                if (mn_dark > best) best = mn_dark;
            }
            const int score = best - 1;
            S[(size_t)b * A + a] = score >= threshold ? score : 0;
        }
}

int vo_detect_fast(const uint8_t *img, int A, int B, double min_contrast, double min_quality, int border,
                   float *ab_out, float *metric_out, int cap)
{
    int *S = (int *)malloc(sizeof(int) * (size_t)A * B);
    vo_fast_score(img, A, B, (int)floor(min_contrast * 255.0 + 0.5), S);
    int smax = 0;
    for (size_t i = 0; i < (size_t)A * B; ++i) if (S[i] > smax) smax = S[i];
    const float thr = (float)min_quality * (float)smax;
    const int lo = border < 1 ? 1 : border;
    int n = 0;
    for (int b = lo; b < B - lo; ++b)
        for (int a = lo; a < A - lo; ++a) {
            const int *c = S + (size_t)b * A + a;
            const int r = c[0];
            if (!((float)r > thr)) continue;
            if (!(r > c[-1] && r > c[1] && r > c[-A] && r > c[A] && r > c[-A - 1] && r > c[-A + 1] &&
                  r > c[A - 1] && r > c[A + 1]))
                continue;
            if (n < cap) {
                ab_out[n] = (float)a;
                ab_out[cap + n] = (float)b;
                metric_out[n] = (float)r;
            }
            ++n;
        }
    free(S);
    return n;
}
