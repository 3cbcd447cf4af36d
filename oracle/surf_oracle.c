/*
 * TEST INFRASTRUCTURE ONLY: the CPU statement the GPU detector is checked against.
 *
 * SURF interest points, the stand-in for detectSURFFeatures as genKeypoints.m:16,31 calls it (the reference's
 * default detector: main_demo.m:25; options {'MetricThreshold', v}: main_eval.m:151,167).  PARITY UNPINNED:
 * a closed-source toolbox function and no keypoint fixtures in the reference.  This project's definition,
 * after Bay, Ess, Tuytelaars, Van Gool, "Speeded-Up Robust Features" (CVIU 2008), with the filter sizes of
 * the toolbox documentation (NumOctaves 3, NumScaleLevels 4):
 *
 *   octave o = 0, 1, 2 samples the image every s = 2^o pixels with box filters of side
 *        L = 9 15 21 27 | 15 27 39 51 | 27 51 75 99,    lobe l = L / 3
 *   at a sample centred on pixel (a, b) whose filter lies inside the image (0 elsewhere):
 *        Dxx = (sum over [a-(L-1)/2, a+(L-1)/2] x [b-l+1, b+l-1]  -  3 * its middle third along a) / (l (2l-1))
 *        Dyy = the same with the axes exchanged
 *        Dxy = (sum of the l x l boxes at the (-,-) and (+,+) corners, one pixel off the centre row and
 *               column, minus those at (+,-) and (-,+)) / l^2
 *        det = Dxx Dyy - 0.81 Dxy^2          (intensities on the 0..255 scale; double, stored as float)
 *   interest point <=> middle level m = 1, 2 of an octave; det > MetricThreshold; det strictly greater than
 *        its 26 neighbours in the 3 x 3 x 3 block of samples (levels m-1, m, m+1 of the same octave); the
 *        filter of level m+1 lies inside the image at every sample of that block; the sample's pixel is at
 *        least `border` pixels from every edge
 *   refinement: x = -H^-1 g of the quadratic through the block (central differences over a, b, level);
 *        the point is dropped when the system is singular or an |x_i| exceeds 1; position (sample + x) * s,
 *        Scale = (1.2 / 9) (L_m + x_level (L_m - L_m-1)), Metric = det
 *   output order: octave, level, b, a ascending.
 * Written as plain loops over an integral image of 64-bit sums.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const int SURF_SIZES[3][4] = {{9, 15, 21, 27}, {15, 27, 39, 51}, {27, 51, 75, 99}};

/* sum of img over [a0, a1) x [b0, b1) from the integral image with a zero first row and column */
static long long box_sum(const long long *S, int A1, int a0, int b0, int a1, int b1)
{
    return S[(size_t)b1 * A1 + a1] - S[(size_t)b0 * A1 + a1] - S[(size_t)b1 * A1 + a0] + S[(size_t)b0 * A1 + a0];
}

/* The response map of (octave, level): (GB, GA) samples, GA = ceil(A / s), GB = ceil(B / s).  R is GA * GB floats. */
void vo_surf_response(const uint8_t *img, int A, int B, int octave, int level, float *R)
{
    const int A1 = A + 1;
    long long *S = (long long *)calloc((size_t)A1 * (B + 1), sizeof(long long));
    for (int b = 0; b < B; ++b) {
        long long row = 0;
        for (int a = 0; a < A; ++a) {
            row += img[(size_t)b * A + a];
            S[(size_t)(b + 1) * A1 + (a + 1)] = S[(size_t)b * A1 + (a + 1)] + row;
        }
    }
    const int s = 1 << octave, L = SURF_SIZES[octave][level], l = L / 3, h = (L - 1) / 2;
    const int GA = (A + s - 1) / s, GB = (B + s - 1) / s;
    const double inv_lobe = 1.0 / (double)(l * (2 * l - 1)), inv_sq = 1.0 / (double)(l * l);
    for (int gb = 0; gb < GB; ++gb)
        for (int ga = 0; ga < GA; ++ga) {
            const int a = ga * s, b = gb * s;
            float r = 0.0f;
            if (a - h >= 0 && a + h <= A - 1 && b - h >= 0 && b + h <= B - 1) {
                const long long nxx = box_sum(S, A1, a - h, b - l + 1, a + h + 1, b + l) -
                                      3 * box_sum(S, A1, a - h + l, b - l + 1, a - h + 2 * l, b + l);
                const long long nyy = box_sum(S, A1, a - l + 1, b - h, a + l, b + h + 1) -
                                      3 * box_sum(S, A1, a - l + 1, b - h + l, a + l, b - h + 2 * l);
                const long long nxy = box_sum(S, A1, a - l, b - l, a, b) + box_sum(S, A1, a + 1, b + 1, a + 1 + l, b + 1 + l) -
                                      box_sum(S, A1, a + 1, b - l, a + 1 + l, b) - box_sum(S, A1, a - l, b + 1, a, b + 1 + l);
                const double dxx = (double)nxx * inv_lobe, dyy = (double)nyy * inv_lobe, dxy = (double)nxy * inv_sq;
                const double t1 = dxx * dyy, t2 = dxy * dxy, t3 = 0.81 * t2;
                r = (float)(t1 - t3);
            }
            R[(size_t)gb * GA + ga] = r;
        }
    free(S);
}

/* Returns the number of points found (may exceed cap; only the first cap are written).
 * ab_out: cap x 2 column-major 0-based (a..., b...) positions; metric_out, scale_out: cap. */
int vo_detect_surf(const uint8_t *img, int A, int B, double metric_threshold, int border, float *ab_out,
                   float *metric_out, float *scale_out, int cap)
{
    const float thr = (float)metric_threshold;
    const int lo = border < 1 ? 1 : border;
    int n = 0;
    for (int o = 0; o < 3; ++o) {
        const int s = 1 << o, GA = (A + s - 1) / s, GB = (B + s - 1) / s;
        float *R[4];
        for (int i = 0; i < 4; ++i) {
            R[i] = (float *)malloc(sizeof(float) * (size_t)GA * GB);
            vo_surf_response(img, A, B, o, i, R[i]);
        }
        for (int m = 1; m <= 2; ++m) {
            const int htop = (SURF_SIZES[o][m + 1] - 1) / 2;
            /* samples at which the filter of level m + 1 lies inside the image */
            const int lo_a = (htop + s - 1) / s, hi_a = (A - 1 - htop) >= 0 ? (A - 1 - htop) / s : -1;
            const int lo_b = lo_a, hi_b = (B - 1 - htop) >= 0 ? (B - 1 - htop) / s : -1;
            for (int gb = lo_b + 1; gb <= hi_b - 1; ++gb)
                for (int ga = lo_a + 1; ga <= hi_a - 1; ++ga) {
                    const int a = ga * s, b = gb * s;
                    if (a < lo || a >= A - lo || b < lo || b >= B - lo) continue;
                    const float v = R[m][(size_t)gb * GA + ga];
                    if (!(v > thr)) continue;
                    int is_max = 1;
                    for (int di = -1; di <= 1 && is_max; ++di)
                        for (int db = -1; db <= 1 && is_max; ++db)
                            for (int da = -1; da <= 1; ++da) {
                                if (!di && !db && !da) continue;
                                if (!(v > R[m + di][(size_t)(gb + db) * GA + (ga + da)])) {
                                    is_max = 0;
                                    break;
                                }
                            }
                    if (!is_max) continue;
#define NB(di, db, da) ((double)R[m + (di)][(size_t)(gb + (db)) * GA + (ga + (da))])
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
                    /* cofactors of the symmetric 3 x 3 matrix [aa ab as; ab bb bs; as bs ss] */
                    const double c_aa = h_bb * h_ss - h_bs * h_bs;
                    const double c_ab = h_as * h_bs - h_ab * h_ss;
                    const double c_as = h_ab * h_bs - h_as * h_bb;
                    const double c_bb = h_aa * h_ss - h_as * h_as;
                    const double c_bs = h_ab * h_as - h_aa * h_bs;
                    const double c_ss = h_aa * h_bb - h_ab * h_ab;
                    const double det3 = (h_aa * c_aa + h_ab * c_ab) + h_as * c_as;
                    if (det3 == 0.0) continue;
                    const double x_a = -(((c_aa * g_a + c_ab * g_b) + c_as * g_s) / det3);
                    const double x_b = -(((c_ab * g_a + c_bb * g_b) + c_bs * g_s) / det3);
                    const double x_s = -(((c_as * g_a + c_bs * g_b) + c_ss * g_s) / det3);
                    if (!(fabs(x_a) <= 1.0 && fabs(x_b) <= 1.0 && fabs(x_s) <= 1.0)) continue;
                    if (n < cap) {
                        const int Lm = SURF_SIZES[o][m], dL = Lm - SURF_SIZES[o][m - 1];
                        ab_out[n] = (float)(((double)ga + x_a) * (double)s);
                        ab_out[cap + n] = (float)(((double)gb + x_b) * (double)s);
                        metric_out[n] = v;
                        if (scale_out) scale_out[n] = (float)((1.2 / 9.0) * ((double)Lm + x_s * (double)dL));
                    }
                    ++n;
                }
        }
        for (int i = 0; i < 4; ++i) free(R[i]);
    }
    return n;
}
