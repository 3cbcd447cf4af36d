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

/*
 * SURF descriptors and their matching, the stand-ins for extractFeatures(I, SURFPoints) (genKeypoints.m:32) and
 * matchFeatures(feat0, feat1) (matching_init.m:26-29, the 'Des_match' bootstrap branch).  PARITY UNPINNED: closed
 * toolbox functions, no fixtures in the reference.  This project's definition, after section 4 of Bay et al.:
 *
 *   the image is addressed as (x, y) whatever its storage; all box sums are clipped to the image;
 *   haarX(x, y, z) = sum over [x, x + z/2) x [y - z/2, y - z/2 + z) - sum over [x - z/2, x) x (same rows)   (z/2 integer)
 *   haarY(x, y, z) = the same with the axes exchanged
 *   S = round(Scale), (X, Y) = (round(x), round(y))
 *   orientation: for the 109 integer offsets (i, j), i^2 + j^2 < 36:  w = exp(-(i^2 + j^2) / 8),
 *        r = w (haarX, haarY)(X + i S, Y + j S, 4 S), in the order j outer, i inner;
 *        42 windows k = 0..41 between the directions 0.15 k and 0.15 k + pi/3: the sum of the r inside
 *        (cross(u_k, r) >= 0 and cross(r, v_k) >= 0), the window with the largest squared sum (the first of equals)
 *        gives the direction (co, si) = sum / |sum|  ((1, 0) when every sum is zero);  Orientation = atan2(si, co)
 *   descriptor: 4 x 4 subregions of 5 x 5 samples, sample (i, j), i, j = -10..9, at offset (i + 0.5, j + 0.5) Scale
 *        in the rotated frame: xs = round(x + (co u - si v)), ys = round(y + (si u + co v)), u = (i + .5) Scale ...
 *        (rx, ry) = (haarX, haarY)(xs, ys, 2 S); rotated back: dx = co rx + si ry, dy = -si rx + co ry;
 *        weight exp(-((i + .5)^2 + (j + .5)^2) / (2 * 3.3^2));
 *        subregion (p, q) = (i + 10) / 5, (j + 10) / 5 collects sum dx, sum dy, sum |dx|, sum |dy| over its samples
 *        in the order j outer, i inner; the 64 numbers in the order q outer, p inner, 4 per subregion;
 *        divided by the square root of the sum of their squares (summed in that order; 0 stays 0).
 *   Doubles in a fixed order, stored as floats.  Upright: (co, si) = (1, 0).
 *
 *   matching: d(i, j) = sum_k (f0[i][k] - f1[j][k])^2 in floats, k ascending; for every i the nearest j (lowest j
 *   of equals) and the second nearest; a pair is kept when d1 <= 4 MatchThreshold / 100 and d1 <= MaxRatio^2 d2
 *   (the ratio test on Euclidean distances; no second neighbour: kept); pairs in the order of i.
 */
static int g_A, g_B, g_swap;

static long long px_box(const long long *S, int x0, int y0, int x1, int y1)
{
    /* sum over x in [x0, x1), y in [y0, y1), clipped; storage: a = x, b = y, or exchanged */
    const int W = g_swap ? g_B : g_A, H = g_swap ? g_A : g_B;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > W) x1 = W;
    if (y1 > H) y1 = H;
    if (x1 <= x0 || y1 <= y0) return 0;
    return g_swap ? box_sum(S, g_A + 1, y0, x0, y1, x1) : box_sum(S, g_A + 1, x0, y0, x1, y1);
}
static double haar_x(const long long *S, int x, int y, int z)
{
    return (double)(px_box(S, x, y - z / 2, x + z / 2, y - z / 2 + z) - px_box(S, x - z / 2, y - z / 2, x, y - z / 2 + z));
}
static double haar_y(const long long *S, int x, int y, int z)
{
    return (double)(px_box(S, x - z / 2, y, x - z / 2 + z, y + z / 2) - px_box(S, x - z / 2, y - z / 2, x - z / 2 + z, y));
}

/* xy: n x 2 column-major (x..., y...), 1-based as the toolbox objects hold them (the definition above is in 0-based
 * coordinates: 1 is subtracted in double); features: n x 64 row-major; orientation: n (may be NULL).
 * swap_xy: the image is stored with y as the fast axis (MATLAB's column-major arrays). */
void vo_surf_descriptors(const uint8_t *img, int A, int B, int swap_xy, const float *xy, const float *scale, int n,
                         int upright, float *features, float *orientation)
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
    g_A = A;
    g_B = B;
    g_swap = swap_xy;
    double wu[42], wv[42], eu[42], ev[42];   /* window edges: first (wu, wv), second (eu, ev) */
    for (int k = 0; k < 42; ++k) {
        wu[k] = cos(0.15 * k);
        wv[k] = sin(0.15 * k);
        eu[k] = cos(0.15 * k + 3.14159265358979323846 / 3.0);
        ev[k] = sin(0.15 * k + 3.14159265358979323846 / 3.0);
    }
    for (int p = 0; p < n; ++p) {
        const double x = (double)xy[p] - 1.0, y = (double)xy[n + p] - 1.0, sc = scale[p];
        const int Sx = (int)floor(sc + 0.5), X = (int)floor(x + 0.5), Y = (int)floor(y + 0.5);
        double co = 1.0, si = 0.0;
        if (!upright) {
            double rx[109], ry[109];
            int m = 0;
            for (int j = -6; j <= 6; ++j)
                for (int i = -6; i <= 6; ++i) {
                    if (i * i + j * j >= 36) continue;
                    const double w = exp(-(double)(i * i + j * j) / 8.0);
                    rx[m] = w * haar_x(S, X + i * Sx, Y + j * Sx, 4 * Sx);
                    ry[m] = w * haar_y(S, X + i * Sx, Y + j * Sx, 4 * Sx);
                    ++m;
                }
            double best = 0.0, bx = 0.0, by = 0.0;
            for (int k = 0; k < 42; ++k) {
                double sx = 0.0, sy = 0.0;
                for (int q = 0; q < 109; ++q) {
                    const double c1 = wu[k] * ry[q] - wv[k] * rx[q], c2 = rx[q] * ev[k] - ry[q] * eu[k];
                    if (c1 >= 0.0 && c2 >= 0.0) {
                        sx += rx[q];
                        sy += ry[q];
                    }
                }
                const double nn = sx * sx + sy * sy;
                if (nn > best) {
                    best = nn;
                    bx = sx;
                    by = sy;
                }
            }
            if (best > 0.0) {
                const double nrm = sqrt(best);
                co = bx / nrm;
                si = by / nrm;
            }
        }
        if (orientation) orientation[p] = (float)atan2(si, co);
        double d[64];
        for (int k = 0; k < 64; ++k) d[k] = 0.0;
        for (int j = -10; j < 10; ++j)
            for (int i = -10; i < 10; ++i) {
                const double u = ((double)i + 0.5) * sc, v = ((double)j + 0.5) * sc;
                const int xs = (int)floor((x + (co * u - si * v)) + 0.5), ys = (int)floor((y + (si * u + co * v)) + 0.5);
                const double w = exp(-((i + 0.5) * (i + 0.5) + (j + 0.5) * (j + 0.5)) / (2.0 * 3.3 * 3.3));
                const double hx = w * haar_x(S, xs, ys, 2 * Sx), hy = w * haar_y(S, xs, ys, 2 * Sx);
                const double dx = co * hx + si * hy, dy = co * hy - si * hx;
                double *c = d + 4 * (((j + 10) / 5) * 4 + (i + 10) / 5);
                c[0] += dx;
                c[1] += dy;
                c[2] += fabs(dx);
                c[3] += fabs(dy);
            }
        double ss = 0.0;
        for (int k = 0; k < 64; ++k) ss += d[k] * d[k];
        const double nrm = sqrt(ss);
        for (int k = 0; k < 64; ++k) features[(size_t)p * 64 + k] = (float)(nrm > 0.0 ? d[k] / nrm : 0.0);
    }
    free(S);
}

/* f0: n0 x dim, f1: n1 x dim row-major.  pairs: up to n0 x 2 (i, j) 0-based, row-major; metric: the pair's distance.
 * Returns the number of pairs. */
int vo_match_features(const float *f0, int n0, const float *f1, int n1, int dim, double match_threshold, double max_ratio,
                      int *pairs, float *metric)
{
    const float thr = (float)(4.0 * match_threshold / 100.0), r2 = (float)(max_ratio * max_ratio);
    int m = 0;
    for (int i = 0; i < n0; ++i) {
        float d1 = INFINITY, d2 = INFINITY;
        int j1 = -1;
        for (int j = 0; j < n1; ++j) {
            float d = 0.0f;
            for (int k = 0; k < dim; ++k) {
                const float e = f0[(size_t)i * dim + k] - f1[(size_t)j * dim + k];
                d += e * e;
            }
            if (d < d1) {
                d2 = d1;
                d1 = d;
                j1 = j;
            } else if (d < d2) {
                d2 = d;
            }
        }
        if (j1 < 0 || !(d1 <= thr)) continue;
        if (n1 > 1 && !(d1 <= r2 * d2)) continue;
        pairs[2 * m] = i;
        pairs[2 * m + 1] = j1;
        metric[m] = d1;
        ++m;
    }
    return m;
}
