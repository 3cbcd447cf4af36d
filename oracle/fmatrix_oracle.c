/*
 * fmatrix_oracle.c -- CPU statement of the fundamental-matrix RANSAC that stands in for
 * estimateFundamentalMatrix(p1, p2, 'NumTrials', 500, 'Method', 'RANSAC', 'DistanceThreshold',
 * 5e-2) as update_landmarks.m:13-14 calls it (only the inlier index is used there).
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
 *
 * PARITY UNPINNED: the toolbox function is closed source and consumes MATLAB's global RNG.  The
 * definition below is this project's own:
 *   - all points are normalised once (centroid to the origin, mean distance sqrt(2));
 *   - each trial takes 8 point pairs (index sets supplied by the caller), solves the 8x9
 *     homogeneous system for its null vector (Gaussian elimination with complete pivoting),
 *     enforces rank 2 (smallest singular value set to zero, one-sided Jacobi), de-normalises;
 *   - a pair is an inlier when its Sampson distance, in pixels^2, is <= the threshold;
 *   - the best trial is the first with the largest inlier count; after every improvement the
 *     trial budget shrinks to ceil(log(1 - c) / log(1 - (inliers / n)^8)) (c = Confidence / 100),
 *     never above NumTrials.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* Hartley normalisation of n points (x_i, y_i): returns s, cx, cy with  x' = s (x - cx). */
static void hartley(const double *xy /*2 x n col-major: x0,y0,x1,...*/, int n, double *s, double *cx, double *cy)
{
    double mx = 0, my = 0;
    for (int i = 0; i < n; ++i) { mx += xy[2 * i]; my += xy[2 * i + 1]; }
    mx /= n; my /= n;
    double md = 0;
    for (int i = 0; i < n; ++i) {
        double dx = xy[2 * i] - mx, dy = xy[2 * i + 1] - my;
        md += sqrt(dx * dx + dy * dy);
    }
    md /= n;
    *s = md > 0 ? sqrt(2.0) / md : 1.0;
    *cx = mx; *cy = my;
}

/* null vector of the 8x9 matrix A (row-major), complete pivoting.  Returns 0 when rank < 8. */
static int null9(double A[8][9], double x[9])
{
    int perm[9];
    for (int c = 0; c < 9; ++c) perm[c] = c;
    for (int k = 0; k < 8; ++k) {
        int pr = k, pc = k;
        double best = 0.0;
        for (int r = k; r < 8; ++r)
            for (int c = k; c < 9; ++c)
                if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); pr = r; pc = c; }
        if (!(best > 0.0)) return 0;
        if (pr != k) for (int c = 0; c < 9; ++c) { double t = A[k][c]; A[k][c] = A[pr][c]; A[pr][c] = t; }
        if (pc != k) {
            for (int r = 0; r < 8; ++r) { double t = A[r][k]; A[r][k] = A[r][pc]; A[r][pc] = t; }
            int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t;
        }
        for (int r = k + 1; r < 8; ++r) {
            double f = A[r][k] / A[k][k];
            for (int c = k; c < 9; ++c) A[r][c] -= f * A[k][c];
        }
    }
    double y[9];
    y[8] = 1.0;
    for (int k = 7; k >= 0; --k) {
        double s = 0.0;
        for (int c = k + 1; c < 9; ++c) s += A[k][c] * y[c];
        y[k] = -s / A[k][k];
    }
    double nn = 0.0;
    for (int c = 0; c < 9; ++c) nn += y[c] * y[c];
    nn = sqrt(nn);
    for (int c = 0; c < 9; ++c) x[perm[c]] = y[c] / nn;
    return 1;
}

/* F (row-major 3x3) <- nearest rank-2 matrix: smallest singular value zeroed. */
static void rank2(double F[3][3])
{
    double A[3][3], V[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    memcpy(A, F, sizeof(A));
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 3; ++r) { al += A[r][i] * A[r][i]; be += A[r][j] * A[r][j]; ga += A[r][i] * A[r][j]; }
                if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    double ai = A[r][i], aj = A[r][j];
                    A[r][i] = c * ai - s * aj; A[r][j] = s * ai + c * aj;
                    double vi = V[r][i], vj = V[r][j];
                    V[r][i] = c * vi - s * vj; V[r][j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    int k = 0;
    double best = INFINITY;
    for (int c = 0; c < 3; ++c) {
        double nn = A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c];
        if (nn < best) { best = nn; k = c; }
    }
    for (int r = 0; r < 3; ++r) A[r][k] = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) F[r][c] = A[r][0] * V[c][0] + A[r][1] * V[c][1] + A[r][2] * V[c][2];
}

/* F from 8 pairs.  nrm = {s1, cx1, cy1, s2, cx2, cy2}.  Returns 0 for a degenerate sample. */
int vo_fmatrix_8point(const double *p1, const double *p2, const int32_t idx[8], const double nrm[6], double F[9])
{
    double A[8][9];
    for (int r = 0; r < 8; ++r) {
        int i = idx[r];
        double x1 = nrm[0] * (p1[2 * i] - nrm[1]), y1 = nrm[0] * (p1[2 * i + 1] - nrm[2]);
        double x2 = nrm[3] * (p2[2 * i] - nrm[4]), y2 = nrm[3] * (p2[2 * i + 1] - nrm[5]);
        double row[9] = { x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0 };
        memcpy(A[r], row, sizeof(row));
    }
    double f[9];
    if (!null9(A, f)) { memset(F, 0, 9 * sizeof(double)); return 0; }
    double Fh[3][3] = { { f[0], f[1], f[2] }, { f[3], f[4], f[5] }, { f[6], f[7], f[8] } };
    rank2(Fh);
    /* F = T2' Fh T1 with T = [s 0 -s cx; 0 s -s cy; 0 0 1] */
    double T1[3][3] = { { nrm[0], 0, -nrm[0] * nrm[1] }, { 0, nrm[0], -nrm[0] * nrm[2] }, { 0, 0, 1 } };
    double T2[3][3] = { { nrm[3], 0, -nrm[3] * nrm[4] }, { 0, nrm[3], -nrm[3] * nrm[5] }, { 0, 0, 1 } };
    double M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = Fh[r][0] * T1[0][c] + Fh[r][1] * T1[1][c] + Fh[r][2] * T1[2][c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) F[3 * r + c] = T2[0][r] * M[0][c] + T2[1][r] * M[1][c] + T2[2][r] * M[2][c];
    return 1;
}

static double sampson(const double F[9], double x1, double y1, double x2, double y2)
{
    double a0 = F[0] * x1 + F[1] * y1 + F[2], a1 = F[3] * x1 + F[4] * y1 + F[5], a2 = F[6] * x1 + F[7] * y1 + F[8];
    double b0 = F[0] * x2 + F[3] * y2 + F[6], b1 = F[1] * x2 + F[4] * y2 + F[7];
    double e = x2 * a0 + y2 * a1 + a2;
    return e * e / (a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1);
}

int vo_fmatrix_count(const double *p1, const double *p2, int n, const double F[9], double thr, uint8_t *mask)
{
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        double d = sampson(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
        int in = d <= thr;
        if (mask) mask[i] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}

double vo_fmatrix_trial_budget(int inliers, int n, int num_trials, double confidence)
{
    double r = (double)inliers / (double)n;
    double r8 = pow(r, 8.0);
    if (r8 >= 1.0) return 1.0;
    if (r8 <= 0.0) return (double)num_trials;
    double t = ceil(log(1.0 - confidence / 100.0) / log(1.0 - r8));
    return t < (double)num_trials ? t : (double)num_trials;
}

/* p1, p2: 2 x n (x; y) column-major.  samples 8 x num_trials (0-based).  mask n bytes.
 * Returns the number of trials executed; *best_count the inlier count (0 = no model). */
int vo_fmatrix_ransac(const double *p1, const double *p2, int n, const int32_t *samples, int num_trials,
                      double thr, double confidence, uint8_t *mask, int *best_count, double Fbest[9],
                      int32_t *counts)
{
    double nrm[6];
    hartley(p1, n, &nrm[0], &nrm[1], &nrm[2]);
    hartley(p2, n, &nrm[3], &nrm[4], &nrm[5]);
    memset(mask, 0, (size_t)n);
    *best_count = 0;
    double budget = (double)num_trials;
    int t = 0;
    while ((double)t < budget && t < num_trials) {
        double F[9];
        int c = 0;
        if (vo_fmatrix_8point(p1, p2, samples + 8 * t, nrm, F)) c = vo_fmatrix_count(p1, p2, n, F, thr, NULL);
        if (counts) counts[t] = c;
        if (c > *best_count) {
            *best_count = c;
            vo_fmatrix_count(p1, p2, n, F, thr, mask);
            if (Fbest) memcpy(Fbest, F, sizeof(F));
            budget = vo_fmatrix_trial_budget(c, n, num_trials, confidence);
        }
        ++t;
    }
    return t;
}
