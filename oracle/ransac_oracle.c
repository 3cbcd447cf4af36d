/*
 * ransac_oracle.c -- CPU restatement of the reference's P3P RANSAC localisation.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).  PARITY UNPINNED versus MATLAB.
 *
 * Build with -ffp-contract=off: every statement below is meant to round exactly
 * where the MATLAB statement it restates rounds.
 *
 * MATLAB semantics that are emulated on purpose:
 *  - sqrt of a negative real and ^(1/3) of a negative real or complex value take the
 *    principal complex branch (solveQuartic.m:50-54,59-64);
 *  - a complex value whose imaginary part is zero is a real value, so a negative
 *    zero imaginary part never selects the lower branch of sqrt;
 *  - only real(.) of the roots and of the poses is used downstream (p3p.m:189-192,
 *    ransacLocalization.m:64-65).
 *
 * The P3P and quartic functions below restate, in the evaluation
 * order of the originals, the closed forms of p3p.m and solveQuartic.m by Laurent Kneip, which are
 * distributed under the following licence; its notice is retained here as the licence requires.
 *
 *   Copyright (c) 2011, Laurent Kneip, ETH Zurich
 *   All rights reserved.
 *
 *   Redistribution and use in source and binary forms, with or without
 *   modification, are permitted provided that the following conditions are met:
 *       * Redistributions of source code must retain the above copyright
 *         notice, this list of conditions and the following disclaimer.
 *       * Redistributions in binary form must reproduce the above copyright
 *         notice, this list of conditions and the following disclaimer in the
 *         documentation and/or other materials provided with the distribution.
 *       * Neither the name of ETH Zurich nor the
 *         names of its contributors may be used to endorse or promote products
 *         derived from this software without specific prior written permission.
 *
 *   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND
 *   ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED
 *   WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE
 *   DISCLAIMED. IN NO EVENT SHALL ETH ZURICH BE LIABLE FOR ANY
 *   DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES
 *   (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES;
 *   LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND
 *   ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT
 *   (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS
 *   SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
 */
#include "vo_oracle.h"

#include <math.h>
#include <string.h>

typedef struct { double re, im; } cplx;

static cplx c_make(double re, double im)
{
    cplx z; z.re = re; z.im = (im == 0.0) ? 0.0 : im; return z;
}
static cplx c_add(cplx a, cplx b) { return c_make(a.re + b.re, a.im + b.im); }
static cplx c_sub(cplx a, cplx b) { return c_make(a.re - b.re, a.im - b.im); }
static cplx c_neg(cplx a) { return c_make(-a.re, -a.im); }
static cplx c_scale(double s, cplx a) { return c_make(s * a.re, s * a.im); }
static int c_is_zero(cplx a) { return a.re == 0.0 && a.im == 0.0; }

/* Smith's algorithm for (a / b). */
static cplx c_div(cplx a, cplx b)
{
    if (b.im == 0.0) return c_make(a.re / b.re, a.im / b.re);
    if (fabs(b.re) >= fabs(b.im)) {
        double r = b.im / b.re;
        double den = b.re + b.im * r;
        return c_make((a.re + a.im * r) / den, (a.im - a.re * r) / den);
    } else {
        double r = b.re / b.im;
        double den = b.re * r + b.im;
        return c_make((a.re * r + a.im) / den, (a.im * r - a.re) / den);
    }
}

/* Principal square root. */
static cplx c_sqrt(cplx z)
{
    if (z.im == 0.0) {
        if (z.re >= 0.0) return c_make(sqrt(z.re), 0.0);
        if (z.re < 0.0) return c_make(0.0, sqrt(-z.re));
        return c_make(z.re, z.re); /* NaN */
    }
    double m = sqrt(z.re * z.re + z.im * z.im);
    if (z.re >= 0.0) {
        double t = sqrt(0.5 * (m + z.re));
        return c_make(t, z.im / (2.0 * t));
    } else {
        double t = sqrt(0.5 * (m - z.re));
        if (z.im < 0.0) t = -t;
        return c_make(z.im / (2.0 * t), t);
    }
}

/* Principal z^(1/3) = |z|^(1/3) * exp(i*arg(z)/3). */
static cplx c_cuberoot(cplx z)
{
    const double third = 1.0 / 3.0;
    if (z.im == 0.0 && z.re >= 0.0) return c_make(pow(z.re, third), 0.0);
    double m = sqrt(z.re * z.re + z.im * z.im);
    double a = atan2(z.im, z.re) * third;
    double r = pow(m, third);
    return c_make(r * cos(a), r * sin(a));
}

/* solveQuartic.m:26-66 */
void vo_solve_quartic(const double f[5], double roots_re[4], double roots_im[4])
{
    double A = f[0], B = f[1], C = f[2], D = f[3], E = f[4];
    double A2 = A * A, B2 = B * B;
    double A3 = A2 * A, B3 = B2 * B;
    double A4 = A3 * A, B4 = B3 * B;

    double alpha = -3.0 * B2 / (8.0 * A2) + C / A;
    double beta = B3 / (8.0 * A3) - B * C / (2.0 * A2) + D / A;
    double gamma = -3.0 * B4 / (256.0 * A4) + B2 * C / (16.0 * A3) - B * D / (4.0 * A2) + E / A;

    double alpha2 = alpha * alpha;
    double alpha3 = alpha2 * alpha;

    double P = -alpha2 / 12.0 - gamma;
    double Q = -alpha3 / 108.0 + alpha * gamma / 3.0 - (beta * beta) / 8.0;
    double disc = (Q * Q) / 4.0 + (P * P * P) / 27.0;
    cplx R = c_add(c_make(-Q / 2.0, 0.0), c_sqrt(c_make(disc, 0.0)));
    cplx U = c_cuberoot(R);

    cplx y;
    if (c_is_zero(U)) {
        y = c_sub(c_make(-5.0 * alpha / 6.0, 0.0), c_cuberoot(c_make(Q, 0.0)));
    } else {
        cplx PU = c_div(c_make(P, 0.0), c_scale(3.0, U));
        y = c_add(c_sub(c_make(-5.0 * alpha / 6.0, 0.0), PU), U);
    }

    cplx w = c_sqrt(c_add(c_make(alpha, 0.0), c_scale(2.0, y)));
    cplx base = c_add(c_make(3.0 * alpha, 0.0), c_scale(2.0, y));
    cplx bw = c_div(c_make(2.0 * beta, 0.0), w);
    cplx s_plus = c_sqrt(c_neg(c_add(base, bw)));
    cplx s_minus = c_sqrt(c_neg(c_sub(base, bw)));
    double shift = -B / (4.0 * A);

    cplx r0 = c_add(c_make(shift, 0.0), c_scale(0.5, c_add(w, s_plus)));
    cplx r1 = c_add(c_make(shift, 0.0), c_scale(0.5, c_sub(w, s_plus)));
    cplx r2 = c_add(c_make(shift, 0.0), c_scale(0.5, c_add(c_neg(w), s_minus)));
    cplx r3 = c_add(c_make(shift, 0.0), c_scale(0.5, c_sub(c_neg(w), s_minus)));
    roots_re[0] = r0.re; roots_im[0] = r0.im;
    roots_re[1] = r1.re; roots_im[1] = r1.im;
    roots_re[2] = r2.re; roots_im[2] = r2.im;
    roots_re[3] = r3.re; roots_im[3] = r3.im;
}

static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static double sumsq3(const double a[3]) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }

/* rows of M (row-major 3x3) times v */
static void mat_rows_vec(const double M[3][3], const double v[3], double out[3])
{
    for (int i = 0; i < 3; ++i) out[i] = M[i][0] * v[0] + M[i][1] * v[1] + M[i][2] * v[2];
}

/* Intermediate camera frame of p3p.m:62-74 (and again :80-92). */
static void camera_frame(const double f1[3], const double f2[3], const double f3_in[3],
                         double T[3][3], double f3[3])
{
    double e3[3], e2[3];
    cross3(f1, f2, e3);
    double n = sqrt(sumsq3(e3));
    e3[0] /= n; e3[1] /= n; e3[2] /= n;
    cross3(e3, f1, e2);
    for (int j = 0; j < 3; ++j) { T[0][j] = f1[j]; T[1][j] = e2[j]; T[2][j] = e3[j]; }
    mat_rows_vec(T, f3_in, f3);
}

/* p3p.m:41-217 */
void vo_p3p(const double world[9], const double bearings[9], double poses[48])
{
    memset(poses, 0, 48 * sizeof(double));

    double P1[3], P2[3], P3[3], f1[3], f2[3], f3[3];
    for (int j = 0; j < 3; ++j) {
        P1[j] = world[j]; P2[j] = world[3 + j]; P3[j] = world[6 + j];
    }
    double v1[3], v2[3], cr[3];
    for (int j = 0; j < 3; ++j) { v1[j] = P2[j] - P1[j]; v2[j] = P3[j] - P1[j]; }
    cross3(v1, v2, cr);
    if (sqrt(sumsq3(cr)) == 0.0) return;

    for (int j = 0; j < 3; ++j) { f1[j] = bearings[j]; f2[j] = bearings[3 + j]; }
    double T[3][3];
    camera_frame(f1, f2, bearings + 6, T, f3);

    if (f3[2] > 0.0) {
        for (int j = 0; j < 3; ++j) {
            f1[j] = bearings[3 + j]; f2[j] = bearings[j];
            P1[j] = world[3 + j]; P2[j] = world[j]; P3[j] = world[6 + j];
        }
        camera_frame(f1, f2, bearings + 6, T, f3);
    }

    /* intermediate world frame, p3p.m:102-110 */
    double n1[3], n2[3], n3[3], d31[3];
    for (int j = 0; j < 3; ++j) { n1[j] = P2[j] - P1[j]; d31[j] = P3[j] - P1[j]; }
    double nn1 = sqrt(sumsq3(n1));
    n1[0] /= nn1; n1[1] /= nn1; n1[2] /= nn1;
    cross3(n1, d31, n3);
    double nn3 = sqrt(sumsq3(n3));
    n3[0] /= nn3; n3[1] /= nn3; n3[2] /= nn3;
    cross3(n3, n1, n2);
    double N[3][3];
    for (int j = 0; j < 3; ++j) { N[0][j] = n1[j]; N[1][j] = n2[j]; N[2][j] = n3[j]; }

    /* known parameters, p3p.m:114-129 */
    double P3n[3];
    mat_rows_vec(N, d31, P3n);
    double d21[3] = { P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2] };
    double d_12 = sqrt(d21[0] * d21[0] + d21[1] * d21[1] + d21[2] * d21[2]);
    double f_1 = f3[0] / f3[2];
    double f_2 = f3[1] / f3[2];
    double p_1 = P3n[0];
    double p_2 = P3n[1];
    double cos_beta = f1[0] * f2[0] + f1[1] * f2[1] + f1[2] * f2[2];
    double b = 1.0 / (1.0 - cos_beta * cos_beta) - 1.0;
    /* sqrt of a negative b would be complex in MATLAB; b >= 0 whenever |cos_beta| <= 1,
     * rounding can only make it a tiny negative, whose complex root has real part 0. */
    double sb = (b < 0.0) ? 0.0 : sqrt(b);
    b = (cos_beta < 0.0) ? -sb : sb;

    double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
    double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
    double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
    double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;

    /* quartic coefficients, p3p.m:146-179 (left-to-right evaluation) */
    double fac[5];
    fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
    fac[1] = 2.0 * p_2_pw3 * d_12 * b + 2.0 * f_2_pw2 * p_2_pw3 * d_12 * b
           - 2.0 * f_2 * p_2_pw3 * f_1 * d_12;
    fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2
           - f_2_pw2 * p_2_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2
           + 2.0 * p_1 * p_2_pw2 * d_12 + 2.0 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b
           - p_2_pw2 * p_1_pw2 * f_1_pw2 + 2.0 * p_1 * p_2_pw2 * f_2_pw2 * d_12
           - p_2_pw2 * d_12_pw2 * b_pw2 - 2.0 * p_1_pw2 * p_2_pw2;
    fac[3] = 2.0 * p_1_pw2 * p_2 * d_12 * b + 2.0 * f_2 * p_2_pw3 * f_1 * d_12
           - 2.0 * f_2_pw2 * p_2_pw3 * d_12 * b - 2.0 * p_1 * p_2 * d_12_pw2 * b;
    fac[4] = -2.0 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2
           + 2.0 * p_1_pw3 * d_12 - p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2
           - p_1_pw4 - 2.0 * f_2_pw2 * p_2_pw2 * p_1 * d_12 + p_2_pw2 * f_1_pw2 * p_1_pw2
           + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;

    double xr[4], xi[4];
    vo_solve_quartic(fac, xr, xi);

    /* back-substitution, p3p.m:187-215 */
    for (int i = 0; i < 4; ++i) {
        double x = xr[i];
        double cot_alpha = (-f_1 * p_1 / f_2 - x * p_2 + d_12 * b)
                         / (-f_1 * x * p_2 / f_2 + p_1 - d_12);
        double cos_theta = x;
        double st2 = 1.0 - x * x;
        /* sqrt(negative) is purely imaginary; its real part, the only part that
         * survives ransacLocalization.m:64-65, is zero. */
        double sin_theta = (st2 < 0.0) ? 0.0 : sqrt(st2);
        double sin_alpha = sqrt(1.0 / (cot_alpha * cot_alpha + 1.0));
        double ca2 = 1.0 - sin_alpha * sin_alpha;
        double cos_alpha = (ca2 < 0.0) ? 0.0 : sqrt(ca2);
        if (cot_alpha < 0.0) cos_alpha = -cos_alpha;

        double Cv[3];
        Cv[0] = d_12 * cos_alpha * (sin_alpha * b + cos_alpha);
        Cv[1] = cos_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha);
        Cv[2] = sin_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha);

        /* C = P1 + N'*C */
        double Cw[3];
        for (int r = 0; r < 3; ++r)
            Cw[r] = P1[r] + (N[0][r] * Cv[0] + N[1][r] * Cv[1] + N[2][r] * Cv[2]);

        double Rm[3][3] = {
            { -cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta },
            {  sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta },
            {  0.0,       -sin_theta,              cos_theta } };

        /* R = (N' * R') * T */
        double NR[3][3], Rw[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                NR[r][c] = N[0][r] * Rm[c][0] + N[1][r] * Rm[c][1] + N[2][r] * Rm[c][2];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rw[r][c] = NR[r][0] * T[0][c] + NR[r][1] * T[1][c] + NR[r][2] * T[2][c];

        double *out = poses + 12 * i; /* 3x4 block, column-major */
        for (int r = 0; r < 3; ++r) {
            out[r] = Cw[r];
            for (int c = 0; c < 3; ++c) out[3 * (c + 1) + r] = Rw[r][c];
        }
    }
}

/* projectPoints.m:11-15 + distortPoints.m:4-16, D = 0 */
void vo_project_points(const double *p, int n, const double K[9], double *out)
{
    const double k1 = 0.0, k2 = 0.0;
    double u0 = K[6], v0 = K[7];
    for (int j = 0; j < n; ++j) {
        const double *q = p + 3 * j;
        double h0 = K[0] * q[0] + K[3] * q[1] + K[6] * q[2];
        double h1 = K[1] * q[0] + K[4] * q[1] + K[7] * q[2];
        double h2 = K[2] * q[0] + K[5] * q[1] + K[8] * q[2];
        double x = h0 / h2, y = h1 / h2;
        double xp = x - u0, yp = y - v0;
        double r2 = xp * xp + yp * yp;
        double s = 1.0 + k1 * r2 + k2 * (r2 * r2);
        out[2 * j] = u0 + xp * s;
        out[2 * j + 1] = v0 + yp * s;
    }
}

/* K \ b for the upper-triangular K of a pin-hole camera (back substitution). */
static void k_backslash(const double K[9], const double b[3], double x[3])
{
    x[2] = b[2] / K[8];
    x[1] = (b[1] - K[7] * x[2]) / K[4];
    x[0] = (b[0] - K[3] * x[1] - K[6] * x[2]) / K[0];
}

static int score_pose(const double *kp_uv, const double *X, int n, const double K[9],
                      const double R[9], const double t[3], uint8_t *mask)
{
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
        const double *x = X + 3 * j;
        double pc[3], uv[2];
        for (int r = 0; r < 3; ++r)
            pc[r] = (R[r] * x[0] + R[3 + r] * x[1] + R[6 + r] * x[2]) + t[r];
        vo_project_points(pc, 1, K, uv);
        double du = kp_uv[2 * j] - uv[0];
        double dv = kp_uv[2 * j + 1] - uv[1];
        double err = du * du + dv * dv;
        int in = err < 100.0; /* pixel_tolerance^2, ransacLocalization.m:19,83 */
        if (mask) mask[j] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}

/* ransacLocalization.m:46-101 for one sample triple */
void vo_ransac_hypothesis(const double *kp_uv, const double *X, int n, const double K[9],
                          const int32_t idx3[3], int counts[4], int *best_sol,
                          double Rt[12], uint8_t *mask)
{
    double world[9], bearings[9], poses[48];
    for (int c = 0; c < 3; ++c) {
        int id = idx3[c];
        double hom[3] = { kp_uv[2 * id], kp_uv[2 * id + 1], 1.0 }, nb[3];
        k_backslash(K, hom, nb);
        double nrm = sqrt(nb[0] * nb[0] + nb[1] * nb[1] + nb[2] * nb[2]);
        for (int r = 0; r < 3; ++r) {
            bearings[3 * c + r] = nb[r] / nrm;
            world[3 * c + r] = X[3 * id + r];
        }
    }
    vo_p3p(world, bearings, poses);

    double Rs[4][9], ts[4][3];
    for (int s = 0; s < 4; ++s) {
        const double *C = poses + 12 * s;     /* t_W_C */
        const double *Rwc = poses + 12 * s + 3; /* R_W_C col-major */
        /* R_C_W = R_W_C' */
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Rs[s][3 * c + r] = Rwc[3 * r + c];
        for (int r = 0; r < 3; ++r)
            ts[s][r] = (-Rs[s][r]) * C[0] + (-Rs[s][3 + r]) * C[1] + (-Rs[s][6 + r]) * C[2];
    }
    int best = 0;
    for (int s = 0; s < 4; ++s) {
        counts[s] = score_pose(kp_uv, X, n, K, Rs[s], ts[s], NULL);
        if (s > 0 && counts[s] > counts[best]) best = s;
    }
    *best_sol = best;
    memcpy(Rt, Rs[best], 9 * sizeof(double));
    memcpy(Rt + 9, ts[best], 3 * sizeof(double));
    if (mask) score_pose(kp_uv, X, n, K, Rs[best], ts[best], mask);
}

/* ransacLocalization.m:118-129 */
double vo_ransac_num_iterations(int max_num_inliers, int n)
{
    const double confidence = 0.9999, upper = 0.95;
    double outlier_ratio = 1.0 - (double)max_num_inliers / (double)n;
    if (upper < outlier_ratio) outlier_ratio = upper;
    double it = log(1.0 - confidence) / log(1.0 - pow(1.0 - outlier_ratio, 3.0));
    if (it > 15000.0) it = 15000.0;
    if (it < 100.0) it = 100.0;
    return it;
}

/* ransacLocalization.m:1-157 */
int vo_ransac_localization(const double *kp_rc, const double *X, int n, const double K[9],
                           const int32_t *samples, int hcap,
                           double R_C_W[9], double t_C_W[3], uint8_t *best_mask, int *found,
                           double *max_hist, double *iter_hist, int32_t *per_hyp_counts)
{
    /* (row, col) -> (u, v), :34 */
    double *kp_uv = (double *)__builtin_malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    uint8_t *mask = (uint8_t *)__builtin_malloc((size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; ++j) { kp_uv[2 * j] = kp_rc[2 * j + 1]; kp_uv[2 * j + 1] = kp_rc[2 * j]; }
    memset(best_mask, 0, (size_t)n);
    *found = 0;

    int max_num_inliers = 0;
    double num_iterations = INFINITY;
    int i = 1, status = 0;
    while (num_iterations > (double)i) {
        if (i > hcap) { status = -1; break; }
        int counts[4], best;
        double Rt[12];
        vo_ransac_hypothesis(kp_uv, X, n, K, samples + 3 * (i - 1), counts, &best, Rt, mask);
        int c = counts[best];
        if (per_hyp_counts) per_hyp_counts[i - 1] = c;
        if (c > max_num_inliers && c >= 30) { /* :104,109-110 */
            max_num_inliers = c;
            memcpy(best_mask, mask, (size_t)n);
            memcpy(R_C_W, Rt, 9 * sizeof(double));
            memcpy(t_C_W, Rt + 9, 3 * sizeof(double));
            *found = 1;
        }
        num_iterations = vo_ransac_num_iterations(max_num_inliers, n);
        iter_hist[i - 1] = num_iterations;
        max_hist[i - 1] = (double)max_num_inliers;
        ++i;
    }
    __builtin_free(kp_uv);
    __builtin_free(mask);
    return status < 0 ? -1 : i - 1;
}
