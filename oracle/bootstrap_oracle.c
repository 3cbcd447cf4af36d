/*
 * bootstrap_oracle.c -- CPU statement of the relative-pose step of the bootstrap.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
 *
 * Stands in for [orientation, location] = relativeCameraPose(E, cameraParams, pts1, pts2) at
 * Initialization/pose_estimation_init.m:29-33.  The toolbox function is closed source and
 * MATLAB is absent, so PARITY IS UNPINNED against the toolbox; the estimator is defined here
 * (the course helpers utils/exe6_8point/decomposeEssentialMatrix.m:12-32 and
 * disambiguateRelativePose.m:23-50 show the intended algorithm):
 *   1. E = U S V' (one-sided Jacobi on the columns of E; columns ordered by decreasing singular
 *      value; u3 = u1 x u2, v3 = v1 x v2 so that det U = det V = +1);
 *   2. R1 = U W V', R2 = U W' V' with W = [0 -1 0; 1 0 0; 0 0 1]; translation candidates -u3, +u3;
 *      candidate order (R1,-u3), (R1,+u3), (R2,-u3), (R2,+u3);
 *   3. every pair is triangulated with the DLT of linearTriangulation.m using M1 = K [I | 0],
 *      M2 = K [R | t]; a point is "behind" when its depth is negative in either camera;
 *   4. the candidate with the fewest points behind wins (first minimum); valid fraction =
 *      1 - behind / n;
 *   5. outputs follow the toolbox convention: orientation = R_C2_C1 (as a plain matrix),
 *      location = -R_C2_C1' t, |t| = 1.
 */
#include <math.h>
#include <string.h>

#include "vo_oracle.h"

static void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* E column-major.  R[2][9] row-major, u3[3].  Returns 0 when E has fewer than two non-zero
 * singular values. */
int vo_decompose_essential(const double E[9], double R[2][9], double u3[3])
{
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = E[3 * c + r];
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 3; ++r) {
                    al += A[r][i] * A[r][i];
                    be += A[r][j] * A[r][j];
                    ga += A[r][i] * A[r][j];
                }
                if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    double ai = A[r][i], aj = A[r][j];
                    A[r][i] = c * ai - s * aj;
                    A[r][j] = s * ai + c * aj;
                    double vi = V[r][i], vj = V[r][j];
                    V[r][i] = c * vi - s * vj;
                    V[r][j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    double nn[3];
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) nn[c] = A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2 - a; ++b)
            if (nn[ord[b + 1]] > nn[ord[b]]) {
                int t = ord[b];
                ord[b] = ord[b + 1];
                ord[b + 1] = t;
            }
    double s1 = sqrt(nn[ord[0]]), s2 = sqrt(nn[ord[1]]);
    if (!(s1 > 0.0 && s2 > 1e-12 * s1)) return 0;
    double u1[3], u2[3], v1[3], v2[3], v3[3];
    for (int r = 0; r < 3; ++r) {
        u1[r] = A[r][ord[0]] / s1;
        u2[r] = A[r][ord[1]] / s2;
        v1[r] = V[r][ord[0]];
        v2[r] = V[r][ord[1]];
    }
    cross3(u1, u2, u3);
    cross3(v1, v2, v3);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            R[0][3 * r + c] = u2[r] * v1[c] - u1[r] * v2[c] + u3[r] * v3[c];
            R[1][3 * r + c] = -u2[r] * v1[c] + u1[r] * v2[c] + u3[r] * v3[c];
        }
    return 1;
}

/* E, K column-major; p1, p2 2 x n (x; y).  orientation column-major 3x3, location[3].
 * Returns 1, or 0 when E cannot be decomposed. */
int vo_relative_pose(const double E[9], const double K[9], const double *p1, const double *p2, int n,
                     double orientation[9], double location[3], double *valid_fraction)
{
    double R[2][9], u3[3];
    if (!vo_decompose_essential(E, R, u3)) return 0;
    double M1[12];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r) M1[3 * c + r] = c < 3 ? K[3 * c + r] : 0.0;
    int best = 0, best_neg = 0x7fffffff;
    double best_Rt[12];
    for (int k = 0; k < 4; ++k) {
        const double *Rk = R[k >> 1];
        double sg = (k & 1) ? 1.0 : -1.0;
        double Rt[12], M2[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Rt[3 * c + r] = Rk[3 * r + c];
            Rt[9 + r] = sg * u3[r];
        }
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 3; ++r)
                M2[3 * c + r] = K[r] * Rt[3 * c] + K[3 + r] * Rt[3 * c + 1] + K[6 + r] * Rt[3 * c + 2];
        int neg = 0;
        for (int i = 0; i < n; ++i) {
            double a[3] = {p1[2 * i], p1[2 * i + 1], 1.0}, b[3] = {p2[2 * i], p2[2 * i + 1], 1.0}, P[4];
            vo_triangulate_one(a, b, M1, M2, P);
            double z2 = Rt[2] * P[0] + Rt[5] * P[1] + Rt[8] * P[2] + Rt[11];
            neg += (P[2] < 0.0 || z2 < 0.0);
        }
        if (neg < best_neg) {
            best_neg = neg;
            best = k;
            memcpy(best_Rt, Rt, sizeof(Rt));
        }
    }
    (void)best;
    for (int q = 0; q < 9; ++q) orientation[q] = best_Rt[q];
    for (int r = 0; r < 3; ++r)
        location[r] = -(best_Rt[3 * r] * best_Rt[9] + best_Rt[3 * r + 1] * best_Rt[10] + best_Rt[3 * r + 2] * best_Rt[11]);
    if (valid_fraction) *valid_fraction = n > 0 ? 1.0 - (double)best_neg / (double)n : 0.0;
    return 1;
}
