/*
 * CPU restatement of the DLT pose branch of the reference's RANSAC localisation.
 *
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h): loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, never by the product.
 *
 * Follows utils/exe7_Ransac/estimatePoseDLT.m:1-60 and the use_p3p = false branch of
 * utils/exe7_Ransac/ransacLocalization.m (:20-24 k = 6, :69-74 hypothesis from 6 samples, :77-83 one pose
 * scored per iteration, :118-129 adaptive bound with exponent k, :145-151 refit on the inliers of the best
 * hypothesis).  The branch is dead in the reference (use_p3p = true at :9), so nothing in the reference
 * exercises it.  PARITY UNPINNED versus MATLAB: the two svd() calls (estimatePoseDLT.m:34, :50) are LAPACK's
 * in MATLAB and one-sided Jacobi rotations here; the tests pin this file against numpy.linalg.svd and
 * against exact correspondences of a known pose.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* One-sided (Hestenes) Jacobi on the m x 12 matrix A (row-major): on return the columns of A are mutually
 * orthogonal (A V0 = A_out V... the product of the rotations is accumulated in V, 12 x 12 row-major), so the
 * column of V that belongs to the column of A with the smallest norm is the right singular vector of the
 * smallest singular value -- V(:, end) of [~, ~, V] = svd(Q) at estimatePoseDLT.m:34-35. */
static void smallest_right_singular_vector(double *A, int m, double v_out[12])
{
    double V[12][12];
    for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) V[r][c] = r == c ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int i = 0; i < 11; ++i)
            for (int j = i + 1; j < 12; ++j) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int r = 0; r < m; ++r) {
                    const double ai = A[12 * (size_t)r + i], aj = A[12 * (size_t)r + j];
                    al += ai * ai;
                    be += aj * aj;
                    ga += ai * aj;
                }
                if (!(ga * ga > 1e-30 * (al * be))) continue;   /* |ga| > 1e-15 sqrt(al be) */
                rotated = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < m; ++r) {
                    const double ai = A[12 * (size_t)r + i], aj = A[12 * (size_t)r + j];
                    A[12 * (size_t)r + i] = c * ai - s * aj;
                    A[12 * (size_t)r + j] = s * ai + c * aj;
                }
                for (int r = 0; r < 12; ++r) {
                    const double vi = V[r][i], vj = V[r][j];
                    V[r][i] = c * vi - s * vj;
                    V[r][j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    int best = 0;
    double best_n = INFINITY;
    for (int c = 0; c < 12; ++c) {
        double nn = 0.0;
        for (int r = 0; r < m; ++r) nn += A[12 * (size_t)r + c] * A[12 * (size_t)r + c];
        if (nn < best_n) {
            best_n = nn;
            best = c;
        }
    }
    for (int r = 0; r < 12; ++r) v_out[r] = V[r][best];
}

/* [U, ~, V] = svd(R); R_tilde = U * V'  (estimatePoseDLT.m:50-51) for a 3 x 3 matrix, row-major in and out. */
static void closest_rotation(const double R[9], double Rt[9])
{
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = R[3 * r + c];
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int i = 0; i < 2; ++i)
            for (int j = i + 1; j < 3; ++j) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 3; ++r) {
                    al += A[r][i] * A[r][i];
                    be += A[r][j] * A[r][j];
                    ga += A[r][i] * A[r][j];
                }
                if (!(ga * ga > 1e-30 * (al * be))) continue;
                rotated = 1;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double ai = A[r][i], aj = A[r][j];
                    A[r][i] = c * ai - s * aj;
                    A[r][j] = s * ai + c * aj;
                    const double vi = V[r][i], vj = V[r][j];
                    V[r][i] = c * vi - s * vj;
                    V[r][j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    /* A = U S: U's columns are A's, normalised; U V' = sum_c u_c v_c' needs no ordering of the singular values */
    double U[3][3];
    for (int c = 0; c < 3; ++c) {
        const double nn = sqrt(A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c]);
        for (int r = 0; r < 3; ++r) U[r][c] = A[r][c] / nn;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[3 * r + c] = U[r][0] * V[c][0] + U[r][1] * V[c][1] + U[r][2] * V[c][2];
}

/* K \ [u; v; 1] for the upper-triangular K of a pin-hole camera (estimatePoseDLT.m:14), K column-major */
static void normalised(const double K[9], double u, double v, double out[2])
{
    const double x2 = 1.0 / K[8];
    const double x1 = (v - K[7] * x2) / K[4];
    const double x0 = (u - K[3] * x1 - K[6] * x2) / K[0];
    /* (the reference uses the first two components as they are: the third is 1 / K(3,3) = 1) */
    out[0] = x0;
    out[1] = x1;
}

/* M_tilde = estimatePoseDLT(p, P, K): p_uv n x 2 (u, v) and P n x 3 as rows, K column-major; M 3 x 4 row-major
 * [R_tilde | alpha t].  Returns 0, or -1 when n < 6. */
int vo_estimate_pose_dlt(const double *p_uv, const double *P, int n, const double K[9], double M[12])
{
    if (n < 6) return -1;
    double *Q = (double *)calloc((size_t)2 * n * 12, sizeof(double));   /* :17-18 */
    for (int i = 0; i < n; ++i) {                                        /* :20-32 */
        double uv[2];
        normalised(K, p_uv[2 * i], p_uv[2 * i + 1], uv);
        const double *X = P + 3 * i;
        double *r0 = Q + 12 * (size_t)(2 * i), *r1 = r0 + 12;
        for (int c = 0; c < 3; ++c) {
            r0[c] = X[c];
            r1[4 + c] = X[c];
            r0[8 + c] = -uv[0] * X[c];
            r1[8 + c] = -uv[1] * X[c];
        }
        r0[3] = 1.0;
        r1[7] = 1.0;
        r0[11] = -uv[0];
        r1[11] = -uv[1];
    }
    double m[12];
    smallest_right_singular_vector(Q, 2 * n, m);                        /* :34-35 */
    free(Q);
    /* reshape(M_tilde, 4, 3)': row r of M = entries 4r .. 4r + 3 (:37) */
    double R[9], t[3];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) R[3 * r + c] = m[4 * r + c];
        t[r] = m[4 * r + 3];
    }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                       R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0.0) {                                                     /* :41-43 */
        for (int q = 0; q < 9; ++q) R[q] = -R[q];
        for (int q = 0; q < 3; ++q) t[q] = -t[q];
    }
    double Rt[9];
    closest_rotation(R, Rt);                                             /* :49-51 */
    double nr = 0.0, nt = 0.0;
    for (int q = 0; q < 9; ++q) {
        nr += R[q] * R[q];
        nt += Rt[q] * Rt[q];
    }
    const double alpha = sqrt(nt) / sqrt(nr);                            /* :55 */
    for (int r = 0; r < 3; ++r) {                                        /* :58 */
        for (int c = 0; c < 3; ++c) M[4 * r + c] = Rt[3 * r + c];
        M[4 * r + 3] = alpha * t[r];
    }
    return 0;
}

/* projectPoints of R X + t against the keypoints, errors < pixel_tolerance^2 (ransacLocalization.m:77-83);
 * M row-major [R | t]. */
static int score_dlt(const double *kp_uv, const double *X, int n, const double K[9], const double M[12], uint8_t *mask)
{
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
        const double *x = X + 3 * j;
        double pc[3], uv[2];
        for (int r = 0; r < 3; ++r) pc[r] = (M[4 * r] * x[0] + M[4 * r + 1] * x[1] + M[4 * r + 2] * x[2]) + M[4 * r + 3];
        vo_project_points(pc, 1, K, uv);
        const double du = kp_uv[2 * j] - uv[0], dv = kp_uv[2 * j + 1] - uv[1];
        const int in = du * du + dv * dv < 100.0;
        if (mask) mask[j] = (uint8_t)in;
        cnt += in;
    }
    return cnt;
}

/* ransacLocalization.m:118-129 with k = 6 */
double vo_ransac_num_iterations_dlt(int max_num_inliers, int n)
{
    const double confidence = 0.9999, upper = 0.95;
    double outlier_ratio = 1.0 - (double)max_num_inliers / (double)n;
    if (upper < outlier_ratio) outlier_ratio = upper;
    double it = log(1.0 - confidence) / log(1.0 - pow(1.0 - outlier_ratio, 6.0));
    if (it > 15000.0) it = 15000.0;
    if (it < 100.0) it = 100.0;
    return it;
}

/* ransacLocalization.m:1-157 with use_p3p = false, tweaked_for_more = adaptive = true.
 * kp_rc n x 2 (row, col) rows; X n x 3 rows; samples hcap x 6 (0-based, 6 distinct indices each).
 * M_hyp (may be NULL): 12 doubles per executed iteration, the hypothesis' [R | t] row-major.
 * Returns the iterations executed, or -1 when the samples ran out before the loop ended. */
int vo_ransac_localization_dlt(const double *kp_rc, const double *X, int n, const double K[9], const int32_t *samples,
                               int hcap, double R_C_W[9], double t_C_W[3], uint8_t *best_mask, int *found,
                               double *max_hist, double *iter_hist, int32_t *per_hyp_counts, double *M_hyp)
{
    double *kp_uv = (double *)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    uint8_t *mask = (uint8_t *)malloc((size_t)(n > 0 ? n : 1));
    for (int j = 0; j < n; ++j) {   /* (row, col) -> (u, v), :34 */
        kp_uv[2 * j] = kp_rc[2 * j + 1];
        kp_uv[2 * j + 1] = kp_rc[2 * j];
    }
    memset(best_mask, 0, (size_t)n);
    *found = 0;
    int max_num_inliers = 0, status = 0, i = 1;
    double num_iterations = INFINITY;
    while (num_iterations > (double)i) {
        if (i > hcap) {
            status = -1;
            break;
        }
        double p6[12], P6[18], M[12];
        for (int c = 0; c < 6; ++c) {                                    /* :46-48 */
            const int id = samples[6 * (size_t)(i - 1) + c];
            p6[2 * c] = kp_uv[2 * id];
            p6[2 * c + 1] = kp_uv[2 * id + 1];
            memcpy(P6 + 3 * c, X + 3 * id, 3 * sizeof(double));
        }
        vo_estimate_pose_dlt(p6, P6, 6, K, M);                           /* :70-73 */
        if (M_hyp) memcpy(M_hyp + 12 * (size_t)(i - 1), M, sizeof(M));
        const int c = score_dlt(kp_uv, X, n, K, M, mask);                /* :77-83 */
        if (per_hyp_counts) per_hyp_counts[i - 1] = c;
        if (c > max_num_inliers && c >= 30) {                            /* :103-116 */
            max_num_inliers = c;
            memcpy(best_mask, mask, (size_t)n);
            *found = 1;
        }
        num_iterations = vo_ransac_num_iterations_dlt(max_num_inliers, n);
        iter_hist[i - 1] = num_iterations;
        max_hist[i - 1] = (double)max_num_inliers;
        ++i;
    }
    if (*found) {                                                        /* :145-151: refit on the inliers */
        double *p = (double *)malloc(sizeof(double) * 2 * (size_t)n), *P = (double *)malloc(sizeof(double) * 3 * (size_t)n);
        int m = 0;
        for (int j = 0; j < n; ++j)
            if (best_mask[j]) {
                p[2 * m] = kp_uv[2 * j];
                p[2 * m + 1] = kp_uv[2 * j + 1];
                memcpy(P + 3 * m, X + 3 * j, 3 * sizeof(double));
                ++m;
            }
        double M[12];
        vo_estimate_pose_dlt(p, P, m, K, M);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) R_C_W[3 * c + r] = M[4 * r + c];   /* column-major, like the P3P branch */
            t_C_W[r] = M[4 * r + 3];
        }
        free(p);
        free(P);
    }
    free(kp_uv);
    free(mask);
    return status < 0 ? -1 : i - 1;
}
