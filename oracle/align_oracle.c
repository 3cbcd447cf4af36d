/*
 * align_oracle.c -- CPU restatement of the trajectory alignment used by the evaluation:
 *   utils/exe9_BA/alignError.m:1-17                 residual of the similarity transform
 *   utils/exe9_BA/alignEstimateToGroundTruth.m:1-27  7-parameter non-linear least squares
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
 *
 * PARITY UNPINNED for the optimiser: the reference minimises with lsqnonlin (closed source,
 * finite-difference Jacobian).  The residual restates alignError.m exactly (through the oracle's
 * twist2HomogMatrix); the minimiser is this project's Levenberg-Marquardt (same rules as
 * vo_pose_refine) with a central-difference Jacobian, which also makes it an independent check
 * of the analytic Jacobian in the HIP kernel.  Pinned by the closed-form similarity alignment
 * (Umeyama) reaching the same minimum, in tests/test_oracle_eval.py.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* alignError.m:6-15.  x[7] = [twist; scale]; pp_G_C, p_V_C 3 x n column-major; error 3n. */
void vo_align_error(const double x[7], const double *pp_G_C, const double *p_V_C, int n, double *error)
{
    double T[16];
    vo_twist2homog(x, T);                                   /* :6 */
    const double s = x[6];                                  /* :7 */
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < 3; ++i) {
            const double *p = p_V_C + 3 * k;
            const double y = s * (T[i] * p[0] + T[4 + i] * p[1] + T[8 + i] * p[2]) + T[12 + i];   /* :10-11 */
            error[3 * k + i] = pp_G_C[3 * k + i] - y;       /* :13 */
        }
}

static int chol7(double *A, double *b)
{
    const int n = 7;
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= A[i * n + k] * b[k];
        b[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * b[k];
        b[i] = s / A[i * n + i];
    }
    return 1;
}

static double sumsq(const double *e, int m)
{
    double c = 0.0;
    for (int i = 0; i < m; ++i) c += e[i] * e[i];
    return c;
}

/* alignEstimateToGroundTruth.m:10-25.  x_out[7], p_G_C 3 x n (the aligned estimate),
 * stats = {accepted iterations, initial cost, final cost}. */
void vo_align_sim3(const double *pp_G_C, const double *p_V_C, int n, int max_iter, double ftol, double xtol,
                   double x_out[7], double *p_G_C, double *stats)
{
    const int m = 3 * n;
    double x[7] = {0, 0, 0, 0, 0, 0, 1};                    /* :10-13 identity twist, scale 1 */
    double *e = malloc(sizeof(double) * m), *ep = malloc(sizeof(double) * m), *em = malloc(sizeof(double) * m);
    double *J = malloc(sizeof(double) * m * 7);
    vo_align_error(x, pp_G_C, p_V_C, n, e);
    double cost = sumsq(e, m), lambda = 1e-3;
    const double cost0 = cost;
    int iter = 0, rejects = 0;
    while (iter < max_iter) {
        for (int a = 0; a < 7; ++a) {
            double xp[7], xm[7];
            const double h = 1e-6 * (1.0 + fabs(x[a]));
            memcpy(xp, x, sizeof(xp));
            memcpy(xm, x, sizeof(xm));
            xp[a] += h;
            xm[a] -= h;
            vo_align_error(xp, pp_G_C, p_V_C, n, ep);
            vo_align_error(xm, pp_G_C, p_V_C, n, em);
            for (int i = 0; i < m; ++i) J[7 * i + a] = (ep[i] - em[i]) / (2.0 * h);
        }
        double H[49], g[7];
        memset(H, 0, sizeof(H));
        memset(g, 0, sizeof(g));
        for (int i = 0; i < m; ++i)
            for (int a = 0; a < 7; ++a) {
                g[a] += J[7 * i + a] * e[i];
                for (int b = 0; b < 7; ++b) H[7 * a + b] += J[7 * i + a] * J[7 * i + b];
            }
        int stop = 0, accepted = 0;
        while (!accepted && !stop) {
            double Hd[49], d[7];
            memcpy(Hd, H, sizeof(H));
            for (int a = 0; a < 7; ++a) {
                Hd[8 * a] *= 1.0 + lambda;
                d[a] = -g[a];
            }
            if (!chol7(Hd, d)) {
                lambda *= 10.0;
                if (++rejects > 10) stop = 1;
                continue;
            }
            double xn[7], dn = 0.0, xnorm = 0.0;
            for (int a = 0; a < 7; ++a) {
                xn[a] = x[a] + d[a];
                dn += d[a] * d[a];
                xnorm += x[a] * x[a];
            }
            vo_align_error(xn, pp_G_C, p_V_C, n, ep);
            const double cn = sumsq(ep, m);
            if (cn < cost) {
                const double dec = cost - cn, cost_old = cost;
                memcpy(x, xn, sizeof(x));
                memcpy(e, ep, sizeof(double) * m);
                cost = cn;
                lambda = lambda * 0.1 > 1e-12 ? lambda * 0.1 : 1e-12;
                ++iter;
                rejects = 0;
                accepted = 1;
                if (dec < ftol * (1.0 + cost_old)) stop = 1;
                if (sqrt(dn) < xtol * (1.0 + sqrt(xnorm))) stop = 1;
            } else {
                lambda *= 10.0;
                if (++rejects > 10) stop = 1;
            }
        }
        if (stop) break;
    }
    memcpy(x_out, x, sizeof(x));
    vo_align_error(x, pp_G_C, p_V_C, n, e);
    for (int i = 0; i < m; ++i) p_G_C[i] = pp_G_C[i] - e[i];   /* :20-25 */
    if (stats) {
        stats[0] = iter;
        stats[1] = cost0;
        stats[2] = cost;
    }
    free(e);
    free(ep);
    free(em);
    free(J);
}
