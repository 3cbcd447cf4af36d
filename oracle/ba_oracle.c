/*
 * ba_oracle.c -- CPU restatement of the reference's bundle adjustment and pose refinement:
 *   utils/exe9_BA/baError.m:2-65        reprojection residuals of the window
 *   utils/exe9_BA/runBA.m:2-88          sliding-window BA over F twists + m landmarks
 *   utils/T_refinement.m:1-21           pose-only refinement
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).
 *
 * PARITY UNPINNED for the optimiser: the reference minimises with lsqnonlin (closed source,
 * trust-region-reflective, finite-difference Jacobians).  The residual function below restates
 * baError.m exactly; the minimiser is this project's own Levenberg-Marquardt:
 *   - unknowns as in the reference: raw twist coordinates [v; w] of every T_W_C, and the
 *     landmark coordinates; landmarks with fewer than 4 residual rows are held fixed
 *     (runBA.m:45-49); no gauge fixing (as in the reference);
 *   - analytic Jacobian with respect to the raw twist coordinates;
 *   - step from (J'J + lambda diag(J'J)) d = -J'r through the Schur complement over landmarks;
 *   - lambda0 = 1e-3, accept (cost decreases): lambda = max(lambda/10, 1e-12), reject:
 *     lambda *= 10, stop after 10 consecutive rejects;
 *   - stop when max_iter steps were accepted, or an accepted step lowers the cost by less than
 *     ftol (1 + cost), or is shorter than xtol (1 + |x|).
 * It is pinned by known-answer tests (perturbed ground truth converges back to the noise floor)
 * and by scipy.optimize.least_squares reaching the same final cost.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- shared small helpers ------------------------------------------------------------- */

static void coeffs(double th2, double *A, double *B, double *C, double *B1, double *C1)
{
    if (th2 < 1e-2) {
        *A = 1.0 + th2 * (-1.0 / 6.0 + th2 * (1.0 / 120.0 + th2 * (-1.0 / 5040.0 + th2 * (1.0 / 362880.0 - th2 / 39916800.0))));
        *B = 0.5 + th2 * (-1.0 / 24.0 + th2 * (1.0 / 720.0 + th2 * (-1.0 / 40320.0 + th2 * (1.0 / 3628800.0 - th2 / 479001600.0))));
        *C = 1.0 / 6.0 + th2 * (-1.0 / 120.0 + th2 * (1.0 / 5040.0 + th2 * (-1.0 / 362880.0 + th2 * (1.0 / 39916800.0 - th2 / 6227020800.0))));
        /* B1 = (A - 2B)/th^2, C1 = (B - 3C)/th^2 */
        *B1 = -1.0 / 12.0 + th2 * (1.0 / 180.0 + th2 * (-1.0 / 6720.0 + th2 * (1.0 / 453600.0 - th2 / 47900160.0)));
        *C1 = -1.0 / 60.0 + th2 * (1.0 / 1260.0 + th2 * (-1.0 / 60480.0 + th2 * (1.0 / 4989600.0 - th2 / 622702080.0)));
    } else {
        double th = sqrt(th2);
        *A = sin(th) / th;
        *B = (1.0 - cos(th)) / th2;
        *C = (th - sin(th)) / (th2 * th);
        *B1 = (*A - 2.0 * *B) / th2;
        *C1 = (*B - 3.0 * *C) / th2;
    }
}

static void cross(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

/* Per-frame quantities for residuals and Jacobians with respect to the twist [v; w] of T_W_C:
 *   p_C = Rcw X + tcw
 *   d p_C / d v_k = Mv[:,k]                       (Mv = -R' V)
 *   d p_C / d w_k = p_C x Jr[:,k] + Dw[:,k]        (Jr = I - B w^ + C w^2, Dw[:,k] = -R' (dV/dw_k) v)
 */
typedef struct { double Rcw[3][3], tcw[3], Mv[3][3], Jr[3][3], Dw[3][3]; } frame_lin;

static void frame_prepare(const double tw[6], frame_lin *f)
{
    const double *v = tw, *w = tw + 3;
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B, C, B1, C1;
    coeffs(th2, &A, &B, &C, &B1, &C1);
    double W[3][3] = { { 0, -w[2], w[1] }, { w[2], 0, -w[0] }, { -w[1], w[0], 0 } };
    double W2[3][3], R[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            W2[i][j] = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double I = i == j ? 1.0 : 0.0;
            R[i][j] = I + A * W[i][j] + B * W2[i][j];
            V[i][j] = I + B * W[i][j] + C * W2[i][j];
            f->Jr[i][j] = I - B * W[i][j] + C * W2[i][j];
        }
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = V[i][0] * v[0] + V[i][1] * v[1] + V[i][2] * v[2];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            f->Rcw[i][j] = R[j][i];
            f->Mv[i][j] = -(R[0][i] * V[0][j] + R[1][i] * V[1][j] + R[2][i] * V[2][j]);
        }
        f->tcw[i] = -(R[0][i] * t[0] + R[1][i] * t[1] + R[2][i] * t[2]);
    }
    double wxv[3], wwxv[3];
    cross(w, v, wxv);
    cross(w, wxv, wwxv);
    for (int k = 0; k < 3; ++k) {
        double e[3] = { 0, 0, 0 }, exv[3], exwxv[3], wxexv[3], dV[3];
        e[k] = 1.0;
        cross(e, v, exv);
        cross(e, wxv, exwxv);
        cross(w, exv, wxexv);
        for (int i = 0; i < 3; ++i)
            dV[i] = B1 * w[k] * wxv[i] + B * exv[i] + C1 * w[k] * wwxv[i] + C * (exwxv[i] + wxexv[i]);
        for (int i = 0; i < 3; ++i) f->Dw[i][k] = -(R[0][i] * dV[0] + R[1][i] * dV[1] + R[2][i] * dV[2]);
    }
}

/* residual (kp - projection) of one observation and, optionally, its Jacobians.
 * Jp 2x6 (twist of the frame), Jl 2x3 (landmark).  K column-major. */
static void observe(const frame_lin *f, const double K[9], const double X[3], double u, double v, double r[2],
                    double Jp[2][6], double Jl[2][3])
{
    double p[3], h[3];
    for (int i = 0; i < 3; ++i) p[i] = f->Rcw[i][0] * X[0] + f->Rcw[i][1] * X[1] + f->Rcw[i][2] * X[2] + f->tcw[i];
    for (int i = 0; i < 3; ++i) h[i] = K[i] * p[0] + K[3 + i] * p[1] + K[6 + i] * p[2];
    double iz = 1.0 / h[2];
    double pu = h[0] * iz, pv = h[1] * iz;
    r[0] = u - pu;
    r[1] = v - pv;
    if (!Jp) return;
    /* d(pu,pv)/dp = (K_row0 - pu K_row2, K_row1 - pv K_row2) / h2 ; residual = kp - proj */
    double D[2][3];
    for (int c = 0; c < 3; ++c) {
        D[0][c] = -(K[3 * c] - pu * K[3 * c + 2]) * iz;
        D[1][c] = -(K[3 * c + 1] - pv * K[3 * c + 2]) * iz;
    }
    for (int k = 0; k < 3; ++k) {
        double dv[3] = { f->Mv[0][k], f->Mv[1][k], f->Mv[2][k] };
        double jr[3] = { f->Jr[0][k], f->Jr[1][k], f->Jr[2][k] }, dw[3];
        cross(p, jr, dw);
        for (int i = 0; i < 3; ++i) dw[i] += f->Dw[i][k];
        for (int a = 0; a < 2; ++a) {
            Jp[a][k] = D[a][0] * dv[0] + D[a][1] * dv[1] + D[a][2] * dv[2];
            Jp[a][3 + k] = D[a][0] * dw[0] + D[a][1] * dw[1] + D[a][2] * dw[2];
        }
    }
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 3; ++c)
            Jl[a][c] = D[a][0] * f->Rcw[0][c] + D[a][1] * f->Rcw[1][c] + D[a][2] * f->Rcw[2][c];
}

/* ---- observation vector of the reference ------------------------------------------------
 * observations = [F; m; for each frame: n_i; n_i x (row, col) interleaved; n_i landmark indices (1-based)]
 * (VO_bundle_adjust.m:47-50, baError.m:16-27). */
typedef struct { int F, m, nobs; int *of, *ol; double *ou, *ov; } obs_t;

static int parse_obs(const double *obs, int len, obs_t *o)
{
    if (len < 2) return -1;
    o->F = (int)obs[0];
    o->m = (int)obs[1];
    int total = 0, pos = 2;
    for (int f = 0; f < o->F; ++f) {
        if (pos >= len) return -1;
        int n = (int)obs[pos];
        total += n;
        pos += 1 + 3 * n;
    }
    if (pos != len) return -1;
    o->nobs = total;
    o->of = (int *)malloc(sizeof(int) * 2 * (size_t)(total + 1));
    o->ol = o->of + total + 1;
    o->ou = (double *)malloc(sizeof(double) * 2 * (size_t)(total + 1));
    o->ov = o->ou + total + 1;
    int k = 0;
    pos = 2;
    for (int f = 0; f < o->F; ++f) {
        int n = (int)obs[pos];
        for (int i = 0; i < n; ++i, ++k) {
            o->of[k] = f;
            o->ov[k] = obs[pos + 1 + 2 * i];     /* row */
            o->ou[k] = obs[pos + 1 + 2 * i + 1]; /* col */
            o->ol[k] = (int)obs[pos + 1 + 2 * n + i] - 1;
        }
        pos += 1 + 3 * n;
    }
    return 0;
}
static void free_obs(obs_t *o) { free(o->of); free(o->ou); }

/* baError.m:2-65.  r_out 2 x Nobs.  Returns Nobs or -1 on a malformed observation vector. */
int vo_ba_error(const double *state, const double *obs, int obs_len, const double K[9], double *r_out)
{
    obs_t o;
    if (parse_obs(obs, obs_len, &o)) return -1;
    frame_lin *fl = (frame_lin *)malloc(sizeof(frame_lin) * (size_t)(o.F > 0 ? o.F : 1));
    for (int f = 0; f < o.F; ++f) frame_prepare(state + 6 * f, fl + f);
    const double *X = state + 6 * o.F;
    for (int k = 0; k < o.nobs; ++k)
        observe(fl + o.of[k], K, X + 3 * o.ol[k], o.ou[k], o.ov[k], r_out + 2 * k, NULL, NULL);
    int n = o.nobs;
    free(fl);
    free_obs(&o);
    return n;
}

static double total_cost(const double *state, const obs_t *o, const double K[9], frame_lin *fl)
{
    for (int f = 0; f < o->F; ++f) frame_prepare(state + 6 * f, fl + f);
    const double *X = state + 6 * o->F;
    double c = 0.0;
    for (int k = 0; k < o->nobs; ++k) {
        double r[2];
        observe(fl + o->of[k], K, X + 3 * o->ol[k], o->ou[k], o->ov[k], r, NULL, NULL);
        c += r[0] * r[0] + r[1] * r[1];
    }
    return c;
}

/* dense Cholesky solve A x = b (A n x n SPD, row-major, destroyed).  Returns 0 if not PD. */
static int chol_solve(double *A, double *b, int n)
{
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

static int inv3_sym(const double V[3][3], double E[3][3])
{
    double c00 = V[1][1] * V[2][2] - V[1][2] * V[2][1];
    double c01 = V[1][2] * V[2][0] - V[1][0] * V[2][2];
    double c02 = V[1][0] * V[2][1] - V[1][1] * V[2][0];
    double det = V[0][0] * c00 + V[0][1] * c01 + V[0][2] * c02;
    if (!(det > 0.0)) return 0;
    double id = 1.0 / det;
    E[0][0] = c00 * id;
    E[0][1] = E[1][0] = c01 * id;
    E[0][2] = E[2][0] = c02 * id;
    E[1][1] = (V[0][0] * V[2][2] - V[0][2] * V[2][0]) * id;
    E[1][2] = E[2][1] = (V[0][2] * V[1][0] - V[0][0] * V[1][2]) * id;
    E[2][2] = (V[0][0] * V[1][1] - V[0][1] * V[1][0]) * id;
    return 1;
}

/* runBA.m:2-88 with this project's LM (see header).  state in/out (6F + 3m).
 * stats[0] = accepted iterations, [1] = initial cost, [2] = final cost, [3] = final lambda.
 * Returns 0, or -1 on a malformed observation vector. */
int vo_run_ba(double *state, const double *obs, int obs_len, const double K[9], int max_iter, double ftol,
              double xtol, double *stats)
{
    obs_t o;
    if (parse_obs(obs, obs_len, &o)) return -1;
    const int F = o.F, m = o.m, np = 6 * F, nx = np + 3 * m;
    frame_lin *fl = (frame_lin *)malloc(sizeof(frame_lin) * (size_t)(F > 0 ? F : 1));
    int *cnt = (int *)calloc((size_t)(m > 0 ? m : 1), sizeof(int));
    for (int k = 0; k < o.nobs; ++k) cnt[o.ol[k]]++;
    /* frozen: fewer than 4 residual rows = fewer than 2 observations, runBA.m:45-49 */
    double *S = (double *)malloc(sizeof(double) * (size_t)np * np);
    double *g = (double *)malloc(sizeof(double) * (size_t)np);
    double *Vm = (double *)calloc((size_t)(m > 0 ? m : 1) * 9, sizeof(double));
    double *gl = (double *)calloc((size_t)(m > 0 ? m : 1) * 3, sizeof(double));
    double *Em = (double *)calloc((size_t)(m > 0 ? m : 1) * 9, sizeof(double));
    double *Wb = (double *)malloc(sizeof(double) * 18 * (size_t)(o.nobs > 0 ? o.nobs : 1));
    double *xn = (double *)malloc(sizeof(double) * (size_t)nx);
    double *delta = (double *)malloc(sizeof(double) * (size_t)nx);
    double *U = (double *)malloc(sizeof(double) * (size_t)F * 36);

    double cost = total_cost(state, &o, K, fl), cost0 = cost, lambda = 1e-3;
    int iter = 0, rejects = 0;
    while (iter < max_iter) {
        /* linearise */
        for (int f = 0; f < F; ++f) frame_prepare(state + 6 * f, fl + f);
        memset(U, 0, sizeof(double) * (size_t)F * 36);
        memset(g, 0, sizeof(double) * (size_t)np);
        memset(Vm, 0, sizeof(double) * (size_t)m * 9);
        memset(gl, 0, sizeof(double) * (size_t)m * 3);
        const double *X = state + np;
        for (int k = 0; k < o.nobs; ++k) {
            int f = o.of[k], j = o.ol[k];
            double r[2], Jp[2][6], Jl[2][3];
            observe(fl + f, K, X + 3 * j, o.ou[k], o.ov[k], r, Jp, Jl);
            int frozen = cnt[j] < 2;
            for (int a = 0; a < 6; ++a) {
                g[6 * f + a] += Jp[0][a] * r[0] + Jp[1][a] * r[1];
                for (int b = 0; b < 6; ++b) U[36 * f + 6 * a + b] += Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b];
                for (int c = 0; c < 3; ++c)
                    Wb[18 * k + 3 * a + c] = frozen ? 0.0 : Jp[0][a] * Jl[0][c] + Jp[1][a] * Jl[1][c];
            }
            if (!frozen)
                for (int a = 0; a < 3; ++a) {
                    gl[3 * j + a] += Jl[0][a] * r[0] + Jl[1][a] * r[1];
                    for (int b = 0; b < 3; ++b) Vm[9 * j + 3 * a + b] += Jl[0][a] * Jl[0][b] + Jl[1][a] * Jl[1][b];
                }
        }
        /* reduced camera system S = U(1+lambda on diag) - sum_j W E W', rhs = -(g - W E gl) */
        memset(S, 0, sizeof(double) * (size_t)np * np);
        for (int f = 0; f < F; ++f)
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b)
                    S[(6 * f + a) * np + 6 * f + b] = U[36 * f + 6 * a + b] * (a == b ? 1.0 + lambda : 1.0);
        int solvable = 1;
        for (int j = 0; j < m; ++j) {
            if (cnt[j] < 2) continue;
            double V[3][3], E[3][3];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) V[a][b] = Vm[9 * j + 3 * a + b] * (a == b ? 1.0 + lambda : 1.0);
            if (!inv3_sym(V, E)) { solvable = 0; break; }
            memcpy(Em + 9 * j, E, sizeof(E));
        }
        if (solvable) {
            /* walk all pairs of observations of each landmark through the (landmark, frame) table */
            int *index = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1) * F);
            for (int q = 0; q < m * F; ++q) index[q] = -1;
            for (int k = 0; k < o.nobs; ++k) index[o.ol[k] * F + o.of[k]] = k;
            for (int k1 = 0; k1 < o.nobs; ++k1) {
                int j = o.ol[k1], f1 = o.of[k1];
                if (cnt[j] < 2) continue;
                const double *E = Em + 9 * j;
                double Y[6][3];
                for (int a = 0; a < 6; ++a)
                    for (int c = 0; c < 3; ++c)
                        Y[a][c] = Wb[18 * k1 + 3 * a] * E[c] + Wb[18 * k1 + 3 * a + 1] * E[3 + c] + Wb[18 * k1 + 3 * a + 2] * E[6 + c];
                for (int a = 0; a < 6; ++a)
                    g[6 * f1 + a] -= Y[a][0] * gl[3 * j] + Y[a][1] * gl[3 * j + 1] + Y[a][2] * gl[3 * j + 2];
                for (int f2 = 0; f2 < F; ++f2) {
                    int k2 = index[j * F + f2];
                    if (k2 < 0) continue;
                    for (int a = 0; a < 6; ++a)
                        for (int b = 0; b < 6; ++b)
                            S[(6 * f1 + a) * np + 6 * f2 + b] -=
                                Y[a][0] * Wb[18 * k2 + 3 * b] + Y[a][1] * Wb[18 * k2 + 3 * b + 1] + Y[a][2] * Wb[18 * k2 + 3 * b + 2];
                }
            }
            free(index);
            for (int i = 0; i < np; ++i) delta[i] = -g[i];
            solvable = chol_solve(S, delta, np);
        }
        if (!solvable) {
            lambda *= 10.0;
            if (++rejects > 10) break;
            continue;
        }
        /* back-substitute landmarks: d_j = -E (gl + W' d_p) */
        for (int j = 0; j < m; ++j) delta[np + 3 * j] = delta[np + 3 * j + 1] = delta[np + 3 * j + 2] = 0.0;
        double *acc = (double *)calloc((size_t)(m > 0 ? m : 1) * 3, sizeof(double));
        for (int k = 0; k < o.nobs; ++k) {
            int j = o.ol[k], f = o.of[k];
            if (cnt[j] < 2) continue;
            for (int c = 0; c < 3; ++c)
                for (int a = 0; a < 6; ++a) acc[3 * j + c] += Wb[18 * k + 3 * a + c] * delta[6 * f + a];
        }
        for (int j = 0; j < m; ++j) {
            if (cnt[j] < 2) continue;
            const double *E = Em + 9 * j;
            double t[3] = { gl[3 * j] + acc[3 * j], gl[3 * j + 1] + acc[3 * j + 1], gl[3 * j + 2] + acc[3 * j + 2] };
            for (int a = 0; a < 3; ++a) delta[np + 3 * j + a] = -(E[3 * a] * t[0] + E[3 * a + 1] * t[1] + E[3 * a + 2] * t[2]);
        }
        free(acc);
        double dn = 0.0, xnorm = 0.0;
        for (int i = 0; i < nx; ++i) {
            xn[i] = state[i] + delta[i];
            dn += delta[i] * delta[i];
            xnorm += state[i] * state[i];
        }
        double cost_new = total_cost(xn, &o, K, fl);
        if (cost_new < cost) {
            double dec = cost - cost_new, cost_old = cost;
            memcpy(state, xn, sizeof(double) * (size_t)nx);
            cost = cost_new;
            lambda = lambda * 0.1 > 1e-12 ? lambda * 0.1 : 1e-12;
            ++iter;
            rejects = 0;
            if (dec < ftol * (1.0 + cost_old)) break;
            if (sqrt(dn) < xtol * (1.0 + sqrt(xnorm))) break;
        } else {
            lambda *= 10.0;
            if (++rejects > 10) break;
        }
    }
    if (stats) { stats[0] = iter; stats[1] = cost0; stats[2] = cost; stats[3] = lambda; }
    free(fl); free(cnt); free(S); free(g); free(Vm); free(gl); free(Em); free(Wb); free(xn); free(delta); free(U);
    free_obs(&o);
    return 0;
}

/* T_refinement.m:1-21: pose-only LM over the twist of T_W_C; residual = keypoints - K T_C_W X
 * (:9-21).  kp_uv 2xN (u, v), X 3xN.  T 4x4 column-major in/out.
 * stats[0] = accepted iterations, [1] = initial cost, [2] = final cost. */
void vo_homog2twist(const double H[16], double twist[6]);
void vo_twist2homog(const double twist[6], double H[16]);

void vo_pose_refine(double T[16], const double *kp_uv, const double *X, int n, const double K[9], int max_iter,
                    double ftol, double xtol, double *stats)
{
    double tw[6];
    vo_homog2twist(T, tw);
    frame_lin fl;
    double cost = 0.0, lambda = 1e-3;
    frame_prepare(tw, &fl);
    for (int k = 0; k < n; ++k) {
        double r[2];
        observe(&fl, K, X + 3 * k, kp_uv[2 * k], kp_uv[2 * k + 1], r, NULL, NULL);
        cost += r[0] * r[0] + r[1] * r[1];
    }
    double cost0 = cost;
    int iter = 0, rejects = 0;
    while (iter < max_iter) {
        double H[36], g[6];
        memset(H, 0, sizeof(H));
        memset(g, 0, sizeof(g));
        frame_prepare(tw, &fl);
        for (int k = 0; k < n; ++k) {
            double r[2], Jp[2][6], Jl[2][3];
            observe(&fl, K, X + 3 * k, kp_uv[2 * k], kp_uv[2 * k + 1], r, Jp, Jl);
            for (int a = 0; a < 6; ++a) {
                g[a] += Jp[0][a] * r[0] + Jp[1][a] * r[1];
                for (int b = 0; b < 6; ++b) H[6 * a + b] += Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b];
            }
        }
        double Hd[36], d[6];
        memcpy(Hd, H, sizeof(H));
        for (int a = 0; a < 6; ++a) { Hd[7 * a] *= 1.0 + lambda; d[a] = -g[a]; }
        if (!chol_solve(Hd, d, 6)) {
            lambda *= 10.0;
            if (++rejects > 10) break;
            continue;
        }
        double twn[6], dn = 0.0, xnorm = 0.0, cn = 0.0;
        for (int a = 0; a < 6; ++a) { twn[a] = tw[a] + d[a]; dn += d[a] * d[a]; xnorm += tw[a] * tw[a]; }
        frame_lin fn;
        frame_prepare(twn, &fn);
        for (int k = 0; k < n; ++k) {
            double r[2];
            observe(&fn, K, X + 3 * k, kp_uv[2 * k], kp_uv[2 * k + 1], r, NULL, NULL);
            cn += r[0] * r[0] + r[1] * r[1];
        }
        if (cn < cost) {
            double dec = cost - cn, cost_old = cost;
            memcpy(tw, twn, sizeof(tw));
            cost = cn;
            lambda = lambda * 0.1 > 1e-12 ? lambda * 0.1 : 1e-12;
            ++iter;
            rejects = 0;
            if (dec < ftol * (1.0 + cost_old)) break;
            if (sqrt(dn) < xtol * (1.0 + sqrt(xnorm))) break;
        } else {
            lambda *= 10.0;
            if (++rejects > 10) break;
        }
    }
    vo_twist2homog(tw, T);
    if (stats) { stats[0] = iter; stats[1] = cost0; stats[2] = cost; }
}
