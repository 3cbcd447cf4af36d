/*
 * geometry_oracle.c -- CPU restatement of the reference's small geometric routines:
 * twist <-> SE(3), linear triangulation (+ non-linear refinement), candidate redundancy
 * filter, and the per-candidate geometry of update_landmarks.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).  PARITY UNPINNED versus MATLAB: expm, logm, svd
 * and lsqnonlin are closed-source; they are restated by closed forms / own solvers and pinned
 * against scipy.linalg.expm/logm, numpy.linalg.svd and scipy.optimize.least_squares in tests.
 */
#include "vo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------
 * SE(3).  twist = [v; w] (linear part first), twist2HomogMatrix.m:10-14.
 * expm([w^ v; 0 0]) = [R, V v; 0 1],  R = I + A w^ + B w^2,  V = I + B w^ + C w^2,
 * A = sin(th)/th, B = (1-cos th)/th^2, C = (th - sin th)/th^3.
 * ------------------------------------------------------------------------------------- */
static void so3_coeffs(double th2, double *A, double *B, double *C)
{
    if (th2 < 1e-2) { /* Taylor to th^10: truncation < 1e-17 relative, no cancellation */
        *A = 1.0 + th2 * (-1.0 / 6.0 + th2 * (1.0 / 120.0 + th2 * (-1.0 / 5040.0 + th2 * (1.0 / 362880.0 - th2 / 39916800.0))));
        *B = 0.5 + th2 * (-1.0 / 24.0 + th2 * (1.0 / 720.0 + th2 * (-1.0 / 40320.0 + th2 * (1.0 / 3628800.0 - th2 / 479001600.0))));
        *C = 1.0 / 6.0 + th2 * (-1.0 / 120.0 + th2 * (1.0 / 5040.0 + th2 * (-1.0 / 362880.0 + th2 * (1.0 / 39916800.0 - th2 / 6227020800.0))));
    } else {
        double th = sqrt(th2);
        *A = sin(th) / th;
        *B = (1.0 - cos(th)) / th2;
        *C = (th - sin(th)) / (th2 * th);
    }
}

/* H 4x4 column-major */
void vo_twist2homog(const double twist[6], double H[16])
{
    const double *v = twist, *w = twist + 3;
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B, C;
    so3_coeffs(th2, &A, &B, &C);
    double W[3][3] = { { 0, -w[2], w[1] }, { w[2], 0, -w[0] }, { -w[1], w[0], 0 } };
    double W2[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            W2[i][j] = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
    memset(H, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i) {
        double t = 0.0;
        for (int j = 0; j < 3; ++j) {
            double I = (i == j) ? 1.0 : 0.0;
            H[4 * j + i] = I + A * W[i][j] + B * W2[i][j];
            t += (I + B * W[i][j] + C * W2[i][j]) * v[j];
        }
        H[12 + i] = t;
    }
    H[15] = 1.0;
}

/* logm of a rigid transform, HomogMatrix2twist.m:14-23.  Rotation angle in [0, pi]. */
void vo_homog2twist(const double H[16], double twist[6])
{
    double R[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[i][j] = H[4 * j + i];
    double t[3] = { H[12], H[13], H[14] };
    /* s = sin(th) * axis */
    double s[3] = { 0.5 * (R[2][1] - R[1][2]), 0.5 * (R[0][2] - R[2][0]), 0.5 * (R[1][0] - R[0][1]) };
    double sn = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    double cs = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1.0);
    double th = atan2(sn, cs);
    double w[3];
    if (sn < 1e-6 && cs > 0.0) {
        /* th ~ 0: w = s * th/sin(th) */
        double k = 1.0 + th * th / 6.0 + 7.0 * th * th * th * th / 360.0;
        for (int i = 0; i < 3; ++i) w[i] = k * s[i];
    } else if (sn < 1e-6) {
        /* th ~ pi: axis from the diagonal of (R + I)/2 = axis axis' (1-cos)/2 + ... */
        double d[3] = { R[0][0], R[1][1], R[2][2] };
        int k = (d[0] >= d[1] && d[0] >= d[2]) ? 0 : (d[1] >= d[2] ? 1 : 2);
        double ax[3];
        double den = sqrt(fmax(0.0, (d[k] - cs) / (1.0 - cs)));
        ax[k] = den;
        for (int i = 0; i < 3; ++i)
            if (i != k) ax[i] = (R[i][k] + R[k][i]) / (2.0 * (1.0 - cs) * den);
        /* sign from the (tiny) antisymmetric part */
        double dot = ax[0] * s[0] + ax[1] * s[1] + ax[2] * s[2];
        double sg = dot < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i) w[i] = sg * th * ax[i];
    } else {
        for (int i = 0; i < 3; ++i) w[i] = s[i] * (th / sn);
    }
    /* v = V^-1 t,  V^-1 = I - w^/2 + D w^2,  D = (1 - A/(2B)) / th^2 */
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B, C, D;
    so3_coeffs(th2, &A, &B, &C);
    if (th2 < 1e-2)
        D = 1.0 / 12.0 + th2 * (1.0 / 720.0 + th2 * (1.0 / 30240.0 + th2 * (1.0 / 1209600.0 + th2 / 47900160.0)));
    else
        D = (1.0 - A / (2.0 * B)) / th2;
    double wxt[3] = { w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0] };
    double wxwxt[3] = { w[1] * wxt[2] - w[2] * wxt[1], w[2] * wxt[0] - w[0] * wxt[2],
                        w[0] * wxt[1] - w[1] * wxt[0] };
    for (int i = 0; i < 3; ++i) {
        twist[i] = t[i] - 0.5 * wxt[i] + D * wxwxt[i];
        twist[3 + i] = w[i];
    }
}

/* ---------------------------------------------------------------------------------------
 * linearTriangulation.m:28-39 for one point with its own pair of projection matrices.
 * A = [p1^ M1; p2^ M2] (6x4); the right singular vector of the smallest singular value
 * by one-sided (Hestenes) Jacobi on the columns of A -- AtA is never formed.
 * ------------------------------------------------------------------------------------- */
static void cross_times_M(const double p[3], const double *M /*3x4 col-major*/, double A[6][4], int r0)
{
    for (int c = 0; c < 4; ++c) {
        const double *m = M + 3 * c;
        A[r0 + 0][c] = -p[2] * m[1] + p[1] * m[2];
        A[r0 + 1][c] = p[2] * m[0] - p[0] * m[2];
        A[r0 + 2][c] = -p[1] * m[0] + p[0] * m[1];
    }
}

void vo_triangulate_one(const double p1[3], const double p2[3], const double *M1, const double *M2,
                        double P[4])
{
    double A[6][4], V[4][4];
    cross_times_M(p1, M1, A, 0);
    cross_times_M(p2, M2, A, 3);
    memset(V, 0, sizeof(V));
    for (int i = 0; i < 4; ++i) V[i][i] = 1.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = i + 1; j < 4; ++j) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 6; ++r) {
                    al += A[r][i] * A[r][i];
                    be += A[r][j] * A[r][j];
                    ga += A[r][i] * A[r][j];
                }
                if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 6; ++r) {
                    double ai = A[r][i], aj = A[r][j];
                    A[r][i] = c * ai - s * aj;
                    A[r][j] = s * ai + c * aj;
                }
                for (int r = 0; r < 4; ++r) {
                    double vi = V[r][i], vj = V[r][j];
                    V[r][i] = c * vi - s * vj;
                    V[r][j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    int k = 0;
    double best = INFINITY;
    for (int c = 0; c < 4; ++c) {
        double nn = 0;
        for (int r = 0; r < 6; ++r) nn += A[r][c] * A[r][c];
        if (nn < best) { best = nn; k = c; }
    }
    for (int r = 0; r < 4; ++r) P[r] = V[r][k] / V[3][k];
}

/* linearTriangulation(p1,p2,M1,M2): p 3xn; M 3x4, or 3x4xn when per_point_M (the way
 * update_landmarks.m:57-61 calls it, one M_first per candidate).  P 4xn. */
void vo_linear_triangulation(const double *p1, const double *p2, const double *M1, const double *M2,
                             int per_point_M1, int per_point_M2, int n, double *P)
{
    for (int j = 0; j < n; ++j)
        vo_triangulate_one(p1 + 3 * j, p2 + 3 * j, M1 + (per_point_M1 ? 12 * j : 0),
                           M2 + (per_point_M2 ? 12 * j : 0), P + 4 * j);
}

/* ---------------------------------------------------------------------------------------
 * linearTriangulation_refinement.m:1-22: per point, minimise the 4 reprojection residuals
 * over the 3 coordinates, at most 20 iterations.  The reference uses lsqnonlin (closed
 * source); this is Levenberg-Marquardt with analytic Jacobians:
 *   lambda0 = 1e-3, accept -> lambda /= 10, reject -> lambda *= 10 (at most 10 rejects per
 *   iteration), damping lambda * diag(JtJ), stop when |step| <= 1e-12 (1 + |P|) or the
 *   gradient inf-norm <= 1e-12.
 * ------------------------------------------------------------------------------------- */
static double reproj4(const double P[3], const double *p1, const double *p2, const double *M1,
                      const double *M2, double r[4], double J[4][3])
{
    const double *Ms[2] = { M1, M2 };
    const double *ps[2] = { p1, p2 };
    double cost = 0.0;
    for (int v = 0; v < 2; ++v) {
        const double *M = Ms[v];
        double h[3];
        for (int i = 0; i < 3; ++i) h[i] = M[i] * P[0] + M[3 + i] * P[1] + M[6 + i] * P[2] + M[9 + i];
        double iz = 1.0 / h[2];
        double u = h[0] * iz, w = h[1] * iz;
        r[2 * v] = u - ps[v][0];
        r[2 * v + 1] = w - ps[v][1];
        cost += r[2 * v] * r[2 * v] + r[2 * v + 1] * r[2 * v + 1];
        if (J)
            for (int c = 0; c < 3; ++c) {
                J[2 * v][c] = (M[3 * c] - u * M[3 * c + 2]) * iz;
                J[2 * v + 1][c] = (M[3 * c + 1] - w * M[3 * c + 2]) * iz;
            }
    }
    return cost;
}

static int solve3(double A[3][3], const double b[3], double x[3])
{
    /* Cholesky of a 3x3 SPD matrix */
    double l00 = A[0][0];
    if (!(l00 > 0)) return 0;
    l00 = sqrt(l00);
    double l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    double d1 = A[1][1] - l10 * l10;
    if (!(d1 > 0)) return 0;
    double l11 = sqrt(d1);
    double l21 = (A[2][1] - l20 * l10) / l11;
    double d2 = A[2][2] - l20 * l20 - l21 * l21;
    if (!(d2 > 0)) return 0;
    double l22 = sqrt(d2);
    double y0 = b[0] / l00;
    double y1 = (b[1] - l10 * y0) / l11;
    double y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
    x[2] = y2 / l22;
    x[1] = (y1 - l21 * x[2]) / l11;
    x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
    return 1;
}

void vo_triangulation_refine_one(const double P_init[3], const double *p1, const double *p2,
                                 const double *M1, const double *M2, int max_iter, double P_out[3])
{
    double P[3] = { P_init[0], P_init[1], P_init[2] };
    double r[4], J[4][3];
    double cost = reproj4(P, p1, p2, M1, M2, r, J);
    double lambda = 1e-3;
    for (int it = 0; it < max_iter; ++it) {
        double H[3][3] = { { 0 } }, g[3] = { 0, 0, 0 };
        for (int k = 0; k < 4; ++k)
            for (int a = 0; a < 3; ++a) {
                g[a] += J[k][a] * r[k];
                for (int b = 0; b < 3; ++b) H[a][b] += J[k][a] * J[k][b];
            }
        double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        if (!(gmax > 1e-12)) break;
        int accepted = 0;
        double step_norm = 0.0;
        for (int tries = 0; tries < 10 && !accepted; ++tries) {
            double Hd[3][3], mg[3] = { -g[0], -g[1], -g[2] }, d[3];
            memcpy(Hd, H, sizeof(H));
            for (int a = 0; a < 3; ++a) Hd[a][a] += lambda * H[a][a];
            if (!solve3(Hd, mg, d)) { lambda *= 10.0; continue; }
            double Pn[3] = { P[0] + d[0], P[1] + d[1], P[2] + d[2] };
            double rn[4];
            double cn = reproj4(Pn, p1, p2, M1, M2, rn, NULL);
            if (cn < cost) {
                accepted = 1;
                step_norm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                memcpy(P, Pn, sizeof(P));
                cost = reproj4(P, p1, p2, M1, M2, r, J);
                lambda *= 0.1;
                if (lambda < 1e-12) lambda = 1e-12;
            } else {
                lambda *= 10.0;
            }
        }
        if (!accepted) break;
        if (step_norm <= 1e-12 * (1.0 + sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]))) break;
    }
    memcpy(P_out, P, 3 * sizeof(double));
}

/* linearTriangulation_refinement(P_init,p1,p2,M1,M2): P_init 4xn (rows 1-3 used), p >= 2 x n
 * given with leading dimension ldp, M 3x4 (shared).  P 4xn, row 4 copied from P_init. */
void vo_triangulation_refinement(const double *P_init, const double *p1, const double *p2, int ldp,
                                 const double *M1, const double *M2, int n, double *P)
{
    for (int j = 0; j < n; ++j) {
        vo_triangulation_refine_one(P_init + 4 * j, p1 + ldp * j, p2 + ldp * j, M1, M2, 20, P + 4 * j);
        P[4 * j + 3] = P_init[4 * j + 3];
    }
}

/* ---------------------------------------------------------------------------------------
 * update_candidate.m:12-27: keep a new detection iff no S.C and no S.P point lies in the closed
 * box of half-width r around it; when S.C is empty keep everything.  new_xy n_new x 2 single
 * (x, y) as cornerPoints.Location; C_rc 2 x nC and P_rc 2 x nP double (row, col).
 * MATLAB evaluates single-double mixed expressions in single: the box edges are single, and
 * the double coordinates are compared after rounding to single.
 * ------------------------------------------------------------------------------------- */
void vo_candidate_filter(const float *new_xy, int n_new, const double *C_rc, int nC, const double *P_rc,
                         int nP, double r, uint8_t *keep)
{
    if (nC == 0) { memset(keep, 1, (size_t)n_new); return; }
    const float rf = (float)r;
    for (int i = 0; i < n_new; ++i) {
        float x = new_xy[i], y = new_xy[n_new + i];
        float u0 = x - rf, u1 = x + rf, v0 = y - rf, v1 = y + rf;
        int near = 0;
        for (int k = 0; k < nC && !near; ++k) {
            float cu = (float)C_rc[2 * k + 1], cv = (float)C_rc[2 * k];
            near = cu >= u0 && cu <= u1 && cv >= v0 && cv <= v1;
        }
        for (int k = 0; k < nP && !near; ++k) {
            float cu = (float)P_rc[2 * k + 1], cv = (float)P_rc[2 * k];
            near = cu >= u0 && cu <= u1 && cv >= v0 && cv <= v1;
        }
        keep[i] = (uint8_t)!near;
    }
}

/* ---------------------------------------------------------------------------------------
 * update_landmarks.m:27-97 for candidates that survived tracking and the F-matrix filter:
 * bearing angle between the current and the first observation, triangulation of those above
 * the threshold, depth (and, with BA, reprojection) filter.
 *   cur_rc, first_rc 2 x n (row, col); T_first 12 x n ([R_W_C(:); t_W_C] at first observation);
 *   T_W_C 4x4 current pose.
 * status[i]: 0 = keep as candidate (angle <= threshold), 1 = new landmark (X_out[:,i] valid),
 *            2 = dropped (angle > threshold but depth / reprojection test failed).
 * angles_deg[i] as computed at :40-42.
 * ------------------------------------------------------------------------------------- */
static void inv_rigid(const double T[16], double Ti[16])
{
    memset(Ti, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Ti[4 * j + i] = T[4 * i + j];
    for (int i = 0; i < 3; ++i)
        Ti[12 + i] = -(Ti[i] * T[12] + Ti[4 + i] * T[13] + Ti[8 + i] * T[14]);
    Ti[15] = 1.0;
}

static void k_backslash3(const double K[9], const double b[3], double x[3])
{
    x[2] = b[2] / K[8];
    x[1] = (b[1] - K[7] * x[2]) / K[4];
    x[0] = (b[0] - K[3] * x[1] - K[6] * x[2]) / K[0];
}

void vo_candidate_triangulate(const double *cur_rc, const double *first_rc, const double *T_first, int n,
                              const double T_W_C[16], const double K[9], double angle_threshold_deg,
                              double min_depth, double max_depth, int is_BA, uint8_t *status,
                              double *X_out, double *angles_deg)
{
    double T_C_W[16];
    inv_rigid(T_W_C, T_C_W);
    /* M_current = K [R_C_W t_C_W], :52 */
    double Mc[12];
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 3; ++i)
            Mc[3 * c + i] = K[i] * T_C_W[4 * c] + K[3 + i] * T_C_W[4 * c + 1] + K[6 + i] * T_C_W[4 * c + 2];
    for (int i = 0; i < n; ++i) {
        double pc[3] = { cur_rc[2 * i + 1], cur_rc[2 * i], 1.0 };
        double pf[3] = { first_rc[2 * i + 1], first_rc[2 * i], 1.0 };
        double bc[3], bf0[3], bf[3];
        k_backslash3(K, pc, bc);
        k_backslash3(K, pf, bf0);
        /* R_C_Cf = R_C_W R_W_Cf, :34-37 */
        const double *Tf = T_first + 12 * i;
        double Rcf[3][3];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                Rcf[a][b] = T_C_W[a] * Tf[3 * b] + T_C_W[4 + a] * Tf[3 * b + 1] + T_C_W[8 + a] * Tf[3 * b + 2];
        for (int a = 0; a < 3; ++a) bf[a] = Rcf[a][0] * bf0[0] + Rcf[a][1] * bf0[1] + Rcf[a][2] * bf0[2];
        double dot = bc[0] * bf[0] + bc[1] * bf[1] + bc[2] * bf[2];
        double nc = sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
        double nf = sqrt(bf[0] * bf[0] + bf[1] * bf[1] + bf[2] * bf[2]);
        double ang = acos(dot / (nc * nf)) * 180.0 / 3.14159265358979323846;
        /* acos of a value rounded above 1 is complex in MATLAB; its real part is 0 */
        if (dot / (nc * nf) > 1.0) ang = 0.0;
        if (angles_deg) angles_deg[i] = ang;
        X_out[3 * i] = X_out[3 * i + 1] = X_out[3 * i + 2] = 0.0;
        if (!(ang > angle_threshold_deg)) { status[i] = 0; continue; }
        /* M_first = K * inv(T_W_Cf)(1:3,:), :57-59 */
        double Tf4[16] = { Tf[0], Tf[1], Tf[2], 0, Tf[3], Tf[4], Tf[5], 0, Tf[6], Tf[7], Tf[8], 0,
                           Tf[9], Tf[10], Tf[11], 1 };
        double Tfi[16], Mf[12];
        inv_rigid(Tf4, Tfi);
        for (int c = 0; c < 4; ++c)
            for (int a = 0; a < 3; ++a)
                Mf[3 * c + a] = K[a] * Tfi[4 * c] + K[3 + a] * Tfi[4 * c + 1] + K[6 + a] * Tfi[4 * c + 2];
        double P[4];
        vo_triangulate_one(pc, pf, Mc, Mf, P);
        double depth = T_C_W[2] * P[0] + T_C_W[6] * P[1] + T_C_W[10] * P[2] + T_C_W[14] * P[3];
        int ok = depth > min_depth && depth < max_depth;
        if (ok && is_BA) { /* :72-82 */
            const double *Ms[2] = { Mc, Mf };
            const double *ps[2] = { pc, pf };
            for (int v = 0; v < 2; ++v) {
                double h[3];
                for (int a = 0; a < 3; ++a)
                    h[a] = Ms[v][a] * P[0] + Ms[v][3 + a] * P[1] + Ms[v][6 + a] * P[2] + Ms[v][9 + a] * P[3];
                double du = h[0] / h[2] - ps[v][0], dv = h[1] / h[2] - ps[v][1], dw = h[2] / h[2] - 1.0;
                ok = ok && sqrt(du * du + dv * dv + dw * dw) < 100.0;
            }
        }
        status[i] = ok ? 1 : 2;
        if (ok) { X_out[3 * i] = P[0]; X_out[3 * i + 1] = P[1]; X_out[3 * i + 2] = P[2]; }
    }
}
