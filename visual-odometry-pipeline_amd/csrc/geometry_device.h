// Per-point device routines shared by the geometry kernels and the frame pipeline.
#pragma once

#include <hip/hip_runtime.h>

#include "se3_device.h"
#include "wave_reduce.h"

namespace vohip_dev {

// linearTriangulation.m:30-36 for one point.  A = [p1^ M1; p2^ M2] is 6x4; its right singular
// vector for the smallest singular value comes from one-sided (Hestenes) Jacobi rotations on
// the columns of A (A'A is never formed, which would square the conditioning).
// M column-major 3x4.  P = v / v(4).
__device__ inline void triangulate_point(const double p1[3], const double p2[3], const double *M1,
                                         const double *M2, double P[4])
{
    double A[6][4], V[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double *m = M1 + 3 * c;
        A[0][c] = -p1[2] * m[1] + p1[1] * m[2];
        A[1][c] = p1[2] * m[0] - p1[0] * m[2];
        A[2][c] = -p1[1] * m[0] + p1[0] * m[1];
        const double *q = M2 + 3 * c;
        A[3][c] = -p2[2] * q[1] + p2[1] * q[2];
        A[4][c] = p2[2] * q[0] - p2[0] * q[2];
        A[5][c] = -p2[1] * q[0] + p2[0] * q[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) V[r][c] = (r == c) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = i + 1; j < 4; ++j) {
                double al = 0, be = 0, ga = 0;
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    al += A[r][i] * A[r][i];
                    be += A[r][j] * A[r][j];
                    ga += A[r][i] * A[r][j];
                }
                if (ga * ga > 1e-30 * (al * be)) {   // |ga| > 1e-15 sqrt(al be)
                    rotated = true;
                    // tangent of the rotation angle, t = sign(zeta) / (|zeta| + sqrt(1 + zeta^2)) with
                    // zeta = (be - al) / (2 ga), multiplied through by |2 ga|: one square root and one
                    // division per rotation (they are most of its dependent chain), then a reciprocal root
                    const double d = be - al, g2 = 2.0 * ga;
                    const double sg = (d == 0.0 || (d > 0.0) == (ga > 0.0)) ? fabs(g2) : -fabs(g2);
                    const double t = sg / (fabs(d) + sqrt(__builtin_fma(d, d, g2 * g2)));
                    const double c = rsqrt(__builtin_fma(t, t, 1.0)), s = c * t;
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const double ai = A[r][i], aj = A[r][j];
                        A[r][i] = c * ai - s * aj;
                        A[r][j] = s * ai + c * aj;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double vi = V[r][i], vj = V[r][j];
                        V[r][i] = c * vi - s * vj;
                        V[r][j] = s * vi + c * vj;
                    }
                }
            }
        }
        if (!rotated) break;
    }
    double best = INFINITY;
    double v[4] = {0, 0, 0, 1};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double nn = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) nn += A[r][c] * A[r][c];
        if (nn < best) {
            best = nn;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = V[r][c];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) P[r] = v[r] / v[3];
}

// The same decomposition by FOUR lanes, an aligned quad of a one-dimensional workgroup whose size is a
// multiple of 64: lane c of the quad owns column c of A and of V, and in each of the three rounds of a sweep
// two disjoint pairs of columns rotate at once ((0,1) (2,3), then (0,2) (1,3), then (0,3) (1,2)); the partner's
// column crosses by DPP quad permutations.  A point's dependent instruction chain, which is what a launch of
// few points lasts, is half as long as in the single-lane routine.  The order of the rotations differs from
// that routine's (cyclic by rows), so the singular vector agrees with it to rounding, not bit for bit.
// Every lane of the quad calls it with the same arguments and receives the same P.
template <int CTRL>
__device__ __forceinline__ int dpp_move_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

template <int CTRL, int MASK>
__device__ __forceinline__ bool quad_jacobi_round(int c, double a[6], double v[4])
{
    double b[6], w[4];
#pragma unroll
    for (int r = 0; r < 6; ++r) b[r] = dpp_move_f64<CTRL>(a[r]);
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = dpp_move_f64<CTRL>(v[r]);
    const bool lo = (c & (MASK == 1 ? 1 : 2)) == 0;   // this lane owns the pair's column of lower index
    double n_own = 0, n_par = 0, ga = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        n_own += a[r] * a[r];
        n_par += b[r] * b[r];
        ga += a[r] * b[r];
    }
    const double al = lo ? n_own : n_par, be = lo ? n_par : n_own;   // (both lanes of a pair hold the same three sums)
    if (!(ga * ga > 1e-30 * (al * be))) return false;
    const double d = be - al, g2 = 2.0 * ga;
    const double sg = (d == 0.0 || (d > 0.0) == (ga > 0.0)) ? fabs(g2) : -fabs(g2);
    const double t = sg / (fabs(d) + sqrt(__builtin_fma(d, d, g2 * g2)));
    const double cs = rsqrt(__builtin_fma(t, t, 1.0)), sn = cs * t;
    const double ss = lo ? -sn : sn;   // column i becomes c a_i - s a_j, column j becomes s a_i + c a_j
#pragma unroll
    for (int r = 0; r < 6; ++r) a[r] = cs * a[r] + ss * b[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = cs * v[r] + ss * w[r];
    return true;
}

__device__ inline void triangulate_point_quad(const double p1[3], const double p2[3], const double m[3],
                                              const double q[3], double P[4])
{
    // m, q: column c of M1 and of M2 (c = this lane's place in its quad)
    const int lane = threadIdx.x & 63, c = lane & 3;
    double a[6], v[4];
    a[0] = -p1[2] * m[1] + p1[1] * m[2];
    a[1] = p1[2] * m[0] - p1[0] * m[2];
    a[2] = -p1[1] * m[0] + p1[0] * m[1];
    a[3] = -p2[2] * q[1] + p2[1] * q[2];
    a[4] = p2[2] * q[0] - p2[0] * q[2];
    a[5] = -p2[1] * q[0] + p2[0] * q[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (r == c) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = quad_jacobi_round<0xB1, 1>(c, a, v);    // quad_perm [1,0,3,2]
        rotated |= quad_jacobi_round<0x4E, 2>(c, a, v);        // quad_perm [2,3,0,1]
        rotated |= quad_jacobi_round<0x1B, 3>(c, a, v);        // quad_perm [3,2,1,0]
        const unsigned long long any = __ballot(rotated);
        if (((any >> (lane & ~3)) & 0xFull) == 0) break;       // (the same decision in the four lanes)
    }
    // the column of least norm, the first of equals (a norm that is not finite never wins: P = (0, 0, 0, 1) then)
    double nn = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) nn += a[r] * a[r];
    const bool finite = nn < INFINITY;
    double bn = finite ? nn : INFINITY;
    int bi = finite ? c : 4;
    {
        const double on = dpp_move_f64<0xB1>(bn);
        const int oi = dpp_move_i32<0xB1>(bi);
        if (on < bn || (on == bn && oi < bi)) {
            bn = on;
            bi = oi;
        }
    }
    {
        const double on = dpp_move_f64<0x4E>(bn);
        const int oi = dpp_move_i32<0x4E>(bi);
        if (on < bn || (on == bn && oi < bi)) {
            bn = on;
            bi = oi;
        }
    }
    const int src = (lane & ~3) | (bi & 3);
    double vs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) vs[r] = __shfl(v[r], src, 64);
    if (bi > 3) {
        vs[0] = vs[1] = vs[2] = 0.0;
        vs[3] = 1.0;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) P[r] = vs[r] / vs[3];
}

// 4 reprojection residuals of a 3-D point in two views and, optionally, their 4x3 Jacobian
// (linearTriangulation_refinement.m:18-22).
template <bool WITH_J>
__device__ inline double reproj4(const double P[3], const double p1[2], const double p2[2], const double *M1,
                                 const double *M2, double r[4], double J[4][3])
{
    double cost = 0.0;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const double *M = v ? M2 : M1;
        const double *p = v ? p2 : p1;
        double h[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) h[i] = M[i] * P[0] + M[3 + i] * P[1] + M[6 + i] * P[2] + M[9 + i];
        const double iz = 1.0 / h[2];
        const double u = h[0] * iz, w = h[1] * iz;
        r[2 * v] = u - p[0];
        r[2 * v + 1] = w - p[1];
        cost += r[2 * v] * r[2 * v] + r[2 * v + 1] * r[2 * v + 1];
        if (WITH_J) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                J[2 * v][c] = (M[3 * c] - u * M[3 * c + 2]) * iz;
                J[2 * v + 1][c] = (M[3 * c + 1] - w * M[3 * c + 2]) * iz;
            }
        }
    }
    return cost;
}

// Cholesky solve of a 3x3 SPD system; false when not positive definite.
__device__ inline bool chol_solve3(const double A[3][3], const double b[3], double x[3])
{
    double l00 = A[0][0];
    if (!(l00 > 0)) return false;
    l00 = sqrt(l00);
    const double l10 = A[1][0] / l00, l20 = A[2][0] / l00;
    const double d1 = A[1][1] - l10 * l10;
    if (!(d1 > 0)) return false;
    const double l11 = sqrt(d1);
    const double l21 = (A[2][1] - l20 * l10) / l11;
    const double d2 = A[2][2] - l20 * l20 - l21 * l21;
    if (!(d2 > 0)) return false;
    const double l22 = sqrt(d2);
    const double y0 = b[0] / l00;
    const double y1 = (b[1] - l10 * y0) / l11;
    const double y2 = (b[2] - l20 * y0 - l21 * y1) / l22;
    x[2] = y2 / l22;
    x[1] = (y1 - l21 * x[2]) / l11;
    x[0] = (y0 - l10 * x[1] - l20 * x[2]) / l00;
    return true;
}

// linearTriangulation_refinement.m:4-10 for one point: Levenberg-Marquardt on the 4 residuals
// (lambda0 1e-3, x0.1 on accept, x10 on reject, at most 10 rejects per iteration, damping
// lambda*diag(J'J); stops on gradient <= 1e-12 or step <= 1e-12 (1+|P|)).
__device__ inline void refine_point(double P[3], const double p1[2], const double p2[2], const double *M1,
                                    const double *M2, int max_iter)
{
    double r[4], J[4][3];
    double cost = reproj4<true>(P, p1, p2, M1, M2, r, J);
    double lambda = 1e-3;
    for (int it = 0; it < max_iter; ++it) {
        double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, g[3] = {0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                g[a] += J[k][a] * r[k];
#pragma unroll
                for (int b = 0; b < 3; ++b) H[a][b] += J[k][a] * J[k][b];
            }
        const double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
        if (!(gmax > 1e-12)) break;
        bool accepted = false;
        double step_norm = 0.0;
        for (int tries = 0; tries < 10 && !accepted; ++tries) {
            double Hd[3][3], mg[3] = {-g[0], -g[1], -g[2]}, d[3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) Hd[a][b] = H[a][b] + (a == b ? lambda * H[a][a] : 0.0);
            if (!chol_solve3(Hd, mg, d)) {
                lambda *= 10.0;
                continue;
            }
            const double Pn[3] = {P[0] + d[0], P[1] + d[1], P[2] + d[2]};
            double rn[4];
            const double cn = reproj4<false>(Pn, p1, p2, M1, M2, rn, nullptr);
            if (cn < cost) {
                accepted = true;
                step_norm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                P[0] = Pn[0];
                P[1] = Pn[1];
                P[2] = Pn[2];
                cost = reproj4<true>(P, p1, p2, M1, M2, r, J);
                lambda *= 0.1;
                if (lambda < 1e-12) lambda = 1e-12;
            } else {
                lambda *= 10.0;
            }
        }
        if (!accepted) break;
        if (step_norm <= 1e-12 * (1.0 + sqrt(P[0] * P[0] + P[1] * P[1] + P[2] * P[2]))) break;
    }
}

// update_landmarks.m:27-97 for one candidate that survived tracking and the epipolar filter.
// cur, first: (row, col); Tf = [R_W_Cf(:); t_W_Cf]; T_W_C 4x4 column-major; K column-major.
// Returns 0 = keep as candidate, 1 = new landmark (X valid), 2 = dropped.
// QUAD: called by the four lanes of an aligned quad with the same arguments (triangulate_point_quad).
template <bool QUAD = false>
__device__ inline int candidate_point(const double cur[2], const double first[2], const double Tf[12],
                                      const double *T_W_C, const double *K, double angle_threshold_deg,
                                      double min_depth, double max_depth, int is_BA, double X[3], double &angle_deg)
{
    const rigid Twc = rigid_from_cm44(T_W_C);
    const rigid Tcw = rigid_inverse(Twc);
    const double pc[3] = {cur[1], cur[0], 1.0};
    const double pf[3] = {first[1], first[0], 1.0};
    double bc[3], bf0[3], bf[3];
    k_backslash(K, pc, bc);
    k_backslash(K, pf, bf0);
    // R_C_Cf = R_C_W R_W_Cf
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const double rcf = Tcw.R[a][0] * Tf[3 * b] + Tcw.R[a][1] * Tf[3 * b + 1] + Tcw.R[a][2] * Tf[3 * b + 2];
            acc = (b == 0) ? rcf * bf0[0] : acc + rcf * bf0[b];
        }
        bf[a] = acc;
    }
    const double dot = bc[0] * bf[0] + bc[1] * bf[1] + bc[2] * bf[2];
    const double nc = sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
    const double nf = sqrt(bf[0] * bf[0] + bf[1] * bf[1] + bf[2] * bf[2]);
    const double cosang = dot / (nc * nf);
    double ang = acos(cosang) * 180.0 / 3.14159265358979323846;
    if (cosang > 1.0) ang = 0.0;  // real part of MATLAB's complex acos
    angle_deg = ang;
    X[0] = X[1] = X[2] = 0.0;
    if (!(ang > angle_threshold_deg)) return 0;

    // M = K [R | t] column-major 3x4 for both views
    double Mc[12], Mf[12];
    rigid Tfw;
    {
        rigid Twf;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Twf.R[i][j] = Tf[3 * j + i];
            Twf.t[i] = Tf[9 + i];
        }
        Tfw = rigid_inverse(Twf);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double c0 = c < 3 ? Tcw.R[0][c] : Tcw.t[0], c1 = c < 3 ? Tcw.R[1][c] : Tcw.t[1],
                         c2 = c < 3 ? Tcw.R[2][c] : Tcw.t[2];
            Mc[3 * c + a] = K[a] * c0 + K[3 + a] * c1 + K[6 + a] * c2;
            const double f0 = c < 3 ? Tfw.R[0][c] : Tfw.t[0], f1 = c < 3 ? Tfw.R[1][c] : Tfw.t[1],
                         f2 = c < 3 ? Tfw.R[2][c] : Tfw.t[2];
            Mf[3 * c + a] = K[a] * f0 + K[3 + a] * f1 + K[6 + a] * f2;
        }
    double P[4];
    if constexpr (QUAD) {
        const int c = threadIdx.x & 3;
        double mc[3], mf[3];   // (selected, not indexed: an index that is not a constant would send the matrices to scratch)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mc[a] = c == 0 ? Mc[a] : c == 1 ? Mc[3 + a] : c == 2 ? Mc[6 + a] : Mc[9 + a];
            mf[a] = c == 0 ? Mf[a] : c == 1 ? Mf[3 + a] : c == 2 ? Mf[6 + a] : Mf[9 + a];
        }
        triangulate_point_quad(pc, pf, mc, mf, P);
    } else {
        triangulate_point(pc, pf, Mc, Mf, P);
    }
    const double depth = Tcw.R[2][0] * P[0] + Tcw.R[2][1] * P[1] + Tcw.R[2][2] * P[2] + Tcw.t[2] * P[3];
    bool ok = depth > min_depth && depth < max_depth;
    if (ok && is_BA) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const double *M = v ? Mf : Mc;
            const double *p = v ? pf : pc;
            double h[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) h[a] = M[a] * P[0] + M[3 + a] * P[1] + M[6 + a] * P[2] + M[9 + a] * P[3];
            const double du = h[0] / h[2] - p[0], dv = h[1] / h[2] - p[1], dw = h[2] / h[2] - 1.0;
            ok = ok && sqrt(du * du + dv * dv + dw * dw) < 100.0;
        }
    }
    if (ok) {
        X[0] = P[0];
        X[1] = P[1];
        X[2] = P[2];
    }
    return ok ? 1 : 2;
}

}  // namespace vohip_dev
