// P3P RANSAC localisation for gfx950.
//
// Stands in for utils/exe7_Ransac/ransacLocalization.m:1-157 with p3p.m:41-217,
// solveQuartic.m:26-66 and projectPoints.m:1-16 of the reference.
//
// Three kernels:
//   p3p_solve_kernel   FOUR LANES per hypothesis: sample -> bearings -> Kneip P3P up to the roots of
//                      the quartic (by each of the four), then one candidate pose per lane:
//                      [R_C_W | t_C_W] and the fused projection K [R | t].
//                      Written statement for statement in the reference's evaluation order and
//                      compiled without FMA contraction, so poses agree with a CPU evaluation
//                      of the same statements up to the last bits of pow/atan2/sincos.
//   p3p_score_kernel   one WAVEFRONT per hypothesis: the 64 lanes stride over the N points,
//                      each point is tested against the 4 candidate poses, counts come from
//                      ballot + popcount.  Points are staged once per workgroup into LDS in
//                      structure-of-arrays form and shared by the workgroup's hypotheses.
//                      The inlier test is the division-free form of the reference test
//                      |kp - proj|^2 < 100:  (x - u z)^2 + (y - v z)^2 < 100 z^2,
//                      13 FMA + 2 MUL + 1 CMP per (pose, point).  It can differ from the
//                      reference's divide-then-distort arithmetic only for points whose squared
//                      error is within rounding of the threshold.
//   inlier_mask_kernel the same test for one pose, writing the mask.
//
// The adaptive stop of ransacLocalization.m:118-135 is a sequential recurrence over the
// per-hypothesis inlier counts; it is replayed on the host in sample order after each chunk
// of hypotheses, so hypotheses past the stop index never influence the result.
//
// The functions p3p_prepare, p3p_solution and solve_quartic_real below restate, in the evaluation
// order of the originals, the closed forms of p3p.m and solveQuartic.m by Laurent Kneip, which are
// distributed under the following licence; its notice is retained here as the licence requires.
//
//   Copyright (c) 2011, Laurent Kneip, ETH Zurich
//   All rights reserved.
//
//   Redistribution and use in source and binary forms, with or without
//   modification, are permitted provided that the following conditions are met:
//       * Redistributions of source code must retain the above copyright
//         notice, this list of conditions and the following disclaimer.
//       * Redistributions in binary form must reproduce the above copyright
//         notice, this list of conditions and the following disclaimer in the
//         documentation and/or other materials provided with the distribution.
//       * Neither the name of ETH Zurich nor the
//         names of its contributors may be used to endorse or promote products
//         derived from this software without specific prior written permission.
//
//   THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND
//   ANY EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED
//   WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A PARTICULAR PURPOSE ARE
//   DISCLAIMED. IN NO EVENT SHALL ETH ZURICH BE LIABLE FOR ANY
//   DIRECT, INDIRECT, INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES
//   (INCLUDING, BUT NOT LIMITED TO, PROCUREMENT OF SUBSTITUTE GOODS OR SERVICES;
//   LOSS OF USE, DATA, OR PROFITS; OR BUSINESS INTERRUPTION) HOWEVER CAUSED AND
//   ON ANY THEORY OF LIABILITY, WHETHER IN CONTRACT, STRICT LIABILITY, OR TORT
//   (INCLUDING NEGLIGENCE OR OTHERWISE) ARISING IN ANY WAY OUT OF THE USE OF THIS
//   SOFTWARE, EVEN IF ADVISED OF THE POSSIBILITY OF SUCH DAMAGE.
#include "vohip_internal.h"
#include "inlier_test.h"
#include "seq_index.h"
#include "ordered_select.h"

#include <cmath>
#include <mutex>
#include <vector>

// Scratch of the allocation-free entry point (vohip_ransac_p3p_score_noalloc): the candidate poses of
// up to NOALLOC_MAX_H hypotheses, [R_C_W | t_C_W] for all of them, then K [R | t] for all of them
// (2 x 4096 x 48 doubles = 3 MB).  Every GPU the library runs on holds its own copy until the process
// ends.  In host code the array's name is only a stand-in: the device address is taken once per
// device with hipGetSymbolAddress (vohip_ransac_scratch_init) and that pointer is what kernels get.
constexpr int NOALLOC_MAX_H = 4096;
static __device__ __align__(256) double scratch[2 * (size_t)NOALLOC_MAX_H * 48];
constexpr int NOALLOC_MAX_DEVICES = 64;
static double *g_scratch_ptr[NOALLOC_MAX_DEVICES] = {nullptr};
static std::mutex g_scratch_mutex;

namespace {

// ------------------------------------------------------------------------------------------
// Complex helpers with MATLAB's conventions (principal branches; a zero imaginary part is a
// real number, so -0.0 imaginary never selects the lower sqrt branch).
// ------------------------------------------------------------------------------------------
#pragma clang fp contract(off)

struct cplx {
    double re, im;
};

__device__ __forceinline__ cplx cmk(double re, double im)
{
    cplx z;
    z.re = re;
    z.im = (im == 0.0) ? 0.0 : im;
    return z;
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return cmk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return cmk(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cplx cneg(cplx a) { return cmk(-a.re, -a.im); }
__device__ __forceinline__ cplx cscale(double s, cplx a) { return cmk(s * a.re, s * a.im); }

// Smith's algorithm.
__device__ __forceinline__ cplx cdiv(cplx a, cplx b)
{
    if (b.im == 0.0) return cmk(a.re / b.re, a.im / b.re);
    if (fabs(b.re) >= fabs(b.im)) {
        double r = b.im / b.re;
        double den = b.re + b.im * r;
        return cmk((a.re + a.im * r) / den, (a.im - a.re * r) / den);
    }
    double r = b.re / b.im;
    double den = b.re * r + b.im;
    return cmk((a.re * r + a.im) / den, (a.im * r - a.re) / den);
}

__device__ __forceinline__ cplx csqrt_principal(cplx z)
{
    if (z.im == 0.0) {
        if (z.re >= 0.0) return cmk(sqrt(z.re), 0.0);
        if (z.re < 0.0) return cmk(0.0, sqrt(-z.re));
        return cmk(z.re, z.re);
    }
    double m = sqrt(z.re * z.re + z.im * z.im);
    if (z.re >= 0.0) {
        double t = sqrt(0.5 * (m + z.re));
        return cmk(t, z.im / (2.0 * t));
    }
    double t = sqrt(0.5 * (m - z.re));
    if (z.im < 0.0) t = -t;
    return cmk(z.im / (2.0 * t), t);
}

__device__ __forceinline__ cplx ccbrt_principal(cplx z)
{
    const double third = 1.0 / 3.0;
    if (z.im == 0.0 && z.re >= 0.0) return cmk(pow(z.re, third), 0.0);
    double m = sqrt(z.re * z.re + z.im * z.im);
    double a = atan2(z.im, z.re) * third;
    double r = pow(m, third);
    double sa, ca;
    sincos(a, &sa, &ca);
    return cmk(r * ca, r * sa);
}

// solveQuartic.m:26-66; only the real parts of the roots are used by p3p.m:189-192.
__device__ void solve_quartic_real(const double f[5], double xr[4])
{
    const double A = f[0], B = f[1], C = f[2], D = f[3], E = f[4];
    const double A2 = A * A, B2 = B * B;
    const double A3 = A2 * A, B3 = B2 * B;
    const double A4 = A3 * A, B4 = B3 * B;

    const double alpha = -3.0 * B2 / (8.0 * A2) + C / A;
    const double beta = B3 / (8.0 * A3) - B * C / (2.0 * A2) + D / A;
    const double gamma =
        -3.0 * B4 / (256.0 * A4) + B2 * C / (16.0 * A3) - B * D / (4.0 * A2) + E / A;

    const double alpha2 = alpha * alpha;
    const double alpha3 = alpha2 * alpha;
    const double P = -alpha2 / 12.0 - gamma;
    const double Q = -alpha3 / 108.0 + alpha * gamma / 3.0 - (beta * beta) / 8.0;
    const double disc = (Q * Q) / 4.0 + (P * P * P) / 27.0;
    const cplx R = cadd(cmk(-Q / 2.0, 0.0), csqrt_principal(cmk(disc, 0.0)));
    const cplx U = ccbrt_principal(R);

    const cplx lead = cmk(-5.0 * alpha / 6.0, 0.0);
    cplx y;
    if (U.re == 0.0 && U.im == 0.0) {
        y = csub(lead, ccbrt_principal(cmk(Q, 0.0)));
    } else {
        y = cadd(csub(lead, cdiv(cmk(P, 0.0), cscale(3.0, U))), U);
    }
    const cplx w = csqrt_principal(cadd(cmk(alpha, 0.0), cscale(2.0, y)));
    const cplx base = cadd(cmk(3.0 * alpha, 0.0), cscale(2.0, y));
    const cplx bw = cdiv(cmk(2.0 * beta, 0.0), w);
    const cplx sp = csqrt_principal(cneg(cadd(base, bw)));
    const cplx sm = csqrt_principal(cneg(csub(base, bw)));
    const cplx shift = cmk(-B / (4.0 * A), 0.0);
    xr[0] = cadd(shift, cscale(0.5, cadd(w, sp))).re;
    xr[1] = cadd(shift, cscale(0.5, csub(w, sp))).re;
    xr[2] = cadd(shift, cscale(0.5, cadd(cneg(w), sm))).re;
    xr[3] = cadd(shift, cscale(0.5, csub(cneg(w), sm))).re;
}

struct v3 {
    double x, y, z;
};
__device__ __forceinline__ v3 vsub(v3 a, v3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ v3 vcross(v3 a, v3 b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 vdivs(v3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }

// 3x3 with named rows; r0,r1,r2 are the rows.
struct m3 {
    v3 r0, r1, r2;
};
__device__ __forceinline__ v3 mrows_times(const m3 &M, v3 v)
{
    return {vdot(M.r0, v), vdot(M.r1, v), vdot(M.r2, v)};
}

// p3p.m:62-74 / :80-92
__device__ __forceinline__ void camera_frame(v3 f1, v3 f2, v3 f3in, m3 &T, v3 &f3)
{
    v3 e3 = vcross(f1, f2);
    e3 = vdivs(e3, sqrt(vdot(e3, e3)));
    v3 e2 = vcross(e3, f1);
    T.r0 = f1;
    T.r1 = e2;
    T.r2 = e3;
    f3 = mrows_times(T, f3in);
}

// Kneip P3P, p3p.m:41-217.  out[s][0..2] = C (camera position), out[s][3..11] = R_W_C
// column-major.  Returns false (and leaves `out` zeroed) for collinear world points.
// What the four solutions of one P3P problem share (p3p.m:41-185): the two intermediate frames, the
// scalars of the quartic and its four roots.
struct p3p_shared {
    v3 P1;
    m3 T, N;
    double f_1, f_2, p_1, p_2, d_12, b;
    double xr[4];
};

__device__ bool p3p_prepare(v3 W1, v3 W2, v3 W3, v3 b1, v3 b2, v3 b3, p3p_shared &q)
{
    v3 cr = vcross(vsub(W2, W1), vsub(W3, W1));
    if (sqrt(vdot(cr, cr)) == 0.0) return false;

    v3 P1 = W1, P2 = W2, P3 = W3, f1 = b1, f2 = b2, f3;
    m3 T;
    camera_frame(f1, f2, b3, T, f3);
    if (f3.z > 0.0) {
        f1 = b2;
        f2 = b1;
        P1 = W2;
        P2 = W1;
        camera_frame(f1, f2, b3, T, f3);
    }

    // intermediate world frame, p3p.m:102-110
    v3 d31 = vsub(P3, P1);
    v3 n1 = vsub(P2, P1);
    n1 = vdivs(n1, sqrt(vdot(n1, n1)));
    v3 n3 = vcross(n1, d31);
    n3 = vdivs(n3, sqrt(vdot(n3, n3)));
    v3 n2 = vcross(n3, n1);
    m3 N = {n1, n2, n3};

    // p3p.m:114-129
    v3 P3n = mrows_times(N, d31);
    v3 d21 = vsub(P2, P1);
    const double d_12 = sqrt(d21.x * d21.x + d21.y * d21.y + d21.z * d21.z);
    const double f_1 = f3.x / f3.z;
    const double f_2 = f3.y / f3.z;
    const double p_1 = P3n.x;
    const double p_2 = P3n.y;
    const double cos_beta = vdot(f1, f2);
    double b = 1.0 / (1.0 - cos_beta * cos_beta) - 1.0;
    const double sb = (b < 0.0) ? 0.0 : sqrt(b);
    b = (cos_beta < 0.0) ? -sb : sb;

    const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2;
    const double p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
    const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2;
    const double d_12_pw2 = d_12 * d_12, b_pw2 = b * b;

    // p3p.m:146-179
    double fac[5];
    fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
    fac[1] = 2.0 * p_2_pw3 * d_12 * b + 2.0 * f_2_pw2 * p_2_pw3 * d_12 * b -
             2.0 * f_2 * p_2_pw3 * f_1 * d_12;
    fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 -
             f_2_pw2 * p_2_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 +
             2.0 * p_1 * p_2_pw2 * d_12 + 2.0 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b -
             p_2_pw2 * p_1_pw2 * f_1_pw2 + 2.0 * p_1 * p_2_pw2 * f_2_pw2 * d_12 -
             p_2_pw2 * d_12_pw2 * b_pw2 - 2.0 * p_1_pw2 * p_2_pw2;
    fac[3] = 2.0 * p_1_pw2 * p_2 * d_12 * b + 2.0 * f_2 * p_2_pw3 * f_1 * d_12 -
             2.0 * f_2_pw2 * p_2_pw3 * d_12 * b - 2.0 * p_1 * p_2 * d_12_pw2 * b;
    fac[4] = -2.0 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 +
             2.0 * p_1_pw3 * d_12 - p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 -
             2.0 * f_2_pw2 * p_2_pw2 * p_1 * d_12 + p_2_pw2 * f_1_pw2 * p_1_pw2 +
             f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;

    solve_quartic_real(fac, q.xr);
    q.P1 = P1;
    q.T = T;
    q.N = N;
    q.f_1 = f_1;
    q.f_2 = f_2;
    q.p_1 = p_1;
    q.p_2 = p_2;
    q.d_12 = d_12;
    q.b = b;
    return true;
}

// p3p.m:187-215 for the root x: out = [C | R_W_C col-major].
__device__ void p3p_solution(const p3p_shared &q, double x, double out[12])
{
    const v3 P1 = q.P1;
    const m3 &T = q.T, &N = q.N;
    const double f_1 = q.f_1, f_2 = q.f_2, p_1 = q.p_1, p_2 = q.p_2, d_12 = q.d_12, b = q.b;
    {
        const double cot_alpha =
            (-f_1 * p_1 / f_2 - x * p_2 + d_12 * b) / (-f_1 * x * p_2 / f_2 + p_1 - d_12);
        const double cos_theta = x;
        const double st2 = 1.0 - x * x;
        const double sin_theta = (st2 < 0.0) ? 0.0 : sqrt(st2);  // real part of a pure-imaginary root
        const double sin_alpha = sqrt(1.0 / (cot_alpha * cot_alpha + 1.0));
        const double ca2 = 1.0 - sin_alpha * sin_alpha;
        double cos_alpha = (ca2 < 0.0) ? 0.0 : sqrt(ca2);
        if (cot_alpha < 0.0) cos_alpha = -cos_alpha;

        const double k = sin_alpha * b + cos_alpha;
        const v3 Cv = {d_12 * cos_alpha * k, cos_theta * d_12 * sin_alpha * k,
                       sin_theta * d_12 * sin_alpha * k};
        // C = P1 + N' * Cv
        out[0] = P1.x + (N.r0.x * Cv.x + N.r1.x * Cv.y + N.r2.x * Cv.z);
        out[1] = P1.y + (N.r0.y * Cv.x + N.r1.y * Cv.y + N.r2.y * Cv.z);
        out[2] = P1.z + (N.r0.z * Cv.x + N.r1.z * Cv.y + N.r2.z * Cv.z);

        const m3 Rm = {{-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta},
                       {sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta},
                       {0.0, -sin_theta, cos_theta}};
        // R = (N' * Rm') * T ;  (N' Rm')[r][c] = sum_k N[k][r] Rm[c][k]
        const v3 Ncol[3] = {{N.r0.x, N.r1.x, N.r2.x}, {N.r0.y, N.r1.y, N.r2.y}, {N.r0.z, N.r1.z, N.r2.z}};
        for (int r = 0; r < 3; ++r) {
            const v3 nr = {vdot(Ncol[r], Rm.r0), vdot(Ncol[r], Rm.r1), vdot(Ncol[r], Rm.r2)};
            const double c0 = nr.x * T.r0.x + nr.y * T.r1.x + nr.z * T.r2.x;
            const double c1 = nr.x * T.r0.y + nr.y * T.r1.y + nr.z * T.r2.y;
            const double c2 = nr.x * T.r0.z + nr.y * T.r1.z + nr.z * T.r2.z;
            out[3 + r] = c0;      // column 0 of R_W_C
            out[6 + r] = c1;      // column 1
            out[9 + r] = c2;      // column 2
        }
    }
}

// p3p.m:41-217: all four solutions (zeros for a degenerate world triangle, p3p.m:59-62).
__device__ bool p3p_solve(v3 W1, v3 W2, v3 W3, v3 b1, v3 b2, v3 b3, double out[4][12])
{
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 12; ++j) out[s][j] = 0.0;
    p3p_shared q;
    if (!p3p_prepare(W1, W2, W3, b1, b2, b3, q)) return false;
    for (int s = 0; s < 4; ++s) p3p_solution(q, q.xr[s], out[s]);
    return true;
}

// K \ [u; v; 1] for an upper-triangular camera matrix, then unit length
// (ransacLocalization.m:52-56).
__device__ __forceinline__ v3 unit_bearing(const vohip_K &K, double u, double v)
{
    v3 n;
    n.z = 1.0 / K.k[8];
    n.y = (v - K.k[7] * n.z) / K.k[4];
    n.x = (u - K.k[3] * n.y - K.k[6] * n.z) / K.k[0];
    return vdivs(n, sqrt(n.x * n.x + n.y * n.y + n.z * n.z));
}

// Decode of ransacLocalization.m:61-68 and the fused projection matrix of one candidate.
// pose = [C | R_W_C col-major];  Rt = [R_C_W col-major | t_C_W];  M = K [R_C_W | t_C_W] row-major.
__device__ __forceinline__ void decode_pose(const vohip_K &K, const double pose[12], double Rt[12],
                                            double M[12])
{
    // R_C_W = R_W_C'
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[3 * c + r] = pose[3 + 3 * r + c];
    for (int r = 0; r < 3; ++r)
        Rt[9 + r] = (-Rt[r]) * pose[0] + (-Rt[3 + r]) * pose[1] + (-Rt[6 + r]) * pose[2];
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 4; ++c)
            M[4 * i + c] = K.k[i] * Rt[3 * c] + K.k[3 + i] * Rt[3 * c + 1] + K.k[6 + i] * Rt[3 * c + 2];
}

__global__ __launch_bounds__(64) void p3p_solve_kernel(const double *__restrict__ kp_rc,
                                                       const double *__restrict__ X, int n,
                                                       const int *__restrict__ n_dev, vohip_K K,
                                                       const int32_t *__restrict__ samples, int h,
                                                       double *__restrict__ Rt_out,
                                                       double *__restrict__ M_out,
                                                       int32_t *__restrict__ zero_counts, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(kp_rc);
    VOHIP_SEQ_STATE(X);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE(samples);
    VOHIP_SEQ_STATE(Rt_out);
    VOHIP_SEQ_STATE(M_out);
    VOHIP_SEQ_STATE_OPT(zero_counts);
    // four lanes per hypothesis: the part the four solutions share is computed by each of them (same
    // operations, same bits), then every lane finishes ONE solution -- the dependent chain of a lane is
    // what this launch lasts, and the four back-substitutions were more than half of it
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int hyp = t >> 2, sol = t & 3;
    if (hyp >= h) return;
    if (zero_counts) zero_counts[t] = 0;   // (for a scoring kernel that adds partial counts)
    if (n_dev) n = *n_dev;  // point count kept on the device (the frame pipeline)
    v3 W[3], bearing[3];
    bool ok = true;
    for (int c = 0; c < 3; ++c) {
        int id = samples[3 * hyp + c];
        if (id < 0 || id >= n) {
            ok = false;
            id = 0;
        }
        W[c] = {X[3 * id], X[3 * id + 1], X[3 * id + 2]};
        bearing[c] = unit_bearing(K, kp_rc[2 * id + 1], kp_rc[2 * id]);
    }
    double pose[12];
    for (int j = 0; j < 12; ++j) pose[j] = 0.0;
    p3p_shared q;
    if (p3p_prepare(W[0], W[1], W[2], bearing[0], bearing[1], bearing[2], q)) {
        const double x = sol == 0 ? q.xr[0] : sol == 1 ? q.xr[1] : sol == 2 ? q.xr[2] : q.xr[3];
        p3p_solution(q, x, pose);
    }
    const double nan = __builtin_nan("");
    double Rt[12], M[12];
    decode_pose(K, pose, Rt, M);
    for (int j = 0; j < 12; ++j) {
        Rt_out[(size_t)hyp * 48 + 12 * sol + j] = ok ? Rt[j] : nan;
        M_out[(size_t)hyp * 48 + 12 * sol + j] = ok ? M[j] : nan;
    }
}

// p3p_solve_kernel for a frame's first hypotheses, in ONE launch with the ordered selection of the valid
// landmark tracks they sample from (main_demo.m:380-381, :387): the solve workgroups (the first nb_solve)
// do not wait for the compacted lists -- they count the valid flags themselves, turn the caller's
// uniform draws into ranks among the valid tracks exactly as the selection does, find the table rows of
// those ranks (per 16 flags a bit mask and a count in LDS, prefix over the counts, binary search) and read
// the sampled points from the track table: the same numbers the compacted lists hold, so the poses are
// the ones p3p_solve_kernel computes from the lists, bit for bit.  The other workgroups are the
// selection (what the scoring kernel behind this launch reads).  S.n <= 16 * SEL_GROUPS.
constexpr int SEL_GROUPS = 512;

__device__ __forceinline__ unsigned nonzero_bytes_to_bits(unsigned w)
{
    w |= w >> 4;
    w |= w >> 2;
    w |= w >> 1;
    w &= 0x01010101u;                     // one bit per non-zero byte
    return (w * 0x01020408u) >> 24;       // ... gathered into the low 4 bits, byte 0 first
}

__global__ __launch_bounds__(256) void p3p_solve_select_kernel(vohip_select_tracked S, vohip_K K, int h, int nb_solve,
                                                               double *__restrict__ Rt_out,
                                                               double *__restrict__ M_out,
                                                               int32_t *__restrict__ zero_counts, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(S.out_xy);
    VOHIP_SEQ_STATE(S.valid);
    VOHIP_SEQ_STATE(S.L_X);
    VOHIP_SEQ_STATE(S.L_id);
    VOHIP_SEQ_STATE(S.mv_rc);
    VOHIP_SEQ_STATE(S.Xv);
    VOHIP_SEQ_STATE(S.idv);
    VOHIP_SEQ_STATE(S.rowv);
    VOHIP_SEQ_STATE(S.n_valid);
    VOHIP_SEQ_STATE(S.idx);
    VOHIP_SEQ_STATE(S.idx_next);
    VOHIP_SEQ_STATE(S.samples);
    VOHIP_SEQ_STATE_OPT(S.n_dev);
    VOHIP_SEQ_PIN(S.u3);
    VOHIP_SEQ_STATE(Rt_out);
    VOHIP_SEQ_STATE(M_out);
    VOHIP_SEQ_STATE_OPT(zero_counts);
    if (S.n_dev) S.n = min(S.n, *S.n_dev);   // (a group of loops: the count of each is known on the device only)
    __shared__ int sh[12];
    __shared__ int sPref[SEL_GROUPS];
    __shared__ unsigned short sMask[SEL_GROUPS];
    __shared__ int sTotal;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nb_solve) {
        // the selection itself
        const int block = blockIdx.x - nb_solve;
        if (block == 0 && tid == 0) *S.idx = *S.idx_next;   // the ids handed out in the previous frame are final now
        int total;
        const uint8_t *__restrict__ valid = S.valid;
        const int r = vohip_dev::ordered_rank_if(S.n, [=](int j) { return valid[j] != 0; }, total, sh, block);
        const int i = block * 256 + tid;
        if (i < S.n && valid[i]) {
            S.mv_rc[2 * r] = S.out_xy[S.ld + i];   // row = y
            S.mv_rc[2 * r + 1] = S.out_xy[i];      // col = x
            S.Xv[3 * r] = S.L_X[3 * i];
            S.Xv[3 * r + 1] = S.L_X[3 * i + 1];
            S.Xv[3 * r + 2] = S.L_X[3 * i + 2];
            S.idv[r] = S.L_id[i];
            S.rowv[r] = i;   // landmark i is row i of the track table
        }
        if (block == 0) {
            if (tid == 0) *S.n_valid = total;
            for (int k = tid; k < S.h0; k += 256)
                vohip_dev::triple_from_uniforms(S.u3 + 3 * (size_t)k, total, S.samples + 3 * (size_t)k);
        }
        return;
    }
    // --- the solve workgroups: where are the valid tracks? ---
    const int groups = (S.n + 15) >> 4;
    for (int g = tid; g < groups; g += 256) {
        const uint4 w = *reinterpret_cast<const uint4 *>(S.valid + 16 * (size_t)g);   // (the flag array is 256-byte aligned and padded)
        unsigned m = nonzero_bytes_to_bits(w.x) | (nonzero_bytes_to_bits(w.y) << 4) | (nonzero_bytes_to_bits(w.z) << 8) |
                     (nonzero_bytes_to_bits(w.w) << 12);
        const int left = S.n - 16 * g;
        if (left < 16) m &= (1u << left) - 1u;
        sMask[g] = (unsigned short)m;
    }
    __syncthreads();
    if (tid < 64) {   // exclusive prefix over the groups' counts: 8 consecutive groups per lane
        int c[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int g = 8 * tid + q;
            c[q] = g < groups ? __popc((unsigned)sMask[g]) : 0;
            sum += c[q];
        }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (tid >= d) incl += t;
        }
        int run = incl - sum;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sPref[8 * tid + q] = run;
            run += c[q];
        }
        if (tid == 63) sTotal = incl;
    }
    __syncthreads();
    const int total = sTotal;
    const int t = blockIdx.x * 256 + tid;
    const int hyp = t >> 2, sol = t & 3;
    const bool live = hyp < h;
    int row = -1;
    if (live && sol < 3) {
        int32_t tri[3];
        vohip_dev::triple_from_uniforms(S.u3 + 3 * (size_t)hyp, total, tri);
        const int target = sol == 0 ? tri[0] : sol == 1 ? tri[1] : tri[2];
        if (target >= 0 && target < total) {
            // the last group whose prefix does not exceed the rank (groups without a valid track share their
            // successor's prefix and are skipped that way)
            int lo = 0, hi = groups - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (sPref[mid] <= target) lo = mid;
                else hi = mid - 1;
            }
            unsigned m = sMask[lo];
            for (int k = target - sPref[lo]; k > 0; --k) m &= m - 1;   // (at most 15 steps)
            row = 16 * lo + (__ffs((int)m) - 1);
        }
    }
    // lanes 4k, 4k + 1, 4k + 2 hold the three rows of hypothesis k
    const int q0 = (tid & 63) & ~3;
    const int id3[3] = {__shfl(row, q0, 64), __shfl(row, q0 + 1, 64), __shfl(row, q0 + 2, 64)};
    if (!live) return;
    if (zero_counts) zero_counts[t] = 0;   // (for a scoring kernel that adds partial counts)
    v3 W[3], bearing[3];
    bool ok = true;
    for (int c = 0; c < 3; ++c) {
        int id = id3[c];
        if (id < 0) {
            ok = false;
            id = 0;
        }
        W[c] = {S.L_X[3 * id], S.L_X[3 * id + 1], S.L_X[3 * id + 2]};
        bearing[c] = unit_bearing(K, S.out_xy[id], S.out_xy[S.ld + id]);
    }
    double pose[12];
    for (int j = 0; j < 12; ++j) pose[j] = 0.0;
    p3p_shared q;
    if (p3p_prepare(W[0], W[1], W[2], bearing[0], bearing[1], bearing[2], q)) {
        const double x = sol == 0 ? q.xr[0] : sol == 1 ? q.xr[1] : sol == 2 ? q.xr[2] : q.xr[3];
        p3p_solution(q, x, pose);
    }
    const double nan = __builtin_nan("");
    double Rt[12], M[12];
    decode_pose(K, pose, Rt, M);
    for (int j = 0; j < 12; ++j) {
        Rt_out[(size_t)hyp * 48 + 12 * sol + j] = ok ? Rt[j] : nan;
        M_out[(size_t)hyp * 48 + 12 * sol + j] = ok ? M[j] : nan;
    }
}

// Stand-alone p3p.m for `count` problems, reference output layout (3x16 column-major).
__global__ __launch_bounds__(64) void p3p_batch_kernel(const double *__restrict__ world,
                                                       const double *__restrict__ bearings, int count,
                                                       double *__restrict__ poses_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double *w = world + 9 * (size_t)i, *b = bearings + 9 * (size_t)i;
    double poses[4][12];
    p3p_solve({w[0], w[1], w[2]}, {w[3], w[4], w[5]}, {w[6], w[7], w[8]}, {b[0], b[1], b[2]},
              {b[3], b[4], b[5]}, {b[6], b[7], b[8]}, poses);
    for (int s = 0; s < 4; ++s)
        for (int j = 0; j < 12; ++j) poses_out[(size_t)i * 48 + 12 * s + j] = poses[s][j];
}

#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------
// Inlier test, shared by the scoring and the mask kernels so that popcount(mask) == count.
// m = K [R | t] row-major 3x4.  13 FMA + 2 MUL + 1 CMP.
// ------------------------------------------------------------------------------------------
using vohip_dev::is_inlier;   // inlier_test.h

// Inlier test with the 9 rotation-part coefficients as wave-uniform scalars (SGPR operands of
// v_fma_f64: one scalar operand per VALU instruction is free) and the 3 translation-part
// coefficients as vector registers.  Same arithmetic, operation for operation, as is_inlier.
struct pose_s {
    double r[9];  // m[0..2], m[4..6], m[8..10]
};
__device__ __forceinline__ bool is_inlier_sv(const pose_s &ms, double t0, double t1, double t2, double X,
                                             double Y, double Z, double u, double v)
{
    const double x = __builtin_fma(ms.r[0], X, __builtin_fma(ms.r[1], Y, __builtin_fma(ms.r[2], Z, t0)));
    const double y = __builtin_fma(ms.r[3], X, __builtin_fma(ms.r[4], Y, __builtin_fma(ms.r[5], Z, t1)));
    const double z = __builtin_fma(ms.r[6], X, __builtin_fma(ms.r[7], Y, __builtin_fma(ms.r[8], Z, t2)));
    const double dx = __builtin_fma(-u, z, x);
    const double dy = __builtin_fma(-v, z, y);
    const double e = __builtin_fma(dy, dy, dx * dx);
    const double zz = z * z;
    return __builtin_fma(-100.0, zz, e) < 0.0;
}

constexpr int SCORE_WAVES = 8;  // default hypotheses per workgroup (chunk granularity of the host loop)

// One wavefront per hypothesis, WAVES hypotheses per workgroup.  The point arrays are staged
// through LDS in tiles of TILE points, structure-of-arrays, double-buffered: the global loads of
// tile t+1 are issued before the arithmetic on tile t and written to the other LDS buffer after it.
template <int WAVES, int TILE>
__global__ __launch_bounds__(WAVES * 64) void p3p_score_kernel(
    const double *__restrict__ kp_rc, const double *__restrict__ X, int n, const int *__restrict__ n_dev,
    const double *__restrict__ M, int h, int32_t *__restrict__ counts, vohip_seqs SQ)
{
    constexpr int NT = WAVES * 64;
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(kp_rc);
    VOHIP_SEQ_STATE(X);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE(M);
    VOHIP_SEQ_STATE(counts);
    if (n_dev) n = *n_dev;
    constexpr int PER_X = (3 * TILE + NT - 1) / NT;
    constexpr int PER_K = (2 * TILE + NT - 1) / NT;
    __shared__ double lds[2][5 * TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hyp_raw = blockIdx.x * WAVES + wave;
    const int hyp = hyp_raw < h ? hyp_raw : h - 1;  // surplus waves redo the last hypothesis, no store

    pose_s ms[4];
    double tv[4][3];
    {
        const double *mp = M + (size_t)hyp * 48;  // wave-uniform address: scalar loads
        // translation column through a per-lane address so that it lives in vector registers
        const int hv = min(blockIdx.x * WAVES + (tid >> 6), h - 1);
        const double *mv = M + (size_t)hv * 48;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ms[s].r[3 * i] = mp[12 * s + 4 * i];
                ms[s].r[3 * i + 1] = mp[12 * s + 4 * i + 1];
                ms[s].r[3 * i + 2] = mp[12 * s + 4 * i + 2];
                tv[s][i] = mv[12 * s + 4 * i + 3];
            }
        }
    }
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;

    double rx[PER_X], rk[PER_K];
    auto load_tile = [&](int base) {
        const int tn = min(TILE, n - base);
#pragma unroll
        for (int i = 0; i < PER_X; ++i) {
            const int idx = tid + i * NT;
            // points past the end are NaN: they fail the test, so the scoring loop needs no bound check
            rx[i] = idx < 3 * tn ? X[(size_t)3 * base + idx] : __builtin_nan("");
        }
#pragma unroll
        for (int i = 0; i < PER_K; ++i) {
            const int idx = tid + i * NT;
            rk[i] = idx < 2 * tn ? kp_rc[(size_t)2 * base + idx] : 0.0;
        }
    };
    auto store_tile = [&](int buf) {
        double *L = lds[buf];
#pragma unroll
        for (int i = 0; i < PER_X; ++i) {
            const int idx = tid + i * NT;
            const int p = idx / 3, c = idx - 3 * p;
            if (idx < 3 * TILE) L[c * TILE + p] = rx[i];
        }
#pragma unroll
        for (int i = 0; i < PER_K; ++i) {
            const int idx = tid + i * NT;
            // kp_rc pairs are (row = v, col = u): array 3 = v, array 4 = u
            if (idx < 2 * TILE) L[(3 + (idx & 1)) * TILE + (idx >> 1)] = rk[i];
        }
    };

    const int ntiles = (n + TILE - 1) / TILE;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int base = t * TILE;
        const int tn = min(TILE, n - base);
        if (t + 1 < ntiles) load_tile(base + TILE);
        const double *L = lds[t & 1] + lane;
        // Wave-uniform trip count (the tile is NaN-padded, so whole groups of 64 points are tested
        // without a bound check), two groups per trip: the LDS reads of one group are in flight
        // while the other one is tested.
        const int pairs = (tn + 127) >> 7;
        double ax = L[0], ay = L[TILE], az = L[2 * TILE], av = L[3 * TILE], au = L[4 * TILE];
        for (int gi = 0; gi < pairs; ++gi) {
            const int o1 = gi * 128 + 64;
            const double bx = L[o1], by = L[TILE + o1], bz = L[2 * TILE + o1];
            const double bv = L[3 * TILE + o1], bu = L[4 * TILE + o1];
            __builtin_amdgcn_sched_barrier(0);   // keep these reads ahead of the arithmetic below
            c0 += __popcll(__ballot(is_inlier_sv(ms[0], tv[0][0], tv[0][1], tv[0][2], ax, ay, az, au, av)));
            c1 += __popcll(__ballot(is_inlier_sv(ms[1], tv[1][0], tv[1][1], tv[1][2], ax, ay, az, au, av)));
            c2 += __popcll(__ballot(is_inlier_sv(ms[2], tv[2][0], tv[2][1], tv[2][2], ax, ay, az, au, av)));
            c3 += __popcll(__ballot(is_inlier_sv(ms[3], tv[3][0], tv[3][1], tv[3][2], ax, ay, az, au, av)));
            const int o2 = gi + 1 < pairs ? (gi + 1) * 128 : 0;   // the last prefetch is a harmless re-read
            ax = L[o2];
            ay = L[TILE + o2];
            az = L[2 * TILE + o2];
            av = L[3 * TILE + o2];
            au = L[4 * TILE + o2];
            __builtin_amdgcn_sched_barrier(0);
            c0 += __popcll(__ballot(is_inlier_sv(ms[0], tv[0][0], tv[0][1], tv[0][2], bx, by, bz, bu, bv)));
            c1 += __popcll(__ballot(is_inlier_sv(ms[1], tv[1][0], tv[1][1], tv[1][2], bx, by, bz, bu, bv)));
            c2 += __popcll(__ballot(is_inlier_sv(ms[2], tv[2][0], tv[2][1], tv[2][2], bx, by, bz, bu, bv)));
            c3 += __popcll(__ballot(is_inlier_sv(ms[3], tv[3][0], tv[3][1], tv[3][2], bx, by, bz, bu, bv)));
        }
        if (t + 1 < ntiles) store_tile((t + 1) & 1);
        __syncthreads();
    }
    if (hyp_raw < h && lane == 0) {
        int4 out = {c0, c1, c2, c3};
        *reinterpret_cast<int4 *>(counts + 4 * (size_t)hyp) = out;
    }
}

// Few hypotheses (a frame's first 104): one wavefront per hypothesis would leave most of the GPU idle
// and last as long as one wavefront needs for all the points, so `split` wavefronts share a
// hypothesis, each tests its share of the points straight from global memory (the point arrays are
// small and stay in the caches) and adds its four counts to the hypothesis' (integer atomics: the
// sums do not depend on the order).  counts must be zero on entry (p3p_solve_kernel clears them).
__global__ __launch_bounds__(256) void p3p_score_split_kernel(const double *__restrict__ kp_rc,
                                                              const double *__restrict__ X, int n,
                                                              const int *__restrict__ n_dev,
                                                              const double *__restrict__ M, int h, int split,
                                                              int32_t *__restrict__ counts, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(kp_rc);
    VOHIP_SEQ_STATE(X);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE(M);
    VOHIP_SEQ_STATE(counts);
    if (n_dev) n = *n_dev;
    const int lane = threadIdx.x & 63;
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int hyp = gw / split, part = gw - hyp * split;
    if (hyp >= h) return;
    const int groups = (n + 63) >> 6, per = (groups + split - 1) / split;
    const int g0 = part * per, g1 = min(groups, g0 + per);
    if (g0 >= g1) return;
    pose_s ms[4];
    double tv[4][3];
    {
        const double *mp = M + (size_t)hyp * 48;  // wave-uniform address: scalar loads
        const double *mv = M + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) / split) * 48;   // per-lane address
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                ms[s].r[3 * i] = mp[12 * s + 4 * i];
                ms[s].r[3 * i + 1] = mp[12 * s + 4 * i + 1];
                ms[s].r[3 * i + 2] = mp[12 * s + 4 * i + 2];
                tv[s][i] = mv[12 * s + 4 * i + 3];
            }
        }
    }
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const double nan = __builtin_nan("");
    auto load = [&](int g, double &x, double &y, double &z, double &u, double &v) {
        const int j = g * 64 + lane;
        const bool in = j < n;
        const size_t jj = in ? j : 0;
        x = X[3 * jj];
        y = X[3 * jj + 1];
        z = X[3 * jj + 2];
        v = kp_rc[2 * jj];       // (row, col) pairs: row = v, col = u
        u = kp_rc[2 * jj + 1];
        if (!in) x = nan;        // fails the test
    };
    double ax, ay, az, au, av;
    load(g0, ax, ay, az, au, av);
    for (int g = g0; g < g1; ++g) {
        double bx = 0, by = 0, bz = 0, bu = 0, bv = 0;
        if (g + 1 < g1) load(g + 1, bx, by, bz, bu, bv);   // in flight while this group is tested
        c0 += __popcll(__ballot(is_inlier_sv(ms[0], tv[0][0], tv[0][1], tv[0][2], ax, ay, az, au, av)));
        c1 += __popcll(__ballot(is_inlier_sv(ms[1], tv[1][0], tv[1][1], tv[1][2], ax, ay, az, au, av)));
        c2 += __popcll(__ballot(is_inlier_sv(ms[2], tv[2][0], tv[2][1], tv[2][2], ax, ay, az, au, av)));
        c3 += __popcll(__ballot(is_inlier_sv(ms[3], tv[3][0], tv[3][1], tv[3][2], ax, ay, az, au, av)));
        ax = bx;
        ay = by;
        az = bz;
        au = bu;
        av = bv;
    }
    if (lane == 0) {
        int32_t *c = counts + 4 * (size_t)hyp;
        if (c0) atomicAdd(c, c0);
        if (c1) atomicAdd(c + 1, c1);
        if (c2) atomicAdd(c + 2, c2);
        if (c3) atomicAdd(c + 3, c3);
    }
}

struct pose_m {
    double m[12];
};

__global__ __launch_bounds__(256) void inlier_mask_kernel(const double *__restrict__ kp_rc,
                                                          const double *__restrict__ X, int n, pose_m P,
                                                          uint8_t *__restrict__ mask)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const bool in = is_inlier(P.m, X[3 * j], X[3 * j + 1], X[3 * j + 2], kp_rc[2 * j + 1], kp_rc[2 * j]);
        mask[j] = in ? 1 : 0;
    }
}

// The same mask for a pose that lives in device memory (m = K [R | t], row-major 3x4) and a
// point count that may live there too.
__global__ __launch_bounds__(256) void inlier_mask_dev_kernel(const double *__restrict__ kp_rc,
                                                              const double *__restrict__ X, int n,
                                                              const int *__restrict__ n_dev,
                                                              const double *__restrict__ m12,
                                                              uint8_t *__restrict__ mask)
{
    if (n_dev) n = *n_dev;
    double m[12];
    for (int q = 0; q < 12; ++q) m[q] = m12[q];
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const bool in = is_inlier(m, X[3 * j], X[3 * j + 1], X[3 * j + 2], kp_rc[2 * j + 1], kp_rc[2 * j]);
        mask[j] = in ? 1 : 0;
    }
}

// K [R | t] on the host with the same statements as decode_pose (no contraction on the host
// build either: the expression is a plain sum of products evaluated left to right).
void host_fused_projection(const double K[9], const double Rt[12], double M[12])
{
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 4; ++c) {
            volatile double a = K[i] * Rt[3 * c];
            volatile double b = K[3 + i] * Rt[3 * c + 1];
            volatile double d = K[6 + i] * Rt[3 * c + 2];
            volatile double ab = a + b;
            M[4 * i + c] = ab + d;
        }
}

// ransacLocalization.m:118-129
double adaptive_num_iterations(int max_num_inliers, int n)
{
    const double confidence = 0.9999, upper = 0.95;
    double outlier_ratio = 1.0 - (double)max_num_inliers / (double)n;
    if (upper < outlier_ratio) outlier_ratio = upper;
    double it = std::log(1.0 - confidence) / std::log(1.0 - std::pow(1.0 - outlier_ratio, 3.0));
    if (it > 15000.0) it = 15000.0;
    if (it < 100.0) it = 100.0;
    return it;
}

int launch_solve(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const double K[9],
                 const int32_t *d_samples, int h)
{
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_RT], (size_t)h * 48 * sizeof(double)));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_M], (size_t)h * 48 * sizeof(double)));
    ctx->ransac_h = h;
    const vohip_K Kv = vohip_make_K(K);
    hipLaunchKernelGGL(p3p_solve_kernel, dim3((h + 15) / 16), dim3(64), 0, ctx->stream, d_kp_rc, d_X, n,
                       (const int *)nullptr, Kv, d_samples, h, (double *)ctx->bufs[B_RANSAC_RT].ptr,
                       (double *)ctx->bufs[B_RANSAC_M].ptr, (int32_t *)nullptr, vohip_seqs());
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

template <int WAVES, int TILE>
void launch_score_t(hipStream_t st, const double *d_kp_rc, const double *d_X, int n, const int *d_n,
                    const double *d_M, int h, int32_t *d_counts, const vohip_seqs &SQ)
{
    hipLaunchKernelGGL((p3p_score_kernel<WAVES, TILE>), dim3((h + WAVES - 1) / WAVES, vohip_seq_count(SQ)), dim3(WAVES * 64), 0,
                       st, d_kp_rc, d_X, n, d_n, d_M, h, d_counts, SQ);
}

int launch_score_variant(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const double *d_M,
                         int h, int32_t *d_counts, const int *d_n = nullptr, const vohip_seqs &SQ = vohip_seqs())
{
    // Geometry: many hypotheses -> 16 per workgroup share each staged tile (fewest L2->LDS bytes);
    // few hypotheses -> small workgroups so that more CUs take part.  Measured on MI355X:
    // 4096 x 8192: variant 2 = 77 us, variant 0 = 83 us; 104 x 2000: variant 3 = 10.7 us, 2 = 21 us.
    int variant = ctx->opt_ransac_variant;
    if (variant < 0) variant = h >= 2048 ? 2 : (h < 512 ? 3 : 0);
    switch (variant) {
    case 1: launch_score_t<8, 512>(ctx->stream, d_kp_rc, d_X, n, d_n, d_M, h, d_counts, SQ); break;
    case 2: launch_score_t<16, 1024>(ctx->stream, d_kp_rc, d_X, n, d_n, d_M, h, d_counts, SQ); break;
    case 3: launch_score_t<4, 512>(ctx->stream, d_kp_rc, d_X, n, d_n, d_M, h, d_counts, SQ); break;
    case 4: launch_score_t<16, 512>(ctx->stream, d_kp_rc, d_X, n, d_n, d_M, h, d_counts, SQ); break;
    case 5: launch_score_t<4, 256>(ctx->stream, d_kp_rc, d_X, n, d_n, d_M, h, d_counts, SQ); break;
    default: launch_score_t<8, 1024>(ctx->stream, d_kp_rc, d_X, n, d_n, d_M, h, d_counts, SQ); break;
    }
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int launch_count(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, int h, int32_t *d_counts)
{
    return launch_score_variant(ctx, d_kp_rc, d_X, n, (const double *)ctx->bufs[B_RANSAC_M].ptr, h, d_counts);
}

}  // namespace

// Solve + score hypotheses [first, first + h) for points whose count may live on the device.
int vohip_ransac_enqueue(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const int *d_n,
                         const double K[9], const int32_t *d_samples, int first, int h, int32_t *d_counts,
                         double *d_Rt_all, double *d_M_all, const vohip_select_tracked *select, const vohip_seqs *seqs)
{
    const vohip_seqs SQ = seqs ? *seqs : vohip_seqs();
    const int ny = vohip_seq_count(SQ);
    if (ny < 1) return VOHIP_OK;
    // the candidate poses go to the caller's arrays (48 doubles per hypothesis each, hypothesis 0 first)
    VOHIP_REQUIRE(ctx, d_Rt_all && d_M_all, "NULL pose buffers");
    double *d_Rt = d_Rt_all + (size_t)first * 48;
    double *d_M = d_M_all + (size_t)first * 48;
    const vohip_K Kv = vohip_make_K(K);
    // few hypotheses: several wavefronts share each (their partial counts are added to counts the solve
    // kernel cleared); many: one wavefront per hypothesis over LDS-staged tiles
    const bool split_points = h < 512 && ctx->opt_ransac_variant < 0;
    if (select) {
        // the frame's first hypotheses and the selection of the tracks they sample from, as one launch
        VOHIP_REQUIRE(ctx, first == 0 && h <= select->h0 && select->n <= 16 * SEL_GROUPS && select->n >= 0 &&
                               (reinterpret_cast<uintptr_t>(select->valid) & 15) == 0,
                      "the selection cannot ride with these hypotheses");
        const int nb_solve = (4 * h + 255) / 256, nb_select = select->n < 1 ? 1 : (select->n + 255) / 256;
        hipLaunchKernelGGL(p3p_solve_select_kernel, dim3(nb_solve + nb_select, ny), dim3(256), 0, ctx->stream, *select, Kv, h,
                           nb_solve, d_Rt, d_M, split_points ? d_counts : (int32_t *)nullptr, SQ);
    } else {
        hipLaunchKernelGGL(p3p_solve_kernel, dim3((h + 15) / 16, ny), dim3(64), 0, ctx->stream, d_kp_rc, d_X, n, d_n, Kv,
                           d_samples + 3 * (size_t)first, h, d_Rt, d_M,
                           split_points ? d_counts + 4 * (size_t)first : (int32_t *)nullptr, SQ);
    }
    if (split_points) {
        const int split = 1024 / h < 1 ? 1 : (1024 / h > 16 ? 16 : 1024 / h);
        hipLaunchKernelGGL(p3p_score_split_kernel, dim3((h * split + 3) / 4, ny), dim3(256), 0, ctx->stream, d_kp_rc, d_X, n,
                           d_n, d_M, h, split, d_counts + 4 * (size_t)first, SQ);
        VOHIP_HIP(ctx, hipGetLastError());
    } else {
        VOHIP_TRY(launch_score_variant(ctx, d_kp_rc, d_X, n, d_M, h, d_counts + 4 * (size_t)first, d_n, SQ));
    }
    return VOHIP_OK;
}

// Inlier counts of `slots` x 4 candidate poses (d_M: 48 doubles per slot, K [R | t] row-major 3x4 each) over n
// correspondences: d_counts 4 ints per slot.  NaN poses count nothing.
int vohip_ransac_score_enqueue(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const double *d_M,
                               int slots, int32_t *d_counts)
{
    return launch_score_variant(ctx, d_kp_rc, d_X, n, d_M, slots, d_counts);
}

int vohip_ransac_mask_enqueue(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const int *d_n,
                              const double *d_M12, uint8_t *d_mask)
{
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(inlier_mask_dev_kernel, dim3(blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks)), dim3(256), 0,
                       ctx->stream, d_kp_rc, d_X, n, d_n, d_M12, d_mask);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

double vohip_ransac_adaptive_iterations(int max_num_inliers, int n) { return adaptive_num_iterations(max_num_inliers, n); }

// Device address of this GPU's copy of `scratch`, taken once per device (called by vohip_create with
// the context's device current).
int vohip_ransac_scratch_init(vohip_ctx *ctx)
{
    VOHIP_REQUIRE(ctx, ctx->device >= 0 && ctx->device < NOALLOC_MAX_DEVICES, "device index beyond the scratch table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[ctx->device]) return VOHIP_OK;
    double *ptr = nullptr;
    VOHIP_HIP(ctx, hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    VOHIP_REQUIRE(ctx, ptr != nullptr, "hipGetSymbolAddress returned no address for the scratch array");
    g_scratch_ptr[ctx->device] = ptr;
    return VOHIP_OK;
}

extern "C" {

int vohip_ransac_p3p_solve_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                               const double K[9], const int32_t *d_samples, int h)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_kp_rc && d_X && K && d_samples, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 3, "need at least 3 points");
    VOHIP_REQUIRE(ctx, h >= 1, "need at least 1 hypothesis");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    return launch_solve(ctx, d_kp_rc, d_X, n, K, d_samples, h);
}

int vohip_ransac_p3p_count_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, int h,
                               int32_t *d_counts)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_kp_rc && d_X && d_counts, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1, "need at least 1 point");
    VOHIP_REQUIRE(ctx, h >= 1 && h <= ctx->ransac_h, "h exceeds the hypotheses solved by the last solve call");
    return launch_count(ctx, d_kp_rc, d_X, n, h, d_counts);
}

int vohip_ransac_p3p_score_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                               const double K[9], const int32_t *d_samples, int h, int32_t *d_counts)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_counts, "NULL argument");
    VOHIP_TRY(vohip_ransac_p3p_solve_dev(ctx, d_kp_rc, d_X, n, K, d_samples, h));
    return launch_count(ctx, d_kp_rc, d_X, n, h, d_counts);
}

int vohip_ransac_p3p_score_noalloc(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                                   const double K[9], const int32_t *d_samples, int h, int32_t *d_counts)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_kp_rc && d_X && K && d_samples && d_counts, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 3, "need at least 3 points");
    VOHIP_REQUIRE(ctx, h >= 1, "need at least 1 hypothesis");
    if (h > NOALLOC_MAX_H)
        return vohip_fail(ctx, VOHIP_ERR_INVALID_ARGUMENT,
                          "%d hypotheses exceed the %d the static scratch holds (use vohip_ransac_p3p_score_dev)", h,
                          NOALLOC_MAX_H);
    VOHIP_REQUIRE(ctx, ctx->device >= 0 && ctx->device < NOALLOC_MAX_DEVICES && g_scratch_ptr[ctx->device],
                  "the scratch address of this device was not taken (vohip_create does it)");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    // every hypothesis [0, h) is written by the solve kernel before the scoring kernel reads it, so
    // nothing of an earlier call's contents has to be cleared
    double *d_Rt = g_scratch_ptr[ctx->device];
    double *d_M = d_Rt + (size_t)NOALLOC_MAX_H * 48;
    const vohip_K Kv = vohip_make_K(K);
    hipLaunchKernelGGL(p3p_solve_kernel, dim3((h + 15) / 16), dim3(64), 0, ctx->stream, d_kp_rc, d_X, n,
                       (const int *)nullptr, Kv, d_samples, h, d_Rt, d_M, (int32_t *)nullptr, vohip_seqs());
    VOHIP_HIP(ctx, hipGetLastError());
    return launch_score_variant(ctx, d_kp_rc, d_X, n, d_M, h, d_counts);
}

int vohip_ransac_p3p_poses(vohip_ctx *ctx, int first, int count, double *Rt)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, Rt, "NULL argument");
    VOHIP_REQUIRE(ctx, first >= 0 && count >= 0 && first + count <= ctx->ransac_h,
                  "pose range outside the last scored batch");
    if (count == 0) return VOHIP_OK;
    VOHIP_HIP(ctx, hipMemcpyAsync(Rt, (const double *)ctx->bufs[B_RANSAC_RT].ptr + (size_t)first * 48,
                                  (size_t)count * 48 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_ransac_inlier_mask_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                                 const double K[9], const double Rt[12], uint8_t *d_mask)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_kp_rc && d_X && K && Rt && d_mask, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1, "need at least 1 point");
    pose_m P;
    host_fused_projection(K, Rt, P.m);
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(inlier_mask_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, ctx->stream,
                       d_kp_rc, d_X, n, P, d_mask);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int vohip_ransac_p3p(vohip_ctx *ctx, const double *kp_rc, const double *X, int n, const double K[9],
                     const int32_t *samples, int hcap, double R_C_W[9], double t_C_W[3], uint8_t *mask,
                     double *max_hist, double *iter_hist, int *n_iter)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, kp_rc && X && K && samples && R_C_W && t_C_W && mask && max_hist && iter_hist && n_iter,
                  "NULL argument");
    *n_iter = 0;
    if (n < 3) return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "P3P needs at least 3 correspondences, got %d", n);
    VOHIP_REQUIRE(ctx, hcap >= 1, "need at least 1 sample triple");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));

    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_KP], (size_t)2 * n * sizeof(double)));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_X], (size_t)3 * n * sizeof(double)));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_SAMPLES], (size_t)3 * hcap * sizeof(int32_t)));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_COUNTS], (size_t)4 * hcap * sizeof(int32_t)));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_MASK], (size_t)n));
    VOHIP_TRY(vohip_reserve_pinned(ctx, ctx->bufs[H_RANSAC_COUNTS], (size_t)4 * hcap * sizeof(int32_t)));
    // the pose buffers must hold every chunk at its own offset
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_RT], (size_t)hcap * 48 * sizeof(double)));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_M], (size_t)hcap * 48 * sizeof(double)));

    double *d_kp = (double *)ctx->bufs[B_RANSAC_KP].ptr;
    double *d_X = (double *)ctx->bufs[B_RANSAC_X].ptr;
    int32_t *d_samples = (int32_t *)ctx->bufs[B_RANSAC_SAMPLES].ptr;
    int32_t *d_counts = (int32_t *)ctx->bufs[B_RANSAC_COUNTS].ptr;
    uint8_t *d_mask = (uint8_t *)ctx->bufs[B_RANSAC_MASK].ptr;
    int32_t *h_counts = (int32_t *)ctx->bufs[H_RANSAC_COUNTS].ptr;
    double *d_Rt = (double *)ctx->bufs[B_RANSAC_RT].ptr;
    double *d_M = (double *)ctx->bufs[B_RANSAC_M].ptr;
    hipStream_t st = ctx->stream;

    VOHIP_HIP(ctx, hipMemcpyAsync(d_kp, kp_rc, (size_t)2 * n * sizeof(double), hipMemcpyHostToDevice, st));
    VOHIP_HIP(ctx, hipMemcpyAsync(d_X, X, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, st));
    VOHIP_HIP(ctx, hipMemcpyAsync(d_samples, samples, (size_t)3 * hcap * sizeof(int32_t), hipMemcpyHostToDevice, st));
    const vohip_K Kv = vohip_make_K(K);

    // Replay state of ransacLocalization.m:37,42-44.
    int max_num_inliers = 0, best_h = -1, best_s = 0;
    double num_iterations = INFINITY;
    int i = 1;          // reference loop counter
    int evaluated = 0;  // hypotheses scored so far
    bool exhausted = false;
    while (num_iterations > (double)i) {
        if (i > evaluated) {
            if (evaluated >= hcap) {
                exhausted = true;
                break;
            }
            // chunk: at least the 99 iterations the clamp at :129 guarantees, then up to the
            // current bound, in multiples of the workgroup's 8 hypotheses.
            int want = evaluated == 0 ? 104 : (int)std::ceil(num_iterations) - 1 - evaluated;
            if (want < 64) want = 64;
            want = (want + SCORE_WAVES - 1) / SCORE_WAVES * SCORE_WAVES;
            if (want > hcap - evaluated) want = hcap - evaluated;
            hipLaunchKernelGGL(p3p_solve_kernel, dim3((want + 15) / 16), dim3(64), 0, st, d_kp, d_X, n,
                               (const int *)nullptr, Kv, d_samples + 3 * (size_t)evaluated, want, d_Rt + (size_t)evaluated * 48,
                               d_M + (size_t)evaluated * 48, (int32_t *)nullptr, vohip_seqs());
            VOHIP_TRY(launch_score_variant(ctx, d_kp, d_X, n, d_M + (size_t)evaluated * 48, want,
                                           d_counts + 4 * (size_t)evaluated));
            VOHIP_HIP(ctx, hipMemcpyAsync(h_counts + 4 * (size_t)evaluated, d_counts + 4 * (size_t)evaluated,
                                          (size_t)4 * want * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            VOHIP_HIP(ctx, hipStreamSynchronize(st));
            evaluated += want;
        }
        // :77-101  pick the solution with the strict '>' rule, lowest index wins ties
        const int32_t *c = h_counts + 4 * (size_t)(i - 1);
        int s_best = 0;
        for (int s = 1; s < 4; ++s)
            if (c[s] > c[s_best]) s_best = s;
        const int cnt = c[s_best];
        if (cnt > max_num_inliers && cnt >= 30) {  // :104,109-110
            max_num_inliers = cnt;
            best_h = i - 1;
            best_s = s_best;
        }
        num_iterations = adaptive_num_iterations(max_num_inliers, n);
        iter_hist[i - 1] = num_iterations;
        max_hist[i - 1] = (double)max_num_inliers;
        ++i;
    }
    ctx->ransac_h = evaluated;
    *n_iter = i - 1;

    if (best_h < 0) {
        std::memset(mask, 0, (size_t)n);
        if (exhausted)
            return vohip_fail(ctx, VOHIP_SAMPLES_EXHAUSTED, "sample triples ran out after %d iterations, no model", i - 1);
        return vohip_fail(ctx, VOHIP_NO_MODEL, "no hypothesis reached 30 inliers in %d iterations", i - 1);
    }
    double Rt[12];
    VOHIP_HIP(ctx, hipMemcpyAsync(Rt, d_Rt + (size_t)best_h * 48 + 12 * best_s, 12 * sizeof(double),
                                  hipMemcpyDeviceToHost, st));
    VOHIP_HIP(ctx, hipStreamSynchronize(st));
    std::memcpy(R_C_W, Rt, 9 * sizeof(double));
    std::memcpy(t_C_W, Rt + 9, 3 * sizeof(double));
    // mask of the winner with the scoring kernel's own projection matrix
    pose_m P;
    VOHIP_HIP(ctx, hipMemcpyAsync(P.m, d_M + (size_t)best_h * 48 + 12 * best_s, 12 * sizeof(double),
                                  hipMemcpyDeviceToHost, st));
    VOHIP_HIP(ctx, hipStreamSynchronize(st));
    const int blocks = (n + 255) / 256;
    hipLaunchKernelGGL(inlier_mask_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256), 0, st, d_kp, d_X, n,
                       P, d_mask);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_HIP(ctx, hipMemcpyAsync(mask, d_mask, (size_t)n, hipMemcpyDeviceToHost, st));
    VOHIP_HIP(ctx, hipStreamSynchronize(st));
    if (exhausted)
        return vohip_fail(ctx, VOHIP_SAMPLES_EXHAUSTED, "sample triples ran out after %d iterations", i - 1);
    return VOHIP_OK;
}

int vohip_p3p(vohip_ctx *ctx, const double *world, const double *bearings, int count, double *poses)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, world && bearings && poses, "NULL argument");
    VOHIP_REQUIRE(ctx, count >= 0, "negative count");
    if (count == 0) return VOHIP_OK;
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_b = (size_t)count * 9 * sizeof(double), out_b = (size_t)count * 48 * sizeof(double);
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_STAGE0], in_b));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_STAGE1], in_b));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_STAGE2], out_b));
    hipStream_t st = ctx->stream;
    VOHIP_HIP(ctx, hipMemcpyAsync(ctx->bufs[B_STAGE0].ptr, world, in_b, hipMemcpyHostToDevice, st));
    VOHIP_HIP(ctx, hipMemcpyAsync(ctx->bufs[B_STAGE1].ptr, bearings, in_b, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(p3p_batch_kernel, dim3((count + 63) / 64), dim3(64), 0, st,
                       (const double *)ctx->bufs[B_STAGE0].ptr, (const double *)ctx->bufs[B_STAGE1].ptr, count,
                       (double *)ctx->bufs[B_STAGE2].ptr);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_HIP(ctx, hipMemcpyAsync(poses, ctx->bufs[B_STAGE2].ptr, out_b, hipMemcpyDeviceToHost, st));
    VOHIP_HIP(ctx, hipStreamSynchronize(st));
    return VOHIP_OK;
}

}  // extern "C"
