// SE(3) helpers for device code: closed-form exp / log of rigid transforms (the reference calls
// MATLAB's expm / logm on 4x4 matrices, twist2HomogMatrix.m:14, HomogMatrix2twist.m:14) and
// small fixed-size linear algebra.  twist = [v; w], linear part first.
#pragma once

#include <hip/hip_runtime.h>

namespace vohip_dev {

// A = sin(th)/th, B = (1-cos th)/th^2, C = (th-sin th)/th^3, D = (1 - A/(2B))/th^2.
// Taylor series to th^10 below th^2 = 1e-2 (truncation < 1e-17 relative, no cancellation).
__device__ __forceinline__ void so3_coeffs(double th2, double &A, double &B, double &C, double &D)
{
    if (th2 < 1e-2) {
        A = 1.0 + th2 * (-1.0 / 6.0 + th2 * (1.0 / 120.0 + th2 * (-1.0 / 5040.0 + th2 * (1.0 / 362880.0 - th2 / 39916800.0))));
        B = 0.5 + th2 * (-1.0 / 24.0 + th2 * (1.0 / 720.0 + th2 * (-1.0 / 40320.0 + th2 * (1.0 / 3628800.0 - th2 / 479001600.0))));
        C = 1.0 / 6.0 + th2 * (-1.0 / 120.0 + th2 * (1.0 / 5040.0 + th2 * (-1.0 / 362880.0 + th2 * (1.0 / 39916800.0 - th2 / 6227020800.0))));
        D = 1.0 / 12.0 + th2 * (1.0 / 720.0 + th2 * (1.0 / 30240.0 + th2 * (1.0 / 1209600.0 + th2 / 47900160.0)));
    } else {
        const double th = sqrt(th2);
        double s, c;
        sincos(th, &s, &c);
        A = s / th;
        B = (1.0 - c) / th2;
        C = (th - s) / (th2 * th);
        D = (1.0 - A / (2.0 * B)) / th2;
    }
}

// Rigid transform as rotation (row-major R[3][3]) + translation.
struct rigid {
    double R[3][3];
    double t[3];
};

__device__ __forceinline__ rigid twist_to_rigid(const double tw[6])
{
    const double *v = tw, *w = tw + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B, C, D;
    so3_coeffs(th2, A, B, C, D);
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    rigid T;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double W2 = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
            const double I = (i == j) ? 1.0 : 0.0;
            T.R[i][j] = I + A * W[i][j] + B * W2;
            t += (I + B * W[i][j] + C * W2) * v[j];
        }
        T.t[i] = t;
    }
    return T;
}

__device__ __forceinline__ void rigid_to_twist(const rigid &T, double tw[6])
{
    const double (*R)[3] = T.R;
    const double s[3] = {0.5 * (R[2][1] - R[1][2]), 0.5 * (R[0][2] - R[2][0]), 0.5 * (R[1][0] - R[0][1])};
    const double sn = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    const double cs = 0.5 * (R[0][0] + R[1][1] + R[2][2] - 1.0);
    const double th = atan2(sn, cs);
    double w[3];
    if (sn < 1e-6 && cs > 0.0) {
        const double k = 1.0 + th * th / 6.0 + 7.0 * th * th * th * th / 360.0;
        for (int i = 0; i < 3; ++i) w[i] = k * s[i];
    } else if (sn < 1e-6) {
        // rotation by ~pi: axis from the symmetric part
        const double d[3] = {R[0][0], R[1][1], R[2][2]};
        const int k = (d[0] >= d[1] && d[0] >= d[2]) ? 0 : (d[1] >= d[2] ? 1 : 2);
        double ax[3];
        const double den = sqrt(fmax(0.0, (d[k] - cs) / (1.0 - cs)));
        for (int i = 0; i < 3; ++i) ax[i] = (i == k) ? den : (R[i][k] + R[k][i]) / (2.0 * (1.0 - cs) * den);
        const double dot = ax[0] * s[0] + ax[1] * s[1] + ax[2] * s[2];
        const double sg = dot < 0.0 ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i) w[i] = sg * th * ax[i];
    } else {
        for (int i = 0; i < 3; ++i) w[i] = s[i] * (th / sn);
    }
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B, C, D;
    so3_coeffs(th2, A, B, C, D);
    const double *t = T.t;
    const double wxt[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
    const double wxwxt[3] = {w[1] * wxt[2] - w[2] * wxt[1], w[2] * wxt[0] - w[0] * wxt[2],
                             w[0] * wxt[1] - w[1] * wxt[0]};
    for (int i = 0; i < 3; ++i) {
        tw[i] = t[i] - 0.5 * wxt[i] + D * wxwxt[i];
        tw[3 + i] = w[i];
    }
}

__device__ __forceinline__ rigid rigid_inverse(const rigid &T)
{
    rigid Ti;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Ti.R[i][j] = T.R[j][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Ti.t[i] = -(Ti.R[i][0] * T.t[0] + Ti.R[i][1] * T.t[1] + Ti.R[i][2] * T.t[2]);
    return Ti;
}

// column-major 4x4 <-> rigid
__device__ __forceinline__ rigid rigid_from_cm44(const double *H)
{
    rigid T;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T.R[i][j] = H[4 * j + i];
        T.t[i] = H[12 + i];
    }
    return T;
}
__device__ __forceinline__ void rigid_to_cm44(const rigid &T, double *H)
{
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            H[4 * j + i] = (i < 3) ? (j < 3 ? T.R[i][j] : T.t[i]) : (j == 3 ? 1.0 : 0.0);
}

// K \ b for an upper-triangular 3x3 (column-major), back substitution.
__device__ __forceinline__ void k_backslash(const double *K, const double b[3], double x[3])
{
    x[2] = b[2] / K[8];
    x[1] = (b[1] - K[7] * x[2]) / K[4];
    x[0] = (b[0] - K[3] * x[1] - K[6] * x[2]) / K[0];
}

}  // namespace vohip_dev
