// The epipolar test of the fundamental-matrix RANSAC (Sampson distance in pixels^2 <= threshold,
// estimateFundamentalMatrix's 'DistanceThreshold' at update_landmarks.m:13-14) in the one form every
// kernel that counts or selects its inliers shares, so that they agree bit for bit: the fused
// multiply-adds are written out, nothing is left to the compiler's contraction.
#pragma once

#include <hip/hip_runtime.h>

namespace vohip_dev {

// F row-major 3x3; (x1, y1) in the first image, (x2, y2) in the second.  An all-zero F gives 0 / 0 = NaN,
// which is no inlier.
__device__ __forceinline__ bool sampson_inlier(const double F[9], double x1, double y1, double x2, double y2,
                                               double thr)
{
    const double a0 = __builtin_fma(F[0], x1, __builtin_fma(F[1], y1, F[2]));
    const double a1 = __builtin_fma(F[3], x1, __builtin_fma(F[4], y1, F[5]));
    const double a2 = __builtin_fma(F[6], x1, __builtin_fma(F[7], y1, F[8]));
    const double b0 = __builtin_fma(F[0], x2, __builtin_fma(F[3], y2, F[6]));
    const double b1 = __builtin_fma(F[1], x2, __builtin_fma(F[4], y2, F[7]));
    const double e = __builtin_fma(x2, a0, __builtin_fma(y2, a1, a2));
    const double den = __builtin_fma(a0, a0, __builtin_fma(a1, a1, __builtin_fma(b0, b0, b1 * b1)));
    const double d = e * e / den;
    return d <= thr;
}

}  // namespace vohip_dev
