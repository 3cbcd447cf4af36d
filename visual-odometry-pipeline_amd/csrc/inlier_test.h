// The reprojection test of ransacLocalization.m:77-101 (pixel_tolerance 10) in the division-free
// form every kernel that counts or selects inliers shares, so that they agree bit for bit.
#pragma once

#include <hip/hip_runtime.h>

namespace vohip_dev {

// m = K [R | t] row-major 3x4.  13 FMA + 2 MUL + 1 CMP.
__device__ __forceinline__ bool is_inlier(const double m[12], double X, double Y, double Z, double u, double v)
{
    const double x = __builtin_fma(m[0], X, __builtin_fma(m[1], Y, __builtin_fma(m[2], Z, m[3])));
    const double y = __builtin_fma(m[4], X, __builtin_fma(m[5], Y, __builtin_fma(m[6], Z, m[7])));
    const double z = __builtin_fma(m[8], X, __builtin_fma(m[9], Y, __builtin_fma(m[10], Z, m[11])));
    const double dx = __builtin_fma(-u, z, x);
    const double dy = __builtin_fma(-v, z, y);
    const double e = __builtin_fma(dy, dy, dx * dx);
    const double zz = z * z;
    return __builtin_fma(-100.0, zz, e) < 0.0;  // pixel_tolerance^2 = 100, strict
}

}  // namespace vohip_dev
