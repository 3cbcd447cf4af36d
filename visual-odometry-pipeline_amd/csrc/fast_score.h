// The segment-test scores of Rosten & Drummond's FAST on a staged tile of integer pixel values.
#pragma once

#include <hip/hip_runtime.h>

namespace vohip_dev {

// FAST 9-16: c points at the pixel, consecutive rows lie `stride` ints apart and the circle of radius 3 is
// staged.  Returns max over the 16 arcs of 9 of max(min(I_q - I_p), min(I_p - I_q)) - 1: the largest threshold
// at which the pixel still passes the test (below 0: at none).  Sliding minima over 9 of the 16 differences by
// doubling (windows of 2, 4, 8, then one more): 4 x 16 min per polarity.
__device__ __forceinline__ int fast_score_9_16(const int *c, int stride)
{
    // circle of radius 3, clockwise from the top: offsets (da, db)
    constexpr int CA[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
    constexpr int CB[16] = {-3, -3, -2, -1, 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3};
    const int p = c[0];
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = c[CB[k] * stride + CA[k]] - p;
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo2[k] = min(d[k], d[(k + 1) & 15]);
        hi2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
        hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
    }
    int bright = -255, dark = -255;   // max over the arcs of min(d) and of min(-d) = -max(d)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        bright = max(bright, lo9);
        dark = max(dark, -hi9);
    }
    return max(bright, dark) - 1;
}

// FAST 5-8: the 8 neighbours, 5 contiguous.  Same score, never below 0.
__device__ __forceinline__ int fast_score_5_8(const int *c, int stride)
{
    constexpr int CA[8] = {0, 1, 1, 1, 0, -1, -1, -1};
    constexpr int CB[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
    const int p = c[0];
    int d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = c[CB[k] * stride + CA[k]] - p;
    int lo2[8], hi2[8], lo4[8], hi4[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo2[k] = min(d[k], d[(k + 1) & 7]);
        hi2[k] = max(d[k], d[(k + 1) & 7]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        lo4[k] = min(lo2[k], lo2[(k + 2) & 7]);
        hi4[k] = max(hi2[k], hi2[(k + 2) & 7]);
    }
    int bright = -255, dark = -255;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        bright = max(bright, min(lo4[k], d[(k + 4) & 7]));
        dark = max(dark, -max(hi4[k], d[(k + 4) & 7]));
    }
    return max(max(bright, dark) - 1, 0);
}

}  // namespace vohip_dev
