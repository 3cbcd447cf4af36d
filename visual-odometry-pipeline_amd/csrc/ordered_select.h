// Ordered selection inside a grid of 256-thread workgroups, shared by the kernels that compact lists
// without changing their order.
#pragma once

#include <hip/hip_runtime.h>

namespace vohip_dev {

__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Returns the number of flagged elements before element i = block * 256 + threadIdx.x and, in
// `total`, the number of flagged elements in [0, n).  Every workgroup re-evaluates the flags before
// its range (at most a few KB, from L2), so there is no inter-workgroup hand-off and the result is
// deterministic.  All 256 threads must call it.  `sh` needs 12 ints.
template <class Pred>
__device__ int ordered_rank_if(int n, Pred pred, int &total, int *sh, int block)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = block * 256;
    int before = 0, all = 0;
    // Eight elements per thread and round, evaluated at clamped indices without a branch, so that their loads
    // are in flight together (one element per round costs a round trip to memory per 256 elements).
    constexpr int U = 8;
    for (int j0 = tid; j0 < n + tid; j0 += 256 * U) {   // (the bound is the same for every thread of the workgroup)
        int f[U];
#pragma unroll
        for (int u = 0; u < U; ++u) f[u] = pred(min(j0 + 256 * u, n - 1)) ? 1 : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = j0 + 256 * u;
            const int g = j < n ? f[u] : 0;
            all += g;
            before += j < first ? g : 0;
        }
    }
    before = wave_sum_i(before);
    all = wave_sum_i(all);
    const int i = first + tid;
    const bool mine = i < n && pred(i);
    const unsigned long long b = __ballot(mine);
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) {
        sh[wave] = before;
        sh[4 + wave] = all;
        sh[8 + wave] = __popcll(b);
    }
    __syncthreads();
    int r = sh[0] + sh[1] + sh[2] + sh[3] + in_wave;
    for (int w = 0; w < wave; ++w) r += sh[8 + w];
    total = sh[4] + sh[5] + sh[6] + sh[7];
    return r;
}

// uniform draws -> 3 distinct indices in [0, n)
__device__ __forceinline__ void triple_from_uniforms(const double *r, int n, int32_t *s)
{
    int a = min((int)(r[0] * (double)n), n - 1);
    int b = min((int)(r[1] * (double)(n - 1)), n - 2);
    int c = min((int)(r[2] * (double)(n - 2)), n - 3);
    b += b >= a;
    const int lo = min(a, b), hi = max(a, b);
    c += c >= lo;
    c += c >= hi;
    s[0] = a;
    s[1] = b;
    s[2] = c;
}

}  // namespace vohip_dev
