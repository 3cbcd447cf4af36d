// Fundamental-matrix RANSAC for gfx950 (fp64).
//
// Stands in for estimateFundamentalMatrix(p1, p2, 'NumTrials', 500, 'Method', 'RANSAC',
// 'DistanceThreshold', 5e-2) at update_landmarks.m:13-14, of which the reference only uses the
// inlier index.  The toolbox function is closed source; the estimator is defined as:
//   Hartley normalisation of all points (once); per trial 8 pairs (index sets supplied by the
//   caller, who owns the RNG) -> null vector of the 8x9 system by Gaussian elimination with
//   complete pivoting -> rank 2 by zeroing the smallest singular value (one-sided Jacobi) ->
//   de-normalise; inlier <=> Sampson distance (pixels^2) <= threshold; winner = first trial with
//   the largest count; trial budget shrinks to ceil(log(1-c)/log(1-(inl/n)^8)) after each
//   improvement (c = Confidence / 100; the toolbox default is 99).
// Kernels: fm_normalise_kernel (one workgroup, two block reductions), fm_solve_kernel (one
// group of 8 lanes per trial), fm_count_kernel (one wavefront per trial, ballot/popcount), fm_mask_kernel.
#include "vohip_internal.h"
#include "sampson_test.h"
#include "seq_index.h"

#include <cmath>

namespace {

__device__ __forceinline__ double wsum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// nrm[6] = {s1, cx1, cy1, s2, cx2, cy2}.  Sums run in a fixed order (per-thread strided partial
// sums, wave butterfly, then waves in order), so the result is reproducible.
__global__ __launch_bounds__(256) void fm_normalise_kernel(const double *__restrict__ p1,
                                                           const double *__restrict__ p2, int n,
                                                           const int *__restrict__ n_dev,
                                                           double *__restrict__ nrm, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(p1);
    VOHIP_SEQ_STATE(p2);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE(nrm);
    __shared__ double sW[4][4];
    if (n_dev) n = *n_dev;  // point count kept on the device (the frame pipeline)
    __shared__ double sMean[4];
    const int tid = threadIdx.x;
    double s[4] = {0, 0, 0, 0};
    for (int i = tid; i < n; i += 256) {
        s[0] += p1[2 * i];
        s[1] += p1[2 * i + 1];
        s[2] += p2[2 * i];
        s[3] += p2[2 * i + 1];
    }
    for (int q = 0; q < 4; ++q) {
        const double t = wsum(s[q]);
        if ((tid & 63) == 0) sW[tid >> 6][q] = t;
    }
    __syncthreads();
    if (tid < 4) sMean[tid] = (sW[0][tid] + sW[1][tid] + sW[2][tid] + sW[3][tid]) / (double)n;
    __syncthreads();
    double d[2] = {0, 0};
    for (int i = tid; i < n; i += 256) {
        const double ax = p1[2 * i] - sMean[0], ay = p1[2 * i + 1] - sMean[1];
        const double bx = p2[2 * i] - sMean[2], by = p2[2 * i + 1] - sMean[3];
        d[0] += sqrt(ax * ax + ay * ay);
        d[1] += sqrt(bx * bx + by * by);
    }
    __syncthreads();
    for (int q = 0; q < 2; ++q) {
        const double t = wsum(d[q]);
        if ((tid & 63) == 0) sW[tid >> 6][q] = t;
    }
    __syncthreads();
    if (tid < 2) {
        const double md = (sW[0][tid] + sW[1][tid] + sW[2][tid] + sW[3][tid]) / (double)n;
        nrm[3 * tid] = md > 0.0 ? sqrt(2.0) / md : 1.0;
        nrm[3 * tid + 1] = sMean[2 * tid];
        nrm[3 * tid + 2] = sMean[2 * tid + 1];
    }
}

// Null vector of the 8x9 system of one trial by Gaussian elimination with complete pivoting, done
// by a group of 8 lanes (lane g owns row g) on a matrix that lives in LDS (the pivot search indexes it
// with data-dependent rows and columns, which registers cannot do).  The operations on every
// element are those of the one-thread formulation, in the same order.  All 64 lanes of the
// wavefront must call it (8 trials per wavefront); `A` is this group's 72 doubles, `perm` its 9
// ints.  Returns false for a singular system.  x (unit norm) is valid in every lane.
__device__ bool null9_group8(double *A, int *perm, double x[9])
{
    const int g = threadIdx.x & 7;
    bool ok = true;
    if (g == 0)
        for (int c = 0; c < 9; ++c) perm[c] = c;
    for (int k = 0; k < 8; ++k) {
        __syncthreads();
        // pivot search: largest |A[r][c]| over r, c >= k; the first one in row-major order wins ties
        double best = -1.0;
        int rc = 0;
        if (g >= k) {
            best = 0.0;
            rc = 16 * g + k;
            for (int c = k; c < 9; ++c) {
                const double v = fabs(A[g * 9 + c]);
                if (v > best) {
                    best = v;
                    rc = 16 * g + c;
                }
            }
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            const double ob = __shfl_xor(best, m, 8);
            const int orc = __shfl_xor(rc, m, 8);
            if (ob > best || (ob == best && orc < rc)) {
                best = ob;
                rc = orc;
            }
        }
        if (!(best > 0.0)) ok = false;   // keep going on garbage: the wavefront stays uniform
        const int pr = rc >> 4, pc = rc & 15;
        __syncthreads();
        if (pr != k) {   // row swap, lane g takes column g (lane 0 also column 8)
            const double t = A[k * 9 + g];
            A[k * 9 + g] = A[pr * 9 + g];
            A[pr * 9 + g] = t;
            if (g == 0) {
                const double u = A[k * 9 + 8];
                A[k * 9 + 8] = A[pr * 9 + 8];
                A[pr * 9 + 8] = u;
            }
        }
        __syncthreads();
        if (pc != k) {   // column swap, lane g takes row g
            const double t = A[g * 9 + k];
            A[g * 9 + k] = A[g * 9 + pc];
            A[g * 9 + pc] = t;
            if (g == 0) {
                const int u = perm[k];
                perm[k] = perm[pc];
                perm[pc] = u;
            }
        }
        __syncthreads();
        if (g > k) {
            const double f = A[g * 9 + k] / A[k * 9 + k];
            for (int c = k; c < 9; ++c) A[g * 9 + c] -= f * A[k * 9 + c];
        }
    }
    __syncthreads();
    double y[9];
    y[8] = 1.0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        double s = 0.0;
#pragma unroll
        for (int c = k + 1; c < 9; ++c) s += A[k * 9 + c] * y[c];
        y[k] = -s / A[k * 9 + k];
    }
    double nn = 0.0;
#pragma unroll
    for (int c = 0; c < 9; ++c) nn += y[c] * y[c];
    nn = sqrt(nn);
    __syncthreads();
    // un-permute through LDS (row 0 of the matrix is no longer needed)
    if (g == 0)
#pragma unroll
        for (int c = 0; c < 9; ++c) A[perm[c]] = y[c] / nn;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 9; ++c) x[c] = A[c];
    return ok;
}

__device__ void rank2(double F[3][3])
{
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = F[r][c];
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = i + 1; j < 3; ++j) {
                double al = 0, be = 0, ga = 0;
                for (int r = 0; r < 3; ++r) {
                    al += A[r][i] * A[r][i];
                    be += A[r][j] * A[r][j];
                    ga += A[r][i] * A[r][j];
                }
                if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < 3; ++r) {
                    const double ai = A[r][i], aj = A[r][j];
                    A[r][i] = c * ai - s * aj;
                    A[r][j] = s * ai + c * aj;
                    const double vi = V[r][i], vj = V[r][j];
                    V[r][i] = c * vi - s * vj;
                    V[r][j] = s * vi + c * vj;
                }
            }
        if (!rotated) break;
    }
    double nn[3];
    for (int c = 0; c < 3; ++c) nn[c] = A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c];
    const int k = (nn[0] <= nn[1] && nn[0] <= nn[2]) ? 0 : (nn[1] <= nn[2] ? 1 : 2);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int q = 0; q < 3; ++q)
                if (q != k) s += A[r][q] * V[c][q];
            F[r][c] = s;
        }
}

// Eight lanes per trial (eight trials per wavefront); F_out[t] row-major 3x3 (all zeros for a
// degenerate sample).
__global__ __launch_bounds__(64) void fm_solve_kernel(const double *__restrict__ p1, const double *__restrict__ p2,
                                                      int n, const int *__restrict__ n_dev,
                                                      const int32_t *__restrict__ samples, int trials,
                                                      const double *__restrict__ nrm, double *__restrict__ F_out,
                                                      vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(p1);
    VOHIP_SEQ_STATE(p2);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE(samples);
    VOHIP_SEQ_STATE(nrm);
    VOHIP_SEQ_STATE(F_out);
    __shared__ double sA[8][72];
    __shared__ int sP[8][9];
    const int g = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const int t_raw = blockIdx.x * 8 + slot;
    const int t = t_raw < trials ? t_raw : trials - 1;   // surplus groups redo the last trial, no store
    if (n_dev) n = *n_dev;
    const double s1 = nrm[0], cx1 = nrm[1], cy1 = nrm[2], s2 = nrm[3], cx2 = nrm[4], cy2 = nrm[5];
    double *A = sA[slot];
    bool ok = true;
    {   // lane g builds row g
        int i = samples[8 * (size_t)t + g];
        if (i < 0 || i >= n) {
            ok = false;
            i = 0;
        }
        const double x1 = s1 * (p1[2 * i] - cx1), y1 = s1 * (p1[2 * i + 1] - cy1);
        const double x2 = s2 * (p2[2 * i] - cx2), y2 = s2 * (p2[2 * i + 1] - cy2);
        double *row = A + 9 * g;
        row[0] = x2 * x1;
        row[1] = x2 * y1;
        row[2] = x2;
        row[3] = y2 * x1;
        row[4] = y2 * y1;
        row[5] = y2;
        row[6] = x1;
        row[7] = y1;
        row[8] = 1.0;
    }
    // a bad index in any of the 8 rows invalidates the trial
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) ok = ok && (__shfl_xor(ok ? 1 : 0, m, 8) != 0);
    double f[9];
    ok = null9_group8(A, sP[slot], f) && ok;
    if (g != 0 || t_raw >= trials) return;
    double F[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ok) {
        double Fh[3][3] = {{f[0], f[1], f[2]}, {f[3], f[4], f[5]}, {f[6], f[7], f[8]}};
        rank2(Fh);
        const double T1[3][3] = {{s1, 0, -s1 * cx1}, {0, s1, -s1 * cy1}, {0, 0, 1}};
        const double T2[3][3] = {{s2, 0, -s2 * cx2}, {0, s2, -s2 * cy2}, {0, 0, 1}};
        double M[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) M[r][c] = Fh[r][0] * T1[0][c] + Fh[r][1] * T1[1][c] + Fh[r][2] * T1[2][c];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) F[3 * r + c] = T2[0][r] * M[0][c] + T2[1][r] * M[1][c] + T2[2][r] * M[2][c];
    }
    for (int q = 0; q < 9; ++q) F_out[9 * (size_t)t + q] = F[q];
}

using vohip_dev::sampson_inlier;   // sampson_test.h

// One wavefront per trial, 4 trials per workgroup.
__global__ __launch_bounds__(256) void fm_count_kernel(const double *__restrict__ p1, const double *__restrict__ p2,
                                                       int n, const int *__restrict__ n_dev,
                                                       const double *__restrict__ Fs, int trials, double thr,
                                                       int32_t *__restrict__ counts, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(p1);
    VOHIP_SEQ_STATE(p2);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE(Fs);
    VOHIP_SEQ_STATE(counts);
    if (n_dev) n = *n_dev;
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (t >= trials) return;
    double F[9];
    bool zero = true;
    for (int q = 0; q < 9; ++q) {
        F[q] = Fs[9 * (size_t)t + q];
        zero = zero && F[q] == 0.0;
    }
    int c = 0;
    if (!zero)
        for (int i = lane; i < n; i += 64)
            c += __popcll(__ballot(sampson_inlier(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], thr)));
    if (lane == 0) counts[t] = c;
}

__global__ __launch_bounds__(256) void fm_mask_kernel(const double *__restrict__ p1, const double *__restrict__ p2,
                                                      int n, const int *__restrict__ n_dev,
                                                      const double *__restrict__ F9, double thr,
                                                      uint8_t *__restrict__ mask)
{
    if (n_dev) n = *n_dev;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double F[9];
    for (int q = 0; q < 9; ++q) F[q] = F9[q];
    mask[i] = sampson_inlier(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], thr) ? 1 : 0;
}

double trial_budget(int inliers, int n, int num_trials, double confidence)
{
    const double r = (double)inliers / (double)n;
    const double r8 = std::pow(r, 8.0);
    if (r8 >= 1.0) return 1.0;
    if (r8 <= 0.0) return (double)num_trials;
    const double t = std::ceil(std::log(1.0 - confidence / 100.0) / std::log(1.0 - r8));
    return t < (double)num_trials ? t : (double)num_trials;
}

}  // namespace

// Normalise + solve + count for `num_trials` trials; n sizes nothing here (one wavefront per
// trial), the point count may live on the device.  d_nrm 6 doubles, d_F 9 x trials, d_counts trials.
int vohip_fm_enqueue(vohip_ctx *ctx, const double *d_p1, const double *d_p2, int n, const int *d_n,
                     const int32_t *d_samples, int num_trials, double threshold, double *d_nrm, double *d_F,
                     int32_t *d_counts, const vohip_seqs *seqs)
{
    hipStream_t s = ctx->stream;
    const vohip_seqs SQ = seqs ? *seqs : vohip_seqs();
    const int ny = vohip_seq_count(SQ);
    if (ny < 1) return VOHIP_OK;
    hipLaunchKernelGGL(fm_normalise_kernel, dim3(1, ny), dim3(256), 0, s, d_p1, d_p2, n, d_n, d_nrm, SQ);
    hipLaunchKernelGGL(fm_solve_kernel, dim3((num_trials + 7) / 8, ny), dim3(64), 0, s, d_p1, d_p2, n, d_n, d_samples,
                       num_trials, (const double *)d_nrm, d_F, SQ);
    hipLaunchKernelGGL(fm_count_kernel, dim3((num_trials + 3) / 4, ny), dim3(256), 0, s, d_p1, d_p2, n, d_n,
                       (const double *)d_F, num_trials, threshold, d_counts, SQ);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

// Inlier mask of the matrix at d_F9 (row-major 3x3, device memory).  n sizes the grid.
int vohip_fm_mask_enqueue(vohip_ctx *ctx, const double *d_p1, const double *d_p2, int n, const int *d_n,
                          const double *d_F9, double threshold, uint8_t *d_mask)
{
    hipLaunchKernelGGL(fm_mask_kernel, dim3((n + 255) / 256 < 1 ? 1 : (n + 255) / 256), dim3(256), 0, ctx->stream, d_p1,
                       d_p2, n, d_n, d_F9, threshold, d_mask);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

extern "C" {

// d_p1, d_p2: 2 x n; d_samples 8 x num_trials; d_mask n bytes (device).  Synchronises once to
// replay the trial budget on the host.  best_count / trials_run are host outputs.
int vohip_fmatrix_ransac_dev(vohip_ctx *ctx, const double *d_p1, const double *d_p2, int n,
                             const int32_t *d_samples, int num_trials, double threshold, double confidence,
                             uint8_t *d_mask, int *best_count, int *trials_run, double *F_best)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_p1 && d_p2 && d_samples && d_mask && best_count, "NULL argument");
    VOHIP_REQUIRE(ctx, num_trials >= 1, "need at least one trial");
    VOHIP_REQUIRE(ctx, confidence > 0.0 && confidence < 100.0, "confidence must lie in (0, 100)");
    *best_count = 0;
    if (trials_run) *trials_run = 0;
    if (n < 8) return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "the 8-point algorithm needs 8 pairs, got %d", n);
    void *d_nrm, *d_F, *d_counts, *h_counts;
    VOHIP_TRY(vohip_upload(ctx, B_FM_NRM, nullptr, 64, &d_nrm));
    VOHIP_TRY(vohip_upload(ctx, B_FM_F, nullptr, sizeof(double) * 9 * (size_t)num_trials, &d_F));
    VOHIP_TRY(vohip_upload(ctx, B_FM_COUNTS, nullptr, sizeof(int32_t) * (size_t)num_trials, &d_counts));
    VOHIP_TRY(vohip_reserve_pinned(ctx, ctx->bufs[H_FM_COUNTS], sizeof(int32_t) * (size_t)num_trials));
    h_counts = ctx->bufs[H_FM_COUNTS].ptr;
    hipStream_t s = ctx->stream;
    VOHIP_TRY(vohip_fm_enqueue(ctx, d_p1, d_p2, n, nullptr, d_samples, num_trials, threshold, (double *)d_nrm,
                               (double *)d_F, (int32_t *)d_counts));
    VOHIP_TRY(vohip_download(ctx, h_counts, d_counts, sizeof(int32_t) * (size_t)num_trials));
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    const int32_t *c = (const int32_t *)h_counts;
    double budget = (double)num_trials;
    int t = 0, best = 0, best_t = -1;
    while ((double)t < budget && t < num_trials) {
        if (c[t] > best) {
            best = c[t];
            best_t = t;
            budget = trial_budget(best, n, num_trials, confidence);
        }
        ++t;
    }
    *best_count = best;
    if (trials_run) *trials_run = t;
    if (best_t < 0) {
        VOHIP_HIP(ctx, hipMemsetAsync(d_mask, 0, (size_t)n, s));
        return vohip_fail(ctx, VOHIP_NO_MODEL, "no trial produced an inlier");
    }
    VOHIP_TRY(vohip_fm_mask_enqueue(ctx, d_p1, d_p2, n, nullptr, (const double *)d_F + 9 * (size_t)best_t, threshold,
                                    d_mask));
    if (F_best) {
        VOHIP_TRY(vohip_download(ctx, F_best, (const double *)d_F + 9 * (size_t)best_t, sizeof(double) * 9));
        VOHIP_HIP(ctx, hipStreamSynchronize(s));
    }
    return VOHIP_OK;
}

int vohip_fmatrix_ransac(vohip_ctx *ctx, const double *p1, const double *p2, int n, const int32_t *samples,
                         int num_trials, double threshold, double confidence, uint8_t *mask, int *best_count,
                         int *trials_run, double *F_best)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, p1 && p2 && samples && mask && best_count, "NULL argument");
    if (n < 8) return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "the 8-point algorithm needs 8 pairs, got %d", n);
    VOHIP_REQUIRE(ctx, num_trials >= 1, "need at least one trial");
    void *d_p1, *d_p2, *d_s, *d_mask;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, p1, sizeof(double) * 2 * n, &d_p1));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, p2, sizeof(double) * 2 * n, &d_p2));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, samples, sizeof(int32_t) * 8 * (size_t)num_trials, &d_s));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, (size_t)n, &d_mask));
    const int st = vohip_fmatrix_ransac_dev(ctx, (const double *)d_p1, (const double *)d_p2, n, (const int32_t *)d_s,
                                            num_trials, threshold, confidence, (uint8_t *)d_mask, best_count, trials_run,
                                            F_best);
    if (st != VOHIP_OK && st != VOHIP_NO_MODEL) return st;
    VOHIP_TRY(vohip_download(ctx, mask, d_mask, (size_t)n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return st;
}

}  // extern "C"
