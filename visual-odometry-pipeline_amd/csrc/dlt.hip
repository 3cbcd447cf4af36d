// DLT camera pose from 2-D / 3-D correspondences and the DLT branch of the RANSAC localisation, for gfx950 (fp64).
//
// Restates utils/exe7_Ransac/estimatePoseDLT.m:1-60 and the use_p3p = false branch of
// utils/exe7_Ransac/ransacLocalization.m: k = 6 samples per hypothesis (:20-24), the hypothesis' pose from
// estimatePoseDLT (:69-74), ONE pose scored per iteration (:77-83), the adaptive bound with exponent k = 6
// (:118-129) and the refit on the inliers of the best hypothesis (:145-151).  The branch is dead in the reference
// (use_p3p = true at :9); it is built because the drop-in's ransacLocalization honours the switch.
//
// estimatePoseDLT's svd(Q) (:34) asks for the right singular vector of the smallest singular value of the
// 2n x 12 measurement matrix.  Here: one-sided (Hestenes) Jacobi rotations on a 12 x 12 matrix by a group of 16
// lanes -- lane c < 12 owns column c of the matrix and of V, the six disjoint column pairs of a round of the
// round-robin tournament rotate at once, partner columns cross by shuffles inside the group -- so a wavefront
// solves four hypotheses at a time.  For six correspondences the matrix is Q itself; for n > 6 (the refit, and
// the stand-alone call) one workgroup first reduces Q to its 12 x 12 triangular factor by Householder
// reflections (Q = H R has the right singular vectors of R; Q'Q, which would square the condition number, is
// never formed).  The 3 x 3 svd of :50 (closest rotation) is a one-lane Jacobi.
// Scoring reuses the P3P scoring kernels: four consecutive DLT hypotheses fill the four pose slots of one
// "hypothesis" of theirs, so the counts come out one per DLT hypothesis.
#include "vohip_internal.h"
#include "inlier_test.h"

#include <cmath>
#include <vector>

namespace {

constexpr int NC = 12;

__device__ __forceinline__ void rotation_cs(double al, double be, double ga, double &cs, double &sn)
{
    const double zeta = (be - al) / (2.0 * ga);
    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    cs = 1.0 / sqrt(1.0 + t * t);
    sn = cs * t;
}

// One-sided Jacobi on a 12 x 12 matrix by an aligned group of 16 lanes (c = lane & 15; lanes 12..15 idle along):
// a[] is column c of the matrix, v[] column c of V (the identity on entry is set here).  On return m[] holds, in
// every lane, the column of V that belongs to the column of the rotated matrix with the smallest norm: the right
// singular vector of the smallest singular value.  All 64 lanes of the wavefront must call it.
__device__ void null12_group16(int c, double a[NC], double m[NC])
{
    double v[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) v[k] = k == c ? 1.0 : 0.0;
    const int shift = threadIdx.x & 48;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int r = 0; r < NC - 1; ++r) {
            // round r of the tournament: column r meets column 11, the others meet (2 r - c) mod 11
            int p = c;
            if (c < NC - 1) {
                p = (2 * r - c) % (NC - 1);
                if (p < 0) p += NC - 1;
                if (p == c) p = NC - 1;
            } else if (c == NC - 1) {
                p = r;
            }
            double b[NC], w[NC];
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                b[k] = __shfl(a[k], p, 16);
                w[k] = __shfl(v[k], p, 16);
            }
            double n_own = 0.0, n_par = 0.0, ga = 0.0;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                n_own += a[k] * a[k];
                n_par += b[k] * b[k];
                ga += a[k] * b[k];
            }
            const bool lo = c < p;   // this lane owns the pair's column of lower index
            const double al = lo ? n_own : n_par, be = lo ? n_par : n_own;
            if (c != p && ga * ga > 1e-30 * (al * be)) {   // |ga| > 1e-15 sqrt(al be); both lanes of a pair agree
                rotated = true;
                double cs, sn;
                rotation_cs(al, be, ga, cs, sn);
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    a[k] = lo ? cs * a[k] - sn * b[k] : sn * b[k] + cs * a[k];
                    v[k] = lo ? cs * v[k] - sn * w[k] : sn * w[k] + cs * v[k];
                }
            }
        }
        if ((((unsigned long long)__ballot(rotated) >> shift) & 0xffffull) == 0ull) break;
    }
    double nn = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) nn += a[k] * a[k];
    int best = 0;
    double best_n = __shfl(nn, 0, 16);
    for (int k = 1; k < NC; ++k) {
        const double t = __shfl(nn, k, 16);
        if (t < best_n) {
            best_n = t;
            best = k;
        }
    }
#pragma unroll
    for (int k = 0; k < NC; ++k) m[k] = __shfl(v[k], best, 16);
}

// [U, ~, V] = svd(R); U * V' (estimatePoseDLT.m:50-51), R and the result row-major.
__device__ void closest_rotation(const double R[9], double Rt[9])
{
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = R[3 * r + c];
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = i + 1; j < 3; ++j) {
                double al = 0, be = 0, ga = 0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    al += A[r][i] * A[r][i];
                    be += A[r][j] * A[r][j];
                    ga += A[r][i] * A[r][j];
                }
                if (!(ga * ga > 1e-30 * (al * be))) continue;
                rotated = true;
                double cs, sn;
                rotation_cs(al, be, ga, cs, sn);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double ai = A[r][i], aj = A[r][j];
                    A[r][i] = cs * ai - sn * aj;
                    A[r][j] = sn * ai + cs * aj;
                    const double vi = V[r][i], vj = V[r][j];
                    V[r][i] = cs * vi - sn * vj;
                    V[r][j] = sn * vi + cs * vj;
                }
            }
        if (!rotated) break;
    }
    double U[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double nn = sqrt(A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c]);
#pragma unroll
        for (int r = 0; r < 3; ++r) U[r][c] = A[r][c] / nn;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[3 * r + c] = U[r][0] * V[c][0] + U[r][1] * V[c][1] + U[r][2] * V[c][2];
}

// estimatePoseDLT.m:37-58 from the singular vector m: Rt = [R_tilde (column-major) | alpha t] and the fused
// projection matrix K [R_tilde | alpha t] (row-major 3 x 4) the scoring kernels take.
__device__ void pose_from_vector(const vohip_K &K, const double m[NC], double Rt[12], double M[12])
{
    double R[9], t[3];
    for (int r = 0; r < 3; ++r) {   // reshape(M_tilde, 4, 3)'
        for (int c = 0; c < 3; ++c) R[3 * r + c] = m[4 * r + c];
        t[r] = m[4 * r + 3];
    }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                       R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0.0) {
        for (int q = 0; q < 9; ++q) R[q] = -R[q];
        for (int q = 0; q < 3; ++q) t[q] = -t[q];
    }
    double Rc[9];
    closest_rotation(R, Rc);
    double nr = 0.0, nt = 0.0;
    for (int q = 0; q < 9; ++q) {
        nr += R[q] * R[q];
        nt += Rc[q] * Rc[q];
    }
    const double alpha = sqrt(nt) / sqrt(nr);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rt[3 * c + r] = Rc[3 * r + c];
        Rt[9 + r] = alpha * t[r];
    }
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 4; ++c) M[4 * i + c] = K.k[i] * Rt[3 * c] + K.k[3 + i] * Rt[3 * c + 1] + K.k[6 + i] * Rt[3 * c + 2];
}

// K \ [u; v; 1] (estimatePoseDLT.m:14), first two components; K upper triangular, column-major
__device__ __forceinline__ void normalised(const vohip_K &K, double u, double v, double &x, double &y)
{
    const double x2 = 1.0 / K.k[8];
    y = (v - K.k[7] * x2) / K.k[4];
    x = (u - K.k[3] * y - K.k[6] * x2) / K.k[0];
}

// Entry `c` of the two rows a correspondence adds to Q (estimatePoseDLT.m:24-31).
__device__ __forceinline__ void q_rows(int c, double X0, double X1, double X2, double u, double v, double &row_u, double &row_v)
{
    const double P4[4] = {X0, X1, X2, 1.0};
    const double p = P4[c & 3];
    row_u = c < 4 ? p : (c < 8 ? 0.0 : -u * p);
    row_v = c < 4 ? 0.0 : (c < 8 ? p : -v * p);
}

// Hypothesis j of [0, 4 * slots): the pose of the six correspondences samples[6 j ..] (ransacLocalization.m:69-74)
// into pose slot (j / 4, j % 4) of Rt_out / M_out; NaN for j >= h (padding of the last slot) and for bad indices.
__global__ __launch_bounds__(64) void dlt_solve_kernel(const double *__restrict__ kp_rc, const double *__restrict__ X, int n,
                                                       vohip_K K, const int32_t *__restrict__ samples, int h, int slots,
                                                       double *__restrict__ Rt_out, double *__restrict__ M_out)
{
    const int c = threadIdx.x & 15;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool live = j < h;
    double a[NC];
    bool ok = live;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        int id = live ? samples[6 * (size_t)j + i] : 0;
        if (id < 0 || id >= n) {
            ok = false;
            id = 0;
        }
        double u, v;
        normalised(K, kp_rc[2 * id + 1], kp_rc[2 * id], u, v);
        q_rows(c < NC ? c : 0, X[3 * id], X[3 * id + 1], X[3 * id + 2], u, v, a[2 * i], a[2 * i + 1]);
    }
    double m[NC], Rt[12], M[12];
    null12_group16(c, a, m);
    pose_from_vector(K, m, Rt, M);
    if (c != 0 || j >= 4 * slots) return;
    const double nan = __builtin_nan("");
    for (int q = 0; q < 12; ++q) {
        Rt_out[12 * (size_t)j + q] = ok ? Rt[q] : nan;
        M_out[12 * (size_t)j + q] = ok ? M[q] : nan;
    }
}

constexpr int RF_THREADS = 512;

// Sums over the workgroup of the NS values every thread holds, left in sOut (two barriers + one).
template <int NS>
__device__ __forceinline__ void block_sums(double (&v)[NS], double *sPart, double *sOut)
{
    constexpr int NW = RF_THREADS / 64;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v[q] += __shfl_xor(v[q], s, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < NS; ++q) sPart[NW * q + (threadIdx.x >> 6)] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += sPart[NW * threadIdx.x + w];
        sOut[threadIdx.x] = t;
    }
    __syncthreads();
}

// estimatePoseDLT over the correspondences whose mask byte is set (mask == NULL: all n): one workgroup.
// work: 2 n x 12 doubles (row-major).  out: Rt[12] then M[12]; out[24] = 1 when fewer than 6 correspondences
// were given (the pose is then NaN).
__global__ __launch_bounds__(RF_THREADS) void dlt_refit_kernel(const double *__restrict__ kp_rc, const double *__restrict__ X,
                                                               int n, const uint8_t *__restrict__ mask, vohip_K K,
                                                               double *__restrict__ work, double *__restrict__ out)
{
    __shared__ double sPart[(RF_THREADS / 64) * NC], sSum[NC];
    const int tid = threadIdx.x;
    const int rows = 2 * n;
    double used[1] = {0.0};
    for (int p = tid; p < n; p += RF_THREADS) {
        const bool on = !mask || mask[p];
        used[0] += on ? 1.0 : 0.0;
        double u, v;
        normalised(K, kp_rc[2 * p + 1], kp_rc[2 * p], u, v);
        double *r0 = work + NC * (size_t)(2 * p), *r1 = r0 + NC;
        for (int c = 0; c < NC; ++c) {
            double a, b;
            q_rows(c, X[3 * p], X[3 * p + 1], X[3 * p + 2], u, v, a, b);
            r0[c] = on ? a : 0.0;   // (a correspondence that is left out adds two zero rows: they change nothing)
            r1[c] = on ? b : 0.0;
        }
    }
    block_sums<1>(used, sPart, sSum);
    const int n_used = (int)sSum[0];
    // Householder reflections, column by column: the rows a thread owns are r = tid, tid + 256, ...
    for (int k = 0; k < NC; ++k) {
        double s[1] = {0.0};
        for (int r = tid; r < rows; r += RF_THREADS)
            if (r >= k) s[0] += work[NC * (size_t)r + k] * work[NC * (size_t)r + k];
        block_sums<1>(s, sPart, sSum);
        const double xx = sSum[0];
        if (!(xx > 0.0)) continue;   // (the column is zero from row k on)
        const double x0 = work[NC * (size_t)k + k];
        const double alpha = x0 >= 0.0 ? -sqrt(xx) : sqrt(xx);
        const double v0 = x0 - alpha;
        const double vtv = (xx - x0 * x0) + v0 * v0;
        if (!(vtv > 0.0)) continue;
        double d[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) d[j] = 0.0;
        for (int r = tid; r < rows; r += RF_THREADS)
            if (r >= k) {
                const double *row = work + NC * (size_t)r;
                const double vr = r == k ? v0 : row[k];
#pragma unroll
                for (int j = 0; j < NC; ++j) d[j] += j > k ? vr * row[j] : 0.0;
            }
        block_sums<NC>(d, sPart, sSum);   // (ends with a barrier: x0 above was read before any write below)
        const double f = 2.0 / vtv;
        for (int r = tid; r < rows; r += RF_THREADS)
            if (r >= k) {
                double *row = work + NC * (size_t)r;
                const double vr = r == k ? v0 : row[k];
#pragma unroll
                for (int j = 0; j < NC; ++j)
                    if (j > k) row[j] -= f * vr * sSum[j];
                row[k] = r == k ? alpha : 0.0;
            }
        __syncthreads();
    }
    __syncthreads();
    if (tid >= 64) return;
    // the 12 x 12 triangular factor, column c to lane c of each group of 16 (every group solves the same problem)
    const int c = tid & 15;
    double a[NC];
#pragma unroll
    for (int r = 0; r < NC; ++r) a[r] = (c < NC && r <= c && r < rows) ? work[NC * (size_t)r + c] : 0.0;
    double m[NC], Rt[12], M[12];
    null12_group16(c, a, m);
    pose_from_vector(K, m, Rt, M);
    if (tid != 0) return;
    const double nan = __builtin_nan("");
    for (int q = 0; q < 12; ++q) {
        out[q] = n_used >= 6 ? Rt[q] : nan;
        out[12 + q] = n_used >= 6 ? M[q] : nan;
    }
    out[24] = n_used >= 6 ? 0.0 : 1.0;
}

// ransacLocalization.m:118-129 with k = 6
double adaptive_iterations_dlt(int max_num_inliers, int n)
{
    const double confidence = 0.9999, upper = 0.95;
    double outlier_ratio = 1.0 - (double)max_num_inliers / (double)n;
    if (upper < outlier_ratio) outlier_ratio = upper;
    double it = std::log(1.0 - confidence) / std::log(1.0 - std::pow(1.0 - outlier_ratio, 6.0));
    if (it > 15000.0) it = 15000.0;
    if (it < 100.0) it = 100.0;
    return it;
}

int refit(vohip_ctx *ctx, const double *d_kp, const double *d_X, int n, const uint8_t *d_mask, const double K[9], double out[25])
{
    void *d_work, *d_out;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE4, nullptr, sizeof(double) * 2 * NC * (size_t)n, &d_work));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE5, nullptr, sizeof(double) * 32, &d_out));
    hipLaunchKernelGGL(dlt_refit_kernel, dim3(1), dim3(RF_THREADS), 0, ctx->stream, d_kp, d_X, n, d_mask, vohip_make_K(K),
                       (double *)d_work, (double *)d_out);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_TRY(vohip_download(ctx, out, d_out, sizeof(double) * 25));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

}  // namespace

extern "C" {

int vohip_estimate_pose_dlt(vohip_ctx *ctx, const double *p_uv, const double *P, int n, const double K[9], double M[12])
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, p_uv && P && K && M, "NULL argument");
    if (n < 6) return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "estimatePoseDLT needs at least 6 correspondences, got %d", n);
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    // MATLAB's n x 2 and n x 3 (column-major) as the kernels' interleaved (row, col) and (X, Y, Z)
    std::vector<double> kp(2 * (size_t)n), X(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        kp[2 * (size_t)i] = p_uv[(size_t)n + i];
        kp[2 * (size_t)i + 1] = p_uv[i];
        for (int q = 0; q < 3; ++q) X[3 * (size_t)i + q] = P[(size_t)q * n + i];
    }
    void *d_kp, *d_X;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, kp.data(), sizeof(double) * kp.size(), &d_kp));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, X.data(), sizeof(double) * X.size(), &d_X));
    double out[25];
    VOHIP_TRY(refit(ctx, (const double *)d_kp, (const double *)d_X, n, nullptr, K, out));   // (synchronises: kp, X may go)
    for (int q = 0; q < 12; ++q) M[q] = out[q];   // [R_tilde(:); alpha t]: the 3 x 4 matrix column-major
    return VOHIP_OK;
}

int vohip_ransac_dlt(vohip_ctx *ctx, const double *kp_rc, const double *X, int n, const double K[9], const int32_t *samples,
                     int hcap, double R_C_W[9], double t_C_W[3], uint8_t *mask, double *max_hist, double *iter_hist,
                     int *n_iter)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, kp_rc && X && K && samples && R_C_W && t_C_W && mask && max_hist && iter_hist && n_iter, "NULL argument");
    *n_iter = 0;
    if (n < 6) return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "the DLT needs at least 6 correspondences, got %d", n);
    VOHIP_REQUIRE(ctx, hcap >= 1, "need at least 1 sample set");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    const int slot_cap = (hcap + 3) / 4;
    void *p_kp, *p_X, *p_s;
    VOHIP_TRY(vohip_upload(ctx, B_RANSAC_KP, kp_rc, sizeof(double) * 2 * (size_t)n, &p_kp));
    VOHIP_TRY(vohip_upload(ctx, B_RANSAC_X, X, sizeof(double) * 3 * (size_t)n, &p_X));
    VOHIP_TRY(vohip_upload(ctx, B_RANSAC_SAMPLES, samples, sizeof(int32_t) * 6 * (size_t)hcap, &p_s));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_COUNTS], sizeof(int32_t) * 4 * (size_t)slot_cap));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_MASK], (size_t)n));
    VOHIP_TRY(vohip_reserve_pinned(ctx, ctx->bufs[H_RANSAC_COUNTS], sizeof(int32_t) * 4 * (size_t)slot_cap));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_RT], sizeof(double) * 48 * (size_t)slot_cap));
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_RANSAC_M], sizeof(double) * 48 * (size_t)slot_cap));
    const double *d_kp = (const double *)p_kp, *d_X = (const double *)p_X;
    const int32_t *d_samples = (const int32_t *)p_s;
    int32_t *d_counts = (int32_t *)ctx->bufs[B_RANSAC_COUNTS].ptr, *h_counts = (int32_t *)ctx->bufs[H_RANSAC_COUNTS].ptr;
    uint8_t *d_mask = (uint8_t *)ctx->bufs[B_RANSAC_MASK].ptr;
    double *d_Rt = (double *)ctx->bufs[B_RANSAC_RT].ptr, *d_M = (double *)ctx->bufs[B_RANSAC_M].ptr;
    hipStream_t st = ctx->stream;
    const vohip_K Kv = vohip_make_K(K);

    // Replay state of ransacLocalization.m:37,42-44.
    int max_num_inliers = 0, best_h = -1;
    double num_iterations = INFINITY;
    int i = 1, evaluated = 0;
    bool exhausted = false;
    while (num_iterations > (double)i) {
        if (i > evaluated) {
            if (evaluated >= hcap) {
                exhausted = true;
                break;
            }
            // chunk: at least the 99 iterations the clamp at :129 guarantees, then up to the current bound, in
            // whole pose slots (four hypotheses each; `evaluated` stays a multiple of four until the samples end)
            int want = evaluated == 0 ? 104 : (int)std::ceil(num_iterations) - 1 - evaluated;
            if (want < 64) want = 64;
            want = (want + 31) / 32 * 32;
            if (want > hcap - evaluated) want = hcap - evaluated;
            const int slots = (want + 3) / 4, slot0 = evaluated / 4;
            hipLaunchKernelGGL(dlt_solve_kernel, dim3(slots), dim3(64), 0, st, d_kp, d_X, n, Kv, d_samples + 6 * (size_t)evaluated,
                               want, slots, d_Rt + 48 * (size_t)slot0, d_M + 48 * (size_t)slot0);
            VOHIP_HIP(ctx, hipGetLastError());
            VOHIP_TRY(vohip_ransac_score_enqueue(ctx, d_kp, d_X, n, d_M + 48 * (size_t)slot0, slots, d_counts + 4 * (size_t)slot0));
            VOHIP_HIP(ctx, hipMemcpyAsync(h_counts + 4 * (size_t)slot0, d_counts + 4 * (size_t)slot0,
                                          sizeof(int32_t) * 4 * (size_t)slots, hipMemcpyDeviceToHost, st));
            VOHIP_HIP(ctx, hipStreamSynchronize(st));
            evaluated += want;
        }
        const int cnt = h_counts[i - 1];
        if (cnt > max_num_inliers && cnt >= 30) {  // :103-110
            max_num_inliers = cnt;
            best_h = i - 1;
        }
        num_iterations = adaptive_iterations_dlt(max_num_inliers, n);
        iter_hist[i - 1] = num_iterations;
        max_hist[i - 1] = (double)max_num_inliers;
        ++i;
    }
    *n_iter = i - 1;
    if (best_h < 0) {
        std::memset(mask, 0, (size_t)n);
        if (exhausted) return vohip_fail(ctx, VOHIP_SAMPLES_EXHAUSTED, "sample sets ran out after %d iterations, no model", i - 1);
        return vohip_fail(ctx, VOHIP_NO_MODEL, "no hypothesis reached 30 inliers in %d iterations", i - 1);
    }
    // best_inlier_mask: the inliers of the best hypothesis, by the scoring kernel's own test (:83, :108)
    VOHIP_TRY(vohip_ransac_mask_enqueue(ctx, d_kp, d_X, n, nullptr, d_M + 12 * (size_t)best_h, d_mask));
    VOHIP_TRY(vohip_download(ctx, mask, d_mask, (size_t)n));
    // :145-151: the pose returned is estimatePoseDLT over those inliers
    double out[25];
    VOHIP_TRY(refit(ctx, d_kp, d_X, n, d_mask, K, out));
    std::memcpy(R_C_W, out, sizeof(double) * 9);
    std::memcpy(t_C_W, out + 9, sizeof(double) * 3);
    if (exhausted) return vohip_fail(ctx, VOHIP_SAMPLES_EXHAUSTED, "sample sets ran out after %d iterations", i - 1);
    return VOHIP_OK;
}

}  // extern "C"
