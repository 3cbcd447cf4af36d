// Small per-point geometry kernels of the hot path (all fp64, one thread per point):
//   twist <-> SE(3)                      utils/exe9_BA/twist2HomogMatrix.m:1-16, HomogMatrix2twist.m:1-25
//   batched linear triangulation         utils/exe6_8point/linearTriangulation.m:28-39
//   non-linear triangulation refinement  utils/linearTriangulation_refinement.m:1-22
//   candidate redundancy filter          Continuous_operation/update_candidate.m:12-27
//   candidate angle test + triangulation + depth filter
//                                        Continuous_operation/update_landmarks.m:27-97
// These are latency-bound at the reference's sizes (10^2..10^4 points): one launch each, every
// lane independent, no LDS except the point tiles of the redundancy filter.
#include "vohip_internal.h"
#include "se3_device.h"
#include "geometry_device.h"
#include "seq_index.h"

using namespace vohip_dev;

namespace {

__global__ __launch_bounds__(64) void twist2homog_kernel(const double *__restrict__ tw, int n,
                                                         double *__restrict__ H)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t[6];
    for (int k = 0; k < 6; ++k) t[k] = tw[6 * (size_t)i + k];
    const rigid T = twist_to_rigid(t);
    rigid_to_cm44(T, H + 16 * (size_t)i);
}

__global__ __launch_bounds__(64) void homog2twist_kernel(const double *__restrict__ H, int n,
                                                         double *__restrict__ tw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const rigid T = rigid_from_cm44(H + 16 * (size_t)i);
    double t[6];
    rigid_to_twist(T, t);
    for (int k = 0; k < 6; ++k) tw[6 * (size_t)i + k] = t[k];
}

__global__ __launch_bounds__(128) void triangulate_kernel(const double *__restrict__ p1,
                                                          const double *__restrict__ p2,
                                                          const double *__restrict__ M1,
                                                          const double *__restrict__ M2, int pp1, int pp2,
                                                          int n, double *__restrict__ P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double a[3], b[3], m1[12], m2[12], out[4];
    for (int k = 0; k < 3; ++k) {
        a[k] = p1[3 * (size_t)j + k];
        b[k] = p2[3 * (size_t)j + k];
    }
    for (int k = 0; k < 12; ++k) {
        m1[k] = M1[(pp1 ? 12 * (size_t)j : 0) + k];
        m2[k] = M2[(pp2 ? 12 * (size_t)j : 0) + k];
    }
    triangulate_point(a, b, m1, m2, out);
    for (int k = 0; k < 4; ++k) P[4 * (size_t)j + k] = out[k];
}

__global__ __launch_bounds__(128) void triangulate_refine_kernel(const double *__restrict__ P_init,
                                                                 const double *__restrict__ p1,
                                                                 const double *__restrict__ p2, int ldp,
                                                                 const double *__restrict__ M1,
                                                                 const double *__restrict__ M2, int n,
                                                                 int max_iter, double *__restrict__ P)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double m1[12], m2[12];
    for (int k = 0; k < 12; ++k) {
        m1[k] = M1[k];
        m2[k] = M2[k];
    }
    const double a[2] = {p1[(size_t)ldp * j], p1[(size_t)ldp * j + 1]};
    const double b[2] = {p2[(size_t)ldp * j], p2[(size_t)ldp * j + 1]};
    double x[3] = {P_init[4 * (size_t)j], P_init[4 * (size_t)j + 1], P_init[4 * (size_t)j + 2]};
    refine_point(x, a, b, m1, m2, max_iter);
    for (int k = 0; k < 3; ++k) P[4 * (size_t)j + k] = x[k];
    P[4 * (size_t)j + 3] = P_init[4 * (size_t)j + 3];
}

// update_candidate.m:14-24.  One wavefront per new detection; its lanes stride over the existing
// points (S.C then S.P, rounded to single exactly as MATLAB's single/double comparison does) and a
// ballot ends the search at the first hit.
// update_candidate.m:12-27: a detection is dropped when an existing candidate or landmark lies in the
// (2r+1)-box around it.  A workgroup of 16 wavefronts serves FILT_DET detections: the existing points
// are staged in LDS as floats (the comparison is made in single precision, as the reference's
// single-precision detections make it), FILT_CHUNK at a time; each wavefront then scans them for
// its detections, one after the other, 64 points per step.
constexpr int FILT_THREADS = 1024;
constexpr int FILT_DET = 16;      // detections per workgroup (one per wavefront)
constexpr int FILT_CHUNK = 4096;  // existing points staged at a time (32 KB)
static_assert(FILT_CHUNK % FILT_THREADS == 0, "whole rounds of the workgroup per chunk");

__global__ __launch_bounds__(FILT_THREADS) void candidate_filter_kernel(const float *__restrict__ new_xy, int ld, int n_new,
                                                                        const double *__restrict__ C_rc, int nC,
                                                                        const double *__restrict__ P_rc, int nP,
                                                                        const int *__restrict__ counts_dev, float r,
                                                                        uint8_t *__restrict__ keep, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(new_xy);
    VOHIP_SEQ_STATE(C_rc);
    VOHIP_SEQ_STATE(P_rc);
    VOHIP_SEQ_STATE_OPT(counts_dev);
    VOHIP_SEQ_STATE(keep);
    __shared__ float2 sP[FILT_CHUNK];   // (col, row) of existing points
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER_WAVE = FILT_DET / (FILT_THREADS / 64);
    const int first = blockIdx.x * FILT_DET + wave * PER_WAVE;
    float xs[PER_WAVE], ys[PER_WAVE];
#pragma unroll
    for (int d = 0; d < PER_WAVE; ++d) {   // (issued before the counts are known: one round trip less)
        const int i = first + d < n_new ? first + d : 0;
        xs[d] = new_xy[i];
        ys[d] = new_xy[ld + i];
    }
    if (counts_dev) {  // {n_new, nC, nP} kept on the device (the frame pipeline); n_new is clamped
        n_new = min(counts_dev[0], n_new);
        nC = counts_dev[1];
        nP = counts_dev[2];
    }
    if (blockIdx.x * FILT_DET >= n_new) return;   // (the whole workgroup)
    float u0[PER_WAVE], u1[PER_WAVE], v0[PER_WAVE], v1[PER_WAVE];
    bool near[PER_WAVE];
#pragma unroll
    for (int d = 0; d < PER_WAVE; ++d) {
        const float x = xs[d], y = ys[d];
        u0[d] = x - r;
        u1[d] = x + r;
        v0[d] = y - r;
        v1[d] = y + r;
        near[d] = false;
    }
    const int total = nC + nP;
    for (int base = 0; base < total; base += FILT_CHUNK) {
        const int cnt = min(FILT_CHUNK, total - base);
        __syncthreads();
        // (a chunk is FILT_CHUNK / FILT_THREADS points per thread: all their loads are issued before the first
        // one is waited for)
        constexpr int PER_THREAD = FILT_CHUNK / FILT_THREADS;
        double pr[PER_THREAD], pc[PER_THREAD];
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int g = base + min(tid + FILT_THREADS * u, cnt - 1);
            const double *src = g < nC ? C_rc + 2 * (size_t)g : P_rc + 2 * (size_t)(g - nC);
            pr[u] = src[0];
            pc[u] = src[1];
        }
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int k = tid + FILT_THREADS * u;
            if (k < cnt) sP[k] = make_float2((float)pc[u], (float)pr[u]);
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < PER_WAVE; ++d) {
            if (first + d >= n_new || __ballot(near[d])) continue;   // (wave-uniform)
            for (int k0 = 0; k0 < cnt; k0 += 256) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 64 * u + lane;
                    const float2 q = sP[k < cnt ? k : 0];
                    near[d] = near[d] || (k < cnt && q.x >= u0[d] && q.x <= u1[d] && q.y >= v0[d] && q.y <= v1[d]);
                }
                if (__ballot(near[d])) break;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < PER_WAVE; ++d) {
        const bool any = __ballot(near[d]) != 0ull;
        if (lane == 0 && first + d < n_new) keep[first + d] = (nC == 0) ? 1 : (any ? 0 : 1);
    }
}

struct cand_params {
    double T_W_C[16];
    double K[9];
    double angle_threshold_deg, min_depth, max_depth;
    int is_BA;
};

__global__ __launch_bounds__(128) void candidate_triangulate_kernel(const double *__restrict__ cur_rc,
                                                                    const double *__restrict__ first_rc,
                                                                    const double *__restrict__ T_first, int n,
                                                                    const int *__restrict__ n_dev,
                                                                    const double *__restrict__ T_dev,
                                                                    cand_params prm, uint8_t *__restrict__ status,
                                                                    double *__restrict__ X_out,
                                                                    double *__restrict__ angles, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(cur_rc);
    VOHIP_SEQ_STATE(first_rc);
    VOHIP_SEQ_STATE(T_first);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE_OPT(T_dev);
    VOHIP_SEQ_STATE(status);
    VOHIP_SEQ_STATE(X_out);
    VOHIP_SEQ_STATE_OPT(angles);
    // four lanes per candidate (the triangulation is shared by the quad; the rest is cheap and repeated)
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;
    if (i >= n) return;   // (n: the rows the arrays hold)
    // the candidate's row is fetched together with the count kept on the device (the frame pipeline), not
    // after it: one round trip to memory instead of two
    if (T_dev)
        for (int q = 0; q < 16; ++q) prm.T_W_C[q] = T_dev[q];
    double Tf[12];
    for (int k = 0; k < 12; ++k) Tf[k] = T_first[12 * (size_t)i + k];
    const double cur[2] = {cur_rc[2 * (size_t)i], cur_rc[2 * (size_t)i + 1]};
    const double fst[2] = {first_rc[2 * (size_t)i], first_rc[2 * (size_t)i + 1]};
    if (n_dev && i >= *n_dev) return;
    double X[3], ang;
    const int st = candidate_point<true>(cur, fst, Tf, prm.T_W_C, prm.K, prm.angle_threshold_deg, prm.min_depth,
                                         prm.max_depth, prm.is_BA, X, ang);
    if ((threadIdx.x & 3) != 0) return;
    status[i] = (uint8_t)st;
    for (int k = 0; k < 3; ++k) X_out[3 * (size_t)i + k] = X[k];
    if (angles) angles[i] = ang;
}

}  // namespace

extern "C" {

int vohip_twist2homog(vohip_ctx *ctx, const double *twists, int n, double *H)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, twists && H && n >= 0, "bad argument");
    if (n == 0) return VOHIP_OK;
    void *d_in, *d_out;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, twists, sizeof(double) * 6 * n, &d_in));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, nullptr, sizeof(double) * 16 * n, &d_out));
    hipLaunchKernelGGL(twist2homog_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, (const double *)d_in, n,
                       (double *)d_out);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_TRY(vohip_download(ctx, H, d_out, sizeof(double) * 16 * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_homog2twist(vohip_ctx *ctx, const double *H, int n, double *twists)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, twists && H && n >= 0, "bad argument");
    if (n == 0) return VOHIP_OK;
    void *d_in, *d_out;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, H, sizeof(double) * 16 * n, &d_in));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, nullptr, sizeof(double) * 6 * n, &d_out));
    hipLaunchKernelGGL(homog2twist_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, (const double *)d_in, n,
                       (double *)d_out);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_TRY(vohip_download(ctx, twists, d_out, sizeof(double) * 6 * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_triangulate_dev(vohip_ctx *ctx, const double *d_p1, const double *d_p2, const double *d_M1,
                          const double *d_M2, int per_point_M1, int per_point_M2, int n, double *d_P)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_p1 && d_p2 && d_M1 && d_M2 && d_P && n >= 0, "bad argument");
    if (n == 0) return VOHIP_OK;
    hipLaunchKernelGGL(triangulate_kernel, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, d_p1, d_p2, d_M1, d_M2,
                       per_point_M1, per_point_M2, n, d_P);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int vohip_triangulate(vohip_ctx *ctx, const double *p1, const double *p2, const double *M1, const double *M2,
                      int per_point_M1, int per_point_M2, int n, double *P)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, p1 && p2 && M1 && M2 && P && n >= 0, "bad argument");
    if (n == 0) return VOHIP_OK;
    void *d_p1, *d_p2, *d_M1, *d_M2, *d_P;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, p1, sizeof(double) * 3 * n, &d_p1));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, p2, sizeof(double) * 3 * n, &d_p2));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, M1, sizeof(double) * 12 * (per_point_M1 ? n : 1), &d_M1));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, M2, sizeof(double) * 12 * (per_point_M2 ? n : 1), &d_M2));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE4, nullptr, sizeof(double) * 4 * n, &d_P));
    VOHIP_TRY(vohip_triangulate_dev(ctx, (const double *)d_p1, (const double *)d_p2, (const double *)d_M1,
                                    (const double *)d_M2, per_point_M1, per_point_M2, n, (double *)d_P));
    VOHIP_TRY(vohip_download(ctx, P, d_P, sizeof(double) * 4 * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_triangulate_refine(vohip_ctx *ctx, const double *P_init, const double *p1, const double *p2, int ldp,
                             const double *M1, const double *M2, int n, double *P)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, P_init && p1 && p2 && M1 && M2 && P && n >= 0 && ldp >= 2, "bad argument");
    if (n == 0) return VOHIP_OK;
    void *d_Pi, *d_p1, *d_p2, *d_M1, *d_M2, *d_P;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, p1, sizeof(double) * ldp * n, &d_p1));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, p2, sizeof(double) * ldp * n, &d_p2));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, M1, sizeof(double) * 12, &d_M1));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, M2, sizeof(double) * 12, &d_M2));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE4, nullptr, sizeof(double) * 4 * n, &d_P));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE5, P_init, sizeof(double) * 4 * n, &d_Pi));
    hipLaunchKernelGGL(triangulate_refine_kernel, dim3((n + 127) / 128), dim3(128), 0, ctx->stream,
                       (const double *)d_Pi, (const double *)d_p1, (const double *)d_p2, ldp, (const double *)d_M1,
                       (const double *)d_M2, n, 20, (double *)d_P);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_TRY(vohip_download(ctx, P, d_P, sizeof(double) * 4 * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_candidate_filter_dev(vohip_ctx *ctx, const float *d_new_xy, int n_new, const double *d_C_rc, int nC,
                               const double *d_P_rc, int nP, double r, uint8_t *d_keep)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, n_new >= 0 && nC >= 0 && nP >= 0 && d_keep, "bad argument");
    VOHIP_REQUIRE(ctx, (n_new == 0 || d_new_xy) && (nC == 0 || d_C_rc) && (nP == 0 || d_P_rc), "NULL array");
    if (n_new == 0) return VOHIP_OK;
    hipLaunchKernelGGL(candidate_filter_kernel, dim3((n_new + FILT_DET - 1) / FILT_DET), dim3(FILT_THREADS), 0, ctx->stream, d_new_xy, n_new,
                       n_new, d_C_rc, nC, d_P_rc, nP, (const int *)nullptr, (float)r, d_keep, vohip_seqs());
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

}  // extern "C"

// Frame-pipeline form: x and y columns `ld` apart, at most n_new_cap detections, the three counts
// {n_new, nC, nP} in device memory.
int vohip_candidate_filter_enqueue(vohip_ctx *ctx, const float *d_new_xy, int ld, int n_new_cap, const double *d_C_rc,
                                   const double *d_P_rc, const int *d_counts3, double r, uint8_t *d_keep,
                                   const vohip_seqs *seqs)
{
    const vohip_seqs SQ = seqs ? *seqs : vohip_seqs();
    const int ny = vohip_seq_count(SQ);
    if (n_new_cap <= 0 || ny < 1) return VOHIP_OK;
    hipLaunchKernelGGL(candidate_filter_kernel, dim3((n_new_cap + FILT_DET - 1) / FILT_DET, ny), dim3(FILT_THREADS), 0, ctx->stream, d_new_xy, ld,
                       n_new_cap, d_C_rc, 0, d_P_rc, 0, d_counts3, (float)r, d_keep, SQ);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

// Frame-pipeline form of the candidate triangulation: count and current pose in device memory.
int vohip_candidate_triangulate_enqueue(vohip_ctx *ctx, const double *d_cur_rc, const double *d_first_rc,
                                        const double *d_T_first, int n_cap, const int *d_n, const double *d_T_W_C,
                                        const double K[9], double angle_threshold_deg, double min_depth,
                                        double max_depth, int is_BA, uint8_t *d_status, double *d_X,
                                        const vohip_seqs *seqs)
{
    const vohip_seqs SQ = seqs ? *seqs : vohip_seqs();
    const int ny = vohip_seq_count(SQ);
    if (n_cap <= 0 || ny < 1) return VOHIP_OK;
    cand_params prm;
    std::memset(prm.T_W_C, 0, sizeof(prm.T_W_C));
    std::memcpy(prm.K, K, sizeof(prm.K));
    prm.angle_threshold_deg = angle_threshold_deg;
    prm.min_depth = min_depth;
    prm.max_depth = max_depth;
    prm.is_BA = is_BA;
    hipLaunchKernelGGL(candidate_triangulate_kernel, dim3((n_cap + 31) / 32, ny), dim3(128), 0, ctx->stream, d_cur_rc,
                       d_first_rc, d_T_first, n_cap, d_n, d_T_W_C, prm, d_status, d_X, (double *)nullptr, SQ);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

extern "C" {

int vohip_candidate_filter(vohip_ctx *ctx, const float *new_xy, int n_new, const double *C_rc, int nC,
                           const double *P_rc, int nP, double r, uint8_t *keep)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, n_new >= 0 && nC >= 0 && nP >= 0 && keep, "bad argument");
    if (n_new == 0) return VOHIP_OK;
    void *d_xy, *d_C, *d_P, *d_keep;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, new_xy, sizeof(float) * 2 * n_new, &d_xy));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, C_rc, sizeof(double) * 2 * nC, &d_C));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, P_rc, sizeof(double) * 2 * nP, &d_P));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, (size_t)n_new, &d_keep));
    VOHIP_TRY(vohip_candidate_filter_dev(ctx, (const float *)d_xy, n_new, (const double *)d_C, nC, (const double *)d_P,
                                         nP, r, (uint8_t *)d_keep));
    VOHIP_TRY(vohip_download(ctx, keep, d_keep, (size_t)n_new));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_candidate_triangulate_dev(vohip_ctx *ctx, const double *d_cur_rc, const double *d_first_rc,
                                    const double *d_T_first, int n, const double T_W_C[16], const double K[9],
                                    double angle_threshold_deg, double min_depth, double max_depth, int is_BA,
                                    uint8_t *d_status, double *d_X, double *d_angles)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, n >= 0 && T_W_C && K && d_status && d_X, "bad argument");
    if (n == 0) return VOHIP_OK;
    VOHIP_REQUIRE(ctx, d_cur_rc && d_first_rc && d_T_first, "NULL array");
    cand_params prm;
    std::memcpy(prm.T_W_C, T_W_C, sizeof(prm.T_W_C));
    std::memcpy(prm.K, K, sizeof(prm.K));
    prm.angle_threshold_deg = angle_threshold_deg;
    prm.min_depth = min_depth;
    prm.max_depth = max_depth;
    prm.is_BA = is_BA;
    hipLaunchKernelGGL(candidate_triangulate_kernel, dim3((n + 31) / 32), dim3(128), 0, ctx->stream, d_cur_rc,
                       d_first_rc, d_T_first, n, (const int *)nullptr, (const double *)nullptr, prm, d_status, d_X,
                       d_angles, vohip_seqs());
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

int vohip_candidate_triangulate(vohip_ctx *ctx, const double *cur_rc, const double *first_rc, const double *T_first,
                                int n, const double T_W_C[16], const double K[9], double angle_threshold_deg,
                                double min_depth, double max_depth, int is_BA, uint8_t *status, double *X,
                                double *angles)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, n >= 0 && status && X, "bad argument");
    if (n == 0) return VOHIP_OK;
    void *d_c, *d_f, *d_T, *d_s, *d_X, *d_a;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, cur_rc, sizeof(double) * 2 * n, &d_c));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, first_rc, sizeof(double) * 2 * n, &d_f));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, T_first, sizeof(double) * 12 * n, &d_T));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, (size_t)n, &d_s));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE4, nullptr, sizeof(double) * 3 * n, &d_X));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE5, nullptr, sizeof(double) * n, &d_a));
    VOHIP_TRY(vohip_candidate_triangulate_dev(ctx, (const double *)d_c, (const double *)d_f, (const double *)d_T, n,
                                              T_W_C, K, angle_threshold_deg, min_depth, max_depth, is_BA,
                                              (uint8_t *)d_s, (double *)d_X, (double *)d_a));
    VOHIP_TRY(vohip_download(ctx, status, d_s, (size_t)n));
    VOHIP_TRY(vohip_download(ctx, X, d_X, sizeof(double) * 3 * n));
    if (angles) VOHIP_TRY(vohip_download(ctx, angles, d_a, sizeof(double) * n));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

}  // extern "C"
