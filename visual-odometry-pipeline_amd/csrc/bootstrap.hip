// Relative camera pose from an essential matrix, for gfx950 (fp64).
//
// Stands in for [orientation, location] = relativeCameraPose(E, cameraParams, pts1, pts2) at
// Initialization/pose_estimation_init.m:29-33.  The toolbox function is closed source; the
// estimator is defined as:
//   E = U S V' by one-sided Jacobi (u3 = u1 x u2), rotations U W V' and U W' V' with the sign
//   fixed to det +1 and translations -u3, +u3 (utils/exe6_8point/decomposeEssentialMatrix.m:12-32);
//   the four candidates are tried in the order of disambiguateRelativePose.m:26-30; every pair is
//   triangulated with the DLT of linearTriangulation.m and the candidate with the fewest points
//   behind either camera wins (first minimum).  The result is returned the way the toolbox
//   returns it: orientation = R_C2_C1, location = -R_C2_C1' t with |t| = 1.
// A workgroup per 64 point pairs: lane 0 decomposes, a wavefront per candidate triangulates, ballots and integer atomics count.
#include "vohip_internal.h"

#include "geometry_device.h"

namespace {

struct pose_candidates {
    double R[2][9];  // row-major R_C2_C1
    double u3[3];
    int ok;
};

__device__ void decompose_essential(const double *E_cm, pose_candidates &out)
{
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = E_cm[3 * c + r];
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int i = 0; i < 2; ++i)
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
    // columns by decreasing norm (stable)
    double nn[3];
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) nn[c] = A[0][c] * A[0][c] + A[1][c] * A[1][c] + A[2][c] * A[2][c];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2 - a; ++b)
            if (nn[ord[b + 1]] > nn[ord[b]]) {
                const int t = ord[b];
                ord[b] = ord[b + 1];
                ord[b + 1] = t;
            }
    const double s1 = sqrt(nn[ord[0]]), s2 = sqrt(nn[ord[1]]);
    out.ok = (s1 > 0.0 && s2 > 1e-12 * s1) ? 1 : 0;
    if (!out.ok) return;
    double U[3][3], Vs[3][3];
    for (int r = 0; r < 3; ++r) {
        U[r][0] = A[r][ord[0]] / s1;
        U[r][1] = A[r][ord[1]] / s2;
        Vs[r][0] = V[r][ord[0]];
        Vs[r][1] = V[r][ord[1]];
    }
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    Vs[0][2] = Vs[1][0] * Vs[2][1] - Vs[2][0] * Vs[1][1];
    Vs[1][2] = Vs[2][0] * Vs[0][1] - Vs[0][0] * Vs[2][1];
    Vs[2][2] = Vs[0][0] * Vs[1][1] - Vs[1][0] * Vs[0][1];
    for (int r = 0; r < 3; ++r) out.u3[r] = U[r][2];
    // U W = [u2, -u1, u3], U W' = [-u2, u1, u3]; det(U) = det(Vs) = +1, so both products are
    // proper rotations already (the sign fix of decomposeEssentialMatrix.m:22-28 is a no-op)
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            out.R[0][3 * r + c] = U[r][1] * Vs[c][0] - U[r][0] * Vs[c][1] + U[r][2] * Vs[c][2];
            out.R[1][3 * r + c] = -U[r][1] * Vs[c][0] + U[r][0] * Vs[c][1] + U[r][2] * Vs[c][2];
        }
}

// result[0..8] orientation (column-major), [9..11] location, [12] valid fraction, [13] status
// (1 = ok, 0 = E has no two non-zero singular values).
// A workgroup serves 64 point pairs: wavefront k triangulates them under candidate k (one triangulation per
// thread; a single workgroup walking all pairs four times took 247 us at n = 2 000).  Every workgroup decomposes E
// itself (one lane, a few microseconds; cheaper than a launch of its own).  cnt[0..3]: points behind a camera per
// candidate, integer atomics, so the sums do not depend on the order; cnt[4]: tickets -- the workgroup that draws
// the last one picks the candidate and writes the result.  cnt arrives zeroed.
__global__ __launch_bounds__(256) void relative_pose_kernel(const double *__restrict__ E_cm,
                                                            const double *__restrict__ K_cm,
                                                            const double *__restrict__ p1,
                                                            const double *__restrict__ p2, int n,
                                                            double *__restrict__ result, int *__restrict__ cnt)
{
    __shared__ pose_candidates cand;
    __shared__ double sM1[12], sM2[4][12], sRt[4][12];
    __shared__ int sLast;
    const int tid = threadIdx.x;
    if (tid == 0) decompose_essential(E_cm, cand);
    __syncthreads();
    if (!cand.ok) {
        if (blockIdx.x == 0 && tid < 14) result[tid] = 0.0;
        return;
    }
    if (tid < 4) {
        // candidate order: (R1, -u3), (R1, +u3), (R2, -u3), (R2, +u3)
        const double *R = cand.R[tid >> 1];
        const double sg = (tid & 1) ? 1.0 : -1.0;
        double Rt[12];  // column-major 3x4 [R | t]
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Rt[3 * c + r] = R[3 * r + c];
            Rt[9 + r] = sg * cand.u3[r];
        }
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 3; ++r) {
                sRt[tid][3 * c + r] = Rt[3 * c + r];
                sM2[tid][3 * c + r] = K_cm[r] * Rt[3 * c] + K_cm[3 + r] * Rt[3 * c + 1] + K_cm[6 + r] * Rt[3 * c + 2];
            }
        if (tid == 0)
            for (int c = 0; c < 4; ++c)
                for (int r = 0; r < 3; ++r) sM1[3 * c + r] = c < 3 ? K_cm[3 * c + r] : 0.0;
    }
    __syncthreads();
    {
        const int k = tid >> 6, i = blockIdx.x * 64 + (tid & 63);
        const bool live = i < n;
        const int ii = live ? i : 0;
        const double a[3] = {p1[2 * ii], p1[2 * ii + 1], 1.0};
        const double b[3] = {p2[2 * ii], p2[2 * ii + 1], 1.0};
        double P[4];
        vohip_dev::triangulate_point(a, b, sM1, sM2[k], P);
        const double *Rt = sRt[k];
        const double z2 = Rt[2] * P[0] + Rt[5] * P[1] + Rt[8] * P[2] + Rt[11];
        const bool behind = live && (P[2] < 0.0 || z2 < 0.0);
        const int neg = __popcll(__ballot(behind));
        if ((tid & 63) == 0 && neg) atomicAdd(&cnt[k], neg);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) sLast = atomicAdd(&cnt[4], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!sLast) return;
    __threadfence();
    if (tid == 0) {
        int best = 0, best_neg = 0x7fffffff;
        for (int k = 0; k < 4; ++k) {
            const int c = __hip_atomic_load(&cnt[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (c < best_neg) {
                best_neg = c;
                best = k;
            }
        }
        const double *Rt = sRt[best];
        for (int q = 0; q < 9; ++q) result[q] = Rt[q];
        for (int r = 0; r < 3; ++r)
            result[9 + r] = -(Rt[3 * r] * Rt[9] + Rt[3 * r + 1] * Rt[10] + Rt[3 * r + 2] * Rt[11]);
        result[12] = n > 0 ? 1.0 - (double)best_neg / (double)n : 0.0;
        result[13] = 1.0;
    }
}

}  // namespace

extern "C" int vohip_relative_pose(vohip_ctx *ctx, const double *E, const double *K, const double *p1,
                                   const double *p2, int n, double *orientation, double *location,
                                   double *valid_fraction)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, E && K && p1 && p2 && orientation && location, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1, "need at least one point pair");
    void *d_E, *d_K, *d_p1, *d_p2, *d_res;
    double head[13];   // E, then the kernel's five integer counters, zeroed (one copy)
    for (int q = 0; q < 9; ++q) head[q] = E[q];
    for (int q = 9; q < 13; ++q) head[q] = 0.0;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, head, sizeof(head), &d_E));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, K, sizeof(double) * 9, &d_K));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, p1, sizeof(double) * 2 * (size_t)n, &d_p1));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, p2, sizeof(double) * 2 * (size_t)n, &d_p2));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE4, nullptr, sizeof(double) * 14, &d_res));
    hipLaunchKernelGGL(relative_pose_kernel, dim3((n + 63) / 64), dim3(256), 0, ctx->stream, (const double *)d_E,
                       (const double *)d_K, (const double *)d_p1, (const double *)d_p2, n, (double *)d_res,
                       reinterpret_cast<int *>((double *)d_E + 9));
    VOHIP_HIP(ctx, hipGetLastError());
    double res[14];
    VOHIP_TRY(vohip_download(ctx, res, d_res, sizeof(res)));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (res[13] == 0.0) return vohip_fail(ctx, VOHIP_NO_MODEL, "the matrix has fewer than two non-zero singular values");
    for (int q = 0; q < 9; ++q) orientation[q] = res[q];
    for (int q = 0; q < 3; ++q) location[q] = res[9 + q];
    if (valid_fraction) *valid_fraction = res[12];
    return VOHIP_OK;
}
