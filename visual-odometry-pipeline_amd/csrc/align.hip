// Similarity alignment of an estimated trajectory to ground truth, for gfx950 (fp64).
//
// utils/exe9_BA/alignEstimateToGroundTruth.m:1-27 with the residual of alignError.m:1-17:
//   x = [twist of T_G_V (v; w); scale],  error_i = pp_G_C_i - (scale * R p_V_C_i + t),
// started from the identity twist and scale 1 (:10-13).  The reference's lsqnonlin call (:17-18) is
// replaced by the library's Levenberg-Marquardt (the one behind T_refinement: lambda0 1e-3, /10 on
// accept with floor 1e-12, x10 on reject, stop after 10 consecutive rejects) with analytic
// Jacobians; one workgroup runs the whole loop in-kernel (block-wide sums of the 7x7 normal
// equations, Cholesky by one lane).
#include "vohip_internal.h"
#include "se3_device.h"

#include <vector>

using namespace vohip_dev;

namespace {

constexpr int AL_THREADS = 256;
constexpr int NP = 7;

// y = s R p + t;  dy/dv_k = V[:,k];  dy/dw_k = s R (Jr[:,k] x p) + dVv[:,k];  dy/ds = R p.
struct sim3_lin {
    double R[3][3], V[3][3], Jr[3][3], dVv[3][3], t[3], s;
};

__device__ __forceinline__ void cross3a(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ void sim3_prepare(const double x[NP], sim3_lin &f)
{
    const double *v = x, *w = x + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B, C, D;
    so3_coeffs(th2, A, B, C, D);
    double B1, C1;   // dB/d(th^2) * 2 and dC/d(th^2) * 2, i.e. (A - 2B)/th^2 and (B - 3C)/th^2
    if (th2 < 1e-2) {
        B1 = -1.0 / 12.0 + th2 * (1.0 / 180.0 + th2 * (-1.0 / 6720.0 + th2 * (1.0 / 453600.0 - th2 / 47900160.0)));
        C1 = -1.0 / 60.0 + th2 * (1.0 / 1260.0 + th2 * (-1.0 / 60480.0 + th2 * (1.0 / 4989600.0 - th2 / 622702080.0)));
    } else {
        B1 = (A - 2.0 * B) / th2;
        C1 = (B - 3.0 * C) / th2;
    }
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double W2 = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
            const double I = i == j ? 1.0 : 0.0;
            f.R[i][j] = I + A * W[i][j] + B * W2;
            f.V[i][j] = I + B * W[i][j] + C * W2;
            f.Jr[i][j] = I - B * W[i][j] + C * W2;
        }
    for (int i = 0; i < 3; ++i) f.t[i] = f.V[i][0] * v[0] + f.V[i][1] * v[1] + f.V[i][2] * v[2];
    double wxv[3], wwxv[3];
    cross3a(w, v, wxv);
    cross3a(w, wxv, wwxv);
    for (int k = 0; k < 3; ++k) {
        double e[3] = {0, 0, 0}, exv[3], exwxv[3], wxexv[3];
        e[k] = 1.0;
        cross3a(e, v, exv);
        cross3a(e, wxv, exwxv);
        cross3a(w, exv, wxexv);
        for (int i = 0; i < 3; ++i)
            f.dVv[i][k] = B1 * w[k] * wxv[i] + B * exv[i] + C1 * w[k] * wwxv[i] + C * (exwxv[i] + wxexv[i]);
    }
    f.s = x[6];
}

__device__ __forceinline__ void sim3_apply(const sim3_lin &f, const double p[3], double y[3])
{
    for (int i = 0; i < 3; ++i) y[i] = f.s * (f.R[i][0] * p[0] + f.R[i][1] * p[1] + f.R[i][2] * p[2]) + f.t[i];
}

// Sums over the workgroup of the NS values every thread holds: butterflies inside each wavefront, the
// wavefronts' partial sums through LDS (one barrier), added in wavefront order by the first NS threads and left
// in sOut (visible after the second barrier).  Two barriers per set of sums, whatever their number.
template <int NS>
__device__ __forceinline__ void al_block_sums(double (&v)[NS], double *sPart, double *sOut)
{
    constexpr int NW = AL_THREADS / 64;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v[q] += __shfl_xor(v[q], m, 64);
    }
    __syncthreads();   // (sPart and sOut of the previous round have been read)
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

// One workgroup per problem.  desc == NULL: a single problem over points [0, n).  Otherwise problem b =
// blockIdx.x reads points [desc[3b], desc[3b] + desc[3b + 1]) of gt / est and writes its aligned points from
// point desc[3b + 2] of `aligned` on, x_out + 7 b and stats + 3 b (the checkpoints of quantitative_eval.m:20-27
// are prefixes of one trajectory: they share their input and run side by side).
// gt, est 3 x n column-major.  x_out[7]; aligned 3 x n; stats = {iterations, cost0, cost}.
__global__ __launch_bounds__(AL_THREADS) void align_sim3_kernel(const double *__restrict__ gt,
                                                                const double *__restrict__ est, int n, int max_iter,
                                                                double ftol, double xtol, double *__restrict__ x_out,
                                                                double *__restrict__ aligned,
                                                                double *__restrict__ stats,
                                                                const int *__restrict__ desc)
{
    constexpr int NH = NP * (NP + 1) / 2, NS = NH + NP;
    __shared__ sim3_lin sF, sFn;
    __shared__ double sX[NP], sXn[NP], sH[NP * NP], sG[NP], sD[NP], sPart[(AL_THREADS / 64) * NS], sSum[NS];
    __shared__ int sOk;
    const int tid = threadIdx.x;
    if (desc) {
        const int b = blockIdx.x;
        gt += 3 * (size_t)desc[3 * b];
        est += 3 * (size_t)desc[3 * b];
        n = desc[3 * b + 1];
        aligned += 3 * (size_t)desc[3 * b + 2];
        x_out += NP * (size_t)b;
        stats += 3 * (size_t)b;
    }
    if (tid == 0) {
        for (int k = 0; k < NP; ++k) sX[k] = k == 6 ? 1.0 : 0.0;
        double x[NP];
        for (int k = 0; k < NP; ++k) x[k] = sX[k];
        sim3_prepare(x, sF);
    }
    __syncthreads();
    auto cost_of = [&](const sim3_lin &f) {
        double c[1] = {0.0};
        for (int k = tid; k < n; k += AL_THREADS) {
            const double p[3] = {est[3 * k], est[3 * k + 1], est[3 * k + 2]};
            double y[3];
            sim3_apply(f, p, y);
            for (int i = 0; i < 3; ++i) {
                const double e = gt[3 * k + i] - y[i];
                c[0] += e * e;
            }
        }
        al_block_sums<1>(c, sPart, sSum);
        return sSum[0];
    };
    double cost = cost_of(sF);
    const double cost0 = cost;
    double lambda = 1e-3;
    int iter = 0, rejects = 0;
    while (iter < max_iter) {
        double hg[NS];   // upper triangle of J'J, then J'e
        for (int q = 0; q < NS; ++q) hg[q] = 0.0;
        for (int k = tid; k < n; k += AL_THREADS) {
            const double p[3] = {est[3 * k], est[3 * k + 1], est[3 * k + 2]};
            double y[3], e[3], J[3][NP];   // J = d error / d x = -dy/dx
            sim3_apply(sF, p, y);
            for (int i = 0; i < 3; ++i) e[i] = gt[3 * k + i] - y[i];
            for (int c = 0; c < 3; ++c) {
                const double jr[3] = {sF.Jr[0][c], sF.Jr[1][c], sF.Jr[2][c]};
                double jxp[3];
                cross3a(jr, p, jxp);
                for (int i = 0; i < 3; ++i) {
                    J[i][c] = -sF.V[i][c];
                    J[i][3 + c] = -(sF.s * (sF.R[i][0] * jxp[0] + sF.R[i][1] * jxp[1] + sF.R[i][2] * jxp[2]) + sF.dVv[i][c]);
                }
            }
            for (int i = 0; i < 3; ++i) J[i][6] = -(sF.R[i][0] * p[0] + sF.R[i][1] * p[1] + sF.R[i][2] * p[2]);
            int q = 0;
#pragma unroll
            for (int a = 0; a < NP; ++a) {
                hg[NH + a] += J[0][a] * e[0] + J[1][a] * e[1] + J[2][a] * e[2];
#pragma unroll
                for (int b = a; b < NP; ++b) hg[q++] += J[0][a] * J[0][b] + J[1][a] * J[1][b] + J[2][a] * J[2][b];
            }
        }
        al_block_sums<NS>(hg, sPart, sSum);
        if (tid == 0) {
            int q = 0;
            for (int a = 0; a < NP; ++a)
                for (int b = a; b < NP; ++b) sH[NP * a + b] = sH[NP * b + a] = sSum[q++];
            for (int a = 0; a < NP; ++a) sG[a] = sSum[NH + a];
        }
        __syncthreads();
        bool accepted = false, stop = false;
        while (!accepted && !stop) {
            __syncthreads();
            if (tid == 0) {
                // 7x7 Cholesky with every loop unrolled, so that A and d stay in registers
                double A[NP * NP], d[NP];
#pragma unroll
                for (int q = 0; q < NP * NP; ++q) A[q] = sH[q];
#pragma unroll
                for (int a = 0; a < NP; ++a) {
                    A[(NP + 1) * a] *= 1.0 + lambda;
                    d[a] = -sG[a];
                }
                int ok = 1;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    double dd = A[j * NP + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) dd -= A[j * NP + k] * A[j * NP + k];
                    if (!(dd > 0.0)) ok = 0;
                    dd = sqrt(dd);
                    A[j * NP + j] = dd;
#pragma unroll
                    for (int i = j + 1; i < NP; ++i) {
                        double s = A[i * NP + j];
#pragma unroll
                        for (int k = 0; k < j; ++k) s -= A[i * NP + k] * A[j * NP + k];
                        A[i * NP + j] = s / dd;
                    }
                }
                if (ok) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        double s = d[i];
#pragma unroll
                        for (int k = 0; k < i; ++k) s -= A[i * NP + k] * d[k];
                        d[i] = s / A[i * NP + i];
                    }
#pragma unroll
                    for (int i = NP - 1; i >= 0; --i) {
                        double s = d[i];
#pragma unroll
                        for (int k = i + 1; k < NP; ++k) s -= A[k * NP + i] * d[k];
                        d[i] = s / A[i * NP + i];
                    }
                    double xn[NP];
                    for (int a = 0; a < NP; ++a) {
                        sD[a] = d[a];
                        xn[a] = sX[a] + d[a];
                        sXn[a] = xn[a];
                    }
                    sim3_prepare(xn, sFn);
                }
                sOk = ok;
            }
            __syncthreads();
            if (!sOk) {
                lambda *= 10.0;
                if (++rejects > 10) stop = true;
                continue;
            }
            const double cn = cost_of(sFn);
            if (cn < cost) {
                double dn = 0.0, xn = 0.0;
                for (int a = 0; a < NP; ++a) {
                    dn += sD[a] * sD[a];
                    xn += sX[a] * sX[a];
                }
                const double dec = cost - cn, cost_old = cost;
                __syncthreads();
                if (tid == 0) {
                    for (int a = 0; a < NP; ++a) sX[a] = sXn[a];
                    sF = sFn;
                }
                __syncthreads();
                cost = cn;
                lambda = lambda * 0.1 > 1e-12 ? lambda * 0.1 : 1e-12;
                ++iter;
                rejects = 0;
                accepted = true;
                if (dec < ftol * (1.0 + cost_old)) stop = true;
                if (sqrt(dn) < xtol * (1.0 + sqrt(xn))) stop = true;
            } else {
                lambda *= 10.0;
                if (++rejects > 10) stop = true;
            }
        }
        if (stop) break;
    }
    __syncthreads();
    for (int k = tid; k < n; k += AL_THREADS) {
        const double p[3] = {est[3 * k], est[3 * k + 1], est[3 * k + 2]};
        double y[3];
        sim3_apply(sF, p, y);
        for (int i = 0; i < 3; ++i) aligned[3 * k + i] = y[i];
    }
    if (tid == 0) {
        for (int k = 0; k < NP; ++k) x_out[k] = sX[k];
        stats[0] = (double)iter;
        stats[1] = cost0;
        stats[2] = cost;
    }
}

}  // namespace

extern "C" {

int vohip_align_sim3(vohip_ctx *ctx, const double *pp_G_C, const double *p_V_C, int n, int max_iter,
                     double ftol, double xtol, double *x, double *p_G_C, double *stats)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, pp_G_C && p_V_C && p_G_C, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && max_iter >= 0, "need at least one point");
    void *d_gt, *d_est, *d_out;
    const size_t bytes = sizeof(double) * 3 * (size_t)n;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, pp_G_C, bytes, &d_gt));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, p_V_C, bytes, &d_est));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, bytes + sizeof(double) * 10, &d_out));
    double *d_x = (double *)d_out + 3 * (size_t)n;
    hipLaunchKernelGGL(align_sim3_kernel, dim3(1), dim3(AL_THREADS), 0, ctx->stream, (const double *)d_gt,
                       (const double *)d_est, n, max_iter, ftol, xtol, d_x, (double *)d_out, d_x + 7, (const int *)nullptr);
    VOHIP_HIP(ctx, hipGetLastError());
    double tail[10];
    VOHIP_TRY(vohip_download(ctx, p_G_C, d_out, bytes));
    VOHIP_TRY(vohip_download(ctx, tail, d_x, sizeof(tail)));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (x)
        for (int k = 0; k < 7; ++k) x[k] = tail[k];
    if (stats)
        for (int k = 0; k < 3; ++k) stats[k] = tail[7 + k];
    return VOHIP_OK;
}

int vohip_align_sim3_prefixes(vohip_ctx *ctx, const double *pp_G_C, const double *p_V_C, int n, int k, const int *lengths,
                              int max_iter, double ftol, double xtol, double *x, double *first_last, double *stats)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, pp_G_C && p_V_C && lengths && first_last, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && max_iter >= 0 && k >= 0 && k <= 4096, "bad sizes");
    if (k == 0) return VOHIP_OK;
    std::vector<int> desc(3 * (size_t)k);
    size_t total = 0;
    for (int b = 0; b < k; ++b) {
        VOHIP_REQUIRE(ctx, lengths[b] >= 1 && lengths[b] <= n, "a prefix holds 1..n points");
        desc[3 * b] = 0;
        desc[3 * b + 1] = lengths[b];
        desc[3 * b + 2] = (int)total;
        total += (size_t)lengths[b];
        VOHIP_REQUIRE(ctx, total < 0x3fffffff, "too many points");
    }
    void *d_gt, *d_est, *d_out, *d_desc;
    const size_t bytes = sizeof(double) * 3 * (size_t)n;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, pp_G_C, bytes, &d_gt));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, p_V_C, bytes, &d_est));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, nullptr, sizeof(double) * (3 * total + 10 * (size_t)k), &d_out));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, desc.data(), sizeof(int) * desc.size(), &d_desc));
    double *d_x = (double *)d_out + 3 * total, *d_stats = d_x + 7 * (size_t)k;
    hipLaunchKernelGGL(align_sim3_kernel, dim3(k), dim3(AL_THREADS), 0, ctx->stream, (const double *)d_gt,
                       (const double *)d_est, n, max_iter, ftol, xtol, d_x, (double *)d_out, d_stats, (const int *)d_desc);
    VOHIP_HIP(ctx, hipGetLastError());
    std::vector<double> tail(10 * (size_t)k);
    for (int b = 0; b < k; ++b) {   // the aligned prefix's first and last point (quantitative_eval.m:25-26 reads no other)
        const double *o = (const double *)d_out + 3 * (size_t)desc[3 * b + 2];
        VOHIP_TRY(vohip_download(ctx, first_last + 6 * (size_t)b, o, sizeof(double) * 3));
        VOHIP_TRY(vohip_download(ctx, first_last + 6 * (size_t)b + 3, o + 3 * ((size_t)lengths[b] - 1), sizeof(double) * 3));
    }
    VOHIP_TRY(vohip_download(ctx, tail.data(), d_x, sizeof(double) * tail.size()));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (desc and tail are released after it)
    if (x) std::memcpy(x, tail.data(), sizeof(double) * 7 * (size_t)k);
    if (stats) std::memcpy(stats, tail.data() + 7 * (size_t)k, sizeof(double) * 3 * (size_t)k);
    return VOHIP_OK;
}

}  // extern "C"
