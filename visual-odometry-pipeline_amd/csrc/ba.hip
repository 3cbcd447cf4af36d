// Sliding-window bundle adjustment and pose-only refinement for gfx950 (fp64).
//
// Stands in for utils/exe9_BA/runBA.m:2-88 + baError.m:2-65 (which the reference minimises with
// lsqnonlin, closed source) and utils/T_refinement.m:1-21.  The residual function is baError's;
// the minimiser is a Levenberg-Marquardt defined here:
//   unknowns: raw twist coordinates [v; w] of every T_W_C and the landmark coordinates; landmarks
//   with fewer than 2 observations are held fixed (runBA.m:45-49); no gauge fixing;
//   analytic Jacobians; step from (J'J + lambda diag(J'J)) d = -J'r via the Schur complement over
//   the landmarks; lambda0 = 1e-3, /10 on accept (floor 1e-12), x10 on reject, stop after 10
//   consecutive rejects, after max_iter accepted steps, when an accepted step lowers the cost by
//   less than ftol (1 + cost) or is shorter than xtol (1 + |x|).
//
// Kernels of one LM trial (no host arithmetic; the host only polls a status word):
//   ba_linearize_kernel  one workgroup per 16 landmarks: residuals and Jacobian blocks of their
//                        observations into LDS, per-landmark V (3x3) and its damped inverse,
//                        Y = W V^-1, then each thread owns entries of the (6F)^2 reduced camera
//                        system and sums U - Y W' over the workgroup's landmarks in a fixed order
//                        -> one partial system per workgroup (deterministic, no atomics).
//   ba_solve_kernel      one workgroup: sums the partial systems in order, Cholesky in LDS,
//                        pose step.
//   ba_update_kernel     one thread per landmark: back-substitution, trial state, cost of the
//                        trial state over the landmark's observations -> partial costs.
//   (ba_decide)          the workgroup of ba_update_kernel that finishes last: accept / reject, lambda update,
//                        stopping tests, commit.
// At the reference's sizes (F = 5 poses, 10^2..10^3 landmarks) the reduced camera system is 30x30:
// far too small for a matrix-core contraction, so everything is fp64 VALU.
//
// pose_refine_kernel: T_refinement as ONE workgroup running the whole LM loop in-kernel
// (one pass over the points per step gives a trial pose its cost and its 6x6 normal equations;
// Cholesky by one lane).
#include "vohip_internal.h"
#include "se3_device.h"
#include "wave_reduce.h"
#include "seq_index.h"

#include <chrono>
#include <cstdlib>
#include <vector>

using namespace vohip_dev;

namespace {

constexpr int FMAX = 8;    // most keyframes in a window
constexpr int LT = 16;     // landmarks per linearisation workgroup

// Per-frame linearisation data for the twist [v; w] of T_W_C:
//   p_C = Rcw X + tcw;  d p_C / d v_k = Mv[:,k];  d p_C / d w_k = p_C x Jr[:,k] + Dw[:,k].
struct frame_lin {
    double Rcw[3][3], tcw[3], Mv[3][3], Jr[3][3], Dw[3][3];
};

__device__ __forceinline__ void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ void frame_prepare(const double tw[6], frame_lin &f)
{
    const double *v = tw, *w = tw + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double A, B, C, D;
    so3_coeffs(th2, A, B, C, D);
    double B1, C1;
    if (th2 < 1e-2) {
        B1 = -1.0 / 12.0 + th2 * (1.0 / 180.0 + th2 * (-1.0 / 6720.0 + th2 * (1.0 / 453600.0 - th2 / 47900160.0)));
        C1 = -1.0 / 60.0 + th2 * (1.0 / 1260.0 + th2 * (-1.0 / 60480.0 + th2 * (1.0 / 4989600.0 - th2 / 622702080.0)));
    } else {
        B1 = (A - 2.0 * B) / th2;
        C1 = (B - 3.0 * C) / th2;
    }
    const double W[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
    double R[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double W2 = W[i][0] * W[0][j] + W[i][1] * W[1][j] + W[i][2] * W[2][j];
            const double I = i == j ? 1.0 : 0.0;
            R[i][j] = I + A * W[i][j] + B * W2;
            V[i][j] = I + B * W[i][j] + C * W2;
            f.Jr[i][j] = I - B * W[i][j] + C * W2;
        }
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = V[i][0] * v[0] + V[i][1] * v[1] + V[i][2] * v[2];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            f.Rcw[i][j] = R[j][i];
            f.Mv[i][j] = -(R[0][i] * V[0][j] + R[1][i] * V[1][j] + R[2][i] * V[2][j]);
        }
        f.tcw[i] = -(R[0][i] * t[0] + R[1][i] * t[1] + R[2][i] * t[2]);
    }
    double wxv[3], wwxv[3];
    cross3(w, v, wxv);
    cross3(w, wxv, wwxv);
    for (int k = 0; k < 3; ++k) {
        double e[3] = {0, 0, 0}, exv[3], exwxv[3], wxexv[3], dV[3];
        e[k] = 1.0;
        cross3(e, v, exv);
        cross3(e, wxv, exwxv);
        cross3(w, exv, wxexv);
        for (int i = 0; i < 3; ++i)
            dV[i] = B1 * w[k] * wxv[i] + B * exv[i] + C1 * w[k] * wwxv[i] + C * (exwxv[i] + wxexv[i]);
        for (int i = 0; i < 3; ++i) f.Dw[i][k] = -(R[0][i] * dV[0] + R[1][i] * dV[1] + R[2][i] * dV[2]);
    }
}

// residual kp - projection (baError.m:36-59) and, when WITH_J, its Jacobians Jp (2x6), Jl (2x3).
template <bool WITH_J>
__device__ __forceinline__ void observe(const frame_lin &f, const double *K, const double X[3], double u, double v,
                                        double r[2], double Jp[2][6], double Jl[2][3])
{
    double p[3], h[3];
    for (int i = 0; i < 3; ++i) p[i] = f.Rcw[i][0] * X[0] + f.Rcw[i][1] * X[1] + f.Rcw[i][2] * X[2] + f.tcw[i];
    for (int i = 0; i < 3; ++i) h[i] = K[i] * p[0] + K[3 + i] * p[1] + K[6 + i] * p[2];
    const double iz = 1.0 / h[2];
    const double pu = h[0] * iz, pv = h[1] * iz;
    r[0] = u - pu;
    r[1] = v - pv;
    if (WITH_J) {
        double D[2][3];
        for (int c = 0; c < 3; ++c) {
            D[0][c] = -(K[3 * c] - pu * K[3 * c + 2]) * iz;
            D[1][c] = -(K[3 * c + 1] - pv * K[3 * c + 2]) * iz;
        }
        for (int k = 0; k < 3; ++k) {
            const double dv[3] = {f.Mv[0][k], f.Mv[1][k], f.Mv[2][k]};
            const double jr[3] = {f.Jr[0][k], f.Jr[1][k], f.Jr[2][k]};
            double dw[3];
            cross3(p, jr, dw);
            for (int i = 0; i < 3; ++i) dw[i] += f.Dw[i][k];
            for (int a = 0; a < 2; ++a) {
                Jp[a][k] = D[a][0] * dv[0] + D[a][1] * dv[1] + D[a][2] * dv[2];
                Jp[a][3 + k] = D[a][0] * dw[0] + D[a][1] * dw[1] + D[a][2] * dw[2];
            }
        }
        for (int a = 0; a < 2; ++a)
            for (int c = 0; c < 3; ++c)
                Jl[a][c] = D[a][0] * f.Rcw[0][c] + D[a][1] * f.Rcw[1][c] + D[a][2] * f.Rcw[2][c];
    }
}

__device__ __forceinline__ bool inv3_sym(const double V[3][3], double E[3][3])
{
    const double c00 = V[1][1] * V[2][2] - V[1][2] * V[2][1];
    const double c01 = V[1][2] * V[2][0] - V[1][0] * V[2][2];
    const double c02 = V[1][0] * V[2][1] - V[1][1] * V[2][0];
    const double det = V[0][0] * c00 + V[0][1] * c01 + V[0][2] * c02;
    if (!(det > 0.0)) return false;
    const double id = 1.0 / det;
    E[0][0] = c00 * id;
    E[0][1] = E[1][0] = c01 * id;
    E[0][2] = E[2][0] = c02 * id;
    E[1][1] = (V[0][0] * V[2][2] - V[0][2] * V[2][0]) * id;
    E[1][2] = E[2][1] = (V[0][2] * V[1][0] - V[0][0] * V[1][2]) * id;
    E[2][2] = (V[0][0] * V[1][1] - V[0][1] * V[1][0]) * id;
    return true;
}

// Device-resident LM state, polled by the host after each trial.
struct ba_status {
    double cost, cost_trial, lambda, step_norm2, x_norm2, cost0;
    int iter, rejects, done, solve_failed, accepted_last, trials;
    int cur;   // which of the two state buffers holds the current estimate (the other one takes the trial step)
};

struct ba_problem {
    int F, m, nobs;
    const int *obs_index;   // m x F: observation id of landmark j in frame f, or -1
    const int *obs_frame;   // nobs
    const int *obs_lm;      // nobs
    const double *obs_uv;   // 2 x nobs
    const int *lm_count;    // m: observations per landmark
    double K[9];
};

__global__ __launch_bounds__(256) void ba_error_kernel(ba_problem P, const double *__restrict__ state,
                                                       double *__restrict__ r_out)
{
    __shared__ frame_lin sF[FMAX];
    if (threadIdx.x < P.F) {
        double tw[6];
        for (int k = 0; k < 6; ++k) tw[k] = state[6 * threadIdx.x + k];
        frame_prepare(tw, sF[threadIdx.x]);
    }
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P.nobs) return;
    const double *X = state + 6 * P.F + 3 * P.obs_lm[k];
    const double Xl[3] = {X[0], X[1], X[2]};
    double r[2];
    observe<false>(sF[P.obs_frame[k]], P.K, Xl, P.obs_uv[2 * k], P.obs_uv[2 * k + 1], r, nullptr, nullptr);
    r_out[2 * k] = r[0];
    r_out[2 * k + 1] = r[1];
}

// Initial cost: partial sums per workgroup of ba_error-like evaluation.
__global__ __launch_bounds__(256) void ba_cost_kernel(ba_problem P, const double *__restrict__ state,
                                                      double *__restrict__ partial)
{
    __shared__ frame_lin sF[FMAX];
    __shared__ double sRed[256];
    if (threadIdx.x < P.F) {
        double tw[6];
        for (int k = 0; k < 6; ++k) tw[k] = state[6 * threadIdx.x + k];
        frame_prepare(tw, sF[threadIdx.x]);
    }
    __syncthreads();
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    double c = 0.0;
    if (k < P.nobs) {
        const double *X = state + 6 * P.F + 3 * P.obs_lm[k];
        const double Xl[3] = {X[0], X[1], X[2]};
        double r[2];
        observe<false>(sF[P.obs_frame[k]], P.K, Xl, P.obs_uv[2 * k], P.obs_uv[2 * k + 1], r, nullptr, nullptr);
        c = r[0] * r[0] + r[1] * r[1];
    }
    sRed[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sRed[threadIdx.x] += sRed[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sRed[0];
}

__global__ __launch_bounds__(256) void ba_init_status_kernel(const double *__restrict__ partial, int nblocks,
                                                             ba_status *__restrict__ st)
{
    if (threadIdx.x == 0) {
        double c = 0.0;
        for (int b = 0; b < nblocks; ++b) c += partial[b];
        st->cost = st->cost0 = c;
        st->cost_trial = 0.0;
        st->lambda = 1e-3;
        st->iter = st->rejects = st->done = st->solve_failed = st->accepted_last = st->trials = st->cur = 0;
        st->step_norm2 = st->x_norm2 = 0.0;
    }
}

// LDS layout of ba_linearize_kernel (dynamic): per (local landmark, frame) slot
//   Jp 12, r 2, W 18, Y 18 doubles; per landmark E 9, gl 3; frames.
__global__ __launch_bounds__(256) void ba_linearize_kernel(ba_problem P, const double *__restrict__ X0,
                                                           const double *__restrict__ X1,
                                                           const ba_status *__restrict__ st,
                                                           double *__restrict__ S_part, double *__restrict__ g_part,
                                                           double *__restrict__ E_out, double *__restrict__ gl_out,
                                                           double *__restrict__ W_out, int *__restrict__ fail_flag)
{
    extern __shared__ double lds[];
    const int F = P.F, np = 6 * F, tid = threadIdx.x;
    frame_lin *sF = reinterpret_cast<frame_lin *>(lds);
    double *sJp = lds + (sizeof(frame_lin) / sizeof(double)) * FMAX;  // [LT*F][12]
    double *sR = sJp + LT * F * 12;                                     // [LT*F][2]
    double *sW = sR + LT * F * 2;                                       // [LT*F][18]
    double *sY = sW + LT * F * 18;                                      // [LT*F][18]
    double *sE = sY + LT * F * 18;                                      // [LT][9]
    double *sGl = sE + LT * 9;                                          // [LT][3]
    if (st->done) return;
    const double *__restrict__ state = st->cur ? X1 : X0;
    const double lambda = st->lambda;
    if (tid < F) {
        double tw[6];
        for (int k = 0; k < 6; ++k) tw[k] = state[6 * tid + k];
        frame_prepare(tw, sF[tid]);
    }
    __syncthreads();
    const int j0 = blockIdx.x * LT;
    // 1. residuals and Jacobian blocks of every (landmark, frame) slot
    for (int s = tid; s < LT * F; s += 256) {
        const int jl = s / F, f = s - jl * F, j = j0 + jl;
        double Jp[2][6] = {{0}}, Jl[2][3] = {{0}}, r[2] = {0, 0};
        int k = -1;
        bool frozen = true;
        if (j < P.m) {
            k = P.obs_index[j * F + f];
            frozen = P.lm_count[j] < 2;
        }
        if (k >= 0) {
            const double *X = state + np + 3 * j;
            const double Xl[3] = {X[0], X[1], X[2]};
            observe<true>(sF[f], P.K, Xl, P.obs_uv[2 * k], P.obs_uv[2 * k + 1], r, Jp, Jl);
        }
        for (int a = 0; a < 6; ++a) {
            sJp[s * 12 + a] = Jp[0][a];
            sJp[s * 12 + 6 + a] = Jp[1][a];
            for (int c = 0; c < 3; ++c) {
                const double w = (k >= 0 && !frozen) ? Jp[0][a] * Jl[0][c] + Jp[1][a] * Jl[1][c] : 0.0;
                sW[s * 18 + 3 * a + c] = w;
                if (k >= 0) W_out[(size_t)k * 18 + 3 * a + c] = w;
            }
        }
        sR[s * 2] = r[0];
        sR[s * 2 + 1] = r[1];
        // stash Jl in the Y slot until step 3 overwrites it
        for (int a = 0; a < 2; ++a)
            for (int c = 0; c < 3; ++c) sY[s * 18 + 3 * a + c] = (k >= 0 && !frozen) ? Jl[a][c] : 0.0;
    }
    __syncthreads();
    // 2. per landmark: V = sum Jl'Jl, gl = sum Jl'r, E = (V + lambda diag V)^-1
    if (tid < LT) {
        const int j = j0 + tid;
        double V[3][3] = {{0}}, gl[3] = {0, 0, 0}, E[3][3] = {{0}};
        for (int f = 0; f < F; ++f) {
            const double *Jl = sY + (tid * F + f) * 18;
            const double *r = sR + (tid * F + f) * 2;
            for (int a = 0; a < 3; ++a) {
                gl[a] += Jl[a] * r[0] + Jl[3 + a] * r[1];
                for (int b = 0; b < 3; ++b) V[a][b] += Jl[a] * Jl[b] + Jl[3 + a] * Jl[3 + b];
            }
        }
        const bool active = j < P.m && P.lm_count[j] >= 2;
        if (active) {
            for (int a = 0; a < 3; ++a) V[a][a] *= 1.0 + lambda;
            if (!inv3_sym(V, E)) {
                *fail_flag = 1;
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) E[a][b] = 0.0;
            }
        }
        for (int a = 0; a < 3; ++a) {
            sGl[tid * 3 + a] = active ? gl[a] : 0.0;
            for (int b = 0; b < 3; ++b) sE[tid * 9 + 3 * a + b] = E[a][b];
        }
        if (j < P.m) {
            for (int a = 0; a < 9; ++a) E_out[(size_t)j * 9 + a] = sE[tid * 9 + a];
            for (int a = 0; a < 3; ++a) gl_out[(size_t)j * 3 + a] = sGl[tid * 3 + a];
        }
    }
    __syncthreads();
    // 3. Y = W E  (6x3 per slot)
    for (int q = tid; q < LT * F * 6; q += 256) {
        const int s = q / 6, a = q - s * 6, jl = s / F;
        const double *w = sW + s * 18 + 3 * a;
        const double *E = sE + jl * 9;
        double y[3];
        for (int c = 0; c < 3; ++c) y[c] = w[0] * E[c] + w[1] * E[3 + c] + w[2] * E[6 + c];
        for (int c = 0; c < 3; ++c) sY[s * 18 + 3 * a + c] = y[c];
    }
    __syncthreads();
    // 4. entries of the partial reduced camera system, fixed summation order over landmarks
    for (int e = tid; e < np * np; e += 256) {
        const int ra = e / np, cb = e - ra * np;
        const int f1 = ra / 6, a = ra - 6 * f1, f2 = cb / 6, b = cb - 6 * f2;
        double u = 0.0, sch = 0.0;
        for (int jl = 0; jl < LT; ++jl) {
            const int s1 = jl * F + f1, s2 = jl * F + f2;
            if (f1 == f2) u += sJp[s1 * 12 + a] * sJp[s1 * 12 + b] + sJp[s1 * 12 + 6 + a] * sJp[s1 * 12 + 6 + b];
            const double *y = sY + s1 * 18 + 3 * a;
            const double *w = sW + s2 * 18 + 3 * b;
            sch += y[0] * w[0] + y[1] * w[1] + y[2] * w[2];
        }
        if (ra == cb) u *= 1.0 + lambda;
        S_part[(size_t)blockIdx.x * np * np + e] = u - sch;
    }
    for (int ra = tid; ra < np; ra += 256) {
        const int f1 = ra / 6, a = ra - 6 * f1;
        double g = 0.0;
        for (int jl = 0; jl < LT; ++jl) {
            const int s1 = jl * F + f1;
            g += sJp[s1 * 12 + a] * sR[s1 * 2] + sJp[s1 * 12 + 6 + a] * sR[s1 * 2 + 1];
            const double *y = sY + s1 * 18 + 3 * a;
            const double *gl = sGl + jl * 3;
            g -= y[0] * gl[0] + y[1] * gl[1] + y[2] * gl[2];
        }
        g_part[(size_t)blockIdx.x * np + ra] = g;
    }
}

// Sums the partial systems, Cholesky in LDS, pose step dp (np doubles).
// Solves S x = g (S symmetric positive definite, NP x NP row-major in LDS, g in LDS) with the calling
// wavefront: lane i keeps row i of the lower triangle in registers (every loop is unrolled, so the
// row is never indexed dynamically), pivots travel by register broadcast, a column's multipliers through LDS.
// Right-looking Cholesky, then the two substitutions column by column.  Returns false (in every
// lane) when a pivot is not positive; x = this lane's unknown.
// (At 42 and 48 rows the compiler declines to unroll the column loop -- the unrolled code would be too large --
// and the row lives in scratch memory instead: correct, slower, and its warning is switched off from here to the
// end of the kernel the routine is inlined into.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <int NP>
__device__ __forceinline__ bool cholesky_solve_rows(double *S, const double *g, double &x)
{
    const int i = threadIdx.x;
    const bool row = i < NP;
    double A[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) A[c] = (row && c <= i) ? S[i * NP + c] : 0.0;
    bool pos = true;
    double diag = 1.0;
    // Column j of the factor travels through LDS: every lane writes its entry and reads the entries of the
    // rows below the pivot back (all lanes read the same address: a broadcast), which costs half the
    // instructions of one register broadcast per entry and none of its waits.  `col`: NP doubles of LDS that
    // nothing else uses meanwhile (the wavefront runs alone by now; LDS operations of one wavefront complete
    // in order).
    double *col = S + NP * NP;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const double d = readlane_f64(A[j], j);
        if (!(d > 0.0)) pos = false;
        const double r = sqrt(d > 0.0 ? d : 1.0);
        const double Lij = i > j ? A[j] / r : (i == j ? r : 0.0);
        A[j] = Lij;
        if (i == j) diag = r;
        if (j + 1 < NP) {
            if (row) col[i] = Lij;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < (NP + 7) / 8; ++c) {   // eight reads in flight, then their eight updates
                double Lk[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) Lk[q] = (8 * c + q > j && 8 * c + q < NP) ? col[8 * c + q] : 0.0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (8 * c + q > j && 8 * c + q < NP) A[8 * c + q] -= i >= 8 * c + q ? Lij * Lk[q] : 0.0;
            }
            __builtin_amdgcn_wave_barrier();   // (the next column is written after every lane has read this one)
        }
    }
    // the factor goes back to LDS: the backward substitution reads it by columns
#pragma unroll
    for (int c = 0; c < NP; ++c)
        if (row && c <= i) S[i * NP + c] = A[c];
    __syncthreads();
    const double inv_diag = 1.0 / diag;
    double gi = row ? g[i] : 0.0;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const double xk = readlane_f64(gi * inv_diag, k);
        gi = (i == k) ? xk : gi - (i > k ? A[k] : 0.0) * xk;
    }
    for (int k = NP - 1; k >= 0; --k) {
        const double lki = (row && i < k) ? S[k * NP + i] : 0.0;
        const double xk = readlane_f64(gi * inv_diag, k);
        gi = (i == k) ? xk : gi - lki * xk;
    }
    x = gi;
    return __builtin_amdgcn_ballot_w64(!pos) == 0;
}

// sum of the partial systems' entries e, in workgroup order; sixteen loads are in flight at a time
__device__ __forceinline__ double ordered_partial_sum(const double *__restrict__ part, size_t stride, int nblocks, int e)
{
    double acc = 0.0;
    int b = 0;
    for (; b + 16 <= nblocks; b += 16) {
        double v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = part[(size_t)(b + q) * stride + e];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
    }
    for (; b < nblocks; ++b) acc += part[(size_t)b * stride + e];
    return acc;
}

// First stage of the ordered sum when there are many partial systems (one per 16 landmarks): workgroup r
// adds the systems of a contiguous range of `chunk` workgroups, in order, entry by entry, so that the
// single workgroup that solves reads RED_GROUPS systems instead of hundreds (it is bound by what one
// compute unit can load: 318 systems of 900 doubles took 64 us).
constexpr int RED_GROUPS = 32;
constexpr int RED_FROM = 64;   // fewer partial systems than this are summed by the solving workgroup itself

__global__ __launch_bounds__(1024) void ba_reduce_kernel(int np, int nblocks, int chunk, const double *__restrict__ S_part,
                                                         const double *__restrict__ g_part,
                                                         const ba_status *__restrict__ st, double *__restrict__ S_red,
                                                         double *__restrict__ g_red)
{
    if (st->done) return;
    const int first = blockIdx.x * chunk;
    const int count = min(chunk, nblocks - first);
    const size_t n2 = (size_t)np * np;
    for (int e = threadIdx.x; e < np * np; e += blockDim.x)
        S_red[blockIdx.x * n2 + e] = count > 0 ? ordered_partial_sum(S_part + first * n2, n2, count, e) : 0.0;
    for (int e = threadIdx.x; e < np; e += blockDim.x)
        g_red[(size_t)blockIdx.x * np + e] = count > 0 ? ordered_partial_sum(g_part + (size_t)first * np, (size_t)np, count, e) : 0.0;
}

// 256 threads share the entries of the reduced system (900 at 5 frames) for the ordered sum over the partial
// systems; one wavefront then factorises.  Not more threads: the register budget of a kernel shrinks with its
// workgroup size, and the factorisation (a row of up to 48 doubles per lane, fully unrolled) spills below
// 256 registers -- measured at 5 frames: 1024 threads 29.7 us, 512 29.0, 256 20.6, 128 23.9.
// One kernel per window size (NP = 6 F unknowns): a kernel's registers are those of its largest code path, so
// one kernel that switches over the sizes would carry the 48-row factorisation's spills at every size.
constexpr int BA_SOLVE_THREADS = 256;
template <int NP>
__global__ __launch_bounds__(BA_SOLVE_THREADS) void ba_solve_kernel(int nblocks, const double *__restrict__ S_part,
                                                                    const double *__restrict__ g_part,
                                                                    ba_status *__restrict__ st,
                                                                    const int *__restrict__ fail_flag,
                                                                    double *__restrict__ dp)
{
    __shared__ double S[NP * NP + NP];   // (+ one column for the factorisation)
    __shared__ double g[NP];
    __shared__ int ok;
    const int tid = threadIdx.x;
    if (st->done) return;
    for (int e = tid; e < NP * NP; e += BA_SOLVE_THREADS) S[e] = ordered_partial_sum(S_part, (size_t)NP * NP, nblocks, e);
    for (int e = tid; e < NP; e += BA_SOLVE_THREADS) g[e] = -ordered_partial_sum(g_part, (size_t)NP, nblocks, e);
    if (tid == 0) ok = *fail_flag ? 0 : 1;
    __syncthreads();
    // The factorisation and the two substitutions are one wavefront's work (NP <= 48 rows, lane i owns
    // row i, in registers); the other wavefronts leave.
    if (tid >= 64) return;
    double x = 0.0;
    const bool good = cholesky_solve_rows<NP>(S, g, x) && ok;
    if (good && tid < NP) dp[tid] = x;
    if (tid == 0) st->solve_failed = good ? 0 : 1;
}

#pragma clang diagnostic pop

void launch_ba_solve(hipStream_t s, int F, int nblocks, const double *S_part, const double *g_part, ba_status *st,
                     const int *fail_flag, double *dp)
{
#define BA_SOLVE_CASE(f)                                                                                              \
    case f:                                                                                                           \
        hipLaunchKernelGGL(ba_solve_kernel<6 * f>, dim3(1), dim3(BA_SOLVE_THREADS), 0, s, nblocks, S_part, g_part, st, \
                           fail_flag, dp);                                                                            \
        break;
    switch (F) {
        BA_SOLVE_CASE(1)
        BA_SOLVE_CASE(2)
        BA_SOLVE_CASE(3)
        BA_SOLVE_CASE(4)
        BA_SOLVE_CASE(5)
        BA_SOLVE_CASE(6)
        BA_SOLVE_CASE(7)
        BA_SOLVE_CASE(8)
    }
#undef BA_SOLVE_CASE
    static_assert(FMAX == 8, "one case per window size");
}

// The decision of a trial, by ONE thread (the last workgroup of ba_update_kernel to finish, or its first one
// when the solve failed).  An accepted step is not copied anywhere: the two state buffers change roles.
// `mirror`: the host's pinned copy of the status (the host follows the run by polling it instead of
// synchronising: the count of decided trials is stored last, with system scope).
__device__ void ba_decide(int nblocks, const double *partial, ba_status *st, int *fail_flag, int max_iter, double ftol,
                          double xtol, ba_status *mirror)
{
    int accept = 0;
    st->trials++;
    bool finished = false;
    if (st->solve_failed) {
        st->lambda *= 10.0;
        if (++st->rejects > 10) finished = true;
    } else {
        double c = 0.0, dn = 0.0, xn = 0.0;
        for (int b = 0; b < nblocks; ++b) {
            c += partial[3 * b];
            dn += partial[3 * b + 1];
            xn += partial[3 * b + 2];
        }
        st->cost_trial = c;
        st->step_norm2 = dn;
        st->x_norm2 = xn;
        if (c < st->cost) {
            const double dec = st->cost - c, cost_old = st->cost;
            accept = 1;
            st->cost = c;
            st->lambda = st->lambda * 0.1 > 1e-12 ? st->lambda * 0.1 : 1e-12;
            st->iter++;
            st->rejects = 0;
            if (dec < ftol * (1.0 + cost_old)) finished = true;
            if (sqrt(dn) < xtol * (1.0 + sqrt(xn))) finished = true;
            if (st->iter >= max_iter) finished = true;
        } else {
            st->lambda *= 10.0;
            if (++st->rejects > 10) finished = true;
        }
    }
    st->accepted_last = accept;
    if (accept) st->cur ^= 1;
    st->done = finished ? 1 : 0;
    *fail_flag = 0;
    if (mirror) {
        ba_status c = *st;
        const int decided = c.trials;
        c.trials = decided - 1;   // (what the host saw last)
        *mirror = c;
        __threadfence_system();
        __hip_atomic_store(&mirror->trials, decided, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Back-substitution, trial state and its cost, one thread per landmark; then the decision of the trial by the
// workgroup that finishes last (every workgroup publishes its three partial sums, then takes a ticket; the
// sums are added in workgroup order, so the result does not depend on which workgroup that is).
// The status is read at the top by every workgroup and rewritten only after the last ticket is taken.
__global__ __launch_bounds__(256) void ba_update_kernel(ba_problem P, double *__restrict__ X0, double *__restrict__ X1,
                                                        const double *__restrict__ dp,
                                                        const double *__restrict__ E_in,
                                                        const double *__restrict__ gl_in,
                                                        const double *__restrict__ W_in, ba_status *st,
                                                        double *partial /* 3 per block */, unsigned *tickets,
                                                        int *fail_flag, int max_iter, double ftol, double xtol,
                                                        ba_status *mirror)
{
    __shared__ frame_lin sF[FMAX];
    __shared__ double sdp[6 * FMAX];
    __shared__ double sRed[3][256];
    const int F = P.F, np = 6 * F, tid = threadIdx.x;
    const int done = st->done, solve_failed = st->solve_failed, cur_buf = st->cur;
    if (done) return;
    if (solve_failed) {   // no trial state: the rejection is decided at once
        if (blockIdx.x == 0 && tid == 0) ba_decide(0, partial, st, fail_flag, max_iter, ftol, xtol, mirror);
        return;
    }
    const double *__restrict__ state = cur_buf ? X1 : X0;   // the trial state goes to the other buffer
    double *__restrict__ trial = cur_buf ? X0 : X1;
    if (tid < np) sdp[tid] = dp[tid];
    __syncthreads();
    if (tid < F) {
        double tw[6];
        for (int k = 0; k < 6; ++k) tw[k] = state[6 * tid + k] + sdp[6 * tid + k];
        frame_prepare(tw, sF[tid]);
    }
    double c = 0.0, dn = 0.0, xn = 0.0;
    if (blockIdx.x == 0 && tid < np) {
        const double x = state[tid], d = sdp[tid];
        trial[tid] = x + d;
        dn += d * d;
        xn += x * x;
    }
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + tid;
    if (j < P.m) {
        const double *X = state + np + 3 * j;
        double Xn[3] = {X[0], X[1], X[2]};
        if (P.lm_count[j] >= 2) {
            double t[3] = {gl_in[3 * j], gl_in[3 * j + 1], gl_in[3 * j + 2]};
            for (int f = 0; f < F; ++f) {
                const int k = P.obs_index[j * F + f];
                if (k < 0) continue;
                const double *W = W_in + (size_t)k * 18;
                for (int cc = 0; cc < 3; ++cc)
                    for (int a = 0; a < 6; ++a) t[cc] += W[3 * a + cc] * sdp[6 * f + a];
            }
            const double *E = E_in + (size_t)j * 9;
            for (int a = 0; a < 3; ++a) {
                const double d = -(E[3 * a] * t[0] + E[3 * a + 1] * t[1] + E[3 * a + 2] * t[2]);
                Xn[a] += d;
                dn += d * d;
            }
        }
        for (int a = 0; a < 3; ++a) {
            xn += X[a] * X[a];
            trial[np + 3 * j + a] = Xn[a];
        }
        for (int f = 0; f < F; ++f) {
            const int k = P.obs_index[j * F + f];
            if (k < 0) continue;
            double r[2];
            observe<false>(sF[f], P.K, Xn, P.obs_uv[2 * k], P.obs_uv[2 * k + 1], r, nullptr, nullptr);
            c += r[0] * r[0] + r[1] * r[1];
        }
    }
    sRed[0][tid] = c;
    sRed[1][tid] = dn;
    sRed[2][tid] = xn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s)
            for (int q = 0; q < 3; ++q) sRed[q][tid] += sRed[q][tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        for (int q = 0; q < 3; ++q) partial[3 * blockIdx.x + q] = sRed[q][0];
        __threadfence();   // the sums are visible to the whole GPU before the ticket is
        if (atomicAdd(tickets, 1u) == gridDim.x - 1) {
            __threadfence();
            *tickets = 0u;
            ba_decide((int)gridDim.x, partial, st, fail_flag, max_iter, ftol, xtol, mirror);
        }
    }
}

// ------------------------------------------------------------------------------------------
// T_refinement: one workgroup, whole LM loop in-kernel.
// ------------------------------------------------------------------------------------------
constexpr int PR_THREADS = 512;

struct pr_params {
    double K[9];
    int n, max_iter;
    double ftol, xtol;
};

// offsets (may be NULL): several independent problems in one launch, one workgroup each -- problem b owns
// points [offsets[b], offsets[b + 1]) of kp_uv and X, pose b of T_io and statistics b.
__global__ __launch_bounds__(PR_THREADS) void pose_refine_kernel(pr_params P, const int *__restrict__ n_dev,
                                                                 const double *__restrict__ kp_uv,
                                                                 const double *__restrict__ X,
                                                                 double *__restrict__ T_io /*16*/,
                                                                 double *__restrict__ stats /*3*/,
                                                                 const int *__restrict__ offsets, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_STATE(kp_uv);
    VOHIP_SEQ_STATE(X);
    VOHIP_SEQ_STATE(T_io);
    VOHIP_SEQ_STATE_OPT(stats);
    if (offsets) {
        const int b = blockIdx.x, first = offsets[b];
        P.n = offsets[b + 1] - first;
        if (P.n == 0) return;   // (an empty problem leaves its pose as it is, like the single call)
        kp_uv += 2 * (size_t)first;
        X += 3 * (size_t)first;
        T_io += 16 * (size_t)b;
        if (stats) stats += 3 * (size_t)b;
    }
    // Two slots (frame, twist, normal equations, cost): the current estimate and the trial step.  One
    // pass over the points gives a slot its cost AND its normal equations, so an accepted step needs
    // no second pass (the next step is solved from the trial slot's own system).
    __shared__ frame_lin sF[2];
    __shared__ double sTw[2][6], sH[2][36], sG[2][6], sCost[2], sD[6], sPart[28 * (PR_THREADS / 8)];
    __shared__ int sOk;
    const int tid = threadIdx.x;
    if (n_dev) P.n = *n_dev;  // point count kept on the device (the frame pipeline)
    if (tid == 0) {
        const rigid T = rigid_from_cm44(T_io);
        double tw[6];
        rigid_to_twist(T, tw);
        for (int k = 0; k < 6; ++k) sTw[0][k] = tw[k];
        frame_prepare(tw, sF[0]);
    }
    __syncthreads();
    // cost, J^T J (21 sums) and J^T r (6 sums) of slot s: three DPP steps leave every group of 8
    // lanes with its sum, the 64 group sums of each quantity go through LDS, 8 threads per quantity
    // add them in a fixed order
    auto linearize = [&](int s) {
        double h[21], g[6], c = 0.0;
#pragma unroll
        for (int q = 0; q < 21; ++q) h[q] = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) g[q] = 0.0;
        for (int k = tid; k < P.n; k += PR_THREADS) {
            const double Xl[3] = {X[3 * k], X[3 * k + 1], X[3 * k + 2]};
            double r[2], Jp[2][6], Jl[2][3];
            observe<true>(sF[s], P.K, Xl, kp_uv[2 * k], kp_uv[2 * k + 1], r, Jp, Jl);
            c += r[0] * r[0] + r[1] * r[1];
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                g[a] += Jp[0][a] * r[0] + Jp[1][a] * r[1];
#pragma unroll
                for (int b = a; b < 6; ++b) h[q++] += Jp[0][a] * Jp[0][b] + Jp[1][a] * Jp[1][b];
            }
        }
#pragma unroll
        for (int q = 0; q < 21; ++q) h[q] = group8_sum_f64(h[q]);
#pragma unroll
        for (int q = 0; q < 6; ++q) g[q] = group8_sum_f64(g[q]);
        c = group8_sum_f64(c);
        __syncthreads();
        if ((tid & 7) == 0) {
            double *col = sPart + (tid >> 3);
#pragma unroll
            for (int q = 0; q < 21; ++q) col[64 * q] = h[q];
#pragma unroll
            for (int q = 0; q < 6; ++q) col[64 * (21 + q)] = g[q];
            col[64 * 27] = c;
        }
        __syncthreads();
        if (tid < 28 * 8) {
            const int q = tid >> 3;
            const double *row = sPart + 64 * q + 8 * (tid & 7);
            double t = ((row[0] + row[1]) + (row[2] + row[3])) + ((row[4] + row[5]) + (row[6] + row[7]));
            t = group8_sum_f64(t);
            if ((tid & 7) == 0) {
                if (q < 21) {
                    // upper-triangle index -> (a, b)
                    int a = 0, rem = q;
                    while (rem >= 6 - a) {
                        rem -= 6 - a;
                        ++a;
                    }
                    const int b = a + rem;
                    sH[s][6 * a + b] = sH[s][6 * b + a] = t;
                } else if (q < 27) {
                    sG[s][q - 21] = t;
                } else {
                    sCost[s] = t;
                }
            }
        }
        __syncthreads();
        return sCost[s];
    };
    int cur = 0;
    double cost = linearize(cur);
    const double cost0 = cost;
    double lambda = 1e-3;
    int iter = 0, rejects = 0;
    bool stop = false;
    while (iter < P.max_iter && !stop) {
        const int nxt = cur ^ 1;
        if (tid == 0) {
            // 6x6 Cholesky with every loop unrolled, so that A and d stay in registers; the reciprocals
            // of the pivots serve the two substitutions as well (no further divisions)
            double A[36], d[6], piv[6];
#pragma unroll
            for (int q = 0; q < 36; ++q) A[q] = sH[cur][q];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                A[7 * a] *= 1.0 + lambda;
                d[a] = -sG[cur][a];
            }
            int ok = 1;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                double dd = A[j * 6 + j];
#pragma unroll
                for (int k = 0; k < j; ++k) dd -= A[j * 6 + k] * A[j * 6 + k];
                if (!(dd > 0.0)) ok = 0;
                const double inv = rsqrt(dd);   // (the pivot itself is never needed, only its reciprocal)
                piv[j] = inv;
#pragma unroll
                for (int i = j + 1; i < 6; ++i) {
                    double t = A[i * 6 + j];
#pragma unroll
                    for (int k = 0; k < j; ++k) t -= A[i * 6 + k] * A[j * 6 + k];
                    A[i * 6 + j] = t * inv;
                }
            }
            if (ok) {
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    double t = d[i];
#pragma unroll
                    for (int k = 0; k < i; ++k) t -= A[i * 6 + k] * d[k];
                    d[i] = t * piv[i];
                }
#pragma unroll
                for (int i = 5; i >= 0; --i) {
                    double t = d[i];
#pragma unroll
                    for (int k = i + 1; k < 6; ++k) t -= A[k * 6 + i] * d[k];
                    d[i] = t * piv[i];
                }
                double twn[6];
                for (int a = 0; a < 6; ++a) {
                    sD[a] = d[a];
                    twn[a] = sTw[cur][a] + d[a];
                    sTw[nxt][a] = twn[a];
                }
                frame_prepare(twn, sF[nxt]);
            }
            sOk = ok;
        }
        __syncthreads();
        if (!sOk) {
            lambda *= 10.0;
            if (++rejects > 10) stop = true;
            __syncthreads();   // (sOk is rewritten by the next solve)
            continue;
        }
        const double cn = linearize(nxt);
        if (cn < cost) {
            double dn = 0.0, xn = 0.0;
            for (int a = 0; a < 6; ++a) {
                dn += sD[a] * sD[a];
                xn += sTw[cur][a] * sTw[cur][a];
            }
            const double dec = cost - cn, cost_old = cost;
            cur = nxt;
            cost = cn;
            lambda = lambda * 0.1 > 1e-12 ? lambda * 0.1 : 1e-12;
            ++iter;
            rejects = 0;
            if (dec < P.ftol * (1.0 + cost_old)) stop = true;
            if (sqrt(dn) < P.xtol * (1.0 + sqrt(xn))) stop = true;
        } else {
            lambda *= 10.0;
            if (++rejects > 10) stop = true;
        }
        __syncthreads();   // every thread has read sD / sTw before the next solve rewrites them
    }
    if (tid == 0) {
        double tw[6];
        for (int k = 0; k < 6; ++k) tw[k] = sTw[cur][k];
        const rigid T = twist_to_rigid(tw);
        rigid_to_cm44(T, T_io);
        if (stats) {
            stats[0] = (double)iter;
            stats[1] = cost0;
            stats[2] = cost;
        }
    }
}

// Host-side view of the reference's serialised observation vector
// ([F; m; per frame: n_i; n_i (row, col) pairs; n_i 1-based landmark ids], VO_bundle_adjust.m:47-50).
struct host_obs {
    int F = 0, m = 0, nobs = 0;
    std::vector<int> frame, lm, index, count;
    std::vector<double> uv;
};

int parse_observations(vohip_ctx *ctx, const double *obs, int len, int state_len, host_obs &o)
{
    VOHIP_REQUIRE(ctx, len >= 2, "observation vector too short");
    o.F = (int)obs[0];
    o.m = (int)obs[1];
    VOHIP_REQUIRE(ctx, o.F >= 1 && o.F <= FMAX, "number of frames must be 1..8");
    VOHIP_REQUIRE(ctx, o.m >= 0, "negative landmark count");
    VOHIP_REQUIRE(ctx, state_len < 0 || state_len == 6 * o.F + 3 * o.m, "state length is not 6F + 3m");
    int pos = 2;
    for (int f = 0; f < o.F; ++f) {
        VOHIP_REQUIRE(ctx, pos < len, "observation vector truncated");
        const int n = (int)obs[pos];
        VOHIP_REQUIRE(ctx, n >= 0 && pos + 1 + 3 * (long)n <= len, "observation vector truncated");
        for (int i = 0; i < n; ++i) {
            const int id = (int)obs[pos + 1 + 2 * n + i] - 1;
            VOHIP_REQUIRE(ctx, id >= 0 && id < o.m, "landmark index out of range");
            o.frame.push_back(f);
            o.lm.push_back(id);
            o.uv.push_back(obs[pos + 1 + 2 * i + 1]);  // u = col
            o.uv.push_back(obs[pos + 1 + 2 * i]);      // v = row
        }
        pos += 1 + 3 * n;
    }
    VOHIP_REQUIRE(ctx, pos == len, "observation vector has trailing data");
    o.nobs = (int)o.frame.size();
    o.index.assign((size_t)o.m * o.F, -1);
    o.count.assign((size_t)o.m, 0);
    for (int k = 0; k < o.nobs; ++k) {
        int &slot = o.index[(size_t)o.lm[k] * o.F + o.frame[k]];
        VOHIP_REQUIRE(ctx, slot < 0, "a landmark is observed twice in one frame");
        slot = k;
        o.count[o.lm[k]]++;
    }
    return VOHIP_OK;
}

int upload_problem(vohip_ctx *ctx, const host_obs &o, const double K[9], ba_problem &P)
{
    void *d;
    P.F = o.F;
    P.m = o.m;
    P.nobs = o.nobs;
    VOHIP_TRY(vohip_upload(ctx, B_BA_INDEX, o.index.data(), sizeof(int) * o.index.size(), &d));
    P.obs_index = (const int *)d;
    VOHIP_TRY(vohip_upload(ctx, B_BA_FRAME, o.frame.data(), sizeof(int) * o.frame.size(), &d));
    P.obs_frame = (const int *)d;
    VOHIP_TRY(vohip_upload(ctx, B_BA_LM, o.lm.data(), sizeof(int) * o.lm.size(), &d));
    P.obs_lm = (const int *)d;
    VOHIP_TRY(vohip_upload(ctx, B_BA_UV, o.uv.data(), sizeof(double) * o.uv.size(), &d));
    P.obs_uv = (const double *)d;
    VOHIP_TRY(vohip_upload(ctx, B_BA_COUNT, o.count.data(), sizeof(int) * o.count.size(), &d));
    P.lm_count = (const int *)d;
    std::memcpy(P.K, K, sizeof(P.K));
    return VOHIP_OK;
}

}  // namespace

extern "C" {

int vohip_ba_error(vohip_ctx *ctx, const double *state, int state_len, const double *obs, int obs_len,
                   const double K[9], double *r_out, int *n_obs_out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, state && obs && K && r_out, "NULL argument");
    host_obs o;
    VOHIP_TRY(parse_observations(ctx, obs, obs_len, state_len, o));
    if (n_obs_out) *n_obs_out = o.nobs;
    if (o.nobs == 0) return VOHIP_OK;
    ba_problem P;
    VOHIP_TRY(upload_problem(ctx, o, K, P));
    void *d_state, *d_r;
    VOHIP_TRY(vohip_upload(ctx, B_BA_STATE, state, sizeof(double) * state_len, &d_state));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, nullptr, sizeof(double) * 2 * o.nobs, &d_r));
    hipLaunchKernelGGL(ba_error_kernel, dim3((o.nobs + 255) / 256), dim3(256), 0, ctx->stream, P,
                       (const double *)d_state, (double *)d_r);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_TRY(vohip_download(ctx, r_out, d_r, sizeof(double) * 2 * o.nobs));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_ba_run(vohip_ctx *ctx, double *state, int state_len, const double *obs, int obs_len, const double K[9],
                 int max_iter, double ftol, double xtol, double *stats)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, state && obs && K, "NULL argument");
    VOHIP_REQUIRE(ctx, max_iter >= 0, "negative max_iter");
    static const bool trace = std::getenv("VOHIP_WINDOW_TRACE") != nullptr;
    const auto t_in = std::chrono::steady_clock::now();
    host_obs o;
    VOHIP_TRY(parse_observations(ctx, obs, obs_len, state_len, o));
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0.0;
    if (o.nobs == 0 || max_iter == 0) return VOHIP_OK;
    ba_problem P;
    VOHIP_TRY(upload_problem(ctx, o, K, P));
    const int F = o.F, m = o.m, np = 6 * F, nx = state_len;
    const int nb_lin = (m + LT - 1) / LT > 0 ? (m + LT - 1) / LT : 1;
    const int nb_upd = (m + 255) / 256 > 0 ? (m + 255) / 256 : 1;
    const int nb_cost = (o.nobs + 255) / 256;
    void *d_state, *d_trial, *d_Sp, *d_gp, *d_E, *d_gl, *d_W, *d_dp, *d_part, *d_st, *d_fail, *h_st;
    VOHIP_TRY(vohip_upload(ctx, B_BA_STATE, state, sizeof(double) * nx, &d_state));
    VOHIP_TRY(vohip_upload(ctx, B_BA_TRIAL, nullptr, sizeof(double) * nx, &d_trial));
    VOHIP_TRY(vohip_upload(ctx, B_BA_SPART, nullptr, sizeof(double) * (size_t)nb_lin * np * np, &d_Sp));
    VOHIP_TRY(vohip_upload(ctx, B_BA_GPART, nullptr, sizeof(double) * (size_t)nb_lin * np, &d_gp));
    VOHIP_TRY(vohip_upload(ctx, B_BA_E, nullptr, sizeof(double) * 9 * (size_t)(m > 0 ? m : 1), &d_E));
    VOHIP_TRY(vohip_upload(ctx, B_BA_GL, nullptr, sizeof(double) * 3 * (size_t)(m > 0 ? m : 1), &d_gl));
    VOHIP_TRY(vohip_upload(ctx, B_BA_W, nullptr, sizeof(double) * 18 * (size_t)o.nobs, &d_W));
    VOHIP_TRY(vohip_upload(ctx, B_BA_DP, nullptr, sizeof(double) * 6 * FMAX, &d_dp));
    const bool two_stage = nb_lin >= RED_FROM;
    const int red_chunk = (nb_lin + RED_GROUPS - 1) / RED_GROUPS;
    void *d_Sr = nullptr, *d_gr = nullptr;
    if (two_stage) {
        VOHIP_TRY(vohip_upload(ctx, B_BA_SRED, nullptr, sizeof(double) * (size_t)RED_GROUPS * np * np, &d_Sr));
        VOHIP_TRY(vohip_upload(ctx, B_BA_GRED, nullptr, sizeof(double) * (size_t)RED_GROUPS * np, &d_gr));
    }
    const int nb_part = nb_upd > nb_cost ? nb_upd : nb_cost;
    VOHIP_TRY(vohip_upload(ctx, B_BA_PART, nullptr, sizeof(double) * 3 * (size_t)nb_part, &d_part));
    VOHIP_TRY(vohip_upload(ctx, B_BA_STATUS, nullptr, 512, &d_st));
    VOHIP_TRY(vohip_reserve_pinned(ctx, ctx->bufs[H_BA_STATUS], sizeof(ba_status)));
    h_st = ctx->bufs[H_BA_STATUS].ptr;
    d_fail = (char *)d_st + 256;
    hipStream_t s = ctx->stream;
    VOHIP_HIP(ctx, hipMemsetAsync(d_st, 0, 512, s));
    hipLaunchKernelGGL(ba_cost_kernel, dim3(nb_cost), dim3(256), 0, s, P, (const double *)d_state, (double *)d_part);
    hipLaunchKernelGGL(ba_init_status_kernel, dim3(1), dim3(64), 0, s, (const double *)d_part, nb_cost,
                       (ba_status *)d_st);
    const size_t lds_lin = sizeof(frame_lin) * FMAX + sizeof(double) * ((size_t)LT * F * (12 + 2 + 18 + 18) + LT * 12);
    ba_status hs;
    std::memset(&hs, 0, sizeof(hs));
    const int max_trials = max_iter + 12 * (max_iter + 1);
    const auto t_loop = std::chrono::steady_clock::now();
    int trials_enqueued = 0;
    // The host follows the run in its pinned copy of the status, which every decision updates: it stays
    // at most two trials ahead of the GPU's decisions (the kernels of a trial return at once when the run
    // has ended, so at most two idle trials are enqueued) and never synchronises inside the loop.
    std::memset(h_st, 0, sizeof(ba_status));
    volatile ba_status *hm = (volatile ba_status *)h_st;
    bool timed_out = false;
    for (int trial = 0; trial < max_trials; ++trial) {
        if (trial >= 2) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(10);
            while (__atomic_load_n(&((ba_status *)h_st)->trials, __ATOMIC_ACQUIRE) < trial - 1 && !hm->done) {
                if (std::chrono::steady_clock::now() > deadline) {
                    timed_out = true;
                    break;
                }
            }
            if (timed_out || hm->done) break;
        }
        ++trials_enqueued;
        hipLaunchKernelGGL(ba_linearize_kernel, dim3(nb_lin), dim3(256), lds_lin, s, P, (const double *)d_state,
                           (const double *)d_trial, (const ba_status *)d_st, (double *)d_Sp, (double *)d_gp, (double *)d_E,
                           (double *)d_gl, (double *)d_W, (int *)d_fail);
        if (two_stage) {
            hipLaunchKernelGGL(ba_reduce_kernel, dim3(RED_GROUPS), dim3(1024), 0, s, np, nb_lin, red_chunk,
                               (const double *)d_Sp, (const double *)d_gp, (const ba_status *)d_st, (double *)d_Sr,
                               (double *)d_gr);
            launch_ba_solve(s, np / 6, RED_GROUPS, (const double *)d_Sr, (const double *)d_gr, (ba_status *)d_st,
                            (const int *)d_fail, (double *)d_dp);
        } else {
            launch_ba_solve(s, np / 6, nb_lin, (const double *)d_Sp, (const double *)d_gp, (ba_status *)d_st,
                            (const int *)d_fail, (double *)d_dp);
        }
        hipLaunchKernelGGL(ba_update_kernel, dim3(nb_upd), dim3(256), 0, s, P, (double *)d_state, (double *)d_trial,
                           (const double *)d_dp, (const double *)d_E, (const double *)d_gl, (const double *)d_W,
                           (ba_status *)d_st, (double *)d_part, (unsigned *)((char *)d_st + 384), (int *)d_fail, max_iter,
                           ftol, xtol, (ba_status *)h_st);
        VOHIP_HIP(ctx, hipGetLastError());
    }
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    if (timed_out) return vohip_fail(ctx, VOHIP_ERR_HIP, "the bundle adjustment did not report a decision within 10 s");
    std::memcpy(&hs, h_st, sizeof(hs));
    const auto t_done = std::chrono::steady_clock::now();
    VOHIP_TRY(vohip_download(ctx, state, hs.cur ? d_trial : d_state, sizeof(double) * nx));
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    if (trace) {
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::micro>(b - a).count();
        };
        std::fprintf(stderr, "vohip runBA trace (us): parse + upload %.0f  LM loop %.0f (%d trials enqueued, %d run, %d accepted)  download %.0f\n",
                     us(t_in, t_loop), us(t_loop, t_done), trials_enqueued, hs.trials, hs.iter,
                     us(t_done, std::chrono::steady_clock::now()));
    }
    if (stats) {
        stats[0] = hs.iter;
        stats[1] = hs.cost0;
        stats[2] = hs.cost;
        stats[3] = hs.lambda;
    }
    return VOHIP_OK;
}

int vohip_pose_refine_dev(vohip_ctx *ctx, double *d_T, const double *d_kp_uv, const double *d_X, int n,
                          const double K[9], int max_iter, double ftol, double xtol, double *d_stats)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, d_T && K && n >= 0, "bad argument");
    if (n == 0) return VOHIP_OK;
    VOHIP_REQUIRE(ctx, d_kp_uv && d_X, "NULL array");
    return vohip_pose_refine_enqueue(ctx, d_T, d_kp_uv, d_X, n, nullptr, K, max_iter, ftol, xtol, d_stats);
}

}  // extern "C"

// n, or *d_n when given (the frame pipeline keeps the count on the device).
int vohip_pose_refine_enqueue(vohip_ctx *ctx, double *d_T, const double *d_kp_uv, const double *d_X, int n,
                              const int *d_n, const double K[9], int max_iter, double ftol, double xtol,
                              double *d_stats, const vohip_seqs *seqs)
{
    const vohip_seqs SQ = seqs ? *seqs : vohip_seqs();
    const int ny = vohip_seq_count(SQ);
    if (ny < 1) return VOHIP_OK;
    pr_params P;
    std::memcpy(P.K, K, sizeof(P.K));
    P.n = n;
    P.max_iter = max_iter;
    P.ftol = ftol;
    P.xtol = xtol;
    hipLaunchKernelGGL(pose_refine_kernel, dim3(1, ny), dim3(PR_THREADS), 0, ctx->stream, P, d_n, d_kp_uv, d_X, d_T,
                       d_stats, (const int *)nullptr, SQ);
    VOHIP_HIP(ctx, hipGetLastError());
    return VOHIP_OK;
}

extern "C" {

int vohip_pose_refine(vohip_ctx *ctx, double T[16], const double *kp_uv, const double *X, int n, const double K[9],
                      int max_iter, double ftol, double xtol, double *stats)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, T && K && n >= 0, "bad argument");
    if (stats) stats[0] = stats[1] = stats[2] = 0.0;
    if (n == 0) return VOHIP_OK;
    VOHIP_REQUIRE(ctx, kp_uv && X, "NULL array");
    void *d_T, *d_kp, *d_X, *d_stats;
    VOHIP_TRY(vohip_upload(ctx, B_STAGE0, T, sizeof(double) * 16, &d_T));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE1, kp_uv, sizeof(double) * 2 * n, &d_kp));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE2, X, sizeof(double) * 3 * n, &d_X));
    VOHIP_TRY(vohip_upload(ctx, B_STAGE3, nullptr, sizeof(double) * 4, &d_stats));
    VOHIP_TRY(vohip_pose_refine_dev(ctx, (double *)d_T, (const double *)d_kp, (const double *)d_X, n, K, max_iter, ftol,
                                    xtol, (double *)d_stats));
    VOHIP_TRY(vohip_download(ctx, T, d_T, sizeof(double) * 16));
    if (stats) VOHIP_TRY(vohip_download(ctx, stats, d_stats, sizeof(double) * 3));
    VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VOHIP_OK;
}

int vohip_pose_refine_batch(vohip_ctx *ctx, int k, double *T, const double *kp_uv, const double *X, const int *n,
                            const double K[9], int max_iter, double ftol, double xtol, double *stats)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, k >= 0 && K, "bad argument");
    if (k == 0) return VOHIP_OK;
    VOHIP_REQUIRE(ctx, k <= 4096 && T && n, "bad argument");
    int *off = (int *)std::malloc(sizeof(int) * ((size_t)k + 1));
    if (!off) return vohip_fail(ctx, VOHIP_ERR_HIP, "out of host memory");
    off[0] = 0;
    bool ok = true;
    for (int b = 0; b < k; ++b) {
        ok = ok && n[b] >= 0 && off[b] <= 0x3fffffff - (n[b] > 0 ? n[b] : 0);
        off[b + 1] = off[b] + (n[b] > 0 ? n[b] : 0);
    }
    const int total = off[k];
    int st = VOHIP_OK;
    if (!ok || (total > 0 && (!kp_uv || !X))) st = vohip_fail(ctx, VOHIP_ERR_INVALID_ARGUMENT, "bad point counts or NULL array");
    if (stats) std::memset(stats, 0, sizeof(double) * 3 * (size_t)k);
    void *d_T = nullptr, *d_kp = nullptr, *d_X = nullptr, *d_stats = nullptr, *d_off = nullptr;
    if (st == VOHIP_OK && total > 0) {
        // (the staging buffers are filled by stream-ordered copies from the caller's arrays; `off` is read by its
        // copy before the synchronisation below, after which it is released)
        st = vohip_upload(ctx, B_STAGE0, T, sizeof(double) * 16 * (size_t)k, &d_T);
        if (st == VOHIP_OK) st = vohip_upload(ctx, B_STAGE1, kp_uv, sizeof(double) * 2 * (size_t)total, &d_kp);
        if (st == VOHIP_OK) st = vohip_upload(ctx, B_STAGE2, X, sizeof(double) * 3 * (size_t)total, &d_X);
        if (st == VOHIP_OK) st = vohip_upload(ctx, B_STAGE3, nullptr, sizeof(double) * 3 * (size_t)k + 8, &d_stats);
        if (st == VOHIP_OK) st = vohip_upload(ctx, B_STAGE4, off, sizeof(int) * ((size_t)k + 1), &d_off);
        if (st == VOHIP_OK) {
            pr_params P;
            std::memcpy(P.K, K, sizeof(P.K));
            P.n = 0;
            P.max_iter = max_iter;
            P.ftol = ftol;
            P.xtol = xtol;
            if (hipMemsetAsync(d_stats, 0, sizeof(double) * 3 * (size_t)k, ctx->stream) != hipSuccess)
                st = vohip_fail(ctx, VOHIP_ERR_HIP, "hipMemsetAsync failed");
            hipLaunchKernelGGL(pose_refine_kernel, dim3(k), dim3(PR_THREADS), 0, ctx->stream, P, (const int *)nullptr,
                               (const double *)d_kp, (const double *)d_X, (double *)d_T, (double *)d_stats,
                               (const int *)d_off, vohip_seqs());
            if (st == VOHIP_OK && hipGetLastError() != hipSuccess) st = vohip_fail(ctx, VOHIP_ERR_HIP, "launch failed");
        }
        if (st == VOHIP_OK) st = vohip_download(ctx, T, d_T, sizeof(double) * 16 * (size_t)k);
        if (st == VOHIP_OK && stats) st = vohip_download(ctx, stats, d_stats, sizeof(double) * 3 * (size_t)k);
        const hipError_t e = hipStreamSynchronize(ctx->stream);   // (also when a step failed: `off` may still be read)
        if (st == VOHIP_OK && e != hipSuccess) st = vohip_fail(ctx, VOHIP_ERR_HIP, "%s", hipGetErrorString(e));
    }
    std::free(off);
    return st;
}

}  // extern "C"
