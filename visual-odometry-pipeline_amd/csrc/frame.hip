// The continuous-operation loop body with the state resident on the GPU (gfx950).
//
// Restates main_demo.m:377-427 (VO_bundle_adjust at :409-411 is a call of its own between frames,
// vohip_frame_bundle_adjust) -- detection (genKeypoints.m), the landmark
// tracker, ransacLocalization.m, T_refinement.m, update_landmarks.m, update_candidate.m and the
// tracker re-initialisation -- as ONE call and ONE synchronisation per frame (vohip_frame_step,
// vohip_frame_step_ahead):
//   context stream  [image -> pyramid -> landmark AND candidate tracking (one launch, from the rows of
//                   the previous frame's track table into this frame's), unless the previous call
//                   enqueued them ahead] -> ordered selection (+ sample triples from the caller's
//                   uniform draws) and P3P solve (one launch) -> scoring -> the adaptive stopping rule replayed by
//                   one wavefront -> inlier selection -> pose refinement -> [join] -> triangulation of
//                   the filtered candidates -> landmark append -> redundancy filter -> candidate
//                   append (+ the frame's summary into the caller's pinned mirror; the call waits on
//                   an event recorded here) -> [step_ahead] the FOLLOWING frame's trackers, with their
//                   counts read from device memory
//   second stream   (after the trackers) ordered selection of the tracked
//                   candidates -> fundamental-matrix RANSAC -> inlier selection
//                   -> corner detection (of the following image with step_ahead; the detections become
//                   rows of the track table)
//   third stream    (step_ahead) the following image, its pyramid and the kept landmarks' tracker
// S.P, S.X, S.C, S.F, S.T never leave the device; every count between the stages stays in device
// memory (kernels are launched over capacity-sized grids and read the count themselves), and
// selections are ordered, so the state is element for element the one the stage-by-stage entry
// points produce.  When the adaptive bound of the localisation asks for more hypotheses than were
// supplied, vohip_frame_retry re-runs the context-stream part from the hypotheses onwards.
#include "vohip_internal.h"
#include "inlier_test.h"
#include "ordered_select.h"
#include "seq_index.h"
#include "sampson_test.h"

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <new>
#include <unordered_map>
#include <vector>

namespace {

constexpr int STAGE_FAILED = 3;
// From this many tracks on the candidate tracker is a launch of its own on the second stream (measured:
// 5 % faster at 6k tracks, 4 % slower at 2k, where one launch lasts as long as its slowest track anyway).
constexpr int SPLIT_TRACKERS_FROM = 4096;

// slots of the device counter array
enum {
    N_V = 0,      // landmarks tracked successfully
    N_IN,         // ... that are inliers of the best pose (= landmarks kept)
    N_CV,         // candidates tracked successfully
    N_CI,         // ... that pass the epipolar filter
    N_L_END,      // landmarks after new ones were appended
    N_C_MID,      // candidates left after triangulation
    NUM_NEW,      // candidates removed by update_landmarks (appended or dropped)
    N_DET,        // corners detected (may exceed the capacity), frames of even parity ...
    N_DET_ODD,    // ... and of odd parity (the next frame's detection can run ahead)
    FILT0, FILT1, FILT2,  // {detections, candidates, landmarks} as the redundancy filter reads them
    N_KEPT,       // detections kept
    N_C_END,      // candidates at the end of the frame
    OVERFLOW,     // bit 0: landmark capacity, bit 1: candidate capacity
    R_ITER, R_FOUND, R_EXHAUSTED, R_BEST_H, R_BEST_S, R_MAX_INL,   // localisation
    F_BEST_T, F_BEST_COUNT, F_TRIALS,                              // epipolar filter
    N_COUNTERS = 32
};
// slots of the device double array
enum {
    D_T = 0,          // T_W_C, 4x4 column-major
    D_M = 16,         // K [R | t] of the winning hypothesis, row-major 3x4
    D_RT = 28,        // [R_C_W (column-major) | t_C_W] of the winning hypothesis
    D_F = 40,         // winning fundamental matrix, row-major
    D_IDX = 49,       // next landmark id
    D_IDX_NEXT = 50,
    D_NUM_ITER = 51,  // adaptive bound after the last replayed iteration
    D_STATS = 52,     // pose refinement: iterations, initial cost, final cost
    D_NRM = 56,       // Hartley normalisation (6)
    N_DOUBLES = 64
};

using vohip_dev::wave_sum_i;              // ordered_select.h
using vohip_dev::triple_from_uniforms;

template <class Pred>
__device__ int ordered_rank_if(int n, Pred pred, int &total, int *sh)
{
    return vohip_dev::ordered_rank_if(n, pred, total, sh, blockIdx.x);
}

template <class Pred>
__device__ int ordered_rank(const uint8_t *__restrict__ flags, int n, Pred pred, int &total, int *sh)
{
    return ordered_rank_if(n, [=](int j) { return pred(flags[j]); }, total, sh);
}

struct is_set {
    __device__ bool operator()(uint8_t f) const { return f != 0; }
};
struct is_one {
    __device__ bool operator()(uint8_t f) const { return f == 1; }
};
struct is_zero {
    __device__ bool operator()(uint8_t f) const { return f == 0; }
};

// main_demo.m:380-381 + the S.X(:, validity) of :387: tracked landmarks that are still valid.
__global__ __launch_bounds__(256) void select_tracked_kernel(const double *__restrict__ out_xy, int ld,
                                                             const uint8_t *__restrict__ valid,
                                                             const double *__restrict__ L_X,
                                                             const double *__restrict__ L_id, int n,
                                                             double *__restrict__ mv_rc, double *__restrict__ Xv,
                                                             double *__restrict__ idv, int32_t *__restrict__ rowv,
                                                             int *__restrict__ cnt, double *__restrict__ dv,
                                                             const double *__restrict__ u3, int h0,
                                                             int32_t *__restrict__ samples,
                                                             const int *__restrict__ n_dev, vohip_seqs SQ)
{
    __shared__ int sh[12];
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(out_xy);
    VOHIP_SEQ_STATE(valid);
    VOHIP_SEQ_STATE(L_X);
    VOHIP_SEQ_STATE(L_id);
    VOHIP_SEQ_STATE(mv_rc);
    VOHIP_SEQ_STATE(Xv);
    VOHIP_SEQ_STATE(idv);
    VOHIP_SEQ_STATE(rowv);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(dv);
    VOHIP_SEQ_STATE(samples);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_PIN(u3);
    if (n_dev) n = min(n, *n_dev);   // (a group of loops: the count of each is known on the device only)
    int total;
    // the ids handed out in the previous frame are final now
    if (blockIdx.x == 0 && threadIdx.x == 0) dv[D_IDX] = dv[D_IDX_NEXT];
    const int r = ordered_rank(valid, n, is_set(), total, sh);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && valid[i]) {
        mv_rc[2 * r] = out_xy[ld + i];   // row = y
        mv_rc[2 * r + 1] = out_xy[i];    // col = x
        Xv[3 * r] = L_X[3 * i];
        Xv[3 * r + 1] = L_X[3 * i + 1];
        Xv[3 * r + 2] = L_X[3 * i + 2];
        idv[r] = L_id[i];
        rowv[r] = i;   // landmark i is row i of the track table
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) cnt[N_V] = total;
        // the first h0 sample triples (every thread of this workgroup knows the count already)
        for (int k = threadIdx.x; k < h0; k += 256) triple_from_uniforms(u3 + 3 * (size_t)k, total, samples + 3 * (size_t)k);
    }
}

// uniform draws -> index triples (3 distinct indices in [0, n)), for hypotheses [first, first + h)
__global__ __launch_bounds__(256) void triples_kernel(const double *__restrict__ u, int first, int h,
                                                      const int *__restrict__ cnt, int32_t *__restrict__ samples,
                                                      vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(samples);
    VOHIP_SEQ_PIN(u);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= h) return;
    triple_from_uniforms(u + 3 * (size_t)(first + k), cnt[N_V], samples + 3 * (size_t)(first + k));
}

// ransacLocalization.m:118-129
__device__ double adaptive_iterations_dev(int max_num_inliers, int n)
{
    const double confidence = 0.9999, upper = 0.95;
    double outlier_ratio = 1.0 - (double)max_num_inliers / (double)n;
    if (upper < outlier_ratio) outlier_ratio = upper;
    double it = log(1.0 - confidence) / log(1.0 - pow(1.0 - outlier_ratio, 3.0));
    if (it > 15000.0) it = 15000.0;
    if (it < 100.0) it = 100.0;
    return it;
}

__device__ __forceinline__ int wave_prefix_max(int v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v = max(v, t);
    }
    return v;
}

// The sequential part of ransacLocalization.m:42-131 over the scored hypotheses: solution choice
// (:77-101, strict '>'), running maximum (:104-110), adaptive bound (:118-129).  The bound is a
// function of the running maximum only, so one wavefront replays 64 iterations at a time: prefix
// maximum, bound per lane, first lane at which the loop condition fails.
// Executed by ONE wavefront (all 64 lanes); `publish`: this wavefront also writes the outcome to the
// device arrays (every workgroup of the selection kernel replays the rule, one of them publishes).
// Returns the index of the winning pose (4 * hypothesis + solution), or -1, in every lane.
__device__ int localize_decide(const int32_t *__restrict__ counts, int evaluated, const double *__restrict__ Rt_all,
                               const double *__restrict__ M_all, int *__restrict__ cnt, double *__restrict__ dv,
                               bool publish)
{
    const int lane = threadIdx.x & 63;
    const int n = cnt[N_V];
    int run_max = 0, best_h = -1, best_s = 0, n_iter = 0;
    double num_iterations = INFINITY;
    bool exhausted = false, stopped = false;
    if (n >= 3) {
        for (int base = 0; !stopped; base += 64) {
            if (base >= evaluated) {
                exhausted = true;   // the loop wants iteration base + 1, which was not scored
                n_iter = evaluated;
                break;
            }
            const int j = base + lane;          // iteration j + 1 of the reference loop
            const bool live = j < evaluated;
            int k = 0, s_best = 0;
            if (live) {
                const int4 c = *reinterpret_cast<const int4 *>(counts + 4 * (size_t)j);
                k = c.x;
                if (c.y > k) { k = c.y; s_best = 1; }
                if (c.z > k) { k = c.z; s_best = 2; }
                if (c.w > k) { k = c.w; s_best = 3; }
            }
            const int e = k >= 30 ? k : 0;      // :104 min_inlier_count
            const int m = max(wave_prefix_max(e, lane), run_max);
            const double bound = adaptive_iterations_dev(m, n);
            // after iteration j + 1 the loop goes on iff bound > j + 2
            const unsigned long long stop = __ballot(live && !(bound > (double)(j + 2)));
            int last = 63;                       // last lane of this chunk that is executed
            if (stop) {
                last = __ffsll((long long)stop) - 1;
                stopped = true;
            } else if (base + 64 >= evaluated) {
                last = evaluated - 1 - base;
            }
            const int m_last = __shfl(m, last, 64);
            if (m_last > run_max) {
                const unsigned long long hit = __ballot(lane <= last && e == m_last);
                const int w = __ffsll((long long)hit) - 1;
                best_h = base + w;
                best_s = __shfl(s_best, w, 64);
                run_max = m_last;
            }
            num_iterations = __shfl(bound, last, 64);
            n_iter = base + last + 1;
            if (!stopped && base + 64 >= evaluated) {
                exhausted = true;
                break;
            }
        }
    }
    const int winner = best_h >= 0 ? 4 * best_h + best_s : -1;
    if (lane != 0 || !publish) return winner;
    cnt[R_ITER] = n_iter;
    cnt[R_FOUND] = best_h >= 0 ? 1 : 0;
    cnt[R_EXHAUSTED] = exhausted ? 1 : 0;
    cnt[R_BEST_H] = best_h;
    cnt[R_BEST_S] = best_s;
    cnt[R_MAX_INL] = run_max;
    dv[D_NUM_ITER] = num_iterations;
    if (best_h >= 0) {
        const double *Rt = Rt_all + (size_t)best_h * 48 + 12 * best_s;
        const double *M = M_all + (size_t)best_h * 48 + 12 * best_s;
        for (int q = 0; q < 12; ++q) {
            dv[D_M + q] = M[q];
            dv[D_RT + q] = Rt[q];
        }
        // T_W_C = inv([R_C_W t_C_W; 0 0 0 1]) (main_demo.m:391), column-major
        double *T = dv + D_T;
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) T[4 * c + r] = Rt[3 * r + c];   // R_W_C = R_C_W'
            T[12 + r] = -(Rt[3 * r] * Rt[9] + Rt[3 * r + 1] * Rt[10] + Rt[3 * r + 2] * Rt[11]);
            T[4 * r + 3] = 0.0;
        }
        T[15] = 1.0;
    } else {
        const double nan = __builtin_nan("");
        for (int q = 0; q < 12; ++q) dv[D_M + q] = nan;   // no point is an inlier of a NaN pose
    }
    return winner;
}

// main_demo.m:394-396: S.P, S.X restricted to the inliers of the winning pose (the same test,
// operation for operation, as the scoring kernel counts with); also the (u, v) copy the refinement reads.
__global__ __launch_bounds__(256) void select_inliers_kernel(const int32_t *__restrict__ counts4, int evaluated,
                                                             const double *__restrict__ Rt_all,
                                                             const double *__restrict__ M_all,
                                                             const double *__restrict__ mv_rc,
                                                             const double *__restrict__ Xv,
                                                             const double *__restrict__ idv,
                                                             const int32_t *__restrict__ rowv,
                                                             double *__restrict__ dv,
                                                             int *__restrict__ cnt, double *__restrict__ L_rc,
                                                             double *__restrict__ L_X, double *__restrict__ L_id,
                                                             int32_t *__restrict__ srcL, double *__restrict__ kp_uv,
                                                             vohip_seqs SQ)
{
    __shared__ int sh[12];
    __shared__ int s_winner;
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(counts4);
    VOHIP_SEQ_STATE(Rt_all);
    VOHIP_SEQ_STATE(M_all);
    VOHIP_SEQ_STATE(mv_rc);
    VOHIP_SEQ_STATE(Xv);
    VOHIP_SEQ_STATE(idv);
    VOHIP_SEQ_STATE(rowv);
    VOHIP_SEQ_STATE(dv);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(L_rc);
    VOHIP_SEQ_STATE(L_X);
    VOHIP_SEQ_STATE(L_id);
    VOHIP_SEQ_STATE(srcL);
    VOHIP_SEQ_STATE(kp_uv);

    const int n = cnt[N_V];
    // the stopping rule of the localisation, replayed by the first wavefront of every workgroup (the
    // scored hypotheses are few: cheaper than a kernel of its own); workgroup 0 publishes the outcome
    if (threadIdx.x < 64) {
        const int w = localize_decide(counts4, evaluated, Rt_all, M_all, cnt, dv, blockIdx.x == 0);
        if (threadIdx.x == 0) s_winner = w;
    }
    __syncthreads();
    double m[12];   // K [R | t] of the winning hypothesis (NaN when there is none: nothing is an inlier)
    for (int q = 0; q < 12; ++q) m[q] = s_winner >= 0 ? M_all[(size_t)s_winner * 12 + q] : __builtin_nan("");
    auto inlier = [&](int j) {
        return vohip_dev::is_inlier(m, Xv[3 * j], Xv[3 * j + 1], Xv[3 * j + 2], mv_rc[2 * j + 1], mv_rc[2 * j]);
    };
    int total;
    const int r = ordered_rank_if(n, inlier, total, sh);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && inlier(i)) {
        const double row = mv_rc[2 * i], col = mv_rc[2 * i + 1];
        L_rc[2 * r] = row;
        L_rc[2 * r + 1] = col;
        kp_uv[2 * r] = col;
        kp_uv[2 * r + 1] = row;
        L_X[3 * r] = Xv[3 * i];
        L_X[3 * r + 1] = Xv[3 * i + 1];
        L_X[3 * r + 2] = Xv[3 * i + 2];
        L_id[r] = idv[i];
        srcL[r] = rowv[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cnt[N_IN] = total;
        // what the mapping stage reads when it has nothing to do (no candidates yet)
        cnt[N_L_END] = total;
        cnt[N_C_MID] = 0;
        cnt[NUM_NEW] = 0;
    }
}

// uniform draws -> a set of 8 distinct indices in [0, n): draw k picks the floor(u_k (n - k))-th
// index not picked before.
__device__ __forceinline__ void octet_from_uniforms(const double *u, int n, int32_t *s)
{
    int chosen[8];   // sorted; every index below is a compile-time constant, so the array stays in registers
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        int idx = min((int)(u[k] * (double)(n - k)), n - k - 1);
#pragma unroll
        for (int m = 0; m < k; ++m) idx += idx >= chosen[m];
        s[k] = idx;
        chosen[k] = idx;
#pragma unroll
        for (int m = k; m > 0; --m) {   // one insertion pass from the end
            const int lo = min(chosen[m - 1], chosen[m]), hi = max(chosen[m - 1], chosen[m]);
            chosen[m - 1] = lo;
            chosen[m] = hi;
        }
    }
}

// update_landmarks.m:10-11,20-25: candidates whose track is valid, with their first observation.
__global__ __launch_bounds__(256) void select_cand_tracked_kernel(const double *__restrict__ out_xy, int ld,
                                                                  const uint8_t *__restrict__ valid,
                                                                  const double *__restrict__ C_rc,
                                                                  const double *__restrict__ F_rc,
                                                                  const double *__restrict__ T12, int n,
                                                                  double *__restrict__ p1_rc, double *__restrict__ m_rc,
                                                                  double *__restrict__ F_t, double *__restrict__ T_t,
                                                                  int row_base, int32_t *__restrict__ row_t,
                                                                  int *__restrict__ cnt, const double *__restrict__ u8,
                                                                  int trials, int32_t *__restrict__ samples8,
                                                                  const int *__restrict__ n_dev, vohip_seqs SQ)
{
    __shared__ int sh[12];
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(out_xy);
    VOHIP_SEQ_STATE(valid);
    VOHIP_SEQ_STATE(C_rc);
    VOHIP_SEQ_STATE(F_rc);
    VOHIP_SEQ_STATE(T12);
    VOHIP_SEQ_STATE(p1_rc);
    VOHIP_SEQ_STATE(m_rc);
    VOHIP_SEQ_STATE(F_t);
    VOHIP_SEQ_STATE(T_t);
    VOHIP_SEQ_STATE(row_t);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(samples8);
    VOHIP_SEQ_STATE_OPT(n_dev);
    VOHIP_SEQ_PIN(u8);
    if (n_dev) n = min(n, *n_dev);   // (a group of loops: the count of each is known on the device only)
    int total;
    const int r = ordered_rank(valid, n, is_set(), total, sh);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && valid[i]) {
        p1_rc[2 * r] = C_rc[2 * i];
        p1_rc[2 * r + 1] = C_rc[2 * i + 1];
        m_rc[2 * r] = out_xy[ld + i];
        m_rc[2 * r + 1] = out_xy[i];
        F_t[2 * r] = F_rc[2 * i];
        F_t[2 * r + 1] = F_rc[2 * i + 1];
        for (int q = 0; q < 12; ++q) T_t[12 * (size_t)r + q] = T12[12 * (size_t)i + q];
        row_t[r] = row_base + i;   // candidate i is row row_base + i of the track table
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[N_CV] = total;
    // the trial samples (update_landmarks.m:13): every thread of the grid knows the count already
    if (total >= 8)
        for (int t = blockIdx.x * 256 + threadIdx.x; t < trials; t += gridDim.x * 256)
            octet_from_uniforms(u8 + 8 * (size_t)t, total, samples8 + 8 * (size_t)t);
}

__device__ double fm_trial_budget_dev(int inliers, int n, int num_trials, double confidence)
{
    const double r8 = pow((double)inliers / (double)n, 8.0);
    if (r8 >= 1.0) return 1.0;
    if (r8 <= 0.0) return (double)num_trials;
    const double b = ceil(log(1.0 - confidence / 100.0) / log(1.0 - r8));
    return b < (double)num_trials ? b : (double)num_trials;
}

// The sequential part of the fundamental-matrix RANSAC: first trial with the largest count inside
// the shrinking trial budget.  The budget is a function of the running maximum only, so a workgroup of NT
// threads replays NT trials at a time: prefix maximum (within each wavefront, then across them), budget per
// thread, first trial at which the loop ends.  Every thread returns the outcome.  `sm`: 3 * NT / 64 + 1 ints.
template <int NT>
__device__ void fm_decide_block(const int32_t *__restrict__ counts, int num_trials, double confidence, int n, int *sm,
                                int &best_t, int &run_max, int &trials)
{
    constexpr int NW = NT / 64;
    int *w_max = sm, *w_stop = sm + NW, *w_hit = sm + 2 * NW, *s_mlast = sm + 3 * NW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    run_max = 0;
    best_t = -1;
    trials = 0;
    if (n < 8) return;
    bool stopped = false;
    for (int base = 0; base < num_trials && !stopped; base += NT) {
        const int t = base + tid;
        const bool live = t < num_trials;
        const int c = live ? counts[t] : 0;
        const int pm = wave_prefix_max(c, lane);
        if (lane == 63) w_max[wave] = pm;
        __syncthreads();
        int m = max(pm, run_max);
        for (int w = 0; w < wave; ++w) m = max(m, w_max[w]);
        const double budget = fm_trial_budget_dev(m, n, num_trials, confidence);
        // after trial t the loop goes on iff t + 1 < budget (and t + 1 < num_trials)
        const unsigned long long stop = __ballot(live && !((double)(t + 1) < budget));
        if (lane == 0) w_stop[wave] = stop ? base + 64 * wave + __ffsll((long long)stop) - 1 : 0x7fffffff;
        __syncthreads();
        int last = 0x7fffffff;
        for (int w = 0; w < NW; ++w) last = min(last, w_stop[w]);
        if (last != 0x7fffffff) stopped = true;
        else last = min(base + NT, num_trials) - 1;
        if (t == last) *s_mlast = m;
        __syncthreads();
        const int m_last = *s_mlast;
        const unsigned long long hit = __ballot(t <= last && c == m_last);
        if (lane == 0) w_hit[wave] = hit ? base + 64 * wave + __ffsll((long long)hit) - 1 : 0x7fffffff;
        __syncthreads();
        if (m_last > run_max) {
            int first = 0x7fffffff;
            for (int w = 0; w < NW; ++w) first = min(first, w_hit[w]);
            best_t = first;
            run_max = m_last;
        }
        trials = last + 1;
        __syncthreads();   // (the arrays are rewritten by the next round)
    }
}

// update_landmarks.m:13-15,20-25: which trial wins (every workgroup replays the trial budget over the trials'
// counts by itself -- 500 integers, cheaper than a launch of its own; workgroup 0 publishes the outcome), then
// the epipolar inliers of the winning matrix (the test the counting kernel applied, evaluated here instead of in
// a mask kernel of its own).  No model flags nothing.
__global__ __launch_bounds__(256) void select_f_inliers_kernel(const double *__restrict__ p1_rc,
                                                               const double *__restrict__ m_rc,
                                                               const double *__restrict__ F_t,
                                                               const double *__restrict__ T_t,
                                                               const int32_t *__restrict__ counts, int num_trials,
                                                               double confidence, const double *__restrict__ F_all,
                                                               double *__restrict__ dv, double thr,
                                                               const int32_t *__restrict__ row_t,
                                                               int *__restrict__ cnt, double *__restrict__ m2_rc,
                                                               double *__restrict__ F2, double *__restrict__ T2,
                                                               int32_t *__restrict__ row2, vohip_seqs SQ)
{
    __shared__ int sh[12];
    __shared__ int sm[3 * 4 + 1];
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(p1_rc);
    VOHIP_SEQ_STATE(m_rc);
    VOHIP_SEQ_STATE(F_t);
    VOHIP_SEQ_STATE(T_t);
    VOHIP_SEQ_STATE(counts);
    VOHIP_SEQ_STATE(F_all);
    VOHIP_SEQ_STATE(dv);
    VOHIP_SEQ_STATE(row_t);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(m2_rc);
    VOHIP_SEQ_STATE(F2);
    VOHIP_SEQ_STATE(T2);
    VOHIP_SEQ_STATE(row2);
    int best_t, best_count, trials;
    fm_decide_block<256>(counts, num_trials, confidence, cnt[N_CV], sm, best_t, best_count, trials);
    double F[9];
    for (int q = 0; q < 9; ++q) F[q] = best_t >= 0 ? F_all[9 * (size_t)best_t + q] : 0.0;
    if (blockIdx.x == 0) {
        if (threadIdx.x < 9) dv[D_F + threadIdx.x] = F[threadIdx.x];
        if (threadIdx.x == 0) {
            cnt[F_BEST_T] = best_t;
            cnt[F_BEST_COUNT] = best_count;
            cnt[F_TRIALS] = trials;
        }
    }
    const int n = best_t >= 0 ? cnt[N_CV] : 0;
    auto inlier = [&](int j) {
        return vohip_dev::sampson_inlier(F, p1_rc[2 * j], p1_rc[2 * j + 1], m_rc[2 * j], m_rc[2 * j + 1], thr);
    };
    int total;
    const int r = ordered_rank_if(n, inlier, total, sh);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && inlier(i)) {
        m2_rc[2 * r] = m_rc[2 * i];
        m2_rc[2 * r + 1] = m_rc[2 * i + 1];
        F2[2 * r] = F_t[2 * i];
        F2[2 * r + 1] = F_t[2 * i + 1];
        for (int q = 0; q < 12; ++q) T2[12 * (size_t)r + q] = T_t[12 * (size_t)i + q];
        row2[r] = row_t[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[N_CI] = total;
}

// update_landmarks.m:83-97,120-123: status 1 -> new landmark (appended to S.X / S.P with a fresh
// id), status 0 -> stays a candidate, status 2 -> dropped.
__global__ __launch_bounds__(256) void split_candidates_kernel(const uint8_t *__restrict__ status,
                                                               const double *__restrict__ m2_rc,
                                                               const double *__restrict__ F2,
                                                               const double *__restrict__ T2,
                                                               const double *__restrict__ Xnew,
                                                               const int32_t *__restrict__ row2,
                                                               int *__restrict__ cnt, const int *__restrict__ n_det,
                                                               double *__restrict__ dv, int capL, int capK,
                                                               double *__restrict__ L_rc, double *__restrict__ L_X,
                                                               double *__restrict__ L_id, int32_t *__restrict__ srcL,
                                                               double *__restrict__ C_rc, double *__restrict__ F_rc,
                                                               double *__restrict__ T12, int32_t *__restrict__ srcC,
                                                               vohip_seqs SQ)
{
    __shared__ int sh[12];
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(status);
    VOHIP_SEQ_STATE(m2_rc);
    VOHIP_SEQ_STATE(F2);
    VOHIP_SEQ_STATE(T2);
    VOHIP_SEQ_STATE(Xnew);
    VOHIP_SEQ_STATE(row2);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(n_det);
    VOHIP_SEQ_STATE(dv);
    VOHIP_SEQ_STATE(L_rc);
    VOHIP_SEQ_STATE(L_X);
    VOHIP_SEQ_STATE(L_id);
    VOHIP_SEQ_STATE(srcL);
    VOHIP_SEQ_STATE(C_rc);
    VOHIP_SEQ_STATE(F_rc);
    VOHIP_SEQ_STATE(T12);
    VOHIP_SEQ_STATE(srcC);
    const int n = cnt[N_CI];
    const int nL = cnt[N_IN];
    const double idx0 = dv[D_IDX];
    int k_new, k_stay;
    const int r_new = ordered_rank(status, n, is_one(), k_new, sh);
    __syncthreads();
    const int r_stay = ordered_rank(status, n, is_zero(), k_stay, sh);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const uint8_t st = status[i];
        if (st == 1 && nL + r_new < capL) {
            const int d = nL + r_new;
            L_rc[2 * d] = m2_rc[2 * i];
            L_rc[2 * d + 1] = m2_rc[2 * i + 1];
            L_X[3 * d] = Xnew[3 * i];
            L_X[3 * d + 1] = Xnew[3 * i + 1];
            L_X[3 * d + 2] = Xnew[3 * i + 2];
            L_id[d] = idx0 + (double)r_new;
            srcL[d] = row2[i];
        } else if (st == 0) {
            C_rc[2 * r_stay] = m2_rc[2 * i];
            C_rc[2 * r_stay + 1] = m2_rc[2 * i + 1];
            F_rc[2 * r_stay] = F2[2 * i];
            F_rc[2 * r_stay + 1] = F2[2 * i + 1];
            for (int q = 0; q < 12; ++q) T12[12 * (size_t)r_stay + q] = T2[12 * (size_t)i + q];
            srcC[r_stay] = row2[i];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int kept = k_new;
        if (nL + kept > capL) {
            kept = capL - nL;
            cnt[OVERFLOW] |= 1;
        }
        cnt[N_L_END] = nL + kept;
        cnt[N_C_MID] = k_stay;
        cnt[NUM_NEW] = n - k_stay;
        dv[D_IDX_NEXT] = idx0 + (double)k_new;
        // the three counts the redundancy filter reads (update_candidate.m:12-24)
        cnt[FILT0] = min(*n_det, capK);
        cnt[FILT1] = k_stay;
        cnt[FILT2] = nL + kept;
    }
}

// The same three counts for a frame without candidates (update_landmarks does not run).
__global__ void filter_counts_kernel(int *__restrict__ cnt, const int *__restrict__ n_det, int capK, int have_map,
                                     double *__restrict__ dv, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(n_det);
    VOHIP_SEQ_STATE(dv);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    cnt[FILT0] = min(*n_det, capK);
    cnt[FILT1] = cnt[N_C_MID];
    cnt[FILT2] = cnt[N_L_END];
    if (!have_map) dv[D_IDX_NEXT] = dv[D_IDX];
}

// update_candidate.m:55-57: kept detections become candidates, first seen here, under this pose.
// Detection i is row det_base + i of the track table.
__global__ __launch_bounds__(256) void append_candidates_kernel(const float *__restrict__ corner_xy, int ld,
                                                                const uint8_t *__restrict__ keep,
                                                                int *__restrict__ cnt, double *__restrict__ dv,
                                                                int capC, int det_base, double *__restrict__ C_rc,
                                                                double *__restrict__ F_rc, double *__restrict__ T12,
                                                                int32_t *__restrict__ srcC, double *__restrict__ mirror_dv,
                                                                int *__restrict__ mirror_cnt,
                                                                const double *__restrict__ L_rc,
                                                                const double *__restrict__ L_X,
                                                                const double *__restrict__ L_id, int capL,
                                                                double *__restrict__ mirror_L, vohip_seqs SQ)
{
    __shared__ int sh[12];
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(corner_xy);
    VOHIP_SEQ_STATE(keep);
    VOHIP_SEQ_STATE(cnt);
    VOHIP_SEQ_STATE(dv);
    VOHIP_SEQ_STATE(C_rc);
    VOHIP_SEQ_STATE(F_rc);
    VOHIP_SEQ_STATE(T12);
    VOHIP_SEQ_STATE(srcC);
    VOHIP_SEQ_STATE(L_rc);
    VOHIP_SEQ_STATE(L_X);
    VOHIP_SEQ_STATE(L_id);
    VOHIP_SEQ_PIN(mirror_dv);
    VOHIP_SEQ_PIN(mirror_cnt);
    const int n = cnt[FILT0];
    const int nC = cnt[N_C_MID];
    int total;
    const int r = ordered_rank(keep, n, is_set(), total, sh);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && keep[i] && nC + r < capC) {
        const int d = nC + r;
        const double row = (double)corner_xy[ld + i], col = (double)corner_xy[i];
        C_rc[2 * d] = row;
        C_rc[2 * d + 1] = col;
        F_rc[2 * d] = row;
        F_rc[2 * d + 1] = col;
        srcC[d] = det_base + i;
        const double *T = dv + D_T;   // [R_W_C(:); t_W_C]
        double *o = T12 + 12 * (size_t)d;
        for (int c = 0; c < 3; ++c)
            for (int rr = 0; rr < 3; ++rr) o[3 * c + rr] = T[4 * c + rr];
        for (int rr = 0; rr < 3; ++rr) o[9 + rr] = T[12 + rr];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int kept = total;
        if (nC + kept > capC) {
            kept = capC - nC;
            cnt[OVERFLOW] |= 2;
        }
        cnt[N_KEPT] = kept;
        cnt[N_C_END] = nC + kept;
    }
    // This is the last kernel of a frame: the frame's summary goes to the caller's (pinned host)
    // mirror from here, which saves a copy command between this kernel and the synchronisation.
    if (blockIdx.x == 0) {
        __syncthreads();
        if (threadIdx.x < N_DOUBLES) mirror_dv[threadIdx.x] = dv[threadIdx.x];
        if (threadIdx.x < N_COUNTERS) mirror_cnt[threadIdx.x] = cnt[threadIdx.x];
    }
    // With bundle adjustment the host reads S.P and S.X after every frame (VO_bundle_adjust.m:14,29,92):
    // they go to its pinned copy from here as well ([S.P | S.X(1:3,:) | S.X(4,:)], capL columns each).
    if (mirror_L) {
        const int nL = min(cnt[N_L_END], capL);
        double *mP = mirror_L, *mX = mirror_L + 2 * (size_t)capL, *mI = mX + 3 * (size_t)capL;
        for (int k = blockIdx.x * 256 + threadIdx.x; k < nL; k += gridDim.x * 256) {
            mP[2 * k] = L_rc[2 * k];
            mP[2 * k + 1] = L_rc[2 * k + 1];
            mX[3 * k] = L_X[3 * k];
            mX[3 * k + 1] = L_X[3 * k + 1];
            mX[3 * k + 2] = L_X[3 * k + 2];
            mI[k] = L_id[k];
        }
    }
}

// VO_bundle_adjust.m:53-57 rewrites the pose of the current keyframe before update_candidate.m:57
// stamps it on the candidates the frame adds: here the frame has already added them (with the pose
// the localisation gave), so the optimised pose replaces it in dv and in the T columns of exactly
// those candidates, [N_C_MID, N_C_END).
__global__ __launch_bounds__(256) void restamp_pose_kernel(const double *__restrict__ T_new, const int *__restrict__ cnt,
                                                           double *__restrict__ dv, double *__restrict__ T12)
{
    const int first = cnt[N_C_MID], last = cnt[N_C_END];
    double o[12];   // [R_W_C(:); t_W_C]
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) o[3 * c + r] = T_new[4 * c + r];
    for (int r = 0; r < 3; ++r) o[9 + r] = T_new[12 + r];
    for (int d = first + blockIdx.x * 256 + threadIdx.x; d < last; d += gridDim.x * 256)
        for (int q = 0; q < 12; ++q) T12[12 * (size_t)d + q] = o[q];
    if (blockIdx.x == 0 && threadIdx.x < 16) dv[D_T + threadIdx.x] = T_new[threadIdx.x];
}

// The track tables, one for even and one for odd frames.  Row r holds a point of that frame's image
// which the tracker can follow into the next image: rows [0, capL) the landmarks in the order the
// frame starts with, rows [capL, capL + capC) the candidates, rows [capL + capC, capL + capC + capK)
// the frame's detections.  xy: x column then y column (leading dimension ld), MATLAB's (x, y); live:
// the row holds a point.  The tracker reads the row of the previous frame's table that a landmark
// (candidate) kept by that frame came from -- srcL (srcC) -- and writes the landmark (candidate)
// rows of the current frame's table.

// Image transfer by a kernel: src is pinned host memory that the GPU can address (16 bytes per lane).
__global__ __launch_bounds__(256) void fetch_image_kernel(const uint8_t *__restrict__ src, size_t bytes,
                                                          uint8_t *__restrict__ dst)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i + 16 <= bytes) {
        *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
    } else {
        for (size_t k = i; k < bytes; ++k) dst[k] = src[k];
    }
}

// Detections -> their rows of the table (main_demo.m:377; every detection may become a candidate).
__global__ __launch_bounds__(256) void detection_rows_kernel(const float *__restrict__ corner_xy, int capK,
                                                             const int *__restrict__ n_det, int base, int ld,
                                                             double *__restrict__ U_xy, uint8_t *__restrict__ U_live)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= capK) return;
    const bool on = i < *n_det;
    if (on) {
        U_xy[base + i] = (double)corner_xy[i];
        U_xy[ld + base + i] = (double)corner_xy[capK + i];
    }
    U_live[base + i] = on ? 1 : 0;
}

// The same for an image whose corners were detected ahead with the whole stored sequence: its list and
// count are copied to where the frame's other kernels read them, and become rows of the table.
__global__ __launch_bounds__(256) void prepared_rows_kernel(const float *__restrict__ src_xy,
                                                            const float *__restrict__ src_metric,
                                                            const int *__restrict__ src_count, int capK,
                                                            float *__restrict__ corner_xy,
                                                            float *__restrict__ corner_metric, int *__restrict__ n_det,
                                                            int base, int ld, double *__restrict__ U_xy,
                                                            uint8_t *__restrict__ U_live, vohip_seqs SQ)
{
    const int seq__ = vohip_dev::seq_index(SQ);
    VOHIP_SEQ_STATE(corner_xy);
    VOHIP_SEQ_STATE(corner_metric);
    VOHIP_SEQ_STATE(n_det);
    VOHIP_SEQ_STATE(U_xy);
    VOHIP_SEQ_STATE(U_live);
    VOHIP_SEQ_PREP(src_xy);
    VOHIP_SEQ_PREP(src_metric);
    VOHIP_SEQ_PREP(src_count);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= capK) return;
    const int n = *src_count;
    if (i == 0) *n_det = n;
    const bool on = i < n;
    if (on) {
        const float x = src_xy[i], y = src_xy[capK + i];
        corner_xy[i] = x;
        corner_xy[capK + i] = y;
        corner_metric[i] = src_metric[i];
        U_xy[base + i] = (double)x;
        U_xy[ld + base + i] = (double)y;
    }
    U_live[base + i] = on ? 1 : 0;
}

// main_demo.m:342-344: the landmarks a sequence starts with, as rows of the table.
__global__ __launch_bounds__(256) void init_rows_kernel(const double *__restrict__ L_rc, int n, int ld,
                                                        double *__restrict__ U_xy, uint8_t *__restrict__ U_live,
                                                        int32_t *__restrict__ srcL)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    U_xy[i] = L_rc[2 * i + 1];
    U_xy[ld + i] = L_rc[2 * i];
    U_live[i] = 1;
    srcL[i] = i;
}

}  // namespace

struct vohip_frame {
    vohip_ctx *ctx = nullptr;
    vohip_frame_params prm;
    double K[9];
    int rows = 0, cols = 0, colmajor = 0;
    int capL = 0, capC = 0, capK = 0, hmax = 0;
    hipStream_t side = nullptr, trk = nullptr;
    hipEvent_t ev_pyr = nullptr, ev_side = nullptr, ev_img = nullptr, ev_next = nullptr, ev_tail = nullptr;
    hipEvent_t ev_adv[2] = {nullptr, nullptr};   // after the trackers INTO the table of that parity
    hipEvent_t ev_advC[2] = {nullptr, nullptr};  // ... the candidate tracker, when it runs on the second stream
    hipEvent_t ev_feed = nullptr;                // what a tracker launch of this call waits for
    hipEvent_t ev_inl = nullptr;                 // after the inlier selection: the landmarks the frame keeps are final
    hipEvent_t ev_advA[2] = {nullptr, nullptr};  // after the early tracker of the kept landmarks INTO the table of that parity
    hipEvent_t tr[8] = {nullptr};   // timing events of the trace option
    hipEvent_t tr_trk[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // ... around the trackers into table [parity]
    hipEvent_t tr_det[2] = {nullptr, nullptr};                            // ... after the detection in the image of [parity]
    bool trace = false;
    void *block = nullptr;       // one device allocation, carved up below
    void *pinned = nullptr;      // host mirror of the counters and doubles
    double *ba_stage = nullptr;  // pinned host copies of S.P, S.X(1:3,:), S.X(4,:) for the bundle adjustment (6 x capL doubles)
    bool ba_mirrored = false;    // ... written by the last kernel of the frame just finished (is_BA)
    // A stored sequence prepared ahead (vohip_frame_prepare): the images on the device, their pyramids and
    // corner lists, all made by batched launches; a call recognises a prepared image by the caller's pointer.
    void *prep_block = nullptr;
    size_t prep_bytes = 0;
    int prep_n = 0;
    uint8_t *prep_imgs = nullptr;
    float *prep_pyr = nullptr, *prep_xy = nullptr, *prep_metric = nullptr;
    int *prep_counts = nullptr;
    size_t prep_pyr_floats = 0;
    std::unordered_map<const void *, int> prep_index;
    bool next_pyramid_stands = false;   // the following image's pyramid was prepared: no event to wait for
    uint8_t *img[2] = {nullptr, nullptr};   // image staging (host images), by frame parity
    double *L_rc, *L_X, *L_id, *C_rc, *F_rc, *T12;
    double *mv_rc, *Xv, *idv, *kp_uv;
    double *p1_rc, *m_rc, *F_t, *T_t, *m2_rc, *F2, *T2, *Xnew;
    double *u8, *fm_F, *dv;
    double *Rt_all, *M_all;      // candidate poses of the frame's hypotheses (48 doubles per hypothesis each)
    void *corner_scratch;        // the corner detector's records (this loop's own: detection runs on its own stream)
    unsigned det_calls = 0;
    double *U_xy[2];             // the track tables of even and odd frames
    uint8_t *U_live[2];
    int32_t *srcL, *srcC;        // table row of every landmark / candidate the next frame starts with
    int32_t *rowv, *row_t, *row2;
    float *corner_xy[2], *corner_metric[2];   // detections, by frame parity
    int32_t *samples3, *counts4, *samples8, *fm_counts;
    int *cnt;
    uint8_t *mask, *status, *keep;
    int capU = 0;                // rows of the track table
    int slot_prev = 0, slot_cur = 0, slot_next = 0;   // pyramids of the previous, the current and the following image
    int par = 0;                 // parity of the frame being processed
    bool ahead_launched = false; // this call prepares the next image
    bool split_used[2] = {false, false};   // the last trackers INTO the table of that parity ran as two launches (ev_advC recorded)
    bool ahead = false;          // the previous call left this image's pyramid and corners on the device and
                                 // enqueued this frame's trackers behind its own last kernel
    int n_L = 0, n_C = 0;        // counts at the start of the frame (known to the host)
    int evaluated = 0;           // hypotheses scored in the current frame
    int stage = 0;               // 0 = between frames, 1 = a frame waits for more hypotheses, 2 = a frame is enqueued,
                                 // STAGE_FAILED = an enqueue failed half-way (only reset and destroy are accepted)
    int h_cur = 0;               // hypotheses supplied with the enqueued frame
    double enqueue_us = 0.0;     // host time the enqueue of the current frame took
    bool have_map = false;       // the frame started with candidates (update_landmarks runs)
    std::chrono::steady_clock::time_point t_entry;   // host-side timing of the current call
    // A group of loops advanced in lockstep (vohip_group_*): nseq state blocks, pinned blocks and prepared
    // sequences of one layout each, a constant number of bytes apart; the members above then describe
    // sequence 0 (pointers) or the whole group (n_L, n_C: the largest count, which sizes the grids).
    bool is_group = false;
    int nseq = 1;
    vohip_seqs Q, Qn;        // the sequences the launches of the current frame serve / those of the look-ahead
    size_t pin_bytes = 0;    // of one sequence's pinned block
    struct seq_state {
        int n_L = 0, n_C = 0, evaluated = 0, n_frames = 0;
        int stage = 0;       // 0 = in step, 1 = waits for vohip_group_retry, STAGE_FAILED
        bool reset = false;  // takes part in the run that is under way
    };
    std::vector<seq_state> seqs;
    int g_max_frames = 0;
    int g_frame = 0;         // index (within the prepared sequences) of the frame processed last
    bool g_inflight = false; // vohip_group_begin without its vohip_group_finish
    int g_h = 0;
    double *prep_xy_all = nullptr;   // (group) corner lists of the prepared frames, sequence 0
};

namespace {

template <class T>
T *carve(char *&p, size_t count)
{
    T *r = reinterpret_cast<T *>(p);
    p += (count * sizeof(T) + 255) / 256 * 256;
    return r;
}

size_t layout(vohip_frame *f, char *base)
{
    char *p = base;
    const size_t L = f->capL, C = f->capC, Kc = f->capK, H = f->hmax, T = f->prm.fm_trials, U = f->capU;
    f->dv = carve<double>(p, N_DOUBLES);   // dv and cnt are adjacent: one copy brings both to the host
    f->cnt = carve<int>(p, N_COUNTERS);
    f->L_rc = carve<double>(p, 2 * L);
    f->L_X = carve<double>(p, 3 * L);
    f->L_id = carve<double>(p, L);
    f->C_rc = carve<double>(p, 2 * C);
    f->F_rc = carve<double>(p, 2 * C);
    f->T12 = carve<double>(p, 12 * C);
    f->U_xy[0] = carve<double>(p, 2 * U);
    f->U_xy[1] = carve<double>(p, 2 * U);
    f->mv_rc = carve<double>(p, 2 * L);
    f->Xv = carve<double>(p, 3 * L);
    f->idv = carve<double>(p, L);
    f->kp_uv = carve<double>(p, 2 * L);
    f->p1_rc = carve<double>(p, 2 * C);
    f->m_rc = carve<double>(p, 2 * C);
    f->F_t = carve<double>(p, 2 * C);
    f->T_t = carve<double>(p, 12 * C);
    f->m2_rc = carve<double>(p, 2 * C);
    f->F2 = carve<double>(p, 2 * C);
    f->T2 = carve<double>(p, 12 * C);
    f->Xnew = carve<double>(p, 3 * C);
    f->corner_scratch = carve<char>(p, f->prm.detector == 3   ? vohip_surf_scratch_bytes(f->rows, f->cols)   // (first: zero from the start)
                                       : f->prm.detector == 4 ? vohip_brisk_scratch_bytes(f->rows, f->cols)
                                                              : vohip_corner_scratch_bytes(f->rows, f->cols));
    f->Rt_all = carve<double>(p, 48 * H);
    f->M_all = carve<double>(p, 48 * H);
    f->u8 = carve<double>(p, 8 * T);
    f->fm_F = carve<double>(p, 9 * T);
    for (int q = 0; q < 2; ++q) {
        f->corner_xy[q] = carve<float>(p, 2 * Kc);
        f->corner_metric[q] = carve<float>(p, Kc);
    }
    f->srcL = carve<int32_t>(p, L);
    f->srcC = carve<int32_t>(p, C);
    f->rowv = carve<int32_t>(p, L);
    f->row_t = carve<int32_t>(p, C);
    f->row2 = carve<int32_t>(p, C);
    f->samples3 = carve<int32_t>(p, 3 * H);
    f->counts4 = carve<int32_t>(p, 4 * H);
    f->samples8 = carve<int32_t>(p, 8 * T);
    f->fm_counts = carve<int32_t>(p, T);
    f->U_live[0] = carve<uint8_t>(p, U);
    f->U_live[1] = carve<uint8_t>(p, U);
    f->mask = carve<uint8_t>(p, L);
    f->status = carve<uint8_t>(p, C);
    f->keep = carve<uint8_t>(p, Kc);
    for (int q = 0; q < 2; ++q) f->img[q] = carve<uint8_t>(p, (size_t)f->rows * f->cols);
    return (size_t)(p - base);
}

inline int blocks256(int n) { return n < 1 ? 1 : (n + 255) / 256; }

// The one threshold option a detector has beside MinQuality: FAST's and BRISK's 'MinContrast', SURF's 'MetricThreshold'.
inline double detector_option(const vohip_frame_params &p) { return p.detector == 3 ? p.metric_threshold : p.min_contrast; }

struct stream_swap {
    vohip_ctx *ctx;
    hipStream_t saved;
    stream_swap(vohip_ctx *c, hipStream_t s) : ctx(c), saved(c->stream) { c->stream = s; }
    ~stream_swap() { ctx->stream = saved; }
};

// pinned host block: [dv mirror | cnt mirror | staging of the P3P uniforms | staging of the trial uniforms]
constexpr size_t PIN_CNT = sizeof(double) * N_DOUBLES;
constexpr size_t PIN_U3 = PIN_CNT + sizeof(int) * N_COUNTERS;
inline double *pin_dv(vohip_frame *f, int seq = 0) { return (double *)((char *)f->pinned + f->pin_bytes * seq); }
inline int *pin_cnt(vohip_frame *f, int seq = 0) { return (int *)((char *)f->pinned + f->pin_bytes * seq + PIN_CNT); }
inline double *pin_u3(vohip_frame *f, int seq = 0) { return (double *)((char *)f->pinned + f->pin_bytes * seq + PIN_U3); }
inline double *pin_u8(vohip_frame *f, int seq = 0) { return pin_u3(f, seq) + 3 * (size_t)f->hmax; }
// the same array of another sequence of the group
template <class T>
inline T *seq_host(vohip_frame *f, T *p, int seq)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(p) + f->Q.stride_state * (size_t)seq);
}
inline int ny_of(const vohip_seqs &q) { return vohip_seq_count(q); }

// A pyramid slot of this loop may show a pyramid inside the prepared block (vohip_pyramid_attach).  Before
// that block is released, such a pyramid is copied into the slot's own buffer, so that the trackers of the
// following frames still find it; with copy = false the slot is just emptied (the loop ends).
int detach_prepared_slots(vohip_frame *f, bool copy)
{
    vohip_ctx *ctx = f->ctx;
    if (!f->prep_block) return VOHIP_OK;
    const char *lo = (const char *)f->prep_block, *hi = lo + f->prep_bytes;
    const int slots[3] = {f->slot_prev, f->slot_cur, f->slot_next};
    for (int q = 0; q < 3; ++q) {
        const int slot = slots[q];
        const char *ext = (const char *)ctx->pyr_ext[slot];
        if (!ext || ext < lo || ext >= hi) continue;
        if (copy) {
            const size_t bytes = sizeof(float) * f->prep_pyr_floats;
            VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[B_PYR0 + slot], bytes));
            VOHIP_HIP(ctx, hipMemcpyAsync(ctx->bufs[B_PYR0 + slot].ptr, ext, bytes, hipMemcpyDeviceToDevice, ctx->stream));
            VOHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        } else {
            ctx->pyr[slot].levels = 0;
        }
        ctx->pyr_ext[slot] = nullptr;
    }
    return VOHIP_OK;
}

int fetch_summary(vohip_frame *f)
{
    vohip_ctx *ctx = f->ctx;
    // (the last kernel of the frame wrote the summary into the pinned mirror; the stream itself may
    // already hold the next frame's trackers, which are not waited for)
    VOHIP_HIP(ctx, hipEventSynchronize(f->ev_tail));
    return VOHIP_OK;
}

// A host image to its device buffer on `stream`.  Pinned memory the GPU can address is fetched by a
// kernel (one launch; a copy command followed by kernels on the same stream costs the host far more),
// anything else goes through hipMemcpyAsync.
int enqueue_image_transfer(vohip_frame *f, const uint8_t *host, uint8_t *dst, hipStream_t stream)
{
    vohip_ctx *ctx = f->ctx;
    const size_t bytes = (size_t)f->rows * f->cols;
    hipPointerAttribute_t attr;
    void *dev = nullptr;
    if (hipPointerGetAttributes(&attr, host) == hipSuccess && attr.type == hipMemoryTypeHost &&
        hipHostGetDevicePointer(&dev, const_cast<uint8_t *>(host), 0) == hipSuccess && dev &&
        (reinterpret_cast<uintptr_t>(dev) & 15) == 0) {
        hipLaunchKernelGGL(fetch_image_kernel, dim3((unsigned)((bytes + 4095) / 4096)), dim3(256), 0, stream,
                           (const uint8_t *)dev, bytes, dst);
        VOHIP_HIP(ctx, hipGetLastError());
        return VOHIP_OK;
    }
    (void)hipGetLastError();   // (a pageable pointer makes the query fail: not an error here)
    VOHIP_HIP(ctx, hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, stream));
    return VOHIP_OK;
}

// Second stream, after the trackers: the epipolar filter over the candidate tracks
// (update_landmarks.m:11-25), which needs neither the new pose nor the localisation draws.
int enqueue_side_filter(vohip_frame *f, const double *u8)
{
    vohip_ctx *ctx = f->ctx;
    const vohip_frame_params &p = f->prm;
    const int T = p.fm_trials;
    (void)u8;   // in the pinned block already
    stream_swap sw(ctx, f->side);
    hipStream_t s = f->side;
    // (redundant when the candidate tracker ran on this very stream)
    VOHIP_HIP(ctx, hipStreamWaitEvent(s, f->ev_adv[f->par], 0));
    if (f->trace) (void)hipEventRecord(f->tr[3], s);
    if (f->have_map) {
        const int ny = ny_of(f->Q);
        hipLaunchKernelGGL(select_cand_tracked_kernel, dim3(blocks256(f->n_C), ny), dim3(256), 0, s, f->U_xy[f->par] + f->capL,
                           f->capU, f->U_live[f->par] + f->capL, f->C_rc, f->F_rc, f->T12, f->n_C, f->p1_rc, f->m_rc, f->F_t,
                           f->T_t, f->capL, f->row_t, f->cnt, pin_u8(f), T, f->samples8,
                           f->is_group ? (const int *)(f->cnt + N_C_END) : (const int *)nullptr, f->Q);
        VOHIP_TRY(vohip_fm_enqueue(ctx, f->p1_rc, f->m_rc, f->n_C, f->cnt + N_CV, f->samples8, T, p.fm_threshold,
                                   f->dv + D_NRM, f->fm_F, f->fm_counts, &f->Q));
        hipLaunchKernelGGL(select_f_inliers_kernel, dim3(blocks256(f->n_C), ny), dim3(256), 0, s, f->p1_rc, f->m_rc, f->F_t,
                           f->T_t, (const int32_t *)f->fm_counts, T, p.fm_confidence, (const double *)f->fm_F, f->dv,
                           p.fm_threshold, f->row_t, f->cnt, f->m2_rc, f->F2, f->T2, f->row2, f->Q);
        VOHIP_HIP(ctx, hipGetLastError());
    }
    VOHIP_HIP(ctx, hipEventRecord(f->ev_side, s));
    if (f->trace) (void)hipEventRecord(f->tr[4], s);   // epipolar filter done
    return VOHIP_OK;
}

// Corner detection (main_demo.m:377) needs only the image; the detections also become rows of the track
// table.  It shares the second stream with the epipolar filter, and the stream's order does the
// synchronising that events would otherwise do: enqueued before the frame's filter (plain call) or after it
// (the following image's, look-ahead), it follows the filter's wait for the trackers that still read the
// table of that parity, and the event the tail waits for -- recorded behind the NEXT filter on this stream --
// covers it.  `par`: parity of the frame the image belongs to; `ready`: event after which the image is on
// the device, or NULL.
int enqueue_detection(vohip_frame *f, const uint8_t *d_img, int par, hipEvent_t ready, int prepared = -1,
                      const vohip_seqs *seqs = nullptr)
{
    vohip_ctx *ctx = f->ctx;
    const vohip_frame_params &p = f->prm;
    hipStream_t s = f->side;
    const vohip_seqs &SQ = seqs ? *seqs : f->Q;
    if (prepared >= 0) {   // detected ahead with the whole sequence: one kernel hands the list over
        hipLaunchKernelGGL(prepared_rows_kernel, dim3(blocks256(f->capK), ny_of(SQ)), dim3(256), 0, s,
                           (const float *)(f->prep_xy + 2 * (size_t)f->capK * prepared),
                           (const float *)(f->prep_metric + (size_t)f->capK * prepared), (const int *)(f->prep_counts + prepared),
                           f->capK, f->corner_xy[par], f->corner_metric[par], f->cnt + N_DET + par, f->capL + f->capC, f->capU,
                           f->U_xy[par], f->U_live[par], SQ);
        VOHIP_HIP(ctx, hipGetLastError());
        if (f->trace) (void)hipEventRecord(f->tr_det[par], s);
        return VOHIP_OK;
    }
    if (ready) VOHIP_HIP(ctx, hipStreamWaitEvent(s, ready, 0));   // (NULL: the image was on the device all along)
    VOHIP_TRY(vohip_detect_corners_enqueue(ctx, s, f->corner_scratch, &f->det_calls, d_img, f->rows, f->cols,
                                           f->colmajor, p.min_quality, p.filter_size, p.border, p.detector,
                                           detector_option(p), f->corner_xy[par], f->corner_metric[par], f->capK, f->cnt + N_DET + par,
                                           p.brisk_octaves));
    hipLaunchKernelGGL(detection_rows_kernel, dim3(blocks256(f->capK)), dim3(256), 0, s, f->corner_xy[par], f->capK,
                       f->cnt + N_DET + par, f->capL + f->capC, f->capU, f->U_xy[par], f->U_live[par]);
    VOHIP_HIP(ctx, hipGetLastError());
    if (f->trace) (void)hipEventRecord(f->tr_det[par], s);   // detection done
    return VOHIP_OK;
}

// Localisation over hypotheses [f->evaluated, h) (ransacLocalization.m, T_refinement.m) on the
// context's stream.  Every kernel reads only buffers an earlier run leaves intact, so it can run
// again with more hypotheses when the adaptive bound asks for them.
// `ready` hypotheses at the front of the range already have their draws on the device and their
// triples made (by the selection kernel).
int enqueue_localize(vohip_frame *f, const double *u3, int h, int ready, bool with_selection = false)
{
    vohip_ctx *ctx = f->ctx;
    hipStream_t s = ctx->stream;
    const vohip_frame_params &p = f->prm;
    const int first = f->evaluated, more = h - first;
    // (the kernels read the draws from the pinned staging block itself: a few KB, no copy command)
    // (u3 == NULL: the caller has put them there itself)
    if (ready == 0 && u3)
        std::memcpy(pin_u3(f) + 3 * (size_t)first, u3 + 3 * (size_t)first, sizeof(double) * 3 * (size_t)more);
    const int ny = ny_of(f->Q);
    if (more > ready)
        hipLaunchKernelGGL(triples_kernel, dim3(blocks256(more - ready), ny), dim3(256), 0, s, pin_u3(f), first + ready,
                           more - ready, f->cnt, f->samples3, f->Q);
    // with_selection: the selection of the valid tracks has not been launched; it rides with the solve kernel
    const vohip_select_tracked sel = {f->U_xy[f->par], f->capU, f->U_live[f->par], f->L_X, f->L_id, f->n_L, f->mv_rc, f->Xv,
                                      f->idv, f->rowv, f->cnt + N_V, f->dv + D_IDX, f->dv + D_IDX_NEXT, pin_u3(f), ready,
                                      f->samples3, f->is_group ? f->cnt + N_L_END : nullptr};
    VOHIP_TRY(vohip_ransac_enqueue(ctx, f->mv_rc, f->Xv, f->n_L, f->cnt + N_V, f->K, f->samples3, first, more,
                                   f->counts4, f->Rt_all, f->M_all, with_selection ? &sel : nullptr, &f->Q));
    f->evaluated = h;
    hipLaunchKernelGGL(select_inliers_kernel, dim3(blocks256(f->n_L), ny), dim3(256), 0, s, f->counts4, h,
                       (const double *)f->Rt_all, (const double *)f->M_all, f->mv_rc,
                       f->Xv, f->idv, f->rowv,
                       f->dv, f->cnt, f->L_rc, f->L_X, f->L_id, f->srcL, f->kp_uv, f->Q);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_HIP(ctx, hipEventRecord(f->ev_inl, s));   // S.P of the inliers is final: their tracker may start
    if (p.refine_pose)   // main_demo.m:398-400
        VOHIP_TRY(vohip_pose_refine_enqueue(ctx, f->dv + D_T, f->kp_uv, f->L_X, f->n_L, f->cnt + N_IN, f->K,
                                            p.refine_max_iter, 1e-6, 1e-6, f->dv + D_STATS, &f->Q));   // lsqnonlin's default tolerances (T_refinement.m:4)
    return VOHIP_OK;
}

// update_landmarks.m:27-125 and update_candidate.m with the new pose, once the other streams are done.
int enqueue_tail(vohip_frame *f)
{
    vohip_ctx *ctx = f->ctx;
    hipStream_t s = ctx->stream;
    const vohip_frame_params &p = f->prm;
    VOHIP_HIP(ctx, hipStreamWaitEvent(s, f->ev_side, 0));   // (the epipolar filter, and this image's corners before it on that stream)
    const int *n_det = f->cnt + N_DET + f->par;
    const int ny = ny_of(f->Q);
    if (f->have_map) {
        VOHIP_TRY(vohip_candidate_triangulate_enqueue(ctx, f->m2_rc, f->F2, f->T2, f->n_C, f->cnt + N_CI, f->dv + D_T,
                                                      f->K, p.angle_threshold_deg, p.min_depth, p.max_depth,
                                                      p.is_BA ? 1 : 0, f->status, f->Xnew, &f->Q));
        hipLaunchKernelGGL(split_candidates_kernel, dim3(blocks256(f->n_C), ny), dim3(256), 0, s, f->status, f->m2_rc, f->F2,
                           f->T2, f->Xnew, f->row2, f->cnt, n_det, f->dv, f->capL, f->capK, f->L_rc, f->L_X, f->L_id,
                           f->srcL, f->C_rc, f->F_rc, f->T12, f->srcC, f->Q);
    } else {
        hipLaunchKernelGGL(filter_counts_kernel, dim3(1, ny), dim3(64), 0, s, f->cnt, n_det, f->capK, 0, f->dv, f->Q);
    }
    VOHIP_TRY(vohip_candidate_filter_enqueue(ctx, f->corner_xy[f->par], f->capK, f->capK, f->C_rc, f->L_rc,
                                             f->cnt + FILT0, p.r_discard_redundant, f->keep, &f->Q));
    hipLaunchKernelGGL(append_candidates_kernel, dim3(blocks256(f->capK), ny), dim3(256), 0, s, f->corner_xy[f->par], f->capK,
                       f->keep, f->cnt, f->dv, f->capC, f->capL + f->capC, f->C_rc, f->F_rc, f->T12, f->srcC, pin_dv(f),
                       pin_cnt(f), (const double *)f->L_rc, (const double *)f->L_X, (const double *)f->L_id, f->capL,
                       f->ba_stage, f->Q);
    f->ba_mirrored = f->ba_stage != nullptr;
    VOHIP_HIP(ctx, hipGetLastError());
    if (f->trace) (void)hipEventRecord(f->tr[6], s);   // tail done (before the event the host waits for: it is read right after)
    VOHIP_HIP(ctx, hipEventRecord(f->ev_tail, s));     // what the frame's single synchronisation waits for
    return VOHIP_OK;
}

// The landmark tracker (main_demo.m:380) and the candidate tracker (update_landmarks.m:10) in one
// launch -- it lasts as long as its slowest track however many tracks it carries -- from the rows of
// table par ^ 1 that the frame before kept (srcL, srcC) into the landmark / candidate rows of table
// par, between the pyramids of slot_from and slot_to.  counts_on_device: nL and nC only bound the
// grid and the counts are the ones the frame before left in device memory (that frame may still be
// running: this is how the trackers of the following frame are enqueued behind a frame's last kernel).
int enqueue_trackers(vohip_frame *f, int par, int slot_from, int slot_to, int nL, int nC, bool counts_on_device,
                     hipEvent_t start, const vohip_seqs *seqs = nullptr, bool kept_apart = true)
{
    vohip_ctx *ctx = f->ctx;
    const vohip_frame_params &p = f->prm;
    const vohip_seqs *SQ = seqs ? seqs : &f->Q;
    const int *dL = counts_on_device ? f->cnt + N_L_END : nullptr, *dC = counts_on_device ? f->cnt + N_C_END : nullptr;
    // ahead of the frame, the landmarks the previous frame kept as inliers have a tracker of their own
    // (enqueue_tracker_kept); this launch then starts at the landmarks that frame appended
    const int *dFirst = counts_on_device && kept_apart ? f->cnt + N_IN : nullptr;
    const vohip_klt_set sets[2] = {
        {f->U_xy[par ^ 1], f->U_xy[par], f->U_live[par], nullptr, f->srcL, nullptr, dL, f->capU, nL, 0, 0, dFirst},
        {f->U_xy[par ^ 1], f->U_xy[par], f->U_live[par], nullptr, f->srcC, nullptr, dC, f->capU, nC, 0, f->capL, nullptr}};
    if (f->trace) (void)hipEventRecord(f->tr_trk[par][0], ctx->stream);
    if (!f->is_group && (counts_on_device ? f->n_C : f->n_L + f->n_C) >= SPLIT_TRACKERS_FROM) {
        // landmarks on the context's stream, candidates on the second stream (where the epipolar filter
        // follows them in stream order); `start`: what the candidate launch waits for on the context's stream
        VOHIP_TRY(vohip_klt_enqueue_sets(ctx, slot_from, slot_to, sets, 1, p.klt_block, p.klt_max_iter,
                                         p.klt_max_bidir_error));
        if (f->trace) (void)hipEventRecord(f->tr_trk[par][1], ctx->stream);
        VOHIP_HIP(ctx, hipEventRecord(f->ev_adv[par], ctx->stream));
        stream_swap sw(ctx, f->side);
        VOHIP_HIP(ctx, hipStreamWaitEvent(f->side, start, 0));
        if (counts_on_device && !f->next_pyramid_stands)
            VOHIP_HIP(ctx, hipStreamWaitEvent(f->side, f->ev_img, 0));   // the following image's pyramid
        VOHIP_TRY(vohip_klt_enqueue_sets(ctx, slot_from, slot_to, sets + 1, 1, p.klt_block, p.klt_max_iter,
                                         p.klt_max_bidir_error));
        VOHIP_HIP(ctx, hipEventRecord(f->ev_advC[par], f->side));
        f->split_used[par] = true;
        return VOHIP_OK;
    }
    f->split_used[par] = false;
    VOHIP_TRY(vohip_klt_enqueue_sets(ctx, slot_from, slot_to, sets, 2, p.klt_block, p.klt_max_iter, p.klt_max_bidir_error, SQ));
    if (f->trace) (void)hipEventRecord(f->tr_trk[par][1], ctx->stream);
    VOHIP_HIP(ctx, hipEventRecord(f->ev_adv[par], ctx->stream));   // the other streams start from here
    return VOHIP_OK;
}

// The landmarks a frame keeps (the inliers of its pose, main_demo.m:394-396) are final long before the
// frame ends -- pose refinement, triangulation and the candidate bookkeeping follow -- so with the
// following image's pyramid on its way their tracker runs beside those stages, on the stream that
// built the pyramid, from the rows srcL[0 .. N_IN) into the first landmark rows of the other table.
int enqueue_tracker_kept(vohip_frame *f)
{
    vohip_ctx *ctx = f->ctx;
    const vohip_frame_params &p = f->prm;
    const int par = f->par ^ 1;   // the table of the following frame
    stream_swap sw(ctx, f->trk);  // (in order behind the following image's pyramid)
    VOHIP_HIP(ctx, hipStreamWaitEvent(f->trk, f->ev_inl, 0));
    const vohip_klt_set set = {f->U_xy[par ^ 1], f->U_xy[par], f->U_live[par], nullptr, f->srcL, nullptr, f->cnt + N_IN,
                               f->capU, f->n_L, 0, 0, nullptr};
    VOHIP_TRY(vohip_klt_enqueue_sets(ctx, f->slot_cur, f->slot_next, &set, 1, p.klt_block, p.klt_max_iter,
                                     p.klt_max_bidir_error, &f->Qn));
    VOHIP_HIP(ctx, hipEventRecord(f->ev_advA[par], f->trk));
    return VOHIP_OK;
}

// With the following image's pyramid on its way (the look-ahead), that frame's trackers go behind this
// frame's last kernel: they need nothing from the host, so the GPU starts them while the host still
// collects this frame's results and prepares the next call.
int enqueue_trackers_ahead(vohip_frame *f)
{
    vohip_ctx *ctx = f->ctx;
    hipStream_t s = ctx->stream;
    if (!f->next_pyramid_stands) VOHIP_HIP(ctx, hipStreamWaitEvent(s, f->ev_img, 0));   // the following image's pyramid
    // the landmarks this launch tracks are the ones the frame appends, which come from its candidates (the
    // kept ones have their own, earlier launch); candidates grow by the detections that are kept
    const long long nLb = (long long)f->n_C, nCb = (long long)f->n_C + f->capK;
    VOHIP_TRY(enqueue_trackers(f, f->par ^ 1, f->slot_cur, f->slot_next, (int)(nLb < f->capL ? nLb : f->capL),
                               (int)(nCb < f->capC ? nCb : f->capC), true, f->ev_tail, &f->Qn));
    return VOHIP_OK;
}

// The frame's single synchronisation and its results.
int complete_frame(vohip_frame *f, int h, vohip_frame_pose *pose, vohip_frame_counts *out)
{
    vohip_ctx *ctx = f->ctx;
    const auto t_wait = std::chrono::steady_clock::now();
    if (const int st = fetch_summary(f)) {
        f->stage = STAGE_FAILED;
        return st;
    }
    const auto t_done = std::chrono::steady_clock::now();
    const int *hc = pin_cnt(f);
    const double *hd = pin_dv(f);
    std::memset(pose, 0, sizeof(*pose));
    std::memset(out, 0, sizeof(*out));
    pose->n_tracked = hc[N_V];
    pose->n_iter = hc[R_ITER];
    pose->found = hc[R_FOUND];
    pose->exhausted = hc[R_EXHAUSTED];
    pose->n_inliers = hc[N_IN];
    pose->max_num_inliers = hc[R_MAX_INL];
    pose->evaluated = h;
    pose->num_iterations = hd[D_NUM_ITER];
    std::memcpy(pose->T_W_C, hd + D_T, sizeof(double) * 16);
    std::memcpy(pose->Rt, hd + D_RT, sizeof(double) * 12);
    std::memcpy(pose->refine_stats, hd + D_STATS, sizeof(double) * 3);
    if (hc[N_V] >= 3 && hc[R_EXHAUSTED] && h < f->hmax)   // stage stays 1: vohip_frame_retry goes on from here
        return vohip_fail(ctx, VOHIP_SAMPLES_EXHAUSTED, "the adaptive bound (%.0f) exceeds the %d hypotheses supplied",
                          hd[D_NUM_ITER], h);
    out->n_landmarks = hc[N_L_END];
    out->n_candidates = hc[N_C_END];
    out->num_new = hc[NUM_NEW];
    out->n_detected = hc[N_DET + f->par];
    out->n_kept = hc[N_KEPT];
    out->n_candidates_tracked = f->have_map ? hc[N_CV] : 0;
    out->n_candidates_inliers = f->have_map ? hc[N_CI] : 0;
    out->fm_trials_run = f->have_map ? hc[F_TRIALS] : 0;
    out->overflow = hc[OVERFLOW];
    out->host_enqueue_us = f->enqueue_us;
    out->host_wait_us = std::chrono::duration<double, std::micro>(t_done - t_wait).count();
    if (f->trace) {
        float ms[8] = {0};
        for (int q = 2; q < 7; ++q)
            if (q != 5) (void)hipEventElapsedTime(&ms[q], f->tr[0], f->tr[q]);
        // (negative when the previous call enqueued this frame's trackers behind its own last kernel)
        (void)hipEventElapsedTime(&ms[7], f->tr[0], f->tr_trk[f->par][0]);
        (void)hipEventElapsedTime(&ms[1], f->tr[0], f->tr_trk[f->par][1]);
        (void)hipEventElapsedTime(&ms[5], f->tr[0], f->tr_det[f->par]);   // (negative too when the detection ran ahead)
        (void)hipGetLastError();   // (an event that was not recorded in this frame is no error of the frame)
        std::fprintf(stderr, "vohip frame trace (us from start): tracks %.0f-%.0f  refined %.0f  filter %.0f-%.0f  detect %.0f  tail %.0f | host enqueue %.0f wait %.0f",
                     1e3 * ms[7], 1e3 * ms[1], 1e3 * ms[2], 1e3 * ms[3], 1e3 * ms[4], 1e3 * ms[5], 1e3 * ms[6], out->host_enqueue_us,
                     out->host_wait_us);
        std::fprintf(stderr, " | rows L %d C %d det %d\n", f->n_L, f->n_C, hc[N_DET + f->par]);
    }
    const int t = f->slot_prev;   // main_demo.m:417-424: this frame's pyramid is the next "previous"
    f->slot_prev = f->slot_cur;
    f->slot_cur = f->slot_next;   // (holds the following image's pyramid when it was prepared ahead)
    f->slot_next = t;
    f->n_L = hc[N_L_END];
    f->n_C = hc[N_C_END];
    f->stage = 0;
    f->ahead = f->ahead_launched;
    const int n_det_now = hc[N_DET + f->par];
    f->par ^= 1;
    // The reference stops at each of these (main_demo.m:389-391 without tracks, the 8-point algorithm
    // without matches) or never meets it (MATLAB arrays grow: update_landmarks.m:83-95 drops nothing).
    // Here the frame has run to its end on truncated or empty lists, so the state no longer is what the
    // reference would hold: only vohip_frame_reset (or destroy) is accepted after any of them.
    const bool lost = hc[N_V] < 3 || !hc[R_FOUND] || (f->have_map && hc[N_CV] < 8);
    const bool over = n_det_now > f->capK || hc[OVERFLOW] != 0;
    if (lost || over) f->stage = STAGE_FAILED;
    if (hc[N_V] < 3) return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "only %d landmarks were tracked", hc[N_V]);
    if (!hc[R_FOUND])
        return vohip_fail(ctx, VOHIP_NO_MODEL, "no hypothesis reached 30 inliers in %d iterations", hc[R_ITER]);
    if (f->have_map && hc[N_CV] < 8)
        return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "the 8-point algorithm needs 8 candidate tracks, got %d", hc[N_CV]);
    if (n_det_now > f->capK)
        return vohip_fail(ctx, VOHIP_CAPACITY_EXCEEDED, "%d corners found, capacity %d", n_det_now, f->capK);
    if (hc[OVERFLOW])
        return vohip_fail(ctx, VOHIP_CAPACITY_EXCEEDED, "landmark or candidate capacity exceeded (flags %d)", hc[OVERFLOW]);
    return VOHIP_OK;
}

int begin_enqueue(vohip_frame *f, const uint8_t *image, int image_on_device, const uint8_t *next_image,
                  int next_on_device, const double *u3, int h, const double *u8);
int create_loops(vohip_ctx *ctx, const double K[9], int rows, int cols, int colmajor, const vohip_frame_params *params,
                 int nseq, vohip_frame **out);

}  // namespace

extern "C" {

void vohip_frame_default_params(vohip_frame_params *p)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->klt_block = 21;            // main_demo.m:342-343
    p->klt_levels = 5;
    p->klt_max_iter = 20;
    p->klt_max_bidir_error = 6.0;
    p->detector = 0;              // Harris
    p->min_quality = 2e-4;
    p->filter_size = 5;
    p->border = 16;
    p->r_discard_redundant = 10.0;   // main_demo.m:72-75 (kitti)
    p->angle_threshold_deg = 1.0;
    p->min_depth = 1.0;
    p->max_depth = 80.0;
    p->refine_pose = 1;
    p->refine_max_iter = 100;
    p->fm_trials = 500;           // update_landmarks.m:13-14
    p->fm_threshold = 5e-2;
    p->fm_confidence = 99.0;
    p->cap_landmarks = 8192;
    p->cap_candidates = 8192;
    p->cap_corners = 16384;
    p->max_hypotheses = 15000;    // ransacLocalization.m:128
    p->slot_a = 6;
    p->slot_b = 7;
    p->slot_c = 5;
    p->is_BA = 0;
    p->min_contrast = 0.2;        // detectFASTFeatures' default
    p->metric_threshold = 1000.0; // detectSURFFeatures' default
    p->brisk_octaves = 4;         // detectBRISKFeatures' default
}

int vohip_frame_create(vohip_ctx *ctx, const double K[9], int rows, int cols, int colmajor,
                       const vohip_frame_params *params, vohip_frame **out)
{
    return create_loops(ctx, K, rows, cols, colmajor, params, 1, out);
}

}  // extern "C"

namespace {

// One loop, or the `nseq` loops of a group: state blocks and pinned blocks of one layout, back to back.
int create_loops(vohip_ctx *ctx, const double K[9], int rows, int cols, int colmajor, const vohip_frame_params *params,
                 int nseq, vohip_frame **out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, K && params && out, "NULL argument");
    VOHIP_REQUIRE(ctx, rows >= 8 && cols >= 8, "image too small");
    const vohip_frame_params &p = *params;
    VOHIP_REQUIRE(ctx, p.cap_landmarks >= 64 && p.cap_candidates >= 64 && p.cap_corners >= 64 && p.cap_corners <= 16384,
                  "capacities out of range");
    VOHIP_REQUIRE(ctx, p.fm_trials >= 1 && p.max_hypotheses >= 104, "bad RANSAC sizes");
    VOHIP_REQUIRE(ctx, p.slot_a != p.slot_b && p.slot_a != p.slot_c && p.slot_b != p.slot_c && p.slot_a >= 0 &&
                           p.slot_a < VOHIP_PYR_SLOTS && p.slot_b >= 0 && p.slot_b < VOHIP_PYR_SLOTS && p.slot_c >= 0 &&
                           p.slot_c < VOHIP_PYR_SLOTS,
                  "bad pyramid slots");
    VOHIP_REQUIRE(ctx, p.klt_block == 21 || p.klt_block == 31, "BlockSize 21 or 31");
    // (the detector's own options are checked here, not at the first frame that reaches it)
    VOHIP_REQUIRE(ctx, p.detector >= 0 && p.detector <= 4, "detector: 0 = Harris, 1 = MinEigen, 2 = FAST, 3 = SURF, 4 = BRISK");
    VOHIP_REQUIRE(ctx, p.min_quality >= 0.0 && p.min_quality <= 1.0, "MinQuality must lie in [0, 1]");
    VOHIP_REQUIRE(ctx, (p.detector != 2 && p.detector != 4) || (p.min_contrast > 0.0 && p.min_contrast < 1.0),
                  "MinContrast must lie in (0, 1)");
    VOHIP_REQUIRE(ctx, p.detector != 3 || p.metric_threshold >= 0.0, "MetricThreshold must be >= 0");
    VOHIP_REQUIRE(ctx, p.detector != 4 || (p.brisk_octaves >= 0 && p.brisk_octaves <= 4), "NumOctaves must be 0..4");
    VOHIP_REQUIRE(ctx, p.detector != 3 || (size_t)rows * cols <= ((size_t)1 << 24), "SURF: at most 2^24 pixels");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    vohip_frame *f = new (std::nothrow) vohip_frame;
    if (!f) return vohip_fail(ctx, VOHIP_ERR_HIP, "out of host memory");
    f->ctx = ctx;
    f->prm = p;
    std::memcpy(f->K, K, sizeof(f->K));
    f->rows = rows;
    f->cols = cols;
    f->colmajor = colmajor ? 1 : 0;
    f->capL = p.cap_landmarks;
    f->capC = p.cap_candidates;
    f->capK = p.cap_corners;
    f->hmax = (p.max_hypotheses + 7) / 8 * 8;
    f->capU = f->capL + f->capC + f->capK;
    f->slot_prev = p.slot_a;
    f->slot_cur = p.slot_b;
    f->slot_next = p.slot_c;
    f->nseq = nseq;
    f->seqs.resize(nseq);
    const size_t bytes = layout(f, nullptr);   // (a multiple of 256)
    f->pin_bytes = (PIN_U3 + sizeof(double) * (3 * (size_t)f->hmax + 8 * (size_t)p.fm_trials) + 255) / 256 * 256;
    f->Q.stride_state = f->Qn.stride_state = bytes;
    f->Q.stride_pin = f->Qn.stride_pin = f->pin_bytes;
    hipError_t e = hipMalloc(&f->block, bytes * nseq);
    if (e == hipSuccess) e = hipMemsetAsync(f->block, 0, bytes * nseq, ctx->stream);
    if (e == hipSuccess) e = hipHostMalloc(&f->pinned, f->pin_bytes * nseq, hipHostMallocDefault);
    if (e == hipSuccess && p.is_BA)   // the host's copy of S.P and S.X, written by every frame's last kernel
        e = hipHostMalloc((void **)&f->ba_stage, sizeof(double) * 6 * (size_t)f->capL, hipHostMallocDefault);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->trk, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_adv[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_adv[1], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_advC[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_advC[1], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_feed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_inl, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_advA[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_advA[1], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_tail, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_next, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_pyr, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_side, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&f->ev_img, hipEventDisableTiming);
    if (e != hipSuccess) {
        vohip_frame_destroy(f);
        return vohip_fail(ctx, VOHIP_ERR_HIP, "frame state allocation failed: %s", hipGetErrorString(e));
    }
    layout(f, (char *)f->block);
    if (std::getenv("VOHIP_FRAME_TRACE")) {   // per-stage GPU timeline of every frame, for tuning
        f->trace = true;
        for (int q = 0; q < 8; ++q) (void)hipEventCreate(&f->tr[q]);
        for (int q = 0; q < 4; ++q) (void)hipEventCreate(&f->tr_trk[q >> 1][q & 1]);
        for (int q = 0; q < 2; ++q) (void)hipEventCreate(&f->tr_det[q]);
    }
    *out = f;
    return VOHIP_OK;
}

}  // namespace

extern "C" {

int vohip_frame_destroy(vohip_frame *f)
{
    if (!f) return VOHIP_OK;
    if (f->ctx) (void)hipSetDevice(f->ctx->device);
    if (f->side) {
        (void)hipStreamSynchronize(f->side);
        (void)hipStreamDestroy(f->side);
    }
    if (f->trk) {
        (void)hipStreamSynchronize(f->trk);
        (void)hipStreamDestroy(f->trk);
    }
    for (int q = 0; q < 2; ++q)
        if (f->ev_adv[q]) (void)hipEventDestroy(f->ev_adv[q]);
    for (int q = 0; q < 2; ++q)
        if (f->ev_advC[q]) (void)hipEventDestroy(f->ev_advC[q]);
    if (f->ev_feed) (void)hipEventDestroy(f->ev_feed);
    if (f->ev_inl) (void)hipEventDestroy(f->ev_inl);
    for (int q = 0; q < 2; ++q)
        if (f->ev_advA[q]) (void)hipEventDestroy(f->ev_advA[q]);
    if (f->ev_tail) (void)hipEventDestroy(f->ev_tail);
    if (f->ev_next) (void)hipEventDestroy(f->ev_next);
    if (f->ctx) (void)hipStreamSynchronize(f->ctx->stream);
    if (f->ev_pyr) (void)hipEventDestroy(f->ev_pyr);
    for (int q = 0; q < 8; ++q)
        if (f->tr[q]) (void)hipEventDestroy(f->tr[q]);
    for (int q = 0; q < 4; ++q)
        if (f->tr_trk[q >> 1][q & 1]) (void)hipEventDestroy(f->tr_trk[q >> 1][q & 1]);
    for (int q = 0; q < 2; ++q)
        if (f->tr_det[q]) (void)hipEventDestroy(f->tr_det[q]);
    if (f->ev_side) (void)hipEventDestroy(f->ev_side);
    if (f->ev_img) (void)hipEventDestroy(f->ev_img);
    if (f->block) (void)hipFree(f->block);
    if (f->pinned) (void)hipHostFree(f->pinned);
    if (f->ba_stage) (void)hipHostFree(f->ba_stage);
    if (f->ctx) (void)detach_prepared_slots(f, false);   // (no slot of the context may keep pointing into the block)
    if (f->prep_block) (void)hipFree(f->prep_block);
    delete f;
    return VOHIP_OK;
}

// main_demo.m:268-303 (the parts the no-BA loop uses) and :342-344.
int vohip_frame_reset(vohip_frame *f, const uint8_t *image, int image_on_device, const double *X, const double *ids,
                      const double *P_rc, int n, const double T_W_C[16])
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, image && X && P_rc && T_W_C, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && n <= f->capL, "landmark count outside the capacity");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    f->ba_mirrored = false;
    VOHIP_HIP(ctx, hipStreamSynchronize(f->side));
    VOHIP_HIP(ctx, hipStreamSynchronize(f->trk));
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    f->ahead = f->ahead_launched = false;
    f->split_used[0] = f->split_used[1] = false;
    f->par = 1;   // the first image of a sequence counts as an even frame
    const uint8_t *d_img = image;
    if (!image_on_device) {
        VOHIP_HIP(ctx, hipMemcpyAsync(f->img[0], image, (size_t)f->rows * f->cols, hipMemcpyHostToDevice, s));
        d_img = f->img[0];
    }
    VOHIP_TRY(vohip_pyramid_build_dev(ctx, f->slot_prev, d_img, f->rows, f->cols, f->colmajor, f->prm.klt_levels));
    VOHIP_HIP(ctx, hipMemsetAsync(f->cnt, 0, sizeof(int) * N_COUNTERS, s));
    VOHIP_HIP(ctx, hipMemsetAsync(f->dv, 0, sizeof(double) * N_DOUBLES, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(f->L_rc, P_rc, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(f->L_X, X, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
    double *hd = pin_u3(f);   // staging
    std::memcpy(hd, T_W_C, sizeof(double) * 16);
    hd[16] = ids ? 0.0 : (double)(n + 1);
    if (ids) {
        VOHIP_HIP(ctx, hipMemcpyAsync(f->L_id, ids, sizeof(double) * n, hipMemcpyHostToDevice, s));
        double mx = 0.0;
        for (int i = 0; i < n; ++i) mx = ids[i] > mx ? ids[i] : mx;
        hd[16] = mx + 1.0;
    } else {
        // 1..n (main_demo.m:275)
        double *tmp = (double *)std::malloc(sizeof(double) * n);
        if (!tmp) return vohip_fail(ctx, VOHIP_ERR_HIP, "out of host memory");
        for (int i = 0; i < n; ++i) tmp[i] = (double)(i + 1);
        hipError_t e = hipMemcpyAsync(f->L_id, tmp, sizeof(double) * n, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        std::free(tmp);
        VOHIP_HIP(ctx, e);
    }
    VOHIP_HIP(ctx, hipMemcpyAsync(f->dv + D_T, hd, sizeof(double) * 16, hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(f->dv + D_IDX, hd + 16, sizeof(double), hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(f->dv + D_IDX_NEXT, hd + 16, sizeof(double), hipMemcpyHostToDevice, s));
    int *hc = pin_cnt(f);
    hc[0] = n;
    VOHIP_HIP(ctx, hipMemcpyAsync(f->cnt + N_L_END, hc, sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(init_rows_kernel, dim3(blocks256(n)), dim3(256), 0, s, f->L_rc, n, f->capU, f->U_xy[0], f->U_live[0],
                       f->srcL);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    f->n_L = n;
    f->n_C = 0;
    f->stage = 0;
    return VOHIP_OK;
}

int vohip_frame_step(vohip_frame *f, const uint8_t *image, int image_on_device, const double *u3, int h,
                     const double *u8, vohip_frame_pose *pose, vohip_frame_counts *counts)
{
    return vohip_frame_step_ahead(f, image, image_on_device, nullptr, 0, u3, h, u8, pose, counts);
}

int vohip_frame_step_ahead(vohip_frame *f, const uint8_t *image, int image_on_device, const uint8_t *next_image,
                           int next_on_device, const double *u3, int h, const double *u8, vohip_frame_pose *pose,
                           vohip_frame_counts *counts)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(f->ctx, pose && counts, "NULL argument");
    VOHIP_TRY(vohip_frame_begin(f, image, image_on_device, next_image, next_on_device, u3, h, u8));
    return vohip_frame_finish(f, pose, counts);
}

int vohip_frame_finish(vohip_frame *f, vohip_frame_pose *pose, vohip_frame_counts *counts)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, pose && counts, "NULL argument");
    VOHIP_REQUIRE(ctx, f->stage == 2, "no frame is enqueued (vohip_frame_begin)");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    f->stage = 1;
    return complete_frame(f, f->h_cur, pose, counts);
}

int vohip_frame_begin(vohip_frame *f, const uint8_t *image, int image_on_device, const uint8_t *next_image,
                      int next_on_device, const double *u3, int h, const double *u8)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, image && u3, "NULL argument");
    VOHIP_REQUIRE(ctx, f->stage != STAGE_FAILED, "an earlier call failed half-way: vohip_frame_reset starts over");
    VOHIP_REQUIRE(ctx, f->stage == 0, "a frame is still waiting for vohip_frame_retry or vohip_frame_finish");
    VOHIP_REQUIRE(ctx, h >= 1 && h <= f->hmax, "hypothesis count outside 1..max_hypotheses");
    VOHIP_REQUIRE(ctx, f->n_C == 0 || u8, "candidates exist: the fundamental-matrix draws are needed");
    const int st = begin_enqueue(f, image, image_on_device, next_image, next_on_device, u3, h, u8);
    // a failure between the first and the last enqueue leaves streams and state half-way through a frame
    if (st != VOHIP_OK) f->stage = STAGE_FAILED;
    return st;
}

}  // extern "C"

namespace {

int begin_enqueue(vohip_frame *f, const uint8_t *image, int image_on_device, const uint8_t *next_image,
                  int next_on_device, const double *u3, int h, const double *u8)
{
    vohip_ctx *ctx = f->ctx;
    f->t_entry = std::chrono::steady_clock::now();
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const vohip_frame_params &p = f->prm;
    const int par = f->par;
    f->have_map = f->n_C > 0;   // main_demo.m:406
    f->evaluated = 0;
    f->stage = 1;
    f->ahead_launched = next_image != nullptr;
    // The order of the enqueues follows the critical paths: what feeds the trackers first, then the
    // context's stream (localisation), the short kernels of the other streams while those run.
    if (f->trace) (void)hipEventRecord(f->tr[0], s);
    // the draws of both RANSAC stages stay in the pinned block: the kernels that turn them into index sets
    // read them there (a few KB each; a copy command would cost the host more than the reads cost the GPU)
    std::memcpy(pin_u3(f), u3, sizeof(double) * 3 * (size_t)h);
    if (f->have_map) std::memcpy(pin_u8(f), u8, sizeof(double) * 8 * (size_t)p.fm_trials);
    // images of a sequence prepared ahead are on the device already, with their pyramids and corner lists
    const size_t img_bytes = (size_t)f->rows * f->cols;
    auto prepared_of = [&](const uint8_t *img) {
        if (!img || f->prep_index.empty()) return -1;
        const auto it = f->prep_index.find(img);
        return it == f->prep_index.end() ? -1 : it->second;
    };
    const int prep_cur = prepared_of(image), prep_next = prepared_of(next_image);
    if (prep_cur >= 0) image_on_device = 1;
    if (prep_next >= 0) next_on_device = 1;
    const uint8_t *d_img = prep_cur >= 0 ? f->prep_imgs + img_bytes * prep_cur : (image_on_device ? image : f->img[par]);
    const uint8_t *d_next = next_image ? (prep_next >= 0 ? f->prep_imgs + img_bytes * prep_next
                                                         : (next_on_device ? next_image : f->img[par ^ 1]))
                                       : nullptr;
    if (!f->ahead) {
        // nothing was done ahead: this image, its pyramid, the trackers from the previous image
        if (!image_on_device) VOHIP_TRY(enqueue_image_transfer(f, image, f->img[par], s));
        VOHIP_HIP(ctx, hipEventRecord(f->ev_pyr, s));
        if (prep_cur >= 0)
            VOHIP_TRY(vohip_pyramid_attach(ctx, f->slot_cur, f->prep_pyr + f->prep_pyr_floats * prep_cur, f->rows, f->cols,
                                           f->colmajor, p.klt_levels));
        else
            VOHIP_TRY(vohip_pyramid_build_dev(ctx, f->slot_cur, d_img, f->rows, f->cols, f->colmajor, p.klt_levels));
        VOHIP_HIP(ctx, hipEventRecord(f->ev_feed, s));
        VOHIP_TRY(enqueue_trackers(f, par, f->slot_prev, f->slot_cur, f->n_L, f->have_map ? f->n_C : 0, false, f->ev_feed));
    }   // (else the previous call enqueued all of that, the trackers behind its own last kernel)
    if (!f->ahead)
        VOHIP_TRY(enqueue_detection(f, d_img, par, image_on_device ? nullptr : f->ev_pyr, f->prep_xy ? prep_cur : -1));
    const int h0 = h < 256 ? h : 256;   // the first triples come out of the selection kernel itself
    static const bool fused_off = std::getenv("VOHIP_FRAME_UNFUSED") != nullptr;   // (A/B measurements)
    if (f->ahead) VOHIP_HIP(ctx, hipStreamWaitEvent(s, f->ev_advA[par], 0));   // the kept landmarks' own tracker
    // the selection of the valid landmark tracks (main_demo.m:381, :387): in the launch that solves the
    // hypotheses when all of them have their triples made there (the usual 104), else a launch of its own
    const bool fused = h <= h0 && f->n_L <= 8192 && !fused_off;
    if (!fused) {
        hipLaunchKernelGGL(select_tracked_kernel, dim3(blocks256(f->n_L)), dim3(256), 0, s, f->U_xy[par], f->capU, f->U_live[par],
                           f->L_X, f->L_id, f->n_L, f->mv_rc, f->Xv, f->idv, f->rowv, f->cnt, f->dv, pin_u3(f), h0, f->samples3,
                           (const int *)nullptr, f->Q);
        VOHIP_HIP(ctx, hipGetLastError());
    }
    VOHIP_TRY(enqueue_localize(f, u3, h, h0, fused));
    if (f->trace) (void)hipEventRecord(f->tr[2], s);   // pose refined
    VOHIP_TRY(enqueue_side_filter(f, u8));
    f->next_pyramid_stands = next_image && prep_next >= 0;
    if (f->next_pyramid_stands) {   // built with the whole sequence: the third slot just shows it
        VOHIP_TRY(vohip_pyramid_attach(ctx, f->slot_next, f->prep_pyr + f->prep_pyr_floats * prep_next, f->rows, f->cols,
                                       f->colmajor, p.klt_levels));
    } else if (next_image) {
        // streams of their own, beside the trackers (which leave most of the GPU idle): the next image,
        // its pyramid (third slot) and its corners, so that the next call starts with its trackers
        stream_swap sw(ctx, f->trk);
        if (!next_on_device) {
            VOHIP_TRY(enqueue_image_transfer(f, next_image, f->img[par ^ 1], f->trk));
            VOHIP_HIP(ctx, hipEventRecord(f->ev_next, f->trk));   // (the corner detection waits for the transfer)
        }
        VOHIP_TRY(vohip_pyramid_build_dev(ctx, f->slot_next, d_next, f->rows, f->cols, f->colmajor, p.klt_levels));
        VOHIP_HIP(ctx, hipEventRecord(f->ev_img, f->trk));
    }
    if (next_image) VOHIP_TRY(enqueue_tracker_kept(f));
    VOHIP_TRY(enqueue_tail(f));
    if (next_image) VOHIP_TRY(enqueue_trackers_ahead(f));
    // (last: nothing waits for the following image's corners before the next call's final kernels)
    if (next_image)
        VOHIP_TRY(enqueue_detection(f, d_next, par ^ 1, next_on_device ? nullptr : f->ev_next, f->prep_xy ? prep_next : -1));
    f->h_cur = h;
    f->stage = 2;
    f->enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f->t_entry).count();
    return VOHIP_OK;
}

}  // namespace

extern "C" {

int vohip_frame_retry(vohip_frame *f, const double *u3, int h, vohip_frame_pose *pose, vohip_frame_counts *counts)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, u3 && pose && counts, "NULL argument");
    VOHIP_REQUIRE(ctx, f->stage == 1, "no frame is waiting for more hypotheses");
    VOHIP_REQUIRE(ctx, h > f->evaluated && h <= f->hmax, "hypothesis count must grow and stay within max_hypotheses");
    f->t_entry = std::chrono::steady_clock::now();
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    // (the first attempt's tracker of the kept landmarks may still read the lists the selection rewrites)
    if (f->ahead_launched) VOHIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, f->ev_advA[f->par ^ 1], 0));
    int st = enqueue_localize(f, u3, h, 0);
    // (the trackers enqueued ahead started from the lists of the first attempt: once more, from the final ones)
    if (st == VOHIP_OK && f->ahead_launched) st = enqueue_tracker_kept(f);
    if (st == VOHIP_OK) st = enqueue_tail(f);
    if (st == VOHIP_OK && f->ahead_launched) st = enqueue_trackers_ahead(f);
    if (st != VOHIP_OK) {
        f->stage = STAGE_FAILED;
        return st;
    }
    f->enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f->t_entry).count();
    return complete_frame(f, h, pose, counts);
}

int vohip_frame_sizes(vohip_frame *f, int *n_landmarks, int *n_candidates)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    if (n_landmarks) *n_landmarks = f->n_L;
    if (n_candidates) *n_candidates = f->n_C;
    return VOHIP_OK;
}

int vohip_frame_state(vohip_frame *f, double *P_rc, double *X, double *ids, double *C_rc, double *F_rc, double *T12)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, f->stage == 0, "the state is complete only between frames");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nL = f->n_L, nC = f->n_C;
    // The state is final once the frame's last kernel has run (vohip_frame_finish waited for it); the
    // copies go on the second stream, which is idle between frames, because the context's stream may
    // already hold the following frame's trackers (which touch none of these arrays).
    hipStream_t s = f->side;
    if (P_rc && nL) VOHIP_HIP(ctx, hipMemcpyAsync(P_rc, f->L_rc, sizeof(double) * 2 * nL, hipMemcpyDeviceToHost, s));
    if (X && nL) VOHIP_HIP(ctx, hipMemcpyAsync(X, f->L_X, sizeof(double) * 3 * nL, hipMemcpyDeviceToHost, s));
    if (ids && nL) VOHIP_HIP(ctx, hipMemcpyAsync(ids, f->L_id, sizeof(double) * nL, hipMemcpyDeviceToHost, s));
    if (C_rc && nC) VOHIP_HIP(ctx, hipMemcpyAsync(C_rc, f->C_rc, sizeof(double) * 2 * nC, hipMemcpyDeviceToHost, s));
    if (F_rc && nC) VOHIP_HIP(ctx, hipMemcpyAsync(F_rc, f->F_rc, sizeof(double) * 2 * nC, hipMemcpyDeviceToHost, s));
    if (T12 && nC) VOHIP_HIP(ctx, hipMemcpyAsync(T12, f->T12, sizeof(double) * 12 * nC, hipMemcpyDeviceToHost, s));
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    return VOHIP_OK;
}

int vohip_frame_apply_ba(vohip_frame *f, const double *X, int n, const double T_W_C[16])
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, f->stage == 0, "the state can be changed only between frames");
    VOHIP_REQUIRE(ctx, X && T_W_C, "NULL argument");
    VOHIP_REQUIRE(ctx, n == f->n_L, "X must hold one column per landmark of the state");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = f->side;   // (idle between frames; see vohip_frame_state)
    double *stage = pin_u3(f);  // the draws of the finished frame are no longer needed
    std::memcpy(stage, T_W_C, sizeof(double) * 16);
    if (n) VOHIP_HIP(ctx, hipMemcpyAsync(f->L_X, X, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(restamp_pose_kernel, dim3(8), dim3(256), 0, s, stage, f->cnt, f->dv, f->T12);
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    return VOHIP_OK;
}

int vohip_frame_bundle_adjust(vohip_frame *f, vohip_window *w, const double T_W_C[16], int *optimised, int *n_poses,
                              int *offsets, double *poses, int cap)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, w && T_W_C && optimised && n_poses, "NULL argument");
    VOHIP_REQUIRE(ctx, vohip_window_context(w) == ctx, "the window belongs to another context");
    VOHIP_REQUIRE(ctx, f->stage == 0, "the state is complete only between frames");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    if (!f->ba_stage) VOHIP_HIP(ctx, hipHostMalloc((void **)&f->ba_stage, sizeof(double) * 6 * (size_t)f->capL, hipHostMallocDefault));
    // The second stream is idle between frames; the context's own may already hold the following frame's
    // trackers, which touch neither the landmarks nor the solvers' buffers, so the window is optimised
    // beside them.
    hipStream_t s = f->side;
    stream_swap sw(ctx, s);
    const size_t n = f->n_L;
    double *P = f->ba_stage, *X = P + 2 * (size_t)f->capL, *ids = X + 3 * (size_t)f->capL;
    if (n && !f->ba_mirrored) {
        VOHIP_HIP(ctx, hipMemcpyAsync(P, f->L_rc, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, s));
        VOHIP_HIP(ctx, hipMemcpyAsync(X, f->L_X, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, s));
        VOHIP_HIP(ctx, hipMemcpyAsync(ids, f->L_id, sizeof(double) * n, hipMemcpyDeviceToHost, s));
        VOHIP_HIP(ctx, hipStreamSynchronize(s));
    }
    VOHIP_TRY(vohip_window_step(w, P, X, ids, (int)n, T_W_C, optimised, n_poses, offsets, poses, cap));
    if (!*optimised) return VOHIP_OK;
    // VO_bundle_adjust.m:53-57: the pose of the current keyframe is the entry with offset 0
    for (int k = 0; k < *n_poses; ++k)
        if (offsets[k] == 0) return vohip_frame_apply_ba(f, X, (int)n, poses + 16 * (size_t)k);
    return vohip_fail(ctx, VOHIP_ERR_INVALID_ARGUMENT, "the optimised window holds no pose for the current frame");
}

int vohip_frame_prepare(vohip_frame *f, const uint8_t *const *images, int n, int on_device)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, n >= 0 && n <= 65535 && (n == 0 || images), "bad argument");
    VOHIP_REQUIRE(ctx, f->stage == 0, "a sequence is prepared between frames");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    VOHIP_HIP(ctx, hipStreamSynchronize(f->side));
    VOHIP_HIP(ctx, hipStreamSynchronize(f->trk));
    f->prep_index.clear();
    VOHIP_TRY(detach_prepared_slots(f, true));   // (a frame in progress keeps the pyramids it still tracks from)
    if (f->prep_block) (void)hipFree(f->prep_block);
    f->prep_block = nullptr;
    f->prep_bytes = 0;
    f->prep_n = 0;
    f->prep_xy = nullptr;
    if (n == 0) return VOHIP_OK;
    const vohip_frame_params &p = f->prm;
    const size_t img_bytes = (size_t)f->rows * f->cols;
    const size_t pyr_floats = vohip_pyramid_floats(f->rows, f->cols, f->colmajor, p.klt_levels);
    const bool corners = f->capK <= 16384;
    auto round256 = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t b_img = round256(img_bytes * n), b_pyr = round256(sizeof(float) * pyr_floats * n),
                 b_xy = corners ? round256(sizeof(float) * 2 * (size_t)f->capK * n) : 0,
                 b_metric = corners ? round256(sizeof(float) * (size_t)f->capK * n) : 0,
                 b_cnt = corners ? round256(sizeof(int) * (size_t)n) : 0;
    VOHIP_HIP(ctx, hipMalloc(&f->prep_block, b_img + b_pyr + b_xy + b_metric + b_cnt));
    f->prep_bytes = b_img + b_pyr + b_xy + b_metric + b_cnt;
    char *q = (char *)f->prep_block;
    f->prep_imgs = (uint8_t *)q;
    q += b_img;
    f->prep_pyr = (float *)q;
    q += b_pyr;
    float *xy = (float *)q;
    q += b_xy;
    f->prep_metric = (float *)q;
    q += b_metric;
    f->prep_counts = (int *)q;
    f->prep_pyr_floats = pyr_floats;
    for (int k = 0; k < n; ++k) {
        VOHIP_REQUIRE(ctx, images[k], "NULL image");
        VOHIP_HIP(ctx, hipMemcpyAsync(f->prep_imgs + img_bytes * k, images[k], img_bytes,
                                      on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    }
    VOHIP_TRY(vohip_pyramid_batch_dev(ctx, f->prep_imgs, n, f->rows, f->cols, f->colmajor, p.klt_levels, f->prep_pyr));
    if (corners) {
        const int chunk = 64;   // (bounds the detector's scratch: records of 64 images at a time)
        for (int c = 0; c < n; c += chunk) {
            const int m = n - c < chunk ? n - c : chunk;
            VOHIP_TRY(vohip_detect_corners_batch_enqueue(ctx, f->prep_imgs + img_bytes * c, m, f->rows, f->cols, f->colmajor,
                                                         p.min_quality, p.filter_size, p.border, p.detector,
                                                         xy + 2 * (size_t)f->capK * c, f->prep_metric + (size_t)f->capK * c, f->capK,
                                                         f->prep_counts + c, detector_option(p), p.brisk_octaves));
        }
    }
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    if (corners) f->prep_xy = xy;
    f->prep_n = n;
    for (int k = 0; k < n; ++k) f->prep_index[images[k]] = k;
    return VOHIP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// A group of loops advanced in lockstep: one launch sequence per frame serves every sequence of the
// group (main_eval.m:22-23 runs independent sequences one after the other; here S of them share the
// launches, their latencies and the host's enqueue work, and the tracker launch fills the GPU).
// ------------------------------------------------------------------------------------------
namespace {

// host-side view of one sequence's state block
template <class T>
inline T *of_seq(vohip_frame *f, T *p, int seq)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(p) + f->Q.stride_state * (size_t)seq);
}
template <class T>
inline T *prep_of_seq(vohip_frame *f, T *p, int seq)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(p) + f->Q.stride_prep * (size_t)seq);
}

int group_check(vohip_frame *f, int seq)
{
    if (!f) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(f->ctx, f->is_group, "not a group of loops");
    VOHIP_REQUIRE(f->ctx, seq >= 0 && seq < f->nseq, "sequence index out of range");
    return VOHIP_OK;
}

// The frame with prepared index k, for the sequences of f->Q (look-ahead for those of f->Qn): the launch
// sequence of begin_enqueue for a prepared, stored sequence.
int group_enqueue(vohip_frame *f, int k, int h)
{
    vohip_ctx *ctx = f->ctx;
    f->t_entry = std::chrono::steady_clock::now();
    hipStream_t s = ctx->stream;
    const vohip_frame_params &p = f->prm;
    const int par = f->par;
    f->evaluated = 0;
    f->ahead_launched = f->Qn.mask != 0;
    auto pyramid = [&](int index) { return f->prep_pyr + f->prep_pyr_floats * (size_t)index; };
    VOHIP_TRY(vohip_pyramid_attach(ctx, f->slot_cur, pyramid(k), f->rows, f->cols, f->colmajor, p.klt_levels));
    if (!f->ahead) {
        // the first frame of a run: its trackers from the image the loops were reset on (index k - 1), counts
        // from device memory (every sequence has its own), and the hand-over of its corner lists
        VOHIP_TRY(vohip_pyramid_attach(ctx, f->slot_prev, pyramid(k - 1), f->rows, f->cols, f->colmajor, p.klt_levels));
        VOHIP_HIP(ctx, hipEventRecord(f->ev_feed, s));
        VOHIP_TRY(enqueue_trackers(f, par, f->slot_prev, f->slot_cur, f->n_L, f->have_map ? f->n_C : 0, true, f->ev_feed,
                                   &f->Q, false));
        VOHIP_TRY(enqueue_detection(f, nullptr, par, nullptr, k, &f->Q));
    } else {
        VOHIP_HIP(ctx, hipStreamWaitEvent(s, f->ev_advA[par], 0));   // the kept landmarks' own tracker
    }
    const int h0 = h < 256 ? h : 256;
    const bool fused = h <= h0 && f->n_L <= 8192;
    if (!fused) {
        hipLaunchKernelGGL(select_tracked_kernel, dim3(blocks256(f->n_L), ny_of(f->Q)), dim3(256), 0, s, f->U_xy[par], f->capU,
                           f->U_live[par], f->L_X, f->L_id, f->n_L, f->mv_rc, f->Xv, f->idv, f->rowv, f->cnt, f->dv, pin_u3(f),
                           h0, f->samples3, (const int *)(f->cnt + N_L_END), f->Q);
        VOHIP_HIP(ctx, hipGetLastError());
    }
    VOHIP_TRY(enqueue_localize(f, nullptr, h, h0, fused));
    VOHIP_TRY(enqueue_side_filter(f, nullptr));
    f->next_pyramid_stands = true;
    if (f->Qn.mask) {
        VOHIP_TRY(vohip_pyramid_attach(ctx, f->slot_next, pyramid(k + 1), f->rows, f->cols, f->colmajor, p.klt_levels));
        VOHIP_TRY(enqueue_tracker_kept(f));
    }
    VOHIP_TRY(enqueue_tail(f));
    if (f->Qn.mask) {
        VOHIP_TRY(enqueue_trackers_ahead(f));
        VOHIP_TRY(enqueue_detection(f, nullptr, par ^ 1, nullptr, k + 1, &f->Qn));
    }
    f->enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f->t_entry).count();
    return VOHIP_OK;
}

// What sequence `seq` reports for the frame just run (its summary is in its pinned block).  Returns the
// sequence's status.
int group_collect(vohip_frame *f, int seq, int h, vohip_frame_pose *pose, vohip_frame_counts *out, double wait_us)
{
    vohip_ctx *ctx = f->ctx;
    vohip_frame::seq_state &q = f->seqs[seq];
    const int *hc = pin_cnt(f, seq);
    const double *hd = pin_dv(f, seq);
    std::memset(pose, 0, sizeof(*pose));
    std::memset(out, 0, sizeof(*out));
    pose->n_tracked = hc[N_V];
    pose->n_iter = hc[R_ITER];
    pose->found = hc[R_FOUND];
    pose->exhausted = hc[R_EXHAUSTED];
    pose->n_inliers = hc[N_IN];
    pose->max_num_inliers = hc[R_MAX_INL];
    pose->evaluated = h;
    pose->num_iterations = hd[D_NUM_ITER];
    std::memcpy(pose->T_W_C, hd + D_T, sizeof(double) * 16);
    std::memcpy(pose->Rt, hd + D_RT, sizeof(double) * 12);
    std::memcpy(pose->refine_stats, hd + D_STATS, sizeof(double) * 3);
    q.evaluated = h;
    if (hc[N_V] >= 3 && hc[R_EXHAUSTED] && h < f->hmax) {
        q.stage = 1;   // vohip_group_retry goes on from here
        return vohip_fail(ctx, VOHIP_SAMPLES_EXHAUSTED, "sequence %d: the adaptive bound (%.0f) exceeds the %d hypotheses supplied",
                          seq, hd[D_NUM_ITER], h);
    }
    const int par = f->par;
    out->n_landmarks = hc[N_L_END];
    out->n_candidates = hc[N_C_END];
    out->num_new = hc[NUM_NEW];
    out->n_detected = hc[N_DET + par];
    out->n_kept = hc[N_KEPT];
    out->n_candidates_tracked = f->have_map ? hc[N_CV] : 0;
    out->n_candidates_inliers = f->have_map ? hc[N_CI] : 0;
    out->fm_trials_run = f->have_map ? hc[F_TRIALS] : 0;
    out->overflow = hc[OVERFLOW];
    out->host_enqueue_us = f->enqueue_us;
    out->host_wait_us = wait_us;
    q.n_L = hc[N_L_END];
    q.n_C = hc[N_C_END];
    q.stage = 0;
    // (the same ends of a run as complete_frame's)
    const bool lost = hc[N_V] < 3 || !hc[R_FOUND] || (f->have_map && hc[N_CV] < 8);
    const bool over = hc[N_DET + par] > f->capK || hc[OVERFLOW] != 0;
    if (lost || over) q.stage = STAGE_FAILED;
    if (hc[N_V] < 3) return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "sequence %d: only %d landmarks were tracked", seq, hc[N_V]);
    if (!hc[R_FOUND])
        return vohip_fail(ctx, VOHIP_NO_MODEL, "sequence %d: no hypothesis reached 30 inliers in %d iterations", seq, hc[R_ITER]);
    if (f->have_map && hc[N_CV] < 8)
        return vohip_fail(ctx, VOHIP_TOO_FEW_POINTS, "sequence %d: the 8-point algorithm needs 8 candidate tracks, got %d", seq,
                          hc[N_CV]);
    if (hc[N_DET + par] > f->capK)
        return vohip_fail(ctx, VOHIP_CAPACITY_EXCEEDED, "sequence %d: %d corners found, capacity %d", seq, hc[N_DET + par], f->capK);
    if (hc[OVERFLOW])
        return vohip_fail(ctx, VOHIP_CAPACITY_EXCEEDED, "sequence %d: landmark or candidate capacity exceeded (flags %d)", seq,
                          hc[OVERFLOW]);
    return VOHIP_OK;
}

}  // namespace

struct vohip_group {
    vohip_frame *f = nullptr;
};

extern "C" {

int vohip_group_create(vohip_ctx *ctx, const double K[9], int rows, int cols, int colmajor,
                       const vohip_frame_params *params, int n_sequences, int max_frames, vohip_group **out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, out && params, "NULL argument");
    VOHIP_REQUIRE(ctx, n_sequences >= 1 && n_sequences <= VOHIP_MAX_SEQ, "1..64 sequences per group");
    VOHIP_REQUIRE(ctx, max_frames >= 2 && max_frames <= 65535, "a sequence holds 2..65535 images");
    VOHIP_REQUIRE(ctx, !params->is_BA, "bundle adjustment is not available in a group of loops");
    VOHIP_REQUIRE(ctx, params->cap_corners <= 16384, "corner capacity of prepared sequences is at most 16384");
    vohip_frame *f = nullptr;
    VOHIP_TRY(create_loops(ctx, K, rows, cols, colmajor, params, n_sequences, &f));
    f->is_group = true;
    f->g_max_frames = max_frames;
    // the prepared sequences: per sequence [images | pyramids | corner (x, y) | corner metric | corner counts]
    const vohip_frame_params &p = f->prm;
    const size_t img_bytes = (size_t)rows * cols;
    const size_t pyr_floats = vohip_pyramid_floats(rows, cols, f->colmajor, p.klt_levels);
    auto round256 = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t b_img = round256(img_bytes * max_frames), b_pyr = round256(sizeof(float) * pyr_floats * max_frames),
                 b_xy = round256(sizeof(float) * 2 * (size_t)f->capK * max_frames),
                 b_metric = round256(sizeof(float) * (size_t)f->capK * max_frames), b_cnt = round256(sizeof(int) * (size_t)max_frames);
    const size_t stride = b_img + b_pyr + b_xy + b_metric + b_cnt;
    const hipError_t e = hipMalloc(&f->prep_block, stride * n_sequences);
    if (e != hipSuccess) {
        vohip_frame_destroy(f);
        return vohip_fail(ctx, VOHIP_ERR_HIP, "%.1f MB for %d prepared sequences of %d frames: %s", stride * n_sequences / 1048576.0,
                          n_sequences, max_frames, hipGetErrorString(e));
    }
    f->prep_bytes = stride * n_sequences;
    char *q = (char *)f->prep_block;
    f->prep_imgs = (uint8_t *)q;
    q += b_img;
    f->prep_pyr = (float *)q;
    q += b_pyr;
    f->prep_xy = (float *)q;
    q += b_xy;
    f->prep_metric = (float *)q;
    q += b_metric;
    f->prep_counts = (int *)q;
    f->prep_pyr_floats = pyr_floats;
    f->prep_n = max_frames;
    f->Q.stride_prep = f->Qn.stride_prep = stride;
    vohip_group *g = new (std::nothrow) vohip_group;
    if (!g) {
        vohip_frame_destroy(f);
        return vohip_fail(ctx, VOHIP_ERR_HIP, "out of host memory");
    }
    g->f = f;
    *out = g;
    return VOHIP_OK;
}

int vohip_group_destroy(vohip_group *g)
{
    if (!g) return VOHIP_OK;
    const int st = vohip_frame_destroy(g->f);
    delete g;
    return st;
}

int vohip_group_prepare(vohip_group *g, int seq, const uint8_t *const *images, int n, int on_device, int detector,
                        double min_quality, double min_contrast)
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, seq));
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, images && n >= 2 && n <= f->g_max_frames, "a sequence holds 2..max_frames images");
    VOHIP_REQUIRE(ctx, !f->g_inflight, "a frame is in flight (vohip_group_finish)");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const vohip_frame_params &p = f->prm;
    if (detector < 0) {
        detector = p.detector;
        min_quality = p.min_quality;
        min_contrast = detector_option(p);
    }
    VOHIP_REQUIRE(ctx, detector >= 0 && detector <= 4, "detector: 0 = Harris, 1 = MinEigen, 2 = FAST, 3 = SURF, 4 = BRISK");
    const size_t img_bytes = (size_t)f->rows * f->cols;
    uint8_t *imgs = prep_of_seq(f, f->prep_imgs, seq);
    for (int k = 0; k < n; ++k) {
        VOHIP_REQUIRE(ctx, images[k], "NULL image");
        VOHIP_HIP(ctx, hipMemcpyAsync(imgs + img_bytes * k, images[k], img_bytes,
                                      on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    }
    VOHIP_TRY(vohip_pyramid_batch_dev(ctx, imgs, n, f->rows, f->cols, f->colmajor, p.klt_levels, prep_of_seq(f, f->prep_pyr, seq)));
    float *xy = prep_of_seq(f, f->prep_xy, seq), *metric = prep_of_seq(f, f->prep_metric, seq);
    int *counts = prep_of_seq(f, f->prep_counts, seq);
    const int chunk = 64;   // (bounds the detector's scratch: records of 64 images at a time)
    for (int c = 0; c < n; c += chunk) {
        const int m = n - c < chunk ? n - c : chunk;
        VOHIP_TRY(vohip_detect_corners_batch_enqueue(ctx, imgs + img_bytes * c, m, f->rows, f->cols, f->colmajor, min_quality,
                                                     p.filter_size, p.border, detector, xy + 2 * (size_t)f->capK * c,
                                                     metric + (size_t)f->capK * c, f->capK, counts + c, min_contrast, p.brisk_octaves));
    }
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    f->seqs[seq].n_frames = n;
    f->seqs[seq].reset = false;
    return VOHIP_OK;
}

int vohip_group_prepare_like(vohip_group *g, int seq, int from_seq, int detector, double min_quality, double min_contrast)
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, seq));
    VOHIP_TRY(group_check(f, from_seq));
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, seq != from_seq && f->seqs[from_seq].n_frames >= 2, "the other sequence must be a prepared one");
    VOHIP_REQUIRE(ctx, !f->g_inflight, "a frame is in flight (vohip_group_finish)");
    VOHIP_REQUIRE(ctx, detector >= -2 && detector <= 4, "detector: -2 = the other sequence's corner lists, -1 = the group's, 0..4");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const vohip_frame_params &p = f->prm;
    const int n = f->seqs[from_seq].n_frames;
    const size_t img_bytes = (size_t)f->rows * f->cols;
    uint8_t *imgs = prep_of_seq(f, f->prep_imgs, seq);
    VOHIP_HIP(ctx, hipMemcpyAsync(imgs, prep_of_seq(f, f->prep_imgs, from_seq), img_bytes * n, hipMemcpyDeviceToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(prep_of_seq(f, f->prep_pyr, seq), prep_of_seq(f, f->prep_pyr, from_seq),
                                  sizeof(float) * f->prep_pyr_floats * n, hipMemcpyDeviceToDevice, s));
    float *xy = prep_of_seq(f, f->prep_xy, seq), *metric = prep_of_seq(f, f->prep_metric, seq);
    int *counts = prep_of_seq(f, f->prep_counts, seq);
    if (detector == -2) {
        VOHIP_HIP(ctx, hipMemcpyAsync(xy, prep_of_seq(f, f->prep_xy, from_seq), sizeof(float) * 2 * (size_t)f->capK * n,
                                      hipMemcpyDeviceToDevice, s));
        VOHIP_HIP(ctx, hipMemcpyAsync(metric, prep_of_seq(f, f->prep_metric, from_seq), sizeof(float) * (size_t)f->capK * n,
                                      hipMemcpyDeviceToDevice, s));
        VOHIP_HIP(ctx, hipMemcpyAsync(counts, prep_of_seq(f, f->prep_counts, from_seq), sizeof(int) * (size_t)n,
                                      hipMemcpyDeviceToDevice, s));
    } else {
        if (detector < 0) {
            detector = p.detector;
            min_quality = p.min_quality;
            min_contrast = detector_option(p);
        }
        const int chunk = 64;
        for (int c = 0; c < n; c += chunk) {
            const int m = n - c < chunk ? n - c : chunk;
            VOHIP_TRY(vohip_detect_corners_batch_enqueue(ctx, imgs + img_bytes * c, m, f->rows, f->cols, f->colmajor, min_quality,
                                                         p.filter_size, p.border, detector, xy + 2 * (size_t)f->capK * c,
                                                         metric + (size_t)f->capK * c, f->capK, counts + c, min_contrast, p.brisk_octaves));
        }
    }
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    f->seqs[seq].n_frames = n;
    f->seqs[seq].reset = false;
    return VOHIP_OK;
}

int vohip_group_reset(vohip_group *g, int seq, const double *X, const double *ids, const double *P_rc, int n,
                      const double T_W_C[16])
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, seq));
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, X && P_rc && T_W_C, "NULL argument");
    VOHIP_REQUIRE(ctx, n >= 1 && n <= f->capL, "landmark count outside the capacity");
    VOHIP_REQUIRE(ctx, f->seqs[seq].n_frames >= 2, "the sequence has not been prepared (vohip_group_prepare)");
    VOHIP_REQUIRE(ctx, !f->g_inflight, "a frame is in flight (vohip_group_finish)");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    VOHIP_HIP(ctx, hipStreamSynchronize(f->side));
    VOHIP_HIP(ctx, hipStreamSynchronize(f->trk));
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    if (f->g_frame > 0 || f->ahead) {
        // a new run of the group starts: only sequences reset from here on take part in it
        for (auto &q : f->seqs) q.reset = false;
        f->g_frame = 0;
    }
    f->ahead = f->ahead_launched = false;
    f->split_used[0] = f->split_used[1] = false;
    f->par = 1;   // the image the loops are reset on counts as an even frame
    f->stage = 0;
    int *cnt = of_seq(f, f->cnt, seq);
    double *dv = of_seq(f, f->dv, seq);
    VOHIP_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(int) * N_COUNTERS, s));
    VOHIP_HIP(ctx, hipMemsetAsync(dv, 0, sizeof(double) * N_DOUBLES, s));
    std::vector<double> stage(17 + (size_t)n);
    std::memcpy(stage.data(), T_W_C, sizeof(double) * 16);
    double next_id = (double)(n + 1);
    for (int i = 0; i < n; ++i) stage[17 + i] = ids ? ids[i] : (double)(i + 1);   // 1..n (main_demo.m:275)
    if (ids) {
        double mx = 0.0;
        for (int i = 0; i < n; ++i) mx = ids[i] > mx ? ids[i] : mx;
        next_id = mx + 1.0;
    }
    stage[16] = next_id;
    VOHIP_HIP(ctx, hipMemcpyAsync(of_seq(f, f->L_rc, seq), P_rc, sizeof(double) * 2 * n, hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(of_seq(f, f->L_X, seq), X, sizeof(double) * 3 * n, hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(of_seq(f, f->L_id, seq), stage.data() + 17, sizeof(double) * n, hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(dv + D_T, stage.data(), sizeof(double) * 16, hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(dv + D_IDX, stage.data() + 16, sizeof(double), hipMemcpyHostToDevice, s));
    VOHIP_HIP(ctx, hipMemcpyAsync(dv + D_IDX_NEXT, stage.data() + 16, sizeof(double), hipMemcpyHostToDevice, s));
    const int n_host = n;
    VOHIP_HIP(ctx, hipMemcpyAsync(cnt + N_L_END, &n_host, sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(init_rows_kernel, dim3(blocks256(n)), dim3(256), 0, s, (const double *)of_seq(f, f->L_rc, seq), n, f->capU,
                       of_seq(f, f->U_xy[0], seq), of_seq(f, f->U_live[0], seq), of_seq(f, f->srcL, seq));
    VOHIP_HIP(ctx, hipGetLastError());
    VOHIP_HIP(ctx, hipStreamSynchronize(s));   // (the staging vector is released on return)
    vohip_frame::seq_state &q = f->seqs[seq];
    q.n_L = n;
    q.n_C = 0;
    q.evaluated = 0;
    q.stage = 0;
    q.reset = true;
    return VOHIP_OK;
}

int vohip_group_begin(vohip_group *g, const double *u3, int h, const double *u8)
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, 0));
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, u3, "NULL argument");
    VOHIP_REQUIRE(ctx, !f->g_inflight, "a frame is in flight (vohip_group_finish)");
    VOHIP_REQUIRE(ctx, f->stage != STAGE_FAILED, "an earlier call failed half-way: vohip_group_reset starts over");
    VOHIP_REQUIRE(ctx, h >= 1 && h <= f->hmax, "hypothesis count outside 1..max_hypotheses");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    const int k = f->g_frame + 1;
    vohip_seqs Q = f->Q, Qn = f->Qn;
    Q.mask = Qn.mask = 0;
    int nL = 0, nC = 0;
    bool some_map = false, all_map = true;
    for (int s = 0; s < f->nseq; ++s) {
        const vohip_frame::seq_state &q = f->seqs[s];
        VOHIP_REQUIRE(ctx, q.stage != 1, "a sequence waits for vohip_group_retry");
        if (!q.reset || q.stage != 0 || k >= q.n_frames) continue;
        Q.mask |= 1ull << s;
        if (k + 1 < q.n_frames) Qn.mask |= 1ull << s;
        nL = q.n_L > nL ? q.n_L : nL;
        nC = q.n_C > nC ? q.n_C : nC;
        some_map = some_map || q.n_C > 0;
        all_map = all_map && q.n_C > 0;
    }
    VOHIP_REQUIRE(ctx, Q.mask != 0, "no sequence of the group has a frame left");
    // update_landmarks runs when a loop holds candidates (main_demo.m:406): in step, either every sequence does
    // (every frame but a run's first) or none
    VOHIP_REQUIRE(ctx, some_map == all_map, "the sequences of a group must all hold candidates, or none of them");
    VOHIP_REQUIRE(ctx, !some_map || u8, "candidates exist: the fundamental-matrix draws are needed");
    if (f->g_frame > 0) {   // the frame before is complete: its pyramid is the next "previous" (main_demo.m:417-424)
        f->par ^= 1;
        f->ahead = true;
    }
    f->Q = Q;
    f->Qn = Qn;
    f->n_L = nL;
    f->n_C = nC;
    f->have_map = some_map;
    // the draws of both RANSAC stages: each sequence reads its own from its pinned block
    for (int s = 0; s < f->nseq; ++s) {
        if (!(Q.mask >> s & 1)) continue;
        std::memcpy(pin_u3(f, s), u3 + 3 * (size_t)h * s, sizeof(double) * 3 * (size_t)h);
        if (some_map) std::memcpy(pin_u8(f, s), u8 + 8 * (size_t)f->prm.fm_trials * s, sizeof(double) * 8 * (size_t)f->prm.fm_trials);
    }
    const int st = group_enqueue(f, k, h);
    if (st != VOHIP_OK) {
        f->stage = STAGE_FAILED;
        return st;
    }
    f->g_h = h;
    f->g_inflight = true;
    return VOHIP_OK;
}

int vohip_group_finish(vohip_group *g, vohip_frame_pose *poses, vohip_frame_counts *counts, int *status)
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, 0));
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, poses && counts && status, "NULL argument");
    VOHIP_REQUIRE(ctx, f->g_inflight, "no frame is enqueued (vohip_group_begin)");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    const auto t_wait = std::chrono::steady_clock::now();
    if (const int st = fetch_summary(f)) {
        f->stage = STAGE_FAILED;
        return st;
    }
    const double wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_wait).count();
    f->g_inflight = false;
    f->g_frame += 1;
    for (int s = 0; s < f->nseq; ++s) {
        if (f->Q.mask >> s & 1) {
            status[s] = group_collect(f, s, f->g_h, poses + s, counts + s, wait_us);
        } else {
            std::memset(poses + s, 0, sizeof(*poses));
            std::memset(counts + s, 0, sizeof(*counts));
            status[s] = VOHIP_SEQUENCE_IDLE;
        }
    }
    return VOHIP_OK;
}

int vohip_group_retry(vohip_group *g, int seq, const double *u3, int h, vohip_frame_pose *pose, vohip_frame_counts *counts)
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, seq));
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, u3 && pose && counts, "NULL argument");
    vohip_frame::seq_state &q = f->seqs[seq];
    VOHIP_REQUIRE(ctx, !f->g_inflight && q.stage == 1, "the sequence is not waiting for more hypotheses");
    VOHIP_REQUIRE(ctx, h > q.evaluated && h <= f->hmax, "hypothesis count must grow and stay within max_hypotheses");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    f->t_entry = std::chrono::steady_clock::now();
    // the launches of the frame once more from the hypotheses onwards, for this sequence alone
    const vohip_seqs Q = f->Q, Qn = f->Qn;
    f->Q.mask = 1ull << seq;
    f->Qn.mask = Qn.mask & (1ull << seq);
    f->evaluated = q.evaluated;
    std::memcpy(pin_u3(f, seq) + 3 * (size_t)q.evaluated, u3 + 3 * (size_t)q.evaluated,
                sizeof(double) * 3 * (size_t)(h - q.evaluated));
    int st = VOHIP_OK;
    // (the first attempt's tracker of the kept landmarks may still read the lists the selection rewrites)
    if (f->Qn.mask && hipStreamWaitEvent(ctx->stream, f->ev_advA[f->par ^ 1], 0) != hipSuccess)
        st = vohip_fail(ctx, VOHIP_ERR_HIP, "hipStreamWaitEvent failed");
    if (st == VOHIP_OK) st = enqueue_localize(f, nullptr, h, 0);
    if (st == VOHIP_OK && f->Qn.mask) st = enqueue_tracker_kept(f);
    if (st == VOHIP_OK) st = enqueue_tail(f);
    if (st == VOHIP_OK && f->Qn.mask) st = enqueue_trackers_ahead(f);
    f->Q = Q;
    f->Qn = Qn;
    if (st != VOHIP_OK) {
        f->stage = STAGE_FAILED;
        return st;
    }
    f->enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - f->t_entry).count();
    const auto t_wait = std::chrono::steady_clock::now();
    if (const int e = fetch_summary(f)) {
        f->stage = STAGE_FAILED;
        return e;
    }
    const double wait_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_wait).count();
    return group_collect(f, seq, h, pose, counts, wait_us);
}

int vohip_group_sizes(vohip_group *g, int seq, int *n_landmarks, int *n_candidates, int *frames_done)
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, seq));
    if (n_landmarks) *n_landmarks = f->seqs[seq].n_L;
    if (n_candidates) *n_candidates = f->seqs[seq].n_C;
    if (frames_done) *frames_done = f->g_frame;
    return VOHIP_OK;
}

int vohip_group_state(vohip_group *g, int seq, double *P_rc, double *X, double *ids, double *C_rc, double *F_rc, double *T12)
{
    vohip_frame *f = g ? g->f : nullptr;
    VOHIP_TRY(group_check(f, seq));
    vohip_ctx *ctx = f->ctx;
    VOHIP_REQUIRE(ctx, !f->g_inflight && f->seqs[seq].stage != 1, "the state is complete only between frames");
    VOHIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nL = f->seqs[seq].n_L, nC = f->seqs[seq].n_C;
    hipStream_t s = f->side;   // (idle between frames; see vohip_frame_state)
    if (P_rc && nL) VOHIP_HIP(ctx, hipMemcpyAsync(P_rc, of_seq(f, f->L_rc, seq), sizeof(double) * 2 * nL, hipMemcpyDeviceToHost, s));
    if (X && nL) VOHIP_HIP(ctx, hipMemcpyAsync(X, of_seq(f, f->L_X, seq), sizeof(double) * 3 * nL, hipMemcpyDeviceToHost, s));
    if (ids && nL) VOHIP_HIP(ctx, hipMemcpyAsync(ids, of_seq(f, f->L_id, seq), sizeof(double) * nL, hipMemcpyDeviceToHost, s));
    if (C_rc && nC) VOHIP_HIP(ctx, hipMemcpyAsync(C_rc, of_seq(f, f->C_rc, seq), sizeof(double) * 2 * nC, hipMemcpyDeviceToHost, s));
    if (F_rc && nC) VOHIP_HIP(ctx, hipMemcpyAsync(F_rc, of_seq(f, f->F_rc, seq), sizeof(double) * 2 * nC, hipMemcpyDeviceToHost, s));
    if (T12 && nC) VOHIP_HIP(ctx, hipMemcpyAsync(T12, of_seq(f, f->T12, seq), sizeof(double) * 12 * nC, hipMemcpyDeviceToHost, s));
    VOHIP_HIP(ctx, hipStreamSynchronize(s));
    return VOHIP_OK;
}

}  // extern "C"
