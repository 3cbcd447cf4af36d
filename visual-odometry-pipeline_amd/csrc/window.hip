// The sliding window of the bundle adjustment, kept on the host.
//
// VO_bundle_adjust.m:1-98 is bookkeeping around two solver calls: which frames are keyframes, which
// landmarks the window still holds, global landmark ids -> columns of the state vector, then runBA
// (VO_bundle_adjust.m:51) and T_refinement (:80) and the write-back of what they return.  The
// bookkeeping is a few thousand integer look-ups per window and stays on the CPU, as in the
// reference; this file is that bookkeeping in C++ (the struct B of main_demo.m:288-303 as
// vohip_window), the two solvers are the library's own GPU routines (vohip_ba_run,
// vohip_pose_refine_batch).  Landmark ids only ever grow (update_landmarks.m:83-87 hands out
// B.new_idx, B.new_idx + 1, ...), so every id row is ascending and the reference's
// find(row == column) look-ups are binary searches here.
#include "vohip_internal.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <deque>
#include <new>
#include <vector>

struct vohip_window {
    vohip_ctx *ctx = nullptr;
    vohip_window_params prm;
    double K[9];
    int num_key = 0;       // B.num_key
    int count_frame = 0;   // B.count_frame
    double new_idx = 1.0;  // B.new_idx: the next id update_landmarks.m:83-87 hands out
    std::vector<double> tau;               // B.tau, 6 per keyframe
    std::vector<double> lm;                // B.landmarks, 4 per column (X; id)
    std::vector<double> lm_id;             // its id row
    std::deque<std::vector<double>> obs;   // B.observation: [n; n (row, col) pairs; n ids] per keyframe
    struct normal_frame {                  // B.normal_frame_refine{k} = {T_W_C, flipud(S.P), S.X(4,:)}
        double T[16];
        std::vector<double> kp_uv, ids;
    };
    std::deque<normal_frame> normal;
    // scratch of a step
    std::vector<double> state, Obs, T_all, kp_all, X_all, H;
    std::vector<int> n_all, col;
};

namespace {

bool ascending(const double *v, int n)
{
    for (int i = 1; i < n; ++i)
        if (!(v[i] > v[i - 1])) return false;
    return true;
}

// column of `row` (ascending) that holds `id`, or -1
int column_of(const std::vector<double> &row, double id)
{
    const auto it = std::lower_bound(row.begin(), row.end(), id);
    return it != row.end() && *it == id ? (int)(it - row.begin()) : -1;
}

// Columns of `row` that hold ids[0 .. n): both ascend, so one walk over the two lists finds them all
// (col[k] = -1 for an id the row does not hold).
void columns_of(const std::vector<double> &row, const double *ids, int n, std::vector<int> &col)
{
    col.resize((size_t)n);
    size_t j = 0;
    const size_t m = row.size();
    for (int k = 0; k < n; ++k) {
        while (j < m && row[j] < ids[k]) ++j;
        col[(size_t)k] = j < m && row[j] == ids[k] ? (int)j : -1;
    }
}

void refresh_ids(vohip_window *w)
{
    const size_t m = w->lm.size() / 4;
    w->lm_id.resize(m);
    for (size_t c = 0; c < m; ++c) w->lm_id[c] = w->lm[4 * c + 3];
}

int twist_of(vohip_window *w, const double T[16], double tw[6]) { return vohip_homog2twist(w->ctx, T, 1, tw); }

// VO_bundle_adjust.m:17-86: slide the window, optimise it, write the result back.
int optimise_window(vohip_window *w, const double *P_rc, double *X, const double *ids, int n, const double T_W_C[16],
                    int *n_poses, int *offsets, double *poses)
{
    vohip_ctx *ctx = w->ctx;
    const int F = w->num_key, d = w->prm.keyframe_d, step = d + 1;
    // VOHIP_WINDOW_TRACE=1: host time of the stages of every optimised window
    static const bool trace = std::getenv("VOHIP_WINDOW_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t0 = now();
    // :18-25 the landmarks that only the dropped keyframe saw leave the window
    {
        const std::vector<double> &o2 = w->obs[1];
        const int m2 = (int)o2[0];
        VOHIP_REQUIRE(ctx, m2 >= 1, "the second keyframe of the window observes no landmark");
        const double *ids2 = o2.data() + o2.size() - m2;
        const int last = column_of(w->lm_id, ids2[m2 - 1]);
        VOHIP_REQUIRE(ctx, last >= 0, "a landmark of the second keyframe is missing from the window");
        std::vector<double> kept;
        kept.reserve(w->lm.size());
        columns_of(w->lm_id, ids2, m2, w->col);
        for (int k = 0; k < m2; ++k) {
            const int c = w->col[(size_t)k];
            if (c >= 0) kept.insert(kept.end(), w->lm.begin() + 4 * (size_t)c, w->lm.begin() + 4 * (size_t)c + 4);
        }
        kept.insert(kept.end(), w->lm.begin() + 4 * ((size_t)last + 1), w->lm.end());
        w->lm.swap(kept);
        refresh_ids(w);
    }
    // :28-31
    double tw[6];
    VOHIP_TRY(twist_of(w, T_W_C, tw));
    w->tau.erase(w->tau.begin(), w->tau.begin() + 6);
    w->tau.insert(w->tau.end(), tw, tw + 6);
    w->obs.pop_front();
    {
        std::vector<double> cur;
        cur.reserve(1 + 3 * (size_t)n);
        cur.push_back((double)n);
        cur.insert(cur.end(), P_rc, P_rc + 2 * (size_t)n);
        cur.insert(cur.end(), ids, ids + n);
        w->obs.push_back(std::move(cur));
    }
    const int m = (int)w->lm_id.size();
    // :43-50 state and observation vectors, global ids -> column numbers (:34-41)
    w->state.resize(6 * (size_t)F + 3 * (size_t)m);
    std::copy(w->tau.begin(), w->tau.end(), w->state.begin());
    for (int c = 0; c < m; ++c)
        for (int a = 0; a < 3; ++a) w->state[6 * (size_t)F + 3 * (size_t)c + a] = w->lm[4 * (size_t)c + a];
    w->Obs.clear();
    {
        size_t total = 2;
        for (int i = 0; i < F; ++i) total += w->obs[i].size();
        w->Obs.reserve(total);
    }
    w->Obs.push_back((double)F);
    w->Obs.push_back((double)m);
    for (int i = 0; i < F; ++i) {
        const std::vector<double> &o = w->obs[i];
        const int ni = (int)o[0];
        w->Obs.insert(w->Obs.end(), o.begin(), o.begin() + 1 + 2 * (size_t)ni);
        columns_of(w->lm_id, o.data() + 1 + 2 * (size_t)ni, ni, w->col);
        for (int k = 0; k < ni; ++k) {
            const int c = w->col[(size_t)k];
            VOHIP_REQUIRE(ctx, c >= 0, "an observed landmark is missing from the window");
            w->Obs.push_back((double)(c + 1));
        }
    }
    // :51
    const auto t1 = now();
    VOHIP_TRY(vohip_ba_run(ctx, w->state.data(), (int)w->state.size(), w->Obs.data(), (int)w->Obs.size(), w->K,
                           w->prm.ba_max_iter, w->prm.ba_ftol, w->prm.ba_xtol, nullptr));
    // :53-57 the keyframes' poses
    const auto t2 = now();
    w->H.resize(16 * (size_t)F);
    VOHIP_TRY(vohip_twist2homog(ctx, w->state.data(), F, w->H.data()));
    int np = 0;
    for (int i = 1; i <= F; ++i) {
        offsets[np] = -step * (F - i);
        std::copy(w->H.begin() + 16 * (size_t)(i - 1), w->H.begin() + 16 * (size_t)i, poses + 16 * (size_t)np);
        ++np;
    }
    // (B.tau keeps the twists the frames were localised with: the reference does not write the optimised ones back)
    // :58
    for (int c = 0; c < m; ++c)
        for (int a = 0; a < 3; ++a) w->lm[4 * (size_t)c + a] = w->state[6 * (size_t)F + 3 * (size_t)c + a];
    // :62-70 an optimised landmark is accepted when its depth in the current camera is plausible, else
    // the window takes the old one back.  Row 3 of T_C_W = inv(T_W_C) for a rigid T_W_C.
    const double r0 = T_W_C[8], r1 = T_W_C[9], r2 = T_W_C[10];
    const double t3 = -(r0 * T_W_C[12] + r1 * T_W_C[13] + r2 * T_W_C[14]);
    columns_of(w->lm_id, ids, n, w->col);
    for (int i = 0; i < n; ++i) {
        const int c = w->col[(size_t)i];
        VOHIP_REQUIRE(ctx, c >= 0, "a tracked landmark is missing from the window");
        double *L = w->lm.data() + 4 * (size_t)c, *Xi = X + 3 * (size_t)i;
        const double depth = r0 * L[0] + r1 * L[1] + r2 * L[2] + t3;
        if (depth > w->prm.min_depth && depth < 1.5 * w->prm.max_depth) {
            Xi[0] = L[0];
            Xi[1] = L[1];
            Xi[2] = L[2];
        } else {
            L[0] = Xi[0];
            L[1] = Xi[1];
            L[2] = Xi[2];
        }
    }
    // :71-86 the buffered normal frames against the optimised landmarks (the frames do not depend on each
    // other: one launch and one synchronisation for all of them)
    const auto t3c = now();
    const int k = (F - 1) * d;
    VOHIP_REQUIRE(ctx, (int)w->normal.size() >= k, "fewer normal frames are buffered than the window spans");
    w->T_all.clear();
    w->kp_all.clear();
    w->X_all.clear();
    w->n_all.clear();
    for (int b = 0; b < k; ++b) {
        const vohip_window::normal_frame &nf = w->normal[b];
        const int nb = (int)nf.ids.size();
        w->T_all.insert(w->T_all.end(), nf.T, nf.T + 16);
        w->kp_all.insert(w->kp_all.end(), nf.kp_uv.begin(), nf.kp_uv.end());
        columns_of(w->lm_id, nf.ids.data(), nb, w->col);
        for (int q = 0; q < nb; ++q) {
            const int c = w->col[(size_t)q];
            VOHIP_REQUIRE(ctx, c >= 0, "a landmark of a buffered normal frame is missing from the window");
            w->X_all.insert(w->X_all.end(), w->lm.begin() + 4 * (size_t)c, w->lm.begin() + 4 * (size_t)c + 3);
        }
        w->n_all.push_back(nb);
    }
    if (k > 0)
        VOHIP_TRY(vohip_pose_refine_batch(ctx, k, w->T_all.data(), w->kp_all.data(), w->X_all.data(), w->n_all.data(), w->K,
                                          w->prm.refine_max_iter, w->prm.refine_ftol, w->prm.refine_xtol, nullptr));
    for (int i = 1; i <= F - 1; ++i)
        for (int j = 1; j <= d; ++j) {
            const int b = d * (i - 1) + j - 1;
            offsets[np] = -step * (F - i) + j;
            std::copy(w->T_all.begin() + 16 * (size_t)b, w->T_all.begin() + 16 * (size_t)(b + 1), poses + 16 * (size_t)np);
            ++np;
        }
    *n_poses = np;
    if (trace)
        std::fprintf(stderr, "vohip window trace (us): slide + pack %.0f  runBA %.0f  poses + depth test %.0f  re-refinement %.0f | F %d m %d\n",
                     us(t0, t1), us(t1, t2), us(t2, t3c), us(t3c, now()), F, m);
    return VOHIP_OK;
}

}  // namespace

vohip_ctx *vohip_window_context(vohip_window *w) { return w ? w->ctx : nullptr; }

extern "C" {

void vohip_window_default_params(vohip_window_params *p)
{
    if (!p) return;
    p->window_size = 5;        // main_BA.m:197
    p->keyframe_d = 1;         // main_BA.m:198
    p->min_depth = 1.0;
    p->max_depth = 100.0;
    p->ba_max_iter = 21;       // runBA.m:65-78: MaxIter 1, then MaxIter 20
    p->ba_ftol = 1e-6;         // lsqnonlin's defaults (runBA.m:77-78)
    p->ba_xtol = 1e-6;
    p->refine_max_iter = 100;  // T_refinement.m:4
    p->refine_ftol = 1e-6;
    p->refine_xtol = 1e-6;
}

int vohip_window_create(vohip_ctx *ctx, const double K[9], const vohip_window_params *params, vohip_window **out)
{
    if (!ctx) return VOHIP_ERR_INVALID_ARGUMENT;
    VOHIP_REQUIRE(ctx, K && params && out, "NULL argument");
    VOHIP_REQUIRE(ctx, params->window_size >= 2 && params->window_size <= 8, "window_size must be 2..8");
    VOHIP_REQUIRE(ctx, params->keyframe_d >= 0, "negative keyframe_d");
    vohip_window *w = new (std::nothrow) vohip_window;
    if (!w) return vohip_fail(ctx, VOHIP_ERR_HIP, "out of host memory");
    w->ctx = ctx;
    w->prm = *params;
    std::memcpy(w->K, K, sizeof(w->K));
    *out = w;
    return VOHIP_OK;
}

int vohip_window_destroy(vohip_window *w)
{
    delete w;
    return VOHIP_OK;
}

int vohip_window_reset(vohip_window *w, const double *X, const double *P_rc, int n, const double T_W_C[16])
{
    if (!w) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = w->ctx;
    VOHIP_REQUIRE(ctx, n >= 0 && T_W_C && (n == 0 || (X && P_rc)), "bad argument");
    w->num_key = 1;
    w->count_frame = 0;
    w->new_idx = (double)(n + 1);
    w->tau.resize(6);
    VOHIP_TRY(twist_of(w, T_W_C, w->tau.data()));
    w->lm.resize(4 * (size_t)n);
    std::vector<double> o(1 + 3 * (size_t)n);
    o[0] = (double)n;
    for (int i = 0; i < n; ++i) {
        for (int a = 0; a < 3; ++a) w->lm[4 * (size_t)i + a] = X[3 * (size_t)i + a];
        w->lm[4 * (size_t)i + 3] = (double)(i + 1);
        // main_demo.m:297-298 stores the first keyframe's keypoints as (u, v); the entry leaves the window
        // before the first optimisation
        o[1 + 2 * (size_t)i] = P_rc[2 * (size_t)i + 1];
        o[2 + 2 * (size_t)i] = P_rc[2 * (size_t)i];
        o[1 + 2 * (size_t)n + i] = (double)(i + 1);
    }
    refresh_ids(w);
    w->obs.clear();
    w->obs.push_back(std::move(o));
    w->normal.clear();
    return VOHIP_OK;
}

int vohip_window_step(vohip_window *w, const double *P_rc, double *X, const double *ids, int n, const double T_W_C[16],
                      int *optimised, int *n_poses, int *offsets, double *poses, int cap)
{
    if (!w) return VOHIP_ERR_INVALID_ARGUMENT;
    vohip_ctx *ctx = w->ctx;
    VOHIP_REQUIRE(ctx, n >= 0 && T_W_C && optimised && n_poses && (n == 0 || (P_rc && X && ids)), "bad argument");
    VOHIP_REQUIRE(ctx, w->num_key >= 1, "vohip_window_reset first");
    VOHIP_REQUIRE(ctx, ascending(ids, n), "landmark ids must ascend");
    *optimised = 0;
    *n_poses = 0;
    // update_landmarks.m:83-95: B.landmarks grows with S.X
    int added = 0;
    for (int i = 0; i < n; ++i)
        if (ids[i] >= w->new_idx) {
            VOHIP_REQUIRE(ctx, w->lm_id.empty() || ids[i] > w->lm_id.back(), "landmark ids must ascend");
            w->lm.insert(w->lm.end(), X + 3 * (size_t)i, X + 3 * (size_t)i + 3);
            w->lm.push_back(ids[i]);
            w->lm_id.push_back(ids[i]);
            ++added;
        }
    w->new_idx += (double)added;
    const int d = w->prm.keyframe_d;
    if (w->count_frame == d) {   // a keyframe, VO_bundle_adjust.m:3
        if (w->num_key < w->prm.window_size) {   // :4-15 the window still fills up
            double tw[6];
            VOHIP_TRY(twist_of(w, T_W_C, tw));
            w->tau.insert(w->tau.end(), tw, tw + 6);
            std::vector<double> cur;
            cur.reserve(1 + 3 * (size_t)n);
            cur.push_back((double)n);
            cur.insert(cur.end(), P_rc, P_rc + 2 * (size_t)n);
            cur.insert(cur.end(), ids, ids + n);
            w->obs.push_back(std::move(cur));
            w->num_key++;
        } else {
            VOHIP_REQUIRE(ctx, offsets && poses && cap >= w->num_key + (w->num_key - 1) * d,
                          "room for window_size + (window_size - 1) keyframe_d poses is needed");
            VOHIP_TRY(optimise_window(w, P_rc, X, ids, n, T_W_C, n_poses, offsets, poses));
            *optimised = 1;
        }
        w->count_frame = 0;
    } else {                     // a normal frame, :89-98
        vohip_window::normal_frame nf;
        std::memcpy(nf.T, T_W_C, sizeof(nf.T));
        nf.kp_uv.resize(2 * (size_t)n);
        for (int i = 0; i < n; ++i) {   // flipud(S.P)
            nf.kp_uv[2 * (size_t)i] = P_rc[2 * (size_t)i + 1];
            nf.kp_uv[2 * (size_t)i + 1] = P_rc[2 * (size_t)i];
        }
        nf.ids.assign(ids, ids + n);
        if ((int)w->normal.size() >= (w->prm.window_size - 1) * d && !w->normal.empty()) w->normal.pop_front();
        w->normal.push_back(std::move(nf));
        w->count_frame++;
    }
    return VOHIP_OK;
}

int vohip_window_state(vohip_window *w, int *num_key, int *m, double *tau, double *landmarks)
{
    if (!w) return VOHIP_ERR_INVALID_ARGUMENT;
    if (num_key) *num_key = w->num_key;
    if (m) *m = (int)w->lm_id.size();
    if (tau) std::copy(w->tau.begin(), w->tau.end(), tau);
    if (landmarks) std::copy(w->lm.begin(), w->lm.end(), landmarks);
    return VOHIP_OK;
}

}  // extern "C"
