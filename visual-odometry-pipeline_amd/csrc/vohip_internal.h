// Internal definitions shared by the translation units of libvohip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "vohip.h"

// A grow-only device (or pinned host) buffer owned by the context.
struct vohip_buf {
    void *ptr = nullptr;
    size_t cap = 0;
    bool pinned_host = false;
};

// Identifiers of the context-owned buffers (device unless the name starts with H_).
enum vohip_buf_id {
    B_RANSAC_RT = 0,   // h x 4 x 12 doubles: [R_C_W | t_C_W] of every candidate pose
    B_RANSAC_M,        // h x 4 x 12 doubles: K [R | t], row-major 3x4
    B_RANSAC_KP,
    B_RANSAC_X,
    B_RANSAC_SAMPLES,
    B_RANSAC_COUNTS,
    B_RANSAC_MASK,
    H_RANSAC_COUNTS,
    B_STAGE0, B_STAGE1, B_STAGE2, B_STAGE3, B_STAGE4, B_STAGE5, B_STAGE6, B_STAGE7,
    H_STAGE0, H_STAGE1, H_STAGE2, H_STAGE3,
    B_IMG_STAGE,
    B_HARRIS_SCRATCH,
    B_HARRIS_BATCH,
    B_SURF_SCRATCH, B_SURF_BATCH, B_SURF_TABLES, B_SURF_MATCH,
    B_BRISK_SCRATCH, B_BRISK_BATCH,
    B_FM_NRM, B_FM_F, B_FM_COUNTS, H_FM_COUNTS,
    B_BA_INDEX, B_BA_FRAME, B_BA_LM, B_BA_UV, B_BA_COUNT, B_BA_STATE, B_BA_TRIAL, B_BA_SPART, B_BA_GPART,
    B_BA_E, B_BA_GL, B_BA_W, B_BA_DP, B_BA_PART, B_BA_STATUS, H_BA_STATUS, B_BA_SRED, B_BA_GRED,
    B_KLT_STATS,
    B_PYR0,                 // first of VOHIP_PYR_SLOTS consecutive ids, one per pyramid slot (see B_PYR_LAST)
    VOHIP_NBUF = 96
};

// One image pyramid (float levels, all in one context buffer B_PYR0 + slot).
constexpr int VOHIP_PYR_SLOTS = 16;
constexpr int B_PYR_LAST = B_PYR0 + VOHIP_PYR_SLOTS - 1;
static_assert(B_PYR_LAST < VOHIP_NBUF, "the pyramid slots must fit into the buffer table");
struct vohip_pyramid {
    int rows = 0, cols = 0, colmajor = 0, levels = 0;
    int A[8] = {0}, B[8] = {0};   // fast / slow axis length per level
    size_t offset[8] = {0};       // in floats from the start of the buffer
};

struct vohip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    char err[512] = {0};
    vohip_buf bufs[VOHIP_NBUF];
    vohip_pyramid pyr[VOHIP_PYR_SLOTS];
    const float *pyr_ext[VOHIP_PYR_SLOTS] = {nullptr};   // a slot that shows a pyramid in the caller's memory (vohip_pyramid_attach)
    int ransac_h = 0;  // hypotheses held by B_RANSAC_RT / B_RANSAC_M
    int opt_ransac_variant = -1;  // scoring-kernel geometry, see vohip_set_option
    int opt_klt_stats = 0;        // count tracker work (two atomics per track), see vohip_klt_stats
    unsigned harris_calls = 0;    // uses of B_HARRIS_SCRATCH (its two maximum slots alternate)
    // tracker launches timed with event pairs on the streams they run on (option "klt_time", see vohip_klt_time)
    int opt_klt_time = 0;
    static constexpr int KLT_TIME_PAIRS = 1024;
    hipEvent_t *klt_ev = nullptr;   // 2 * KLT_TIME_PAIRS events, created with the option
    int klt_ev_used = 0;            // pairs recorded since the last reset
};

// The independent sequences one launch of a frame-loop kernel serves (a group of loops advanced in lockstep,
// vohip_group_*).  A kernel of the loop is launched with gridDim.y = the number of bits set in `mask`; the
// workgroups with blockIdx.y = j work on the sequence whose index is the position of the j-th set bit, and
// find its arrays by adding index * stride to the pointers they were given (those of sequence 0): the
// per-sequence state blocks, pinned staging blocks and prepared-sequence blocks all have one layout and lie a
// constant number of bytes apart.  A single loop is the group {sequence 0}: mask 1, gridDim.y 1, offsets 0.
constexpr int VOHIP_MAX_SEQ = 64;
struct vohip_seqs {
    unsigned long long stride_state = 0;   // bytes between the device state blocks of consecutive sequences
    unsigned long long stride_pin = 0;     // ... between their pinned host blocks (draws in, summary out)
    unsigned long long stride_prep = 0;    // ... between their prepared data (images, pyramids, corner lists)
    unsigned long long mask = 1;
};
static inline int vohip_seq_count(const vohip_seqs &q) { return __builtin_popcountll(q.mask); }

int vohip_fail(vohip_ctx *ctx, int status, const char *fmt, ...);
int vohip_reserve(vohip_ctx *ctx, vohip_buf &b, size_t bytes);
int vohip_reserve_pinned(vohip_ctx *ctx, vohip_buf &b, size_t bytes);
void vohip_release(vohip_buf &b);

#define VOHIP_HIP(ctx, call)                                                                 \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return vohip_fail((ctx), VOHIP_ERR_HIP, "%s failed: %s (%s:%d)", #call,          \
                              hipGetErrorString(e__), __FILE__, __LINE__);                   \
    } while (0)

#define VOHIP_TRY(expr)                     \
    do {                                    \
        int s__ = (expr);                   \
        if (s__ != VOHIP_OK) return s__;    \
    } while (0)

#define VOHIP_REQUIRE(ctx, cond, msg)                                                \
    do {                                                                             \
        if (!(cond)) return vohip_fail((ctx), VOHIP_ERR_INVALID_ARGUMENT, "%s", msg); \
    } while (0)

// Reserve context buffer `id` for `bytes` and (when src != NULL) enqueue the host-to-device copy.
static inline int vohip_upload(vohip_ctx *ctx, int id, const void *src, size_t bytes, void **dptr)
{
    VOHIP_TRY(vohip_reserve(ctx, ctx->bufs[id], bytes ? bytes : 1));
    *dptr = ctx->bufs[id].ptr;
    if (src && bytes)
        VOHIP_HIP(ctx, hipMemcpyAsync(*dptr, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return VOHIP_OK;
}
static inline int vohip_download(vohip_ctx *ctx, void *dst, const void *dsrc, size_t bytes)
{
    if (bytes) VOHIP_HIP(ctx, hipMemcpyAsync(dst, dsrc, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return VOHIP_OK;
}

// Launchers shared between translation units.  They enqueue on ctx->stream and never synchronise;
// a count argument `n` sizes the grid (an upper bound) and `d_n`, when not NULL, points to the
// actual count in device memory.
int vohip_klt_enqueue(vohip_ctx *ctx, int slot_prev, int slot_cur, const double *d_pts_xy, int ld, int n,
                      const int *d_n, int block_size, int max_iter, double max_bidir_error, double *d_out_xy,
                      uint8_t *d_valid, double *d_fb_err);
// One point set of a tracker launch.  Track i of the set is read from row r = rows ? rows[i] :
// base + i of pts_xy (x column then y column, leading dimension ld) and of live; its results go to row
// out_base + i of out_xy / valid / fb_err, or to row r when out_base < 0.  Rows whose `live` byte is 0
// are not tracked (valid = 0).
struct vohip_klt_set {
    const double *pts_xy;
    double *out_xy;
    uint8_t *valid;
    double *fb_err;        // may be NULL
    const int32_t *rows;   // may be NULL
    const uint8_t *live;   // may be NULL
    const int *n_dev;      // may be NULL; else the device-side count (<= n, n then only sizes the grid)
    int ld, n, base, out_base;
    const int *first_dev;  // may be NULL; else the set starts at track *first_dev (device side) and ends before *n_dev
};
// Up to three sets in one launch (a launch lasts as long as its slowest track however many tracks
// it carries).
// `seqs` (may be NULL: a single sequence): the sequences of a group the launch serves; the sets then name the
// arrays of sequence 0 and the pyramid slots show sequence 0's pyramids.
int vohip_klt_enqueue_sets(vohip_ctx *ctx, int slot_prev, int slot_cur, const vohip_klt_set *sets, int nsets,
                           int block_size, int max_iter, double max_bidir_error, const vohip_seqs *seqs = nullptr);
// The ordered selection of the valid landmark tracks (main_demo.m:380-381, :387) as it rides in the launch
// that solves a frame's first hypotheses: track i of the table (x column then y column, leading dimension
// ld) with valid[i] != 0 goes, in order, to mv_rc (row, col), Xv, idv and rowv; *n_valid receives their
// number, *idx = *idx_next, and the first h0 sample triples are made from the uniform draws u3.
// `valid` must be 16-byte aligned and readable up to the next multiple of 16 past n.
struct vohip_select_tracked {
    const double *out_xy;
    int ld;
    const uint8_t *valid;
    const double *L_X, *L_id;
    int n;
    double *mv_rc, *Xv, *idv;
    int32_t *rowv;
    int *n_valid;
    double *idx;
    const double *idx_next;
    const double *u3;
    int h0;
    int32_t *samples;
    const int *n_dev;   // may be NULL; else the table holds min(n, *n_dev) tracks (n then only sizes the grid)
};
// `select` (may be NULL): see vohip_select_tracked; then first must be 0, h <= select->h0 and d_kp_rc, d_X, d_n
// name the lists the selection writes.
int vohip_ransac_enqueue(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const int *d_n,
                         const double K[9], const int32_t *d_samples, int first, int h, int32_t *d_counts,
                         double *d_Rt_all, double *d_M_all, const vohip_select_tracked *select,
                         const vohip_seqs *seqs = nullptr);
int vohip_ransac_mask_enqueue(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const int *d_n,
                              const double *d_M12, uint8_t *d_mask);
int vohip_ransac_score_enqueue(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, const double *d_M,
                               int slots, int32_t *d_counts);
double vohip_ransac_adaptive_iterations(int max_num_inliers, int n);
int vohip_ransac_scratch_init(vohip_ctx *ctx);   // once per device, with that device current
int vohip_fm_enqueue(vohip_ctx *ctx, const double *d_p1, const double *d_p2, int n, const int *d_n,
                     const int32_t *d_samples, int num_trials, double threshold, double *d_nrm, double *d_F,
                     int32_t *d_counts, const vohip_seqs *seqs = nullptr);
int vohip_fm_mask_enqueue(vohip_ctx *ctx, const double *d_p1, const double *d_p2, int n, const int *d_n,
                          const double *d_F9, double threshold, uint8_t *d_mask);
int vohip_candidate_filter_enqueue(vohip_ctx *ctx, const float *d_new_xy, int ld, int n_new_cap, const double *d_C_rc,
                                   const double *d_P_rc, const int *d_counts3, double r, uint8_t *d_keep,
                                   const vohip_seqs *seqs = nullptr);
int vohip_candidate_triangulate_enqueue(vohip_ctx *ctx, const double *d_cur_rc, const double *d_first_rc,
                                        const double *d_T_first, int n_cap, const int *d_n, const double *d_T_W_C,
                                        const double K[9], double angle_threshold_deg, double min_depth,
                                        double max_depth, int is_BA, uint8_t *d_status, double *d_X,
                                        const vohip_seqs *seqs = nullptr);
size_t vohip_corner_scratch_bytes(int rows, int cols);
int vohip_detect_corners_enqueue(vohip_ctx *ctx, hipStream_t st, void *scratch, unsigned *calls, const uint8_t *d_img,
                                 int rows, int cols, int colmajor, double min_quality, int filter_size, int border,
                                 int method, double min_contrast, float *d_xy, float *d_metric, int cap, int *d_count,
                                 int brisk_octaves = 4);
// Slot `slot` shows the pyramid at d_pyr (layout of vohip_pyramid_batch_dev) until the slot is built again; nothing is copied.
int vohip_pyramid_attach(vohip_ctx *ctx, int slot, const float *d_pyr, int rows, int cols, int colmajor, int levels);
int vohip_detect_corners_batch_enqueue(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                       double min_quality, int filter_size, int border, int method, float *d_xy,
                                       float *d_metric, int cap, int *d_counts, double min_contrast = 0.2,
                                       int brisk_octaves = 4);
// SURF (detector 3): scratch of a single image; detection on `st` with caller-owned scratch; a stored sequence.
size_t vohip_surf_scratch_bytes(int rows, int cols);
int vohip_detect_surf_enqueue(vohip_ctx *ctx, hipStream_t st, void *scratch, const uint8_t *d_img, int rows, int cols,
                              int colmajor, double metric_threshold, int border, float *d_xy, float *d_metric,
                              float *d_scale, int cap, int *d_count);
int vohip_detect_surf_batch_enqueue(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                    double metric_threshold, int border, float *d_xy, float *d_metric, int cap,
                                    int *d_counts);
// BRISK (detector 4): as for SURF.
size_t vohip_brisk_scratch_bytes(int rows, int cols);
int vohip_detect_brisk_enqueue(vohip_ctx *ctx, hipStream_t st, void *scratch, const uint8_t *d_img, int rows, int cols,
                               int colmajor, double min_contrast, double min_quality, int num_octaves, int border,
                               float *d_xy, float *d_metric, float *d_scale, int cap, int *d_count);
int vohip_detect_brisk_batch_enqueue(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                     double min_contrast, double min_quality, int num_octaves, int border, float *d_xy,
                                     float *d_metric, int cap, int *d_counts);
int vohip_pose_refine_enqueue(vohip_ctx *ctx, double *d_T, const double *d_kp_uv, const double *d_X, int n,
                              const int *d_n, const double K[9], int max_iter, double ftol, double xtol,
                              double *d_stats, const vohip_seqs *seqs = nullptr);

vohip_ctx *vohip_window_context(vohip_window *w);

// Camera matrix passed to kernels by value (column-major, as MATLAB stores it).
struct vohip_K {
    double k[9];
};
static inline vohip_K vohip_make_K(const double K[9])
{
    vohip_K r;
    for (int i = 0; i < 9; ++i) r.k[i] = K[i];
    return r;
}
