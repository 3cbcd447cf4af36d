/*
 * vohip.h -- C ABI of libvohip.so, the MI355X (gfx950) implementation of the per-frame
 * continuous-operation hot path of the MATLAB visual-odometry pipeline.
 *
 * The reference has no FFI; its boundary is the set of MATLAB function signatures.
 * Every entry point below names the reference function (file:line) it stands in for.
 * A MEX gateway (or ctypes, as the bundled Python host layer does) binds these
 * symbols one to one.
 *
 * Conventions
 *  - All matrices are column-major, exactly as MATLAB stores them: "3xN" means N
 *    consecutive groups of 3 values.
 *  - Every function returns a vohip_status; 0 is success.  vohip_last_error(ctx)
 *    returns a human-readable message for the last non-zero status of that context.
 *  - The caller owns every buffer it passes.  The library never keeps a host pointer
 *    after a call returns.  Device scratch memory belongs to the context.
 *  - A context is bound to one GPU and one HIP stream and must be used from one host
 *    thread at a time; different contexts are independent.
 *  - Functions whose name ends in _dev take DEVICE pointers for the arrays documented
 *    as such, enqueue their kernels on the context's stream and return without
 *    synchronising.  All other functions take host pointers and return with the
 *    results in place.
 *  - Index arguments are 0-based (a MATLAB caller subtracts 1).
 */
#ifndef VOHIP_H
#define VOHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vohip_ctx vohip_ctx;

typedef enum vohip_status {
    VOHIP_OK = 0,
    VOHIP_ERR_INVALID_ARGUMENT = 1,
    VOHIP_ERR_HIP = 2,              /* a HIP runtime call failed; see vohip_last_error */
    VOHIP_NO_MODEL = 3,             /* RANSAC: no hypothesis reached the minimum inlier count
                                       (the reference returns R = t = [], ransacLocalization.m:138-140) */
    VOHIP_SAMPLES_EXHAUSTED = 4,    /* RANSAC: the adaptive stop was not reached within the supplied
                                       sample triples; outputs hold the best model so far */
    VOHIP_TOO_FEW_POINTS = 5,
    VOHIP_ERR_NO_DEVICE = 6,
    VOHIP_CAPACITY_EXCEEDED = 7,    /* an output list was longer than the caller's capacity; the
                                       first `cap` entries are valid but their choice is arbitrary */
    VOHIP_SEQUENCE_IDLE = 8         /* vohip_group_finish: this sequence of the group took no part in the frame
                                       (not reset, ended earlier, or stopped by an error of its own) */
} vohip_status;

/* ---- context ------------------------------------------------------------------------------ */

/* Creates a context on GPU `device_id`.  `hip_stream` is a hipStream_t to enqueue on, or NULL to
 * let the context create (and own) a stream.  No reference counterpart: the reference keeps all
 * state in MATLAB workspace variables. */
int vohip_create(vohip_ctx **out, int device_id, void *hip_stream);
int vohip_destroy(vohip_ctx *ctx);
const char *vohip_last_error(const vohip_ctx *ctx);
/* The hipStream_t all kernels of this context are enqueued on. */
void *vohip_stream(vohip_ctx *ctx);
int vohip_synchronize(vohip_ctx *ctx);
/* Tuning knobs (no reference counterpart).  "ransac_score_variant" 0..5 (-1 = automatic, the default) selects the workgroup
 * geometry (hypotheses per workgroup x points per LDS tile) of the RANSAC scoring kernel; results
 * do not depend on it.  "klt_stats" 1 makes the tracker count its work (see vohip_klt_stats; two
 * atomics per track, off by default).  "klt_time" 1 brackets every tracker launch with a pair of timing
 * events on the stream it runs on (see vohip_klt_time; measurement only, off by default). */
int vohip_set_option(vohip_ctx *ctx, const char *name, int value);
/* Library version, "major.minor". */
const char *vohip_version(void);

/* ---- P3P RANSAC localisation ---------------------------------------------------------------
 * Replaces utils/exe7_Ransac/ransacLocalization.m:1-157 together with p3p.m:41-217,
 * solveQuartic.m:26-66 and the projectPoints.m:1-16 calls inside it (use_p3p = true,
 * tweaked_for_more = true, adaptive = true, the only configuration the reference runs).
 *
 * The reference draws each 3-point sample with datasample from MATLAB's global RNG
 * (ransacLocalization.m:46-47).  Here the caller draws the index triples (with the same toolbox
 * call, so the hypothesis set is identical) and passes them in: samples is 3 x hcap, 0-based.
 */

/* Full adaptive RANSAC on host buffers.
 *   kp_rc   2xN keypoints in (row, col) order, as the reference passes them (main_demo.m:387-388)
 *   X       3xN landmarks
 *   K       3x3 camera matrix
 *   R_C_W (3x3), t_C_W (3): pose of the best hypothesis; untouched when VOHIP_NO_MODEL
 *   mask    N bytes, 1 = inlier of the best hypothesis (all 0 when VOHIP_NO_MODEL)
 *   max_num_inliers_history, num_iteration_history: hcap doubles each, first *n_iter filled
 *   n_iter  iterations the reference loop would have executed (i-1 at ransacLocalization.m:155)
 * Hypotheses are evaluated in chunks on the GPU; the adaptive-stop recurrence of
 * ransacLocalization.m:118-135 is then replayed in sample order, so hypotheses past the stop
 * index never influence the result. */
int vohip_ransac_p3p(vohip_ctx *ctx, const double *kp_rc, const double *X, int n, const double K[9],
                     const int32_t *samples, int hcap,
                     double R_C_W[9], double t_C_W[3], uint8_t *mask,
                     double *max_num_inliers_history, double *num_iteration_history, int *n_iter);

/* The DLT branch of the same function (use_p3p = false, ransacLocalization.m:9,20-24,69-74,145-151): six samples
 * per hypothesis (samples is 6 x hcap, 0-based, six distinct indices each, as datasample(..., 6, 2, 'Replace',
 * false) draws them), the hypothesis' pose from estimatePoseDLT, one pose scored per iteration, the adaptive bound
 * with exponent 6, and the returned pose re-estimated by estimatePoseDLT over the inliers of the best hypothesis
 * (mask is that hypothesis' inlier mask, as in the reference).  Arguments and status codes as vohip_ransac_p3p;
 * VOHIP_TOO_FEW_POINTS below 6 correspondences.  The branch is dead in the reference (use_p3p = true at :9). */
int vohip_ransac_dlt(vohip_ctx *ctx, const double *kp_rc, const double *X, int n, const double K[9],
                     const int32_t *samples, int hcap, double R_C_W[9], double t_C_W[3], uint8_t *mask,
                     double *max_num_inliers_history, double *num_iteration_history, int *n_iter);
/* M_tilde = estimatePoseDLT(p, P, K), utils/exe7_Ransac/estimatePoseDLT.m:1-60: p n x 2 undistorted (u, v), P n x 3,
 * K 3 x 3, M 3 x 4 = [R_tilde | alpha t], all column-major as MATLAB passes them; n >= 6. */
int vohip_estimate_pose_dlt(vohip_ctx *ctx, const double *p_uv, const double *P, int n, const double K[9],
                            double M[12]);

/* Hypothesis generation + inlier scoring only (ransacLocalization.m:46-101 for h hypotheses at
 * once), everything resident on the GPU: d_kp_rc (2xN), d_X (3xN), d_samples (3xh int32) and
 * d_counts (4xh int32, out: inlier count of each of the 4 P3P solutions) are device pointers.
 * Asynchronous on the context's stream.  Equivalent to _solve_dev followed by _count_dev. */
int vohip_ransac_p3p_score_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                               const double K[9], const int32_t *d_samples, int h,
                               int32_t *d_counts);

/* vohip_ransac_p3p_score_dev for callers that cannot allocate (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments,
 * same kernels and bit-identical counts, but the candidate poses go to a static 3 MB device array
 * instead of context buffers, so the call makes no hipMalloc / hipFree: h <= 4096, else
 * VOHIP_ERR_INVALID_ARGUMENT.  Every call on one GPU shares that GPU's array (whatever the context):
 * calls that overlap on one GPU race, and the poses are not retrievable (vohip_ransac_p3p_poses and
 * _count_dev keep referring to the last _solve_dev / _score_dev call).  Asynchronous. */
int vohip_ransac_p3p_score_noalloc(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                                   const double K[9], const int32_t *d_samples, int h,
                                   int32_t *d_counts);

/* First half of _score_dev: sample -> bearings -> p3p -> 4 candidate poses per hypothesis
 * (ransacLocalization.m:46-68, p3p.m, solveQuartic.m), kept in the context. */
int vohip_ransac_p3p_solve_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                               const double K[9], const int32_t *d_samples, int h);

/* Second half of _score_dev: inlier counts of the poses of the last _solve_dev call against
 * all n points (ransacLocalization.m:77-101), one hypothesis per wavefront. */
int vohip_ransac_p3p_count_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n, int h,
                               int32_t *d_counts);

/* Poses behind the counts of the last vohip_ransac_p3p_score_dev call: copies, for hypotheses
 * [first, first+count), the 4 candidate poses as 12 doubles each ([R_C_W col-major | t_C_W]) to
 * the host array Rt (count x 4 x 12).  Synchronises the stream. */
int vohip_ransac_p3p_poses(vohip_ctx *ctx, int first, int count, double *Rt);

/* Inlier mask of one pose over device-resident points, with exactly the arithmetic the scoring
 * kernel counts with.  Rt = [R_C_W col-major | t_C_W].  d_mask: n bytes, device.  Asynchronous. */
int vohip_ransac_inlier_mask_dev(vohip_ctx *ctx, const double *d_kp_rc, const double *d_X, int n,
                                 const double K[9], const double Rt[12], uint8_t *d_mask);

/* p3p.m:41-217 for `count` independent problems (host buffers): world and bearings are
 * count x (3x3), poses count x (3x16) as the reference lays them out
 * ([C1 R1 C2 R2 C3 R3 C4 R4], real part). */
int vohip_p3p(vohip_ctx *ctx, const double *world, const double *bearings, int count, double *poses);

/* ---- SE(3) ---------------------------------------------------------------------------------- */

/* utils/exe9_BA/twist2HomogMatrix.m:1-16 for n twists (6xn, [v; w]) -> n 4x4 matrices.  Closed
 * form of the reference's expm call. */
int vohip_twist2homog(vohip_ctx *ctx, const double *twists, int n, double *H);
/* utils/exe9_BA/HomogMatrix2twist.m:1-25 for n 4x4 rigid transforms -> 6xn twists (rotation
 * angle in [0, pi]).  Closed form of the reference's logm call. */
int vohip_homog2twist(vohip_ctx *ctx, const double *H, int n, double *twists);

/* ---- triangulation -------------------------------------------------------------------------- */

/* utils/exe6_8point/linearTriangulation.m:1-41.  p1, p2 3xn homogeneous image points, M1, M2
 * 3x4 projection matrices -- or one 3x4 per point (3x4xn) when the per_point flag is non-zero,
 * which is how update_landmarks.m:57-61 calls it (n = 1, a different M_first each time).
 * P 4xn, dehomogenised. */
int vohip_triangulate(vohip_ctx *ctx, const double *p1, const double *p2, const double *M1,
                      const double *M2, int per_point_M1, int per_point_M2, int n, double *P);
int vohip_triangulate_dev(vohip_ctx *ctx, const double *d_p1, const double *d_p2, const double *d_M1,
                          const double *d_M2, int per_point_M1, int per_point_M2, int n, double *d_P);

/* utils/linearTriangulation_refinement.m:1-22.  P_init 4xn (rows 1-3 used), p1, p2 ldp x n with
 * ldp >= 2 (rows 1-2 used), M1, M2 3x4.  P 4xn (row 4 copied).  The reference's lsqnonlin call is
 * replaced by a per-point Levenberg-Marquardt with analytic Jacobians, at most 20 iterations. */
int vohip_triangulate_refine(vohip_ctx *ctx, const double *P_init, const double *p1, const double *p2,
                             int ldp, const double *M1, const double *M2, int n, double *P);

/* ---- candidate bookkeeping ------------------------------------------------------------------ */

/* Continuous_operation/update_candidate.m:12-27.  new_xy n_new x 2 single, (x, y) as
 * cornerPoints.Location; C_rc 2xnC = S.C and P_rc 2xnP = S.P in (row, col).  keep[i] = 1 iff no
 * S.C and no S.P point lies in the closed box of half-width r around detection i; all 1 when
 * S.C is empty (update_candidate.m:25-27). */
int vohip_candidate_filter(vohip_ctx *ctx, const float *new_xy, int n_new, const double *C_rc, int nC,
                           const double *P_rc, int nP, double r, uint8_t *keep);
int vohip_candidate_filter_dev(vohip_ctx *ctx, const float *d_new_xy, int n_new, const double *d_C_rc,
                               int nC, const double *d_P_rc, int nP, double r, uint8_t *d_keep);

/* Continuous_operation/update_landmarks.m:27-97 for the n candidates that survived tracking and
 * the epipolar filter: bearing-angle test (:33-43), triangulation of those above the threshold
 * (:56-61), depth filter and -- with is_BA -- the 100 px reprojection filter (:72-97).
 *   cur_rc, first_rc 2xn (row, col); T_first 12xn = S.T; T_W_C 4x4 current pose; K 3x3.
 *   status[i]: 0 = stays a candidate, 1 = becomes a landmark (X[:, i] valid), 2 = dropped.
 *   X 3xn; angles n (degrees, may be NULL). */
int vohip_candidate_triangulate(vohip_ctx *ctx, const double *cur_rc, const double *first_rc,
                                const double *T_first, int n, const double T_W_C[16], const double K[9],
                                double angle_threshold_deg, double min_depth, double max_depth, int is_BA,
                                uint8_t *status, double *X, double *angles);
int vohip_candidate_triangulate_dev(vohip_ctx *ctx, const double *d_cur_rc, const double *d_first_rc,
                                    const double *d_T_first, int n, const double T_W_C[16],
                                    const double K[9], double angle_threshold_deg, double min_depth,
                                    double max_depth, int is_BA, uint8_t *d_status, double *d_X,
                                    double *d_angles);

/* ---- the frame loop with device-resident state -------------------------------------------------
 * The loop body main_demo.m:377-427 as one call per frame.  S.P, S.X, S.C, S.F
 * and S.T (main_demo.m:272-279) stay in GPU memory between frames; the stages are the kernels behind
 * the entry points above, chained through device-side counts, so the state after each frame is the
 * one the stage-by-stage calls produce (genKeypoints.m, ransacLocalization.m, T_refinement.m,
 * update_landmarks.m, update_candidate.m).  With is_BA the caller runs VO_bundle_adjust.m between
 * frames: vohip_frame_bundle_adjust (the window kept by the library), or its own bookkeeping over
 * vohip_frame_state, vohip_ba_run and vohip_frame_apply_ba.
 *
 * Random numbers: the caller passes uniform draws in [0, 1) instead of index sets, because the
 * sizes they index are only known on the device: 3 per P3P hypothesis (mapped to 3 distinct
 * landmark indices: a = floor(u0 n), b = floor(u1 (n-1)) skipping a, c = floor(u2 (n-2)) skipping
 * both) and 8 per fundamental-matrix trial (draw k picks the floor(u_k (n-k))-th index not picked
 * before).
 *
 * The two sets of draws are independent of each other (the trial draws do not wait for the number
 * of hypotheses the localisation consumes), which lets the epipolar filter run beside the
 * localisation on a second stream.
 */
typedef struct vohip_frame vohip_frame;

typedef struct vohip_frame_params {
    int klt_block, klt_levels, klt_max_iter;   /* vision.PointTracker arguments, main_demo.m:342-343 */
    double klt_max_bidir_error;
    int detector;                              /* 0 Harris, 1 MinEigen, 2 FAST, 3 SURF, 4 BRISK (genKeypoints.m:16-20,31-32) */
    double min_quality;
    int filter_size, border;
    double r_discard_redundant;                /* hyper_paras, main_demo.m:36-43, 72-75 */
    double angle_threshold_deg, min_depth, max_depth;
    int refine_pose, refine_max_iter;          /* hyper_paras.is_refine_pose; T_refinement.m:4 */
    int fm_trials;                             /* update_landmarks.m:13-14 */
    double fm_threshold, fm_confidence;
    int cap_landmarks, cap_candidates, cap_corners;   /* capacities of the device arrays */
    int max_hypotheses;                        /* 15000, ransacLocalization.m:128 */
    int slot_a, slot_b, slot_c;                /* the three pyramid slots of the context this loop uses */
    int is_BA;                                 /* hyper_paras.is_BA: update_landmarks.m:78-88 then also asks for
                                                  reprojection errors below 100 px in both views */
    double min_contrast;                       /* detector 2: detectFASTFeatures' 'MinContrast' (main_eval.m:172-173) */
    double metric_threshold;                   /* detector 3: detectSURFFeatures' 'MetricThreshold' (main_eval.m:151,167) */
    int brisk_octaves;                         /* detector 4: detectBRISKFeatures' 'NumOctaves' (with min_contrast and min_quality) */
} vohip_frame_params;

typedef struct vohip_frame_pose {
    double T_W_C[16];          /* 4x4 column-major, after refinement when enabled */
    double Rt[12];             /* [R_C_W (column-major) | t_C_W] of the winning hypothesis */
    double refine_stats[3];    /* iterations, initial cost, final cost */
    double num_iterations;     /* the adaptive bound when the loop stopped */
    int n_tracked, n_inliers, max_num_inliers, n_iter, found, exhausted, evaluated;
} vohip_frame_pose;

typedef struct vohip_frame_counts {
    int n_landmarks, n_candidates;   /* size(S.X, 2), size(S.C, 2) at the end of the frame */
    int num_new;                     /* S.num_new(end), update_landmarks.m:124 */
    int n_detected, n_kept;          /* corners found / kept as new candidates */
    int n_candidates_tracked, n_candidates_inliers, fm_trials_run;
    int overflow;
    double host_enqueue_us, host_wait_us;   /* host time spent enqueuing / waiting in this call */
} vohip_frame_counts;

void vohip_frame_default_params(vohip_frame_params *p);
/* rows x cols uint8 images, row-major (colmajor = 0) or MATLAB's column-major (colmajor = 1). */
int vohip_frame_create(vohip_ctx *ctx, const double K[9], int rows, int cols, int colmajor,
                       const vohip_frame_params *params, vohip_frame **out);
int vohip_frame_destroy(vohip_frame *f);
/* State initialisation, main_demo.m:268-303, 342-344: X 3xn, ids n (NULL = 1..n), P_rc 2xn (row;
 * col) in `image`, T_W_C 4x4 column-major.  image on the host, or on the device when the flag is set. */
int vohip_frame_reset(vohip_frame *f, const uint8_t *image, int image_on_device, const double *X, const double *ids,
                      const double *P_rc, int n, const double T_W_C[16]);
/* One frame: u3 = 3 x h uniforms for the P3P hypotheses, u8 = 8 x fm_trials uniforms for the
 * fundamental-matrix trials (may be NULL while there are no candidates yet).  One synchronisation.
 * An image passed as device memory must be complete when the call is made: the corner detection reads it
 * on a stream of the loop's own, which does not wait for work the caller still has in flight on the
 * context's stream.
 * VOHIP_SAMPLES_EXHAUSTED: the adaptive rule wants more than h hypotheses (pose->num_iterations tells
 * how many); call vohip_frame_retry with a longer u3 whose first h triples are the ones already
 * passed.  VOHIP_NO_MODEL: no hypothesis reached 30 inliers (the reference stops there too).
 * VOHIP_TOO_FEW_POINTS: fewer than 3 landmark tracks (main_demo.m:389-391 stops) or fewer than 8
 * candidate tracks for the 8-point algorithm.  VOHIP_CAPACITY_EXCEEDED: the frame found more corners,
 * landmarks or candidates than cap_corners / cap_landmarks / cap_candidates hold (counts->overflow: bit 0
 * landmarks, bit 1 candidates); the surplus was dropped, which the reference never does
 * (update_landmarks.m:83-95 and update_candidate.m grow their arrays).  After VOHIP_NO_MODEL,
 * VOHIP_TOO_FEW_POINTS or VOHIP_CAPACITY_EXCEEDED the state is no longer the reference's: every call
 * but vohip_frame_reset and vohip_frame_destroy is refused (create the loop with larger capacities and
 * start it again from the last good state). */
int vohip_frame_step(vohip_frame *f, const uint8_t *image, int image_on_device, const double *u3, int h,
                     const double *u8, vohip_frame_pose *pose, vohip_frame_counts *counts);
int vohip_frame_retry(vohip_frame *f, const double *u3, int h, vohip_frame_pose *pose, vohip_frame_counts *counts);
/* vohip_frame_step for a caller that already holds the following image (a stored sequence, as the
 * loop of main_demo.m:352-375 reads it): while this frame is localised and mapped, `next_image` is
 * transferred, its pyramid is built and its corners are detected on streams of their own, and the
 * following frame's trackers are enqueued behind this frame's last kernel (the call returns without
 * waiting for them; the context's stream may still be busy with them afterwards).  The results are
 * those of vohip_frame_step, bit for bit; the
 * next call must pass `next_image` as its `image` (it is not transferred again), and a host
 * `next_image` must stay readable until that call returns.  next_image = NULL: exactly vohip_frame_step. */
int vohip_frame_step_ahead(vohip_frame *f, const uint8_t *image, int image_on_device, const uint8_t *next_image,
                           int next_on_device, const double *u3, int h, const double *u8, vohip_frame_pose *pose,
                           vohip_frame_counts *counts);
/* vohip_frame_step_ahead in two halves, for a host that has work of its own to do while the GPU
 * processes the frame (the next frame's draws, bookkeeping of the previous one): vohip_frame_begin
 * enqueues the frame and returns at once, vohip_frame_finish waits for it and reports exactly as
 * vohip_frame_step does (including VOHIP_SAMPLES_EXHAUSTED, answered with vohip_frame_retry).  u3 and
 * u8 are copied before vohip_frame_begin returns; the images are read until vohip_frame_finish (the
 * following image until the next frame's) returns. */
int vohip_frame_begin(vohip_frame *f, const uint8_t *image, int image_on_device, const uint8_t *next_image,
                      int next_on_device, const double *u3, int h, const double *u8);
int vohip_frame_finish(vohip_frame *f, vohip_frame_pose *pose, vohip_frame_counts *counts);
/* For a stored sequence (the loop of main_demo.m:352-375 reads its images from a directory): the n images
 * (pointers to host memory, or to device memory when the flag is set) are copied to the device and the
 * pyramids and corner lists of ALL of them are made by one batched launch sequence each (image = a grid
 * dimension).  A later vohip_frame_begin /
 * _step / _step_ahead whose `image` or `next_image` is one of these pointers then transfers nothing, builds no
 * pyramid and detects no corners for it; the results are those of the unprepared calls, bit for bit.  The
 * prepared data (about 2.7 MB per KITTI-shape image) lives until the next vohip_frame_prepare (n = 0 frees
 * it) or vohip_frame_destroy; the images themselves are not read again.  No counterpart in the reference. */
int vohip_frame_prepare(vohip_frame *f, const uint8_t *const *images, int n, int on_device);
int vohip_frame_sizes(vohip_frame *f, int *n_landmarks, int *n_candidates);
/* Copies of S.P (2xn), S.X(1:3,:) (3xn), S.X(4,:), S.C, S.F (2xm), S.T (12xm); any pointer may be NULL. */
int vohip_frame_state(vohip_frame *f, double *P_rc, double *X, double *ids, double *C_rc, double *F_rc, double *T12);

/* ---- a group of loops advanced in lockstep -------------------------------------------------------
 * main_eval.m:22-23 runs independent sequences (feature variant x repetition, each the loop body
 * main_demo.m:377-424 over its own images with its own random draws) one after the other.  A group holds
 * n_sequences such loops of one image shape, camera matrix and parameter set, and advances all of them by
 * one frame per call: every kernel of the frame is launched once, with the sequence as a grid dimension, so
 * the launches, their latencies and the host's enqueue work are shared by the sequences and the tracker
 * launch fills the GPU.  Each sequence's state, counts, ids and poses are those of its own vohip_frame_*
 * loop, bit for bit (a sequence reads nothing of another).
 *
 * The sequences are stored ones: vohip_group_prepare copies sequence `seq`'s n images (n <= max_frames;
 * image 0 is the one the loop is reset on, main_demo.m:342-344, images 1.. are the loop's frames) to the
 * device and makes all their pyramids and corner lists by batched launches, as vohip_frame_prepare does
 * (detector < 0: the group's detector and thresholds; else 0 Harris / 1 MinEigen / 2 FAST / 3 SURF / 4 BRISK with
 * min_quality and the detector's other threshold in the last argument -- FAST's and BRISK's 'MinContrast', SURF's
 * 'MetricThreshold': the feature variants of main_eval.m:165-174 may share a group, the loop only reads the
 * corner lists).
 * vohip_group_reset gives sequence `seq` its initial state (as vohip_frame_reset; the image is prepared
 * image 0).  The first reset after a frame was processed starts a new run of the group: sequences that
 * are not reset again stay idle in it.
 *
 * vohip_group_begin enqueues frame k = 1, 2, ... for every sequence that was reset, has not failed and holds
 * an image k; u3 = n_sequences x (3 x h) uniforms, u8 = n_sequences x (8 x fm_trials) (blocks of idle
 * sequences are not read).  vohip_group_finish waits for the frame and fills poses[n_sequences],
 * counts[n_sequences] and status[n_sequences] (the status vohip_frame_step would return for that sequence, or
 * VOHIP_SEQUENCE_IDLE; the call's own return value only reports failures of the runtime).  A sequence whose
 * status is VOHIP_SAMPLES_EXHAUSTED must be answered with vohip_group_retry (more hypotheses for that
 * sequence alone, as vohip_frame_retry) before the next vohip_group_begin; one that failed otherwise stays
 * idle for the rest of the run.  No bundle adjustment in a group (is_BA must be 0).
 * Memory: max_frames x (pyramid + image + corner lists, about 3.2 MB at 1241 x 376) per sequence. */
typedef struct vohip_group vohip_group;
int vohip_group_create(vohip_ctx *ctx, const double K[9], int rows, int cols, int colmajor,
                       const vohip_frame_params *params, int n_sequences, int max_frames, vohip_group **out);
int vohip_group_destroy(vohip_group *g);
int vohip_group_prepare(vohip_group *g, int seq, const uint8_t *const *images, int n, int on_device, int detector,
                        double min_quality, double min_contrast_or_metric_threshold);
/* vohip_group_prepare for a sequence that holds the same images as the prepared sequence from_seq (the
 * repetitions and feature variants of main_eval.m:22-23 run over one image sequence): images and pyramids
 * are copied on the device instead of being uploaded and built again; detector -2 copies the corner lists
 * too, otherwise they are detected as vohip_group_prepare would. */
int vohip_group_prepare_like(vohip_group *g, int seq, int from_seq, int detector, double min_quality,
                             double min_contrast_or_metric_threshold);
int vohip_group_reset(vohip_group *g, int seq, const double *X, const double *ids, const double *P_rc, int n,
                      const double T_W_C[16]);
int vohip_group_begin(vohip_group *g, const double *u3, int h, const double *u8);
int vohip_group_finish(vohip_group *g, vohip_frame_pose *poses, vohip_frame_counts *counts, int *status);
int vohip_group_retry(vohip_group *g, int seq, const double *u3, int h, vohip_frame_pose *pose,
                      vohip_frame_counts *counts);
int vohip_group_sizes(vohip_group *g, int seq, int *n_landmarks, int *n_candidates, int *frames_done);
int vohip_group_state(vohip_group *g, int seq, double *P_rc, double *X, double *ids, double *C_rc, double *F_rc,
                      double *T12);

/* The device-side half of VO_bundle_adjust.m:53-70 for a loop whose state lives on the GPU.  The
 * keyframe bookkeeping and the runBA call stay with the caller (which reads S.P, S.X with
 * vohip_frame_state after each frame); when a window was optimised, this call puts the result back:
 * X 3xn replaces S.X(1:3,:) (n = size(S.X, 2); :62-70, landmarks the depth test rejected keep their old
 * column) and T_W_C (4x4 column-major), the optimised pose of the current keyframe (:53-57), replaces the
 * pose that update_candidate.m:57 stamped on the candidates added by the frame just finished.  Call it
 * between frames (after vohip_frame_finish, before the next vohip_frame_begin).  Synchronises a side
 * stream only: trackers already enqueued for the following frame keep running. */
int vohip_frame_apply_ba(vohip_frame *f, const double *X, int n, const double T_W_C[16]);

/* ---- the bundle-adjustment window ---------------------------------------------------------------
 * Continuous_operation/VO_bundle_adjust.m:1-98 with the struct B of main_demo.m:288-303 kept by the
 * library: keyframe / normal-frame bookkeeping, window slide, global landmark ids -> columns of the
 * state vector on the host (the reference's find(row == column) look-ups as binary searches: id rows
 * ascend, update_landmarks.m:83-87), runBA (:51) and the re-refinement of the buffered normal frames
 * (:71-86, one batched launch) on the GPU.
 */
typedef struct vohip_window vohip_window;

typedef struct vohip_window_params {
    int window_size, keyframe_d;        /* main_BA.m:197-198 (5, 1); main_demo.m:288-289 */
    double min_depth, max_depth;        /* hyper_paras, VO_bundle_adjust.m:65 */
    int ba_max_iter;                    /* 21 = runBA.m:65-78 */
    double ba_ftol, ba_xtol;            /* lsqnonlin's defaults, 1e-6 */
    int refine_max_iter;                /* T_refinement.m:4 */
    double refine_ftol, refine_xtol;
} vohip_window_params;

void vohip_window_default_params(vohip_window_params *p);
/* K 3x3 column-major. */
int vohip_window_create(vohip_ctx *ctx, const double K[9], const vohip_window_params *params, vohip_window **out);
int vohip_window_destroy(vohip_window *w);
/* main_demo.m:288-303: the window holds the bootstrap keyframe; X 3xn (ids 1..n), P_rc 2xn (row; col),
 * T_W_C 4x4 column-major. */
int vohip_window_reset(vohip_window *w, const double *X, const double *P_rc, int n, const double T_W_C[16]);
/* [S, B] = VO_bundle_adjust(S, B, T_W_C, K, hyper_paras) for the frame whose landmarks are P_rc (S.P, 2xn),
 * X (S.X(1:3,:), 3xn, updated in place by :62-70) and ids (S.X(4,:), ascending), after update_landmarks
 * (ids at or above B.new_idx join the window first, update_landmarks.m:83-95).  When the frame is a
 * keyframe of a full window, *optimised = 1 and the poses VO_bundle_adjust.m:53-57,80-84 rewrites come
 * back as *n_poses pairs: offsets[k] <= 0 is the column of S.est_trans / S.est_rot counted from the
 * last one (0 = this frame), poses + 16 k its 4x4 column-major T_W_C; cap >= window_size +
 * (window_size - 1) keyframe_d entries.  A call that fails (ids that do not ascend, a landmark missing
 * from the window, a solver error) may leave the window half-way through the step: start it again with
 * vohip_window_reset. */
int vohip_window_step(vohip_window *w, const double *P_rc, double *X, const double *ids, int n, const double T_W_C[16],
                      int *optimised, int *n_poses, int *offsets, double *poses, int cap);
/* Copies of B.num_key, B.m = size(B.landmarks, 2), B.tau (6 x num_key) and B.landmarks (4 x m, [X; id]);
 * any pointer may be NULL (call once for the sizes). */
int vohip_window_state(vohip_window *w, int *num_key, int *m, double *tau, double *landmarks);
/* vohip_window_step for a loop whose state lives on the GPU (main_BA.m:288-291): S.P and S.X come down
 * in one round trip, and after an optimised window the landmarks and the keyframe's pose go back up
 * (vohip_frame_apply_ba).  Call it between frames; it runs on a side stream, beside the trackers
 * already enqueued for the following frame. */
int vohip_frame_bundle_adjust(vohip_frame *f, vohip_window *w, const double T_W_C[16], int *optimised, int *n_poses,
                              int *offsets, double *poses, int cap);

/* ---- trajectory alignment (evaluation) ---------------------------------------------------------
 * utils/exe9_BA/alignEstimateToGroundTruth.m:1-27 with the residual of alignError.m:1-17, as
 * utils/quantitative_eval.m:23-24 calls it: finds x = [twist of T_G_V (v; w); scale] minimising
 * |pp_G_C - (scale * R * p_V_C + t)|^2 from the identity twist and scale 1, and returns the aligned
 * points p_G_C.  pp_G_C, p_V_C, p_G_C 3xn column-major; x[7] and stats[3] = {iterations, initial
 * cost, final cost} may be NULL.  The reference's lsqnonlin call is replaced by the
 * Levenberg-Marquardt of vohip_pose_refine with analytic Jacobians. */
int vohip_align_sim3(vohip_ctx *ctx, const double *pp_G_C, const double *p_V_C, int n, int max_iter, double ftol,
                     double xtol, double *x, double *p_G_C, double *stats);
/* The alignments of utils/quantitative_eval.m:20-27 for one trajectory at once: checkpoint b aligns the first
 * lengths[b] points of the estimate to the first lengths[b] points of the ground truth (:23-24), k checkpoints
 * side by side in one launch, each exactly as vohip_align_sim3 computes it.  x (7 x k), stats (3 x k) may be
 * NULL; first_last (6 x k) receives the first and the last aligned point of every prefix, the two points
 * :25-26 read. */
int vohip_align_sim3_prefixes(vohip_ctx *ctx, const double *pp_G_C, const double *p_V_C, int n, int k, const int *lengths,
                              int max_iter, double ftol, double xtol, double *x, double *first_last, double *stats);

/* ---- relative pose (bootstrap) ---------------------------------------------------------------
 * Stands in for [orientation, location] = relativeCameraPose(E, cameraParams, pts1, pts2) at
 * Initialization/pose_estimation_init.m:29-33 (decomposition as in
 * utils/exe6_8point/decomposeEssentialMatrix.m:12-32; the candidate with the fewest triangulated
 * points behind either camera wins).  E, K 3x3 column-major; p1, p2 2xn (x; y) pixel coordinates
 * in images 1 and 2.  orientation 3x3 column-major and location[3] follow the toolbox convention:
 * x_cam2 = orientation * (x_cam1 - location), |location| = 1.  valid_fraction (may be NULL) is the
 * share of points in front of both cameras.  VOHIP_NO_MODEL when E has fewer than two non-zero
 * singular values. */
int vohip_relative_pose(vohip_ctx *ctx, const double *E, const double *K, const double *p1, const double *p2,
                        int n, double *orientation, double *location, double *valid_fraction);

/* ---- image pyramids and KLT tracking -------------------------------------------------------
 * Stand in for vision.PointTracker as the reference uses it: constructed at
 * main_demo.m:342-343,348-349 ('BlockSize',[21 21],'NumPyramidLevels',5,'MaxIterations',20,
 * 'MaxBidirectionalError',6), initialize(obj, points, I) at main_demo.m:344,418,423,
 * [points, validity] = obj(I) at main_demo.m:380 and update_landmarks.m:10.
 * Because the reference re-initialises both trackers on every frame, tracking is a pure
 * function of (previous image, current image, points): pyramids live in numbered slots of the
 * context so that one pyramid per frame serves both trackers and the next frame.
 * Images are uint8, rows x cols; colmajor != 0 means MATLAB's column-major storage (no
 * transpose is made: the tracker is symmetric in the axes).  Points are (x, y), 1-based, stored
 * n x 2 column-major exactly like MATLAB's N-by-2 matrices.
 */
int vohip_pyramid_build(vohip_ctx *ctx, int slot, const uint8_t *img, int rows, int cols, int colmajor,
                        int levels);
int vohip_pyramid_build_dev(vohip_ctx *ctx, int slot, const uint8_t *d_img, int rows, int cols,
                            int colmajor, int levels);
/* Pyramids of n images of one size in one set of launches (no counterpart in the reference, as above):
 * d_pyramids receives n x vohip_pyramid_floats(...) floats, each pyramid laid out as a slot's (levels back to
 * back, each starting on a multiple of 64 floats; level l has ceil(A / 2^l) x ceil(B / 2^l) samples).
 * Asynchronous. */
size_t vohip_pyramid_floats(int rows, int cols, int colmajor, int levels);
int vohip_pyramid_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                            int levels, float *d_pyramids);
/* Copies one level (float, fast-axis-major) to `out` (may be NULL to query the size only);
 * A = fast-axis length, B = slow-axis length. */
int vohip_pyramid_level(vohip_ctx *ctx, int slot, int level, float *out, int *A, int *B);

/* Tracks n points from the image in slot_prev to the image in slot_cur.  valid[i] = 1 iff the
 * normal matrix is well conditioned in both directions, the new position lies inside the image
 * and the forward-backward distance is <= max_bidir_error.  fb_err may be NULL.
 * block_size: 21 or 31 (BlockSize [b b]). */
int vohip_klt_track(vohip_ctx *ctx, int slot_prev, int slot_cur, const double *pts_xy, int n,
                    int block_size, int max_iter, double max_bidir_error, double *out_xy,
                    uint8_t *valid, double *fb_err);
int vohip_klt_track_dev(vohip_ctx *ctx, int slot_prev, int slot_cur, const double *d_pts_xy, int n,
                        int block_size, int max_iter, double max_bidir_error, double *d_out_xy,
                        uint8_t *d_valid, double *d_fb_err);

/* Work counters of the tracker since the last reset (no reference counterpart; measurement only):
 * tracks processed and Gauss-Newton iterations executed over all levels and both directions.
 * Synchronises the device. */
int vohip_klt_stats(vohip_ctx *ctx, long long *tracks, long long *iterations, int reset);
/* With the option "klt_time": the tracker launches since the last reset (at most 1024 are recorded) -- the sum
 * and the longest of their durations in milliseconds, as the event pair around each launch measured it on the
 * launch's own stream, and their number.  Synchronises the device.  No reference counterpart. */
int vohip_klt_time(vohip_ctx *ctx, double *total_ms, double *longest_ms, int *launches, int reset);

/* ---- fundamental-matrix RANSAC -----------------------------------------------------------
 * Stands in for [F, inliersIndex] = estimateFundamentalMatrix(p1, p2, 'NumTrials', 500, 'Method',
 * 'RANSAC', 'DistanceThreshold', 5e-2) at update_landmarks.m:13-14 (Sampson distance; the toolbox
 * default Confidence is 99) and for the 2000-trial, 1e-2, Confidence 99.9 call of the bootstrap,
 * Initialization/pose_estimation_init.m:7-12.  confidence is in percent, in (0, 100).
 * p1, p2 2xn (one column per point, any consistent axis order); samples
 * 8 x num_trials 0-based index sets drawn by the caller (the toolbox draws them from MATLAB's
 * global RNG); mask n bytes; best_count the inlier count of the winning trial; trials_run (may
 * be NULL) the trials the adaptive budget consumed; F_best (may be NULL) the winning matrix,
 * row-major 3x3, for points in the axis order given.  Needs n >= 8 (VOHIP_TOO_FEW_POINTS; the
 * toolbox raises an error there too).
 */
int vohip_fmatrix_ransac(vohip_ctx *ctx, const double *p1, const double *p2, int n, const int32_t *samples,
                         int num_trials, double threshold, double confidence, uint8_t *mask, int *best_count,
                         int *trials_run, double *F_best);
/* d_p1, d_p2, d_samples, d_mask on the device; best_count, trials_run, F_best on the host.
 * Synchronises (the trial budget is replayed on the host). */
int vohip_fmatrix_ransac_dev(vohip_ctx *ctx, const double *d_p1, const double *d_p2, int n,
                             const int32_t *d_samples, int num_trials, double threshold, double confidence,
                             uint8_t *d_mask, int *best_count, int *trials_run, double *F_best);

/* ---- bundle adjustment and pose refinement ------------------------------------------------
 * hidden_state = [twists of T_W_C (6 x F, [v; w]); landmarks (3 x m)], observations =
 * [F; m; per frame: n_i; n_i (row, col) pairs; n_i 1-based landmark indices] -- exactly the
 * vectors VO_bundle_adjust.m:43-50 builds.  F <= 8.
 */

/* utils/exe9_BA/baError.m:2-65: r_out 2 x Nobs reprojection residuals (keypoint - projection,
 * u then v).  n_obs_out (may be NULL) receives Nobs. */
int vohip_ba_error(vohip_ctx *ctx, const double *state, int state_len, const double *obs, int obs_len,
                   const double K[9], double *r_out, int *n_obs_out);

/* utils/exe9_BA/runBA.m:2-88: minimises the sum of squared residuals over all twists and over
 * the landmarks seen in at least 2 frames (the others stay fixed, runBA.m:45-49); state is
 * updated in place.  The reference's two lsqnonlin calls (1 + 20 iterations; the second one only
 * if the first took under 0.5 s of wall-clock, runBA.m:64-87) are replaced by one
 * Levenberg-Marquardt run of at most max_iter accepted steps (21 = the reference's budget) with
 * tolerances ftol (cost decrease relative to 1 + cost) and xtol (step relative to 1 + |x|).
 * stats (may be NULL): [accepted iterations, initial cost, final cost, final lambda]. */
int vohip_ba_run(vohip_ctx *ctx, double *state, int state_len, const double *obs, int obs_len,
                 const double K[9], int max_iter, double ftol, double xtol, double *stats);

/* utils/T_refinement.m:1-21: refines T_W_C (4x4, in place) by minimising keypoints - K T_C_W X
 * over the 6 twist coordinates; kp_uv 2xN in (u, v) order (callers pass flipud(S.P),
 * main_demo.m:399), X 3xN.  max_iter 100 in the reference.  stats (may be NULL):
 * [accepted iterations, initial cost, final cost]. */
int vohip_pose_refine(vohip_ctx *ctx, double T_W_C[16], const double *kp_uv, const double *X, int n,
                      const double K[9], int max_iter, double ftol, double xtol, double *stats);
/* T_refinement for k independent frames in one launch and one synchronisation, as VO_bundle_adjust.m:71-86
 * re-refines the buffered normal frames after a window was optimised (the reference calls T_refinement once per
 * frame there; the frames do not depend on each other).  Frame b has n[b] points; its columns follow frame
 * b - 1's in kp_uv (2 x sum n) and X (3 x sum n); T holds k 4x4 matrices (in/out), stats 3k doubles or NULL.
 * The results are those of k calls of vohip_pose_refine, bit for bit. */
int vohip_pose_refine_batch(vohip_ctx *ctx, int k, double *T_W_C, const double *kp_uv, const double *X, const int *n,
                            const double K[9], int max_iter, double ftol, double xtol, double *stats);
/* Device-resident form (d_T 16 doubles in/out, d_stats 3 doubles or NULL).  Asynchronous. */
int vohip_pose_refine_dev(vohip_ctx *ctx, double *d_T_W_C, const double *d_kp_uv, const double *d_X, int n,
                          const double K[9], int max_iter, double ftol, double xtol, double *d_stats);

/* ---- corner detection ----------------------------------------------------------------------
 * Stands in for genKeypoints.m:1-33 with method 'Harris' or 'MinEigen': detectHarrisFeatures /
 * detectMinEigenFeatures(image, 'MinQuality', q [, 'FilterSize', 5]) followed by extractFeatures,
 * of which the continuous-operation loop only uses valid_key_points.Location
 * (update_candidate.m:10,26,57) -- i.e. the detections that survive extractFeatures' border drop.
 *   image      uint8 rows x cols, colmajor != 0 for MATLAB storage
 *   border     detections closer than this to an image edge are dropped (extractFeatures' margin)
 *   method     0 = Harris, 1 = MinEigen
 *   xy_out     n x 2 single, (x, y) 1-based, column-major -- cornerPoints.Location
 *   metric_out n single -- cornerPoints.Metric
 *   cap        capacity of the two outputs (<= 16384); VOHIP_CAPACITY_EXCEEDED if more were found
 * Output order: ascending linear index of the pixel in storage order (for a column-major image
 * that is the order of MATLAB's find on the corner mask).
 */
int vohip_detect_corners(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor,
                         double min_quality, int filter_size, int border, int method, float *xy_out,
                         float *metric_out, int cap, int *n_out);
/* Device-resident form: d_xy is cap x 2 (x column then y column, cap-strided), d_metric cap,
 * d_count one int that receives the total number found (may exceed cap).  Asynchronous. */
int vohip_detect_corners_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor,
                             double min_quality, int filter_size, int border, int method, float *d_xy,
                             float *d_metric, int cap, int *d_count);
/* vohip_detect_corners_dev for n images of one size at once (d_imgs: image after image), for a caller that
 * holds a whole stored sequence (main_eval.m reads one image per iteration of its loop; no counterpart there):
 * d_xy n x (cap x 2), d_metric n x cap, d_counts n ints.  Every image's lists are those of a single call, bit
 * for bit; one launch of each of the three kernels serves the batch, so the GPU is filled instead of waiting
 * on a 0.47 MB image's latency.  Harris and MinEigen.  Asynchronous. */
int vohip_detect_corners_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                   double min_quality, int filter_size, int border, int method, float *d_xy,
                                   float *d_metric, int cap, int *d_counts);

/* Stands in for genKeypoints.m:1-33 with method 'FAST' (dispatch table genKeypoints.m:19; main_eval.m:172-173
 * passes {'MinContrast', 0.11}): detectFASTFeatures(image, 'MinContrast', c, 'MinQuality', q) followed by
 * extractFeatures' border drop.  The segment test of Rosten & Drummond on the 16-pixel circle with 9
 * contiguous pixels, threshold t = round(255 c); Metric = the largest threshold the pixel still passes;
 * strict 3x3 non-maximum suppression on the metric; Metric > q * max(Metric); integer pixel positions.
 * The other arguments, the output layout and order are those of vohip_detect_corners. */
int vohip_detect_fast(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double min_contrast,
                      double min_quality, int border, float *xy_out, float *metric_out, int cap, int *n_out);
int vohip_detect_fast_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double min_contrast,
                          double min_quality, int border, float *d_xy, float *d_metric, int cap, int *d_count);
/* vohip_detect_fast_dev for n images of one size stored one after the other, with the output layout of
 * vohip_detect_corners_batch_dev: the lists of n single calls, bit for bit (main_eval.m:172-173's variant over a
 * stored sequence; no counterpart in the reference, which reads one image per iteration).  Asynchronous. */
int vohip_detect_fast_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                double min_contrast, double min_quality, int border, float *d_xy, float *d_metric, int cap,
                                int *d_counts);

/* Stands in for genKeypoints.m:1-33 with method 'SURF', the reference's default detector (dispatch table
 * genKeypoints.m:16; main_demo.m:25; main_eval.m:151,167 pass {'MetricThreshold', 10 / 200}):
 * detectSURFFeatures(image, 'MetricThreshold', t) with the toolbox defaults NumOctaves 3, NumScaleLevels 4.
 * Fast-Hessian detector of Bay et al. (CVIU 2008): box-filter second derivatives of side 9 15 21 27 |
 * 15 27 39 51 | 27 51 75 99 sampled every 1 | 2 | 4 pixels from an integral image, Metric = Dxx Dyy - 0.81 Dxy^2
 * on the 0..255 intensity scale with every box normalised by its area, strict maxima of the 3 x 3 x 3 sample
 * blocks of the two middle levels of every octave above MetricThreshold, refined by the quadratic through the
 * block (dropped when an offset exceeds one sample).  Only the positions are used by the pipeline
 * (genKeypoints.m:32 returns valid_points; the loop reads .Location); scale_out / d_scale (may be NULL)
 * receives SURFPoints' Scale.  Points closer than `border` pixels to an edge are dropped as for the other
 * detectors.  Output layout as vohip_detect_corners; order: octave, level, then pixel order.  At most 2^24
 * pixels.  The batch call serves n images stored one after the other (lists of n single calls, bit for bit). */
int vohip_detect_surf(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double metric_threshold,
                      int border, float *xy_out, float *metric_out, float *scale_out, int cap, int *n_out);
int vohip_detect_surf_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double metric_threshold,
                          int border, float *d_xy, float *d_metric, float *d_scale, int cap, int *d_count);
int vohip_detect_surf_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                double metric_threshold, int border, float *d_xy, float *d_metric, int cap, int *d_counts);

/* Stands in for genKeypoints.m:1-33 with method 'BRISK' (dispatch table genKeypoints.m:20; main_eval.m:170-171 passes
 * {'MinQuality', 2e-3}): detectBRISKFeatures(image, 'MinContrast', c, 'MinQuality', q, 'NumOctaves', n) with the toolbox
 * defaults 0.2, 0.1 and 4.  The scale-space FAST detector of Leutenegger et al. (ICCV 2011): 2 n layers (octaves by
 * half-sampling, intra-octaves from a 2/3 resampling of the image) at the scales 1, 1.5, 2, 3, 4, 6, 8, 12; the FAST 9-16
 * score of every layer's pixels (threshold round(255 c)); a keypoint is a strict maximum of its 8 neighbours in its layer
 * whose score also exceeds the scores of the layers above and below inside its 3 x 3 patch and q times the largest score of
 * all layers; sub-pixel position by a parabola per axis, mapped to the image; Scale (scale_out, may be NULL) by a parabola
 * through the three layers' scores (the layer below the image: FAST 5-8 at scale 2/3), in units of the image's pixel.
 * Only the positions are used by the pipeline.  Points closer than `border` pixels to an edge are dropped as for the other
 * detectors.  Output layout as vohip_detect_corners; order: layer, then pixel order.  At most 2^26 pixels.  The batch
 * call serves n images stored one after the other (lists of n single calls, bit for bit). */
int vohip_detect_brisk(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, double min_contrast,
                       double min_quality, int num_octaves, int border, float *xy_out, float *metric_out, float *scale_out,
                       int cap, int *n_out);
int vohip_detect_brisk_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, double min_contrast,
                           double min_quality, int num_octaves, int border, float *d_xy, float *d_metric, float *d_scale,
                           int cap, int *d_count);
int vohip_detect_brisk_batch_dev(vohip_ctx *ctx, const uint8_t *d_imgs, int n, int rows, int cols, int colmajor,
                                 double min_contrast, double min_quality, int num_octaves, int border, float *d_xy,
                                 float *d_metric, int cap, int *d_counts);

/* Stands in for extractFeatures(image, SURFPoints) as genKeypoints.m:32 calls it (64-element SURF descriptors,
 * 'Upright' false), the features the 'Des_match' bootstrap branch matches (matching_init.m:26-29).  A closed toolbox
 * function; built as section 4 of Bay et al. describes it: the dominant direction of the Haar-wavelet responses
 * (side 4 s) at the 109 sample points within 6 s of the point, Gaussian-weighted (sigma 2 s), taken from 42 sliding
 * windows of pi/3; then a square of side 20 s turned to that direction, 4 x 4 subregions of 5 x 5 samples each,
 * Haar responses of side 2 s weighted by a Gaussian (sigma 3.3 s) about the point, (sum dx, sum dy, sum |dx|,
 * sum |dy|) per subregion, the 64 numbers normalised to unit length.  s = Scale; sample positions and sizes rounded
 * to pixels; boxes clipped to the image, so every point gets a descriptor (no point is dropped).
 * xy: n x 2 (x column then y column, leading dimension n / ld), 1-based as SURFPoints.Location; scale: n
 * (SURFPoints.Scale); features: n x 64 with a point's 64 numbers contiguous (the transpose of MATLAB's N x 64 array
 * in memory); orientation (may be NULL): n angles in radians, atan2 of the dominant direction in image coordinates
 * (x right, y down).  upright != 0 skips the direction.  At most 2^24 pixels; 0 < Scale <= 1e4 and positions within 1e6
 * pixels of the image (checked by the host form; the device form trusts its caller). */
int vohip_extract_surf_features(vohip_ctx *ctx, const uint8_t *img, int rows, int cols, int colmajor, const float *xy,
                                const float *scale, int n, int upright, float *features_out, float *orientation_out);
int vohip_extract_surf_features_dev(vohip_ctx *ctx, const uint8_t *d_img, int rows, int cols, int colmajor, const float *d_xy,
                                    int ld, const float *d_scale, int n, int upright, float *d_features,
                                    float *d_orientation);

/* Stands in for indexPairs = matchFeatures(features1, features2) with the toolbox defaults for non-binary features
 * (matching_init.m:29: 'Exhaustive', metric SSD, 'MatchThreshold' 1.0 percent, 'MaxRatio' 0.6, 'Unique' false): for
 * every feature of the first set its nearest neighbour in the second by sum of squared differences (the lowest index
 * of equals); the pair is kept when that distance is at most match_threshold percent of 4, the largest distance
 * between unit vectors, and at most max_ratio^2 times the second nearest distance (the ratio test on Euclidean
 * distances).  f0: n0 x dim, f1: n1 x dim, a feature's numbers contiguous.  pairs: up to n0 rows (i, j), 0-based,
 * in the order of i; metric: their distances; *n_pairs_out / *d_count: how many. */
int vohip_match_features(vohip_ctx *ctx, const float *f0, int n0, const float *f1, int n1, int dim, double match_threshold,
                         double max_ratio, int *pairs_out, float *metric_out, int *n_pairs_out);
int vohip_match_features_dev(vohip_ctx *ctx, const float *d_f0, int n0, const float *d_f1, int n1, int dim,
                             double match_threshold, double max_ratio, int *d_pairs, float *d_metric, int *d_count);

#ifdef __cplusplus
}
#endif
#endif
