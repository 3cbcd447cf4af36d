/*
 * vo_oracle.h -- CPU restatement (plain C, fp64) of the reference MATLAB hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls
 * this code; it is the checker the HIP path is compared against (tests/, the smoke
 * check and the cpu_baseline leg of bench.py).
 *
 * PARITY UNPINNED versus MATLAB itself: the reference ships no tests, golden vectors
 * or fixtures and MATLAB/Octave are not available, so every function here is pinned
 * only by (a) analytic known-answer tests and (b) independent numpy/scipy
 * implementations (tests/test_oracle_*.py).  Each function cites the reference
 * file:line it restates.  Arrays are column-major, exactly as MATLAB stores them.
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* utils/exe7_Ransac/solveQuartic.m:26-66.  factors[5] = A..E, roots as (re, im) pairs. */
void vo_solve_quartic(const double factors[5], double roots_re[4], double roots_im[4]);

/* utils/exe7_Ransac/p3p.m:41-217.  world 3x3 (columns = points), bearings 3x3 (unit
 * columns).  poses 3x16 column-major = 4 x [C | R_W_C]; only the real part is kept
 * (ransacLocalization.m:64-65 discards the imaginary part).  All zeros when the world
 * points are collinear (p3p.m:56-58). */
void vo_p3p(const double world[9], const double bearings[9], double poses[48]);

/* utils/exe1/projectPoints.m:1-16 + distortPoints.m:1-18 with D = 0 (as every call
 * site on the hot path uses it).  p 3xN -> out 2xN. */
void vo_project_points(const double *p, int n, const double K[9], double *out);

/* Per-hypothesis part of ransacLocalization.m:46-101: sample -> bearings -> p3p ->
 * decode 4 poses -> score all N points for each -> pick the solution as :95 does.
 * kp_uv is 2xN in (u,v) order (i.e. after the flipud at :34).  idx3 is 0-based.
 * Outputs: counts[4] inlier count per solution, *best_sol the index kept by the
 * strict '>' rule of :95, Rt[12] = [R_C_W (col-major 9) | t_C_W (3)] of that solution,
 * mask[N] (may be NULL) inlier mask of that solution. */
void vo_ransac_hypothesis(const double *kp_uv, const double *X, int n, const double K[9],
                          const int32_t idx3[3], int counts[4], int *best_sol,
                          double Rt[12], uint8_t *mask);

/* utils/exe7_Ransac/ransacLocalization.m:1-157 with use_p3p = tweaked_for_more =
 * adaptive = true.  kp_rc 2xN (row; col), X 3xN, K 3x3 column-major.  The reference
 * draws its index triples from MATLAB's global RNG (datasample, :46-47); here the
 * caller supplies them: samples 3 x hcap, 0-based.  Returns the number of iterations
 * executed (i-1 at :155); -1 if the supplied triples ran out before the adaptive
 * stop.  found = 0 reproduces the empty R/t of :138-140.  hist arrays need hcap
 * entries.  per_hyp_counts (may be NULL) receives the winning-solution inlier count
 * of every executed iteration. */
int vo_ransac_localization(const double *kp_rc, const double *X, int n, const double K[9],
                           const int32_t *samples, int hcap,
                           double R_C_W[9], double t_C_W[3], uint8_t *best_mask, int *found,
                           double *max_num_inliers_history, double *num_iteration_history,
                           int32_t *per_hyp_counts);

/* The adaptive iteration bound of ransacLocalization.m:118-129 for a given running
 * maximum inlier count and point count. */
double vo_ransac_num_iterations(int max_num_inliers, int n);

/* ---- dlt_oracle.c ---------------------------------------------------------------------- */

/* M_tilde = estimatePoseDLT(p, P, K), utils/exe7_Ransac/estimatePoseDLT.m:1-60: p_uv n x 2 (u, v) and P n x 3
 * as rows, K column-major; M 3 x 4 row-major [R_tilde | alpha t].  Returns 0, or -1 when n < 6. */
int vo_estimate_pose_dlt(const double *p_uv, const double *P, int n, const double K[9], double M[12]);

/* utils/exe7_Ransac/ransacLocalization.m:1-157 with use_p3p = false (k = 6 samples per hypothesis, one pose per
 * iteration, refit on the best hypothesis' inliers at :145-151); arguments as vo_ransac_localization, samples
 * 6 x hcap.  M_hyp (may be NULL): the [R | t] (row-major 3 x 4) of every executed iteration. */
int vo_ransac_localization_dlt(const double *kp_rc, const double *X, int n, const double K[9], const int32_t *samples,
                               int hcap, double R_C_W[9], double t_C_W[3], uint8_t *best_mask, int *found,
                               double *max_hist, double *iter_hist, int32_t *per_hyp_counts, double *M_hyp);
double vo_ransac_num_iterations_dlt(int max_num_inliers, int n);

/* ---- geometry_oracle.c ---------------------------------------------------------------- */

/* utils/exe9_BA/twist2HomogMatrix.m:1-16 (expm of the se(3) matrix; twist = [v; w]). */
void vo_twist2homog(const double twist[6], double H[16]);
/* utils/exe9_BA/HomogMatrix2twist.m:1-25 (logm; rotation angle in [0, pi]). */
void vo_homog2twist(const double H[16], double twist[6]);

/* utils/exe6_8point/linearTriangulation.m:28-39 for one point. */
void vo_triangulate_one(const double p1[3], const double p2[3], const double *M1, const double *M2,
                        double P[4]);
/* linearTriangulation.m:1-41; M 3x4 shared, or 3x4xn when the per_point flag is set (the way
 * update_landmarks.m:57-61 calls it). */
void vo_linear_triangulation(const double *p1, const double *p2, const double *M1, const double *M2,
                             int per_point_M1, int per_point_M2, int n, double *P);
/* utils/linearTriangulation_refinement.m:1-22. */
void vo_triangulation_refine_one(const double P_init[3], const double *p1, const double *p2,
                                 const double *M1, const double *M2, int max_iter, double P_out[3]);
void vo_triangulation_refinement(const double *P_init, const double *p1, const double *p2, int ldp,
                                 const double *M1, const double *M2, int n, double *P);
/* Continuous_operation/update_candidate.m:12-27. */
void vo_candidate_filter(const float *new_xy, int n_new, const double *C_rc, int nC, const double *P_rc,
                         int nP, double r, uint8_t *keep);
/* Continuous_operation/update_landmarks.m:27-97 (angle test, triangulation, depth filter). */
void vo_candidate_triangulate(const double *cur_rc, const double *first_rc, const double *T_first, int n,
                              const double T_W_C[16], const double K[9], double angle_threshold_deg,
                              double min_depth, double max_depth, int is_BA, uint8_t *status,
                              double *X_out, double *angles_deg);

/* ---- klt_oracle.c ---------------------------------------------------------------------- */

/* This project's definition of the pyramidal KLT that stands in for vision.PointTracker
 * (main_demo.m:342-343,380; update_landmarks.m:10); see the header of klt_oracle.c.
 * Images are img[b * A + a] (a = fast axis); a column-major MATLAB HxW image has A = H, B = W,
 * a = row, b = column.  Coordinates are 0-based. */
void vo_pyr_down(const float *src, int A, int B, float *dst);
void vo_pyramid_level(const uint8_t *img, int A, int B, int level, float *out, int *Aout, int *Bout);
int vo_set_threads(int n);
void vo_klt_track(const uint8_t *prev, const uint8_t *cur, int A, int B, const double *pts, int n, int W,
                  int levels, int max_iter, double max_bidir_error, double *out, uint8_t *valid,
                  double *fb_err);

/* ---- ba_oracle.c ------------------------------------------------------------------------- */

/* utils/exe9_BA/baError.m:2-65.  observations as the reference serialises them
 * ([F; m; per frame: n_i; n_i (row, col) pairs; n_i 1-based landmark indices]).  r_out 2 x Nobs
 * (u residual, v residual per observation).  Returns Nobs, -1 if the vector is malformed. */
int vo_ba_error(const double *state, const double *obs, int obs_len, const double K[9], double *r_out);
/* utils/exe9_BA/runBA.m:2-88 with this project's Levenberg-Marquardt (see ba_oracle.c). */
int vo_run_ba(double *state, const double *obs, int obs_len, const double K[9], int max_iter, double ftol,
              double xtol, double *stats);
/* utils/T_refinement.m:1-21 with the same Levenberg-Marquardt on the 6 twist coordinates. */
void vo_pose_refine(double T[16], const double *kp_uv, const double *X, int n, const double K[9],
                    int max_iter, double ftol, double xtol, double *stats);

/* ---- fmatrix_oracle.c --------------------------------------------------------------------- */

/* This project's definition of the fundamental-matrix RANSAC that stands in for
 * estimateFundamentalMatrix(..., 'NumTrials', 500, 'Method', 'RANSAC', 'DistanceThreshold', 5e-2)
 * at update_landmarks.m:13-14; see the header of fmatrix_oracle.c. */
int vo_fmatrix_8point(const double *p1, const double *p2, const int32_t idx[8], const double nrm[6], double F[9]);
int vo_fmatrix_count(const double *p1, const double *p2, int n, const double F[9], double thr, uint8_t *mask);
double vo_fmatrix_trial_budget(int inliers, int n, int num_trials, double confidence);
int vo_fmatrix_ransac(const double *p1, const double *p2, int n, const int32_t *samples, int num_trials,
                      double thr, double confidence, uint8_t *mask, int *best_count, double Fbest[9],
                      int32_t *counts);

/* ---- bootstrap_oracle.c -------------------------------------------------------------------- */

/* This project's definition of the relative-pose step that stands in for relativeCameraPose at
 * Initialization/pose_estimation_init.m:29-33; see the header of bootstrap_oracle.c. */
int vo_decompose_essential(const double E[9], double R[2][9], double u3[3]);
int vo_relative_pose(const double E[9], const double K[9], const double *p1, const double *p2, int n,
                     double orientation[9], double location[3], double *valid_fraction);

/* ---- align_oracle.c ------------------------------------------------------------------------ */

/* utils/exe9_BA/alignError.m:1-17. */
void vo_align_error(const double x[7], const double *pp_G_C, const double *p_V_C, int n, double *error);
/* utils/exe9_BA/alignEstimateToGroundTruth.m:1-27 with this project's Levenberg-Marquardt. */
void vo_align_sim3(const double *pp_G_C, const double *p_V_C, int n, int max_iter, double ftol, double xtol,
                   double x_out[7], double *p_G_C, double *stats);

/* ---- harris_oracle.c ------------------------------------------------------------------- */

/* This project's definition of the corner detector that stands in for detectHarrisFeatures /
 * detectMinEigenFeatures + extractFeatures' border drop (genKeypoints.m:16-18,31-32); see the
 * header of harris_oracle.c.  method 0 = Harris, 1 = MinEigen. */
void vo_gaussian_taps(int filter_size, float *w);
void vo_corner_response(const uint8_t *img, int A, int B, int filter_size, int method, float *R);
int vo_detect_corners(const uint8_t *img, int A, int B, double min_quality, int filter_size, int border,
                      int method, float *ab_out, float *metric_out, int cap);
/* This project's definition of the FAST detector that stands in for detectFASTFeatures (genKeypoints.m:19);
 * see harris_oracle.c.  S: A x B segment-test scores (0 where the pixel is no corner at `threshold`). */
void vo_fast_score(const uint8_t *img, int A, int B, int threshold, int *S);
int vo_detect_fast(const uint8_t *img, int A, int B, double min_contrast, double min_quality, int border,
                   float *ab_out, float *metric_out, int cap);

/* ---- surf_oracle.c --------------------------------------------------------------------- */

/* This project's definition of the SURF detector that stands in for detectSURFFeatures (genKeypoints.m:16);
 * see surf_oracle.c.  R: the response map of (octave 0..2, level 0..3), ceil(A / 2^octave) x ceil(B / 2^octave). */
void vo_surf_response(const uint8_t *img, int A, int B, int octave, int level, float *R);
int vo_detect_surf(const uint8_t *img, int A, int B, double metric_threshold, int border, float *ab_out,
                   float *metric_out, float *scale_out, int cap);
/* This project's definitions of the SURF descriptor (extractFeatures on SURFPoints, genKeypoints.m:32) and of
 * matchFeatures with its defaults (matching_init.m:29); see surf_oracle.c. */
void vo_surf_descriptors(const uint8_t *img, int A, int B, int swap_xy, const float *xy, const float *scale, int n,
                         int upright, float *features, float *orientation);
int vo_match_features(const float *f0, int n0, const float *f1, int n1, int dim, double match_threshold, double max_ratio,
                      int *pairs, float *metric);

/* ---- brisk_oracle.c -------------------------------------------------------------------- */

/* This project's definition of the BRISK detector that stands in for detectBRISKFeatures (genKeypoints.m:20); see
 * brisk_oracle.c.  Layers 0..7 in the order of their scale 1, 1.5, 2, 3, 4, 6, 8, 12. */
void vo_brisk_half(const uint8_t *src, int A, int B, uint8_t *dst);
void vo_brisk_twothirds(const uint8_t *src, int A, int B, uint8_t *dst);
void vo_brisk_layer(const uint8_t *img, int A, int B, int k, uint8_t *out, int *Ak, int *Bk);
void vo_fast58_score(const uint8_t *img, int A, int B, int *S);
int vo_detect_brisk(const uint8_t *img, int A, int B, double min_contrast, double min_quality, int num_octaves, int border,
                    float *ab_out, float *metric_out, float *scale_out, int cap);

#ifdef __cplusplus
}
#endif
#endif
