"""ctypes binding of libvohip.so -- the one place the C ABI of include/vohip.h is declared to
Python.  There is no CPU fallback: if the library is missing or no GPU is visible, calls fail.
"""
import atexit
import ctypes as C
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# VOHIP_LIB: another build of the library (kernel tuning experiments)
LIB_PATH = os.environ.get("VOHIP_LIB") or os.path.join(HERE, "lib", "libvohip.so")

VOHIP_OK = 0
VOHIP_ERR_INVALID_ARGUMENT = 1
VOHIP_ERR_HIP = 2
VOHIP_NO_MODEL = 3
VOHIP_SAMPLES_EXHAUSTED = 4
VOHIP_TOO_FEW_POINTS = 5
VOHIP_ERR_NO_DEVICE = 6
VOHIP_CAPACITY_EXCEEDED = 7
VOHIP_SEQUENCE_IDLE = 8

STATUS_NAMES = {
    0: "VOHIP_OK", 1: "VOHIP_ERR_INVALID_ARGUMENT", 2: "VOHIP_ERR_HIP", 3: "VOHIP_NO_MODEL",
    4: "VOHIP_SAMPLES_EXHAUSTED", 5: "VOHIP_TOO_FEW_POINTS", 6: "VOHIP_ERR_NO_DEVICE", 7: "VOHIP_CAPACITY_EXCEEDED",
    8: "VOHIP_SEQUENCE_IDLE",
}

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_int_p = C.POINTER(C.c_int)



class FrameParams(C.Structure):
    """vohip_frame_params"""
    _fields_ = [("klt_block", C.c_int), ("klt_levels", C.c_int), ("klt_max_iter", C.c_int),
                ("klt_max_bidir_error", C.c_double), ("detector", C.c_int), ("min_quality", C.c_double),
                ("filter_size", C.c_int), ("border", C.c_int), ("r_discard_redundant", C.c_double),
                ("angle_threshold_deg", C.c_double), ("min_depth", C.c_double), ("max_depth", C.c_double),
                ("refine_pose", C.c_int), ("refine_max_iter", C.c_int), ("fm_trials", C.c_int),
                ("fm_threshold", C.c_double), ("fm_confidence", C.c_double), ("cap_landmarks", C.c_int),
                ("cap_candidates", C.c_int), ("cap_corners", C.c_int), ("max_hypotheses", C.c_int),
                ("slot_a", C.c_int), ("slot_b", C.c_int), ("slot_c", C.c_int), ("is_BA", C.c_int),
                ("min_contrast", C.c_double), ("metric_threshold", C.c_double), ("brisk_octaves", C.c_int)]


class FramePose(C.Structure):
    """vohip_frame_pose"""
    _fields_ = [("T_W_C", C.c_double * 16), ("Rt", C.c_double * 12), ("refine_stats", C.c_double * 3),
                ("num_iterations", C.c_double), ("n_tracked", C.c_int), ("n_inliers", C.c_int),
                ("max_num_inliers", C.c_int), ("n_iter", C.c_int), ("found", C.c_int), ("exhausted", C.c_int),
                ("evaluated", C.c_int)]


class FrameCounts(C.Structure):
    """vohip_frame_counts"""
    _fields_ = [("n_landmarks", C.c_int), ("n_candidates", C.c_int), ("num_new", C.c_int), ("n_detected", C.c_int),
                ("n_kept", C.c_int), ("n_candidates_tracked", C.c_int), ("n_candidates_inliers", C.c_int),
                ("fm_trials_run", C.c_int), ("overflow", C.c_int), ("host_enqueue_us", C.c_double),
                ("host_wait_us", C.c_double)]


class WindowParams(C.Structure):
    """vohip_window_params"""
    _fields_ = [("window_size", C.c_int), ("keyframe_d", C.c_int), ("min_depth", C.c_double), ("max_depth", C.c_double),
                ("ba_max_iter", C.c_int), ("ba_ftol", C.c_double), ("ba_xtol", C.c_double), ("refine_max_iter", C.c_int),
                ("refine_ftol", C.c_double), ("refine_xtol", C.c_double)]


# name -> (restype, argtypes); mirrors include/vohip.h one to one
SIGNATURES = {
    "vohip_window_default_params": (None, [C.POINTER(WindowParams)]),
    "vohip_window_create": (C.c_int, [C.c_void_p, c_double_p, C.POINTER(WindowParams), C.POINTER(C.c_void_p)]),
    "vohip_window_destroy": (C.c_int, [C.c_void_p]),
    "vohip_window_reset": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, c_double_p]),
    "vohip_window_step": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p, c_int_p, c_int_p,
                                     c_int_p, c_double_p, C.c_int]),
    "vohip_window_state": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_double_p, c_double_p]),
    "vohip_frame_bundle_adjust": (C.c_int, [C.c_void_p, C.c_void_p, c_double_p, c_int_p, c_int_p, c_int_p, c_double_p,
                                             C.c_int]),
    "vohip_klt_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int]),
    "vohip_klt_time": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_int_p, C.c_int]),
    "vohip_frame_default_params": (None, [C.POINTER(FrameParams)]),
    "vohip_frame_create": (C.c_int, [C.c_void_p, c_double_p, C.c_int, C.c_int, C.c_int, C.POINTER(FrameParams),
                                      C.POINTER(C.c_void_p)]),
    "vohip_frame_destroy": (C.c_int, [C.c_void_p]),
    "vohip_frame_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int,
                                     c_double_p]),
    "vohip_frame_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_double_p, C.c_int, c_double_p,
                                    C.POINTER(FramePose), C.POINTER(FrameCounts)]),
    "vohip_frame_retry": (C.c_int, [C.c_void_p, c_double_p, C.c_int, C.POINTER(FramePose), C.POINTER(FrameCounts)]),
    "vohip_frame_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, c_double_p, C.c_int,
                                     c_double_p]),
    "vohip_frame_finish": (C.c_int, [C.c_void_p, C.POINTER(FramePose), C.POINTER(FrameCounts)]),
    "vohip_frame_step_ahead": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, c_double_p, C.c_int,
                                          c_double_p, C.POINTER(FramePose), C.POINTER(FrameCounts)]),
    "vohip_frame_prepare": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "vohip_frame_sizes": (C.c_int, [C.c_void_p, c_int_p, c_int_p]),
    "vohip_frame_state": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                     c_double_p]),
    "vohip_frame_apply_ba": (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p]),
    "vohip_align_sim3_prefixes": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int, c_int_p, C.c_int, C.c_double,
                                            C.c_double, c_double_p, c_double_p, c_double_p]),
    "vohip_group_create": (C.c_int, [C.c_void_p, c_double_p, C.c_int, C.c_int, C.c_int, C.POINTER(FrameParams), C.c_int,
                                     C.c_int, C.POINTER(C.c_void_p)]),
    "vohip_group_destroy": (C.c_int, [C.c_void_p]),
    "vohip_group_prepare": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_double,
                                      C.c_double]),
    "vohip_group_prepare_like": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]),
    "vohip_group_reset": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p]),
    "vohip_group_begin": (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p]),
    "vohip_group_finish": (C.c_int, [C.c_void_p, C.POINTER(FramePose), C.POINTER(FrameCounts), c_int_p]),
    "vohip_group_retry": (C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_int, C.POINTER(FramePose),
                                    C.POINTER(FrameCounts)]),
    "vohip_group_sizes": (C.c_int, [C.c_void_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    "vohip_group_state": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                    c_double_p]),
    "vohip_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    "vohip_destroy": (C.c_int, [C.c_void_p]),
    "vohip_last_error": (C.c_char_p, [C.c_void_p]),
    "vohip_stream": (C.c_void_p, [C.c_void_p]),
    "vohip_synchronize": (C.c_int, [C.c_void_p]),
    "vohip_version": (C.c_char_p, []),
    "vohip_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "vohip_ransac_p3p": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, c_double_p, c_int32_p,
                                    C.c_int, c_double_p, c_double_p, c_uint8_p, c_double_p, c_double_p,
                                    c_int_p]),
    "vohip_ransac_p3p_score_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_double_p,
                                              C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_ransac_p3p_score_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_double_p,
                                                  C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_ransac_p3p_solve_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_double_p,
                                              C.c_void_p, C.c_int]),
    "vohip_ransac_p3p_count_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p]),
    "vohip_ransac_p3p_poses": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p]),
    "vohip_ransac_inlier_mask_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_double_p,
                                                c_double_p, C.c_void_p]),
    "vohip_p3p": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, c_double_p]),
    "vohip_ransac_dlt": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, c_double_p, c_int32_p, C.c_int, c_double_p,
                                   c_double_p, c_uint8_p, c_double_p, c_double_p, c_int_p]),
    "vohip_estimate_pose_dlt": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, c_double_p, c_double_p]),
    "vohip_twist2homog": (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p]),
    "vohip_homog2twist": (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p]),
    "vohip_triangulate": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int,
                                     C.c_int, C.c_int, c_double_p]),
    "vohip_triangulate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_int, C.c_void_p]),
    "vohip_triangulate_refine": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p,
                                            c_double_p, C.c_int, c_double_p]),
    "vohip_candidate_filter": (C.c_int, [C.c_void_p, c_float_p, C.c_int, c_double_p, C.c_int, c_double_p,
                                          C.c_int, C.c_double, c_uint8_p]),
    "vohip_candidate_filter_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_int, C.c_double, C.c_void_p]),
    "vohip_candidate_triangulate": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int,
                                               c_double_p, c_double_p, C.c_double, C.c_double, C.c_double,
                                               C.c_int, c_uint8_p, c_double_p, c_double_p]),
    "vohip_align_sim3": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                    c_double_p, c_double_p, c_double_p]),
    "vohip_relative_pose": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int,
                                       c_double_p, c_double_p, c_double_p]),
    "vohip_fmatrix_ransac": (C.c_int, [C.c_void_p, c_double_p, c_double_p, C.c_int, c_int32_p, C.c_int, C.c_double,
                                        C.c_double, c_uint8_p, c_int_p, c_int_p, c_double_p]),
    "vohip_fmatrix_ransac_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                            C.c_double, C.c_double, C.c_void_p, c_int_p, c_int_p, c_double_p]),
    "vohip_ba_error": (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p, C.c_int, c_double_p, c_double_p,
                                  c_int_p]),
    "vohip_ba_run": (C.c_int, [C.c_void_p, c_double_p, C.c_int, c_double_p, C.c_int, c_double_p, C.c_int,
                                C.c_double, C.c_double, c_double_p]),
    "vohip_pose_refine": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p, C.c_int,
                                     C.c_double, C.c_double, c_double_p]),
    "vohip_pose_refine_batch": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_int_p, c_double_p,
                                           C.c_int, C.c_double, C.c_double, c_double_p]),
    "vohip_pose_refine_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_double_p,
                                         C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "vohip_detect_corners": (C.c_int, [C.c_void_p, c_uint8_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                        C.c_int, C.c_int, c_float_p, c_float_p, C.c_int, c_int_p]),
    "vohip_detect_corners_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_detect_fast": (C.c_int, [C.c_void_p, c_uint8_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                     c_float_p, c_float_p, C.c_int, c_int_p]),
    "vohip_detect_fast_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_detect_fast_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                               C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_detect_surf": (C.c_int, [C.c_void_p, c_uint8_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, c_float_p,
                                     c_float_p, c_float_p, C.c_int, c_int_p]),
    "vohip_detect_surf_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_detect_surf_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                               C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_detect_brisk": (C.c_int, [C.c_void_p, c_uint8_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int,
                                      c_float_p, c_float_p, c_float_p, C.c_int, c_int_p]),
    "vohip_detect_brisk_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_detect_brisk_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                                C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_extract_surf_features": (C.c_int, [C.c_void_p, c_uint8_p, C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, C.c_int,
                                               C.c_int, c_float_p, c_float_p]),
    "vohip_extract_surf_features_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vohip_match_features": (C.c_int, [C.c_void_p, c_float_p, C.c_int, c_float_p, C.c_int, C.c_int, C.c_double, C.c_double,
                                        c_int_p, c_float_p, c_int_p]),
    "vohip_match_features_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double,
                                            C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vohip_pyramid_build": (C.c_int, [C.c_void_p, C.c_int, c_uint8_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vohip_pyramid_build_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "vohip_pyramid_floats": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "vohip_pyramid_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vohip_detect_corners_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                                  C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vohip_pyramid_level": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_float_p, c_int_p, c_int_p]),
    "vohip_klt_track": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, C.c_int, C.c_int, C.c_int,
                                   C.c_double, c_double_p, c_uint8_p, c_double_p]),
    "vohip_klt_track_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vohip_candidate_triangulate_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                   c_double_p, c_double_p, C.c_double, C.c_double, C.c_double,
                                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


class VohipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


def load():
    """Loads libvohip.so and declares every entry point.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                "%s not found: build it with `python %s`" % (LIB_PATH, os.path.join(HERE, "build.py")))
        # PyTorch ships its own copy of the HIP runtime and finds it by file name, so a process that
        # loaded the system's copy first would end up with two runtimes (and torch with no GPU).
        # Loading torch's first makes this library resolve to the same one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def colmajor(mat):
    """MATLAB (column-major) memory image of a 2-D array given in math layout."""
    return np.ascontiguousarray(np.asarray(mat, dtype=np.float64).T)


# Objects that own GPU resources, closed in order (frame loops before their contexts) when the
# interpreter exits: a destructor that runs after the HIP runtime has shut down would abort the process.
_live_frames = weakref.WeakSet()
_live_contexts = weakref.WeakSet()


@atexit.register
def _close_all():
    for group in (_live_frames, _live_contexts):
        for obj in list(group):
            try:
                obj.close()
            except Exception:
                pass


class Context:
    """One GPU + one HIP stream.  `stream` is a raw hipStream_t (int) or None."""

    def __init__(self, device=0, stream=None):
        lib = load()
        h = C.c_void_p()
        st = lib.vohip_create(C.byref(h), int(device), C.c_void_p(stream) if stream else None)
        if st != VOHIP_OK:
            raise VohipError(st, lib.vohip_last_error(None).decode())
        self._h = h
        self.device = device
        _live_contexts.add(self)

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError("context destroyed")
        return self._h

    def check(self, status, allow=()):
        if status != VOHIP_OK and status not in allow:
            raise VohipError(status, load().vohip_last_error(self._h).decode())
        return status

    def set_option(self, name, value):
        self.check(load().vohip_set_option(self.handle, name.encode(), int(value)))

    def synchronize(self):
        self.check(load().vohip_synchronize(self.handle))

    def close(self):
        if self._h is not None:
            # frame loops and groups created on this context hold device state through it: they go first (one
            # that outlives its context would touch freed memory when it is closed later)
            for obj in list(_live_frames):
                if getattr(obj, "ctx", None) is self:
                    try:
                        obj.close()
                    except Exception:
                        pass
            load().vohip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Process-wide context on the GPU named by LOCAL_RANK (one process per GPU), else GPU 0."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx
