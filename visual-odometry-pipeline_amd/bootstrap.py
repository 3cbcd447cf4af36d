"""Host-side mirror of the reference's bootstrap (main_demo.m:252-264):

  matching_init          Initialization/matching_init.m:1-45        ('KLT' and 'Des_match' branches)
  pose_estimation_init   Initialization/pose_estimation_init.m:1-37 ('fundamental' and 'essential' branches)
  triangulation_init     Initialization/triangulation_init.m:1-34
  compProjMat            Initialization/compProjMat.m:1-6
  relativeCameraPose     the toolbox call at pose_estimation_init.m:29-33

Detection, tracking, the fundamental-matrix RANSAC, the pose disambiguation, triangulation and
pose refinement all run in libvohip.so on the GPU; this file only does the reference's own
bookkeeping.  Points are N-by-2 (x, y), 1-based, as the toolbox objects hold them.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import colmajor, dptr
from .ba import T_refinement
from .fmatrix import estimateFundamentalMatrix
from .geometry import linearTriangulation
from .keypoints import genKeypoints, matchFeatures
from .klt import PointTracker


def relativeCameraPose(E, K, inlierPoints1, inlierPoints2, ctx=None):
    """[relativeOrientation, relativeLocation, validPointsFraction] =
    relativeCameraPose(E, cameraParams, inlierPoints1, inlierPoints2) with cameraParams built from
    K.  Toolbox convention: x_cam2 = orientation * (x_cam1 - location'), location 1x3 of unit
    length."""
    ctx = ctx or _lib.default_context()
    p1 = np.ascontiguousarray(np.asarray(inlierPoints1, dtype=np.float64).reshape(-1, 2))
    p2 = np.ascontiguousarray(np.asarray(inlierPoints2, dtype=np.float64).reshape(-1, 2))
    if p1.shape != p2.shape:
        raise ValueError("point sets differ in count")
    R = np.zeros(9)
    loc = np.zeros(3)
    frac = C.c_double(0.0)
    ctx.check(_lib.load().vohip_relative_pose(ctx.handle, dptr(colmajor(E)), dptr(colmajor(K)), dptr(p1), dptr(p2),
                                              p1.shape[0], dptr(R), dptr(loc), C.byref(frac)))
    return R.reshape(3, 3).T.copy(), loc.reshape(1, 3).copy(), frac.value


def matching_init(img0, img_seqs, hyper_paras, ctx=None):
    """[init_points, matched_points] = matching_init(img0, img_seqs, hyper_paras)"""
    method = getattr(hyper_paras, "init_matching_method", "KLT")
    img1 = img_seqs[-1]
    if method == "Des_match":                                                              # :25-32
        feat0, valid0 = genKeypoints(img0, hyper_paras.feature_extract, hyper_paras.feature_extract_options, ctx=ctx,
                                     with_features=True)
        feat1, valid1 = genKeypoints(img1, hyper_paras.feature_extract, hyper_paras.feature_extract_options, ctx=ctx,
                                     with_features=True)
        indexPairs, _ = matchFeatures(feat0, feat1, ctx=ctx)
        return valid0.Location[indexPairs[:, 0]], valid1.Location[indexPairs[:, 1]]
    if method != "KLT":
        raise NotImplementedError("init_matching_method '%s' is not built ('KLT' and 'Des_match' are)" % method)
    tracker = PointTracker(NumPyramidLevels=5, MaxBidirectionalError=3, ctx=ctx)          # :8-10
    _, valid_key_candidates0 = genKeypoints(img0, hyper_paras.feature_extract, hyper_paras.feature_extract_options,
                                            ctx=ctx)                                       # :12
    init_points_ = valid_key_candidates0.Location                                          # :13
    tracker.initialize(init_points_, img0)                                                 # :14
    for img_intermediate in img_seqs[:-1]:                                                 # :16-19
        tracker(img_intermediate)
    matched_points1_, validity_img1 = tracker(img1)                                        # :21
    tracker.release()
    return init_points_[validity_img1], matched_points1_[validity_img1]                    # :23-24


def estimateEssentialMatrix(matchedPoints1, matchedPoints2, K, MaxNumTrials=500, MaxDistance=0.1, Confidence=99.0,
                            samples=None, rng=None, ctx=None):
    """[E, inliersIndex] = estimateEssentialMatrix(matchedPoints1, matchedPoints2, cameraParams, 'Confidence', c)
    (pose_estimation_init.m:19; MaxNumTrials 500 and MaxDistance 0.1 px^2 Sampson are the toolbox function's
    documented defaults).

    The toolbox function is closed source (a five-point minimal solver inside M-estimator sample consensus).  This
    project's definition, stated in full: the fundamental-matrix RANSAC of estimateFundamentalMatrix (normalised
    eight-point hypotheses, Sampson distance in pixels^2, adaptive trial budget) with those trial / distance /
    confidence settings, then E = K' F K moved to the nearest essential matrix (singular values (1, 1, 0)).  The
    inlier test is the toolbox's (Sampson distance against MaxDistance); the hypothesis generator is not, so the
    inlier SETS agree with MATLAB's only in kind."""
    K = np.asarray(K, dtype=np.float64)
    F, inliers = estimateFundamentalMatrix(matchedPoints1, matchedPoints2, Method="RANSAC", NumTrials=MaxNumTrials,
                                           DistanceThreshold=MaxDistance, Confidence=Confidence, samples=samples, rng=rng,
                                           ctx=ctx)
    U, _, Vt = np.linalg.svd(K.T @ F @ K)
    return U @ np.diag([1.0, 1.0, 0.0]) @ Vt, inliers


def pose_estimation_init(init_points, matched_points, K, hyper_paras, rng=None, ctx=None):
    """[T_W_C, init_points_valid, matched_points_valid] =
    pose_estimation_init(init_points, matched_points, K, hyper_paras)

    As in the reference, the toolbox's orientation matrix and location vector are placed
    side by side without conversion (:35), so the rotation block of the returned matrix is the
    toolbox orientation (R_C2_C1), not its transpose."""
    method = getattr(hyper_paras, "sfm_pose", "fundamental")
    K = np.asarray(K, dtype=np.float64)
    if method == "fundamental":
        fRANSAC, inliers = estimateFundamentalMatrix(init_points, matched_points, Method="RANSAC", NumTrials=2000,
                                                     DistanceThreshold=1e-2, Confidence=99.9, rng=rng, ctx=ctx)   # :7-12
        essMat = K.T @ fRANSAC @ K                                                         # :15
    elif method == "essential":
        essMat, inliers = estimateEssentialMatrix(init_points, matched_points, K, Confidence=99.9, rng=rng, ctx=ctx)   # :19
    else:
        raise NotImplementedError("sfm_pose '%s' is not built ('fundamental' and 'essential' are)" % method)
    init_points_valid = np.asarray(init_points)[inliers]                                   # :25-26
    matched_points_valid = np.asarray(matched_points)[inliers]
    R_W_C, t_W_C, _ = relativeCameraPose(essMat, K, init_points_valid, matched_points_valid, ctx=ctx)   # :29-33
    T_W_C = np.eye(4)
    T_W_C[:3, :3] = R_W_C                                                                  # :35
    T_W_C[:3, 3] = t_W_C[0]
    return T_W_C, init_points_valid, matched_points_valid


def compProjMat(rot_mat, trans_vec, intrinsic_mat):
    """proj_mat = intrinsic_mat * [rot_mat, trans_vec]"""
    return np.asarray(intrinsic_mat, dtype=np.float64) @ np.hstack(
        [np.asarray(rot_mat, dtype=np.float64), np.asarray(trans_vec, dtype=np.float64).reshape(3, 1)])


def triangulation_init(init_points_valid, matched_points_valid, T_init_WC, K, hyper_paras, ctx=None):
    """[pts3d, matched_points_valid] = triangulation_init(init_points_valid, matched_points_valid,
    T_init_WC, K, hyper_paras): pts3d N-by-3 in the frame of the first bootstrap camera."""
    T_C_W = np.linalg.inv(np.asarray(T_init_WC, dtype=np.float64))                         # :2-4
    proj_mat0 = compProjMat(np.eye(3), np.zeros(3), K)                                     # :7-8
    proj_mat1 = compProjMat(T_C_W[:3, :3], T_C_W[:3, 3], K)
    p0 = np.asarray(init_points_valid, dtype=np.float64).reshape(-1, 2)
    p1 = np.asarray(matched_points_valid, dtype=np.float64).reshape(-1, 2)
    n = p0.shape[0]
    pts3d = linearTriangulation(np.vstack([p0.T, np.ones((1, n))]), np.vstack([p1.T, np.ones((1, n))]), proj_mat0,
                                proj_mat1, ctx=ctx)                                        # :11-18
    pts3d = pts3d[:3].T                                                                    # :20
    valid_idx = (pts3d[:, 2] >= hyper_paras.min_depth) & (pts3d[:, 2] <= hyper_paras.max_depth)   # :28
    return pts3d[valid_idx], np.asarray(matched_points_valid)[valid_idx]                   # :29-31


def bootstrap(img0, img_seqs, K, hyper_paras, rng=None, ctx=None):
    """main_demo.m:252-264: returns (p_W_landmarks N x 3, keypoints N x 2 (x, y) in img_seqs[-1],
    T_init_WC 4x4) -- the three inputs of the continuous-operation state."""
    init_points, matched_points = matching_init(img0, img_seqs, hyper_paras, ctx=ctx)                       # :252
    T_init_WC, init_valid, matched_valid = pose_estimation_init(init_points, matched_points, K, hyper_paras,
                                                                rng=rng, ctx=ctx)                           # :254
    pts3d, matched_valid = triangulation_init(init_valid, matched_valid, T_init_WC, K, hyper_paras, ctx=ctx)  # :256
    p_W_landmarks = np.asarray(pts3d, dtype=np.float64)                                                     # :259-260
    keypoints = np.asarray(matched_valid, dtype=np.float64)
    if hyper_paras.is_refine_pose:                                                                          # :262-264
        T_init_WC = T_refinement(T_init_WC, keypoints.T, p_W_landmarks.T, K, ctx=ctx)
    return p_W_landmarks, keypoints, T_init_WC
