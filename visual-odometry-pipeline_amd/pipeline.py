"""The reference's continuous-operation stage functions and frame loop, as a host layer over
libvohip.so (every numeric step runs on the GPU; this file only does the bookkeeping that the
reference does in MATLAB):

  update_landmarks   Continuous_operation/update_landmarks.m:1-125
  update_candidate   Continuous_operation/update_candidate.m:1-59
  VO_bundle_adjust   Continuous_operation/VO_bundle_adjust.m:1-98
  VOPipeline         state initialisation main_demo.m:268-303,342-349 and the loop body
                     main_demo.m:377-427 (identical in main_BA.m:261-305, main_eval.m:395-442)

Arrays keep MATLAB's math layout (S.P 2xN (row; col), S.X 4xN, S.T 12xM, ...).  Indices stored
inside observation vectors stay 1-based, exactly as the reference serialises them.
"""
from types import SimpleNamespace

import numpy as np

from . import _lib
from .ba import T_refinement, T_refinement_batch, runBA
from .fmatrix import estimateFundamentalMatrix
from .geometry import (HomogMatrix2twist, candidate_filter, candidate_triangulate, twist2HomogMatrix,
                       twist2HomogMatrix_many)
from .keypoints import genKeypoints
from .klt import PointTracker, Pyramid
from .ransac import ransacLocalization


def default_hyper_paras(dataset="kitti", is_BA=False):
    """hyper_paras of main_demo.m:21-46 with the per-dataset overrides of :62-75 (kitti),
    :92-97 (malaga), :107-112 (parking); the detector is Harris (main_eval.m:168-169)."""
    hp = SimpleNamespace(is_BA=is_BA, is_refine_pose=True, feature_extract="Harris", init_matching_method="KLT",
                         show_matching_res=False, sfm_pose="fundamental",
                         feature_extract_options=["MinQuality", 2e-4], min_depth=0.0, max_depth=100.0,
                         r_discard_redundant=5.0, angle_threshold=5.0)
    if dataset == "kitti":
        hp.min_depth, hp.r_discard_redundant, hp.max_depth, hp.angle_threshold = 1.0, 10.0, 80.0, 1.0
    elif dataset in ("malaga", "parking"):
        hp.min_depth, hp.r_discard_redundant, hp.max_depth, hp.angle_threshold = 2.0, 10.0, 50.0, 0.8
    return hp


def _pose12(R_W_C, t_W_C):
    return np.concatenate([R_W_C.reshape(-1, order="F"), np.asarray(t_W_C).reshape(3)])


def init_state(p_W_landmarks, keypoints_xy, T_init_WC, window_size=3, keyframe_d=2, ctx=None):
    """S and B as main_demo.m:268-303 builds them.  p_W_landmarks N x 3, keypoints_xy N x 2 (x, y)."""
    X = np.asarray(p_W_landmarks, dtype=np.float64).reshape(-1, 3).T
    n = X.shape[1]
    S = SimpleNamespace()
    S.P = np.asarray(keypoints_xy, dtype=np.float64).reshape(-1, 2)[:, ::-1].T.copy()   # (row; col)
    S.X = np.vstack([X, np.arange(1, n + 1, dtype=np.float64)])
    S.C = np.zeros((2, 0))
    S.F = np.zeros((2, 0))
    S.T = np.zeros((12, 0))
    T = np.asarray(T_init_WC, dtype=np.float64)
    S.est_trans = T[:3, 3].reshape(3, 1).copy()
    S.est_rot = T[:3, :3].reshape(9, 1, order="F").copy()
    S.num_X = [n]
    S.num_C = [0]
    S.num_new = [0]
    B = SimpleNamespace(window_size=window_size, keyframe_d=keyframe_d, num_key=1, count_frame=0, m=n)
    B.tau = HomogMatrix2twist(T, ctx=ctx).reshape(6, 1)
    B.landmarks = S.X.copy()
    B.observation = [None] * window_size
    uv = S.P[::-1]
    B.observation[0] = np.concatenate([[float(n)], uv.reshape(-1, order="F"), np.arange(1, n + 1, dtype=np.float64)])
    B.normal_frame_refine = []
    B.num_normal_frame = 0
    B.new_idx = n + 1
    return S, B


def update_landmarks(S, B, KLT_tracker_C, image, K, hyper_paras, rng=None, ctx=None):
    """[S, B] = update_landmarks(S, B, KLT_tracker_C, image, K, hyper_paras)"""
    R_W_C = S.est_rot[:, -1].reshape(3, 3, order="F")
    T_W_C = np.eye(4)
    T_W_C[:3, :3] = R_W_C
    T_W_C[:3, 3] = S.est_trans[:, -1]

    pts, valid = KLT_tracker_C(image)                           # :10
    matched = pts[valid][:, ::-1]                               # (u, v) -> (row, col), :11
    _, inl = estimateFundamentalMatrix(S.C[:, valid].T, matched, NumTrials=500, DistanceThreshold=5e-2,
                                       rng=rng, ctx=ctx)        # :13-14
    matched = matched[inl]
    nv = len(valid)
    S.F = S.F[:, :nv][:, valid][:, inl]                         # :20-25 (S.T may be wider than S.C)
    S.T = S.T[:, :nv][:, valid][:, inl]

    status, X_new, _ = candidate_triangulate(matched.T, S.F, S.T, T_W_C, K, hyper_paras.angle_threshold,
                                             hyper_paras.min_depth, hyper_paras.max_depth, hyper_paras.is_BA,
                                             ctx=ctx)           # :27-97
    append = status != 0
    new = status == 1
    k = int(new.sum())
    if k:
        ids = B.new_idx + np.arange(k, dtype=np.float64)
        cols = np.vstack([X_new[:, new], ids])
        S.X = np.hstack([S.X, cols])                            # :83-87 / :91-95
        B.landmarks = np.hstack([B.landmarks, cols])
        B.new_idx += k
        S.P = np.hstack([S.P, matched[new].T])
    B.m = B.landmarks.shape[1]
    S.C = matched[~append].T.copy()                             # :120-123
    S.F = S.F[:, ~append]
    S.T = S.T[:, ~append]
    S.num_new.append(int(append.sum()))
    return S, B


def update_candidate(S, valid_key_candidates, image, K, hyper_paras, ctx=None):
    """S = update_candidate(S, valid_key_candidates, image, K, hyper_paras) -- image and K are
    unused, as in the reference."""
    R_W_C = S.est_rot[:, -1].reshape(3, 3, order="F")
    t_W_C = S.est_trans[:, -1]
    loc = np.asarray(valid_key_candidates.Location, dtype=np.float32).reshape(-1, 2)
    if S.C.shape[1] > 0:                                        # :12-24
        keep = candidate_filter(loc, S.C, S.P, hyper_paras.r_discard_redundant, ctx=ctx)
        new_valid = loc[keep][:, ::-1].T.astype(np.float64)
    else:                                                       # :25-27
        new_valid = loc[:, ::-1].T.astype(np.float64)
    S.C = np.hstack([S.C, new_valid])                           # :55-57
    S.F = np.hstack([S.F, new_valid])
    S.T = np.hstack([S.T, np.tile(_pose12(R_W_C, t_W_C).reshape(12, 1), (1, loc.shape[0]))])
    return S


def _columns_with_ids(id_row, ids):
    """[~, col] = find(id_row == ids(:)): ascending column indices whose id is in `ids`.  Landmark ids only
    ever grow (update_landmarks.m:83-87 appends B.new_idx, B.new_idx + 1, ...), so id_row is ascending and a
    binary search finds the columns; anything else takes the general route."""
    id_row = np.asarray(id_row)
    ids = np.asarray(ids)
    if id_row.size and ids.size and np.all(id_row[1:] > id_row[:-1]) and np.all(ids[1:] > ids[:-1]):
        cols = np.minimum(np.searchsorted(id_row, ids), id_row.size - 1)
        return cols[id_row[cols] == ids]
    return np.flatnonzero(np.isin(id_row, ids))


def VO_bundle_adjust(S, B, T_W_C, K, hyper_paras, ctx=None):
    """[S, B] = VO_bundle_adjust(S, B, T_W_C, K, hyper_paras)"""
    if B.count_frame == B.keyframe_d:                           # keyframe, :3
        obs_current = np.concatenate([[float(S.P.shape[1])], S.P.reshape(-1, order="F"), S.X[3]])
        if B.num_key < B.window_size:                           # :4-15
            B.num_key += 1
            tw = HomogMatrix2twist(T_W_C, ctx=ctx).reshape(6, 1)
            if B.tau.shape[1] < B.num_key:
                B.tau = np.hstack([B.tau, np.zeros((6, B.num_key - B.tau.shape[1]))])
            B.tau[:, B.num_key - 1] = tw[:, 0]
            B.observation[B.num_key - 1] = obs_current
        else:
            # slide the window, :17-31
            o2 = B.observation[1]
            m2 = int(o2[0])
            ids2 = o2[len(o2) - m2:]
            last_idx = int(np.flatnonzero(B.landmarks[3] == ids2[-1])[0]) + 1
            remaining = _columns_with_ids(B.landmarks[3], ids2)
            B.landmarks = np.hstack([B.landmarks[:, :last_idx][:, remaining], B.landmarks[:, last_idx:]])
            B.tau = np.hstack([B.tau[:, 1:], HomogMatrix2twist(T_W_C, ctx=ctx).reshape(6, 1)])
            B.observation = B.observation[1:] + [obs_current]
            B.m = B.landmarks.shape[1]
            # global ids -> column numbers of B.landmarks, :33-41
            obs_loc = []
            for obs in B.observation[:B.num_key]:
                n = int(obs[0])
                cols = _columns_with_ids(B.landmarks[3], obs[len(obs) - n:])
                if len(cols) != n:
                    raise RuntimeError("VO_bundle_adjust: an observed landmark is missing from the window")
                o = obs.copy()
                o[len(o) - n:] = cols + 1.0
                obs_loc.append(o)
            state = np.concatenate([B.tau.reshape(-1, order="F"), B.landmarks[:3].reshape(-1, order="F")])
            Obs = np.concatenate([[float(B.num_key), float(B.m)]] + obs_loc)
            opt = runBA(state, Obs, K, ctx=ctx)                 # :51
            step = B.keyframe_d + 1
            ncol = S.est_rot.shape[1]
            Ts = twist2HomogMatrix_many(opt[:6 * B.num_key].reshape(B.num_key, 6), ctx=ctx)   # (one call for the window)
            for i in range(1, B.num_key + 1):                   # :53-57
                T = Ts[i - 1]
                col = ncol - 1 - step * (B.num_key - i)
                S.est_rot[:, col] = T[:3, :3].reshape(-1, order="F")
                S.est_trans[:, col] = T[:3, 3]
            B.landmarks[:3] = opt[B.num_key * 6:].reshape(3, B.m, order="F")
            # accept / revert each optimised landmark by its depth in the current camera, :62-70
            T_C_W = np.linalg.inv(np.asarray(T_W_C, dtype=np.float64))
            cols = np.searchsorted(B.landmarks[3], S.X[3])
            if not np.array_equal(B.landmarks[3, cols], S.X[3]):
                raise RuntimeError("VO_bundle_adjust: a tracked landmark is missing from the window")
            depth = T_C_W[2, :3] @ B.landmarks[:3, cols] + T_C_W[2, 3]
            ok = (depth > hyper_paras.min_depth) & (depth < 1.5 * hyper_paras.max_depth)
            S.X[:3, ok] = B.landmarks[:3, cols[ok]]
            B.landmarks[:3, cols[~ok]] = S.X[:3, ~ok]
            # re-refine the buffered normal frames against the optimised landmarks, :71-86 (the frames are
            # independent of each other: one launch and one synchronisation for all of them)
            problems, cols_out = [], []
            for i in range(1, B.num_key):
                for j in range(1, B.keyframe_d + 1):
                    T_f, kp_f, ids_f = B.normal_frame_refine[B.keyframe_d * (i - 1) + j - 1]
                    idx = _columns_with_ids(B.landmarks[3], ids_f)
                    problems.append((T_f, kp_f, B.landmarks[:3, idx]))
                    cols_out.append(ncol - 1 - step * (B.num_key - i) + j)
            for col, T_r in zip(cols_out, T_refinement_batch(problems, K, ctx=ctx)):
                S.est_rot[:, col] = T_r[:3, :3].reshape(-1, order="F")
                S.est_trans[:, col] = T_r[:3, 3]
        B.count_frame = 0
    else:                                                       # normal frame, :89-98
        entry = (np.array(T_W_C, dtype=np.float64), S.P[::-1].copy(), S.X[3].copy())
        if B.num_normal_frame < (B.window_size - 1) * B.keyframe_d:
            B.num_normal_frame += 1
            B.normal_frame_refine.append(entry)
        else:
            B.normal_frame_refine = B.normal_frame_refine[1:] + [entry]
        B.count_frame += 1
    return S, B


class VOPipeline:
    """Continuous operation: the loop body of main_demo.m:377-427.  Construct it from the
    bootstrap result (landmarks, their keypoints in the last bootstrap image, T_init_WC), then
    call process_frame(image) per frame.  One image pyramid is built per frame and shared by both
    trackers and by the next frame."""

    def __init__(self, K, first_image, p_W_landmarks, keypoints_xy, T_init_WC, hyper_paras=None, window_size=3,
                 keyframe_d=2, rng=None, ctx=None, tracker_options=None, rng_map=None):
        self.ctx = ctx or _lib.default_context()
        self.K = np.asarray(K, dtype=np.float64)
        self.hp = hyper_paras or default_hyper_paras()
        self.rng = rng or np.random.default_rng()      # sample triples of the localisation
        self.rng_map = rng_map or self.rng.spawn(1)[0]  # trials of the epipolar filter: an independent stream
        self.S, self.B = init_state(p_W_landmarks, keypoints_xy, T_init_WC, window_size, keyframe_d, ctx=self.ctx)
        opts = dict(BlockSize=[21, 21], NumPyramidLevels=5, MaxIterations=20, MaxBidirectionalError=6)  # :342-349
        opts.update(tracker_options or {})
        self.levels = opts["NumPyramidLevels"]
        self.KLT_tracker_L = PointTracker(ctx=self.ctx, **opts)
        self.KLT_tracker_C = PointTracker(ctx=self.ctx, **opts)
        self._pyr = Pyramid(first_image, self.levels, ctx=self.ctx)
        self.KLT_tracker_L.initialize(self.S.P[::-1].T, self._pyr)
        self.frames = 0

    def process_frame(self, image):
        S, B, K, hp, ctx = self.S, self.B, self.K, self.hp, self.ctx
        _, valid_key_candidates = genKeypoints(image, hp.feature_extract, hp.feature_extract_options, ctx=ctx)  # :377
        pyr = Pyramid(image, self.levels, ctx=ctx)
        matched_points, validity = self.KLT_tracker_L(pyr)      # :380
        matched_valid = matched_points[validity][:, ::-1]       # (row, col), :381
        R_C_W, t_C_W, inlier_mask, _, _ = ransacLocalization(matched_valid.T, S.X[:3, validity], K, rng=self.rng,
                                                             ctx=ctx)   # :387-388
        if R_C_W.size == 0:
            raise RuntimeError("ransacLocalization found no pose (the reference fails here too, main_demo.m:389-391)")
        T_C_W = np.eye(4)
        T_C_W[:3, :3] = R_C_W
        T_C_W[:3, 3] = t_C_W[:, 0]
        T_W_C = np.linalg.inv(T_C_W)                            # :391
        inl = inlier_mask[0] > 0
        S.P = matched_valid[inl].T.copy()                       # :394-396
        S.X = S.X[:, validity][:, inl]
        if hp.is_refine_pose:
            T_W_C = T_refinement(T_W_C, S.P[::-1], S.X[:3], K, ctx=ctx)     # :398-400
        S.est_trans = np.hstack([S.est_trans, T_W_C[:3, 3:4]])  # :401-403
        S.est_rot = np.hstack([S.est_rot, T_W_C[:3, :3].reshape(9, 1, order="F")])
        if S.C.shape[1] > 0:                                    # :406-408
            S, B = update_landmarks(S, B, self.KLT_tracker_C, pyr, K, hp, rng=self.rng_map, ctx=ctx)
        if hp.is_BA:                                            # :409-411
            S, B = VO_bundle_adjust(S, B, T_W_C, K, hp, ctx=ctx)
        S = update_candidate(S, valid_key_candidates, image, K, hp, ctx=ctx)    # :413
        self.KLT_tracker_L.release()                            # :417-424
        self.KLT_tracker_L.initialize(S.P[::-1].T, pyr)
        if S.C.shape[1] > 0:
            self.KLT_tracker_C.release()
            self.KLT_tracker_C.initialize(S.C[::-1].T, pyr)
        self._pyr.release()
        self._pyr = pyr
        S.num_X.append(S.X.shape[1])                            # :426-427
        S.num_C.append(S.C.shape[1])
        self.S, self.B = S, B
        self.frames += 1
        return T_W_C

    def trajectory(self):
        """est_trans (3 x T) and est_rot (9 x T), as saved by main_demo.m:473-476."""
        return self.S.est_trans.copy(), self.S.est_rot.copy()

    def close(self):
        self.KLT_tracker_L.release()
        self.KLT_tracker_C.release()
        self._pyr.release()
