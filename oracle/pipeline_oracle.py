"""CPU restatement of the reference's continuous-operation loop (main_demo.m:268-303, 342-349,
377-427) and its stage functions (update_landmarks.m, update_candidate.m, VO_bundle_adjust.m),
written statement by statement after the MATLAB sources on top of the C oracle primitives.

TEST INFRASTRUCTURE ONLY: used by tests/ and by the cpu_baseline leg of bench.py.
PARITY UNPINNED versus MATLAB for the toolbox calls inside (tracker, detector, F-matrix RANSAC,
lsqnonlin, RNG): those are this project's own definitions (see the C files); the bookkeeping
around them follows the reference literally, including its 1-based indices inside observation
vectors and the `find(row == column)` idiom (ascending column indices).
"""
import numpy as np

from . import vo_oracle as orc


class Struct:
    pass


def matlab_find_cols(row, ids):
    """[~, c] = find(row == ids(:)) for a row vector and a column vector: column-major scan of
    the comparison matrix, so column indices come out ascending (0-based here)."""
    ids = np.asarray(ids).reshape(-1)
    hit = (np.asarray(row).reshape(1, -1) == ids.reshape(-1, 1))
    return np.nonzero(hit.T)[0]


def init_state(p_W_landmarks, keypoints_xy, T_init_WC, window_size, keyframe_d):
    S, B = Struct(), Struct()
    S.X = np.asarray(p_W_landmarks, dtype=np.float64).T                      # main_demo.m:272
    S.P = np.fliplr(np.asarray(keypoints_xy, dtype=np.float64)).T           # :273
    n = S.X.shape[1]
    S.X = np.vstack([S.X, np.arange(1, n + 1)])                             # :275
    S.C = np.zeros((2, 0)); S.F = np.zeros((2, 0)); S.T = np.zeros((12, 0))  # :276-279
    S.est_trans = T_init_WC[:3, 3].reshape(3, 1).copy()                     # :280
    S.est_rot = T_init_WC[:3, :3].reshape(9, 1, order="F").copy()           # :281-282
    S.num_X, S.num_C, S.num_new = [n], [0], [0]                             # :284-286
    B.window_size, B.keyframe_d, B.num_key, B.count_frame = window_size, keyframe_d, 1, 0   # :289-292
    B.m = n
    B.tau = orc.homog2twist(T_init_WC).reshape(6, 1)                        # :294
    B.landmarks = S.X.copy()
    B.observation = [None] * window_size
    temp = np.flipud(S.P)
    B.observation[0] = np.concatenate([[B.m], temp.reshape(-1, order="F"), np.arange(1, B.m + 1)])   # :297-298
    B.normal_frame_refine = []
    B.num_normal_frame = 0
    B.new_idx = B.m + 1                                                     # :303
    return S, B


def klt_step(prev_img, cur_img, pts_xy):
    """vision.PointTracker step with the reference's parameters (main_demo.m:342-343)."""
    out, valid, _ = orc.klt_track(prev_img, cur_img, np.asarray(pts_xy, dtype=np.float64) - 1.0, block=21, levels=5,
                                  max_iter=20, max_bidir_error=6.0)
    return out + 1.0, valid


def update_landmarks(S, B, prev_img, image, K, hp, draw_octets, rng):
    R_W_C = S.est_rot[:, -1].reshape(3, 3, order="F")                       # update_landmarks.m:3-5
    t_W_C = S.est_trans[:, -1]
    T_W_C = np.eye(4); T_W_C[:3, :3] = R_W_C; T_W_C[:3, 3] = t_W_C
    pts, valid = klt_step(prev_img, image, np.fliplr(S.C.T))               # :10 (tracker seeded with fliplr(S.C'))
    matched = np.fliplr(pts[valid])                                         # :11
    p1 = S.C[:, valid].T
    if p1.shape[0] < 8:
        raise RuntimeError("estimateFundamentalMatrix needs at least 8 points")
    res = orc.fmatrix_ransac(p1.T, matched.T, draw_octets(rng, p1.shape[0], 500), 5e-2)    # :13-14
    inl = res["mask"]
    matched = matched[inl]                                                  # :15
    nv = len(valid)
    S.F = S.F[:, :nv][:, valid][:, inl]                                     # :20-25
    S.T = S.T[:, :nv][:, valid][:, inl]
    status, X_new, _ = orc.candidate_triangulate(matched.T, S.F, S.T, T_W_C, K, hp.angle_threshold, hp.min_depth,
                                                 hp.max_depth, hp.is_BA)    # :27-97
    append = status != 0                                                    # whehter_append
    for ii in np.flatnonzero(append):                                       # :56-97, growth in loop order
        if status[ii] == 1:
            col = np.concatenate([X_new[:, ii], [B.new_idx]]).reshape(4, 1)
            S.X = np.hstack([S.X, col])
            B.landmarks = np.hstack([B.landmarks, col])
            B.new_idx += 1
            B.m = B.landmarks.shape[1]
            S.P = np.hstack([S.P, matched[ii].reshape(2, 1)])
    S.C = matched[~append].T                                                # :120-123
    S.F = S.F[:, ~append]
    S.T = S.T[:, ~append]
    S.num_new.append(int(append.sum()))
    return S, B


def update_candidate(S, location_xy, hp):
    R_W_C = S.est_rot[:, -1].reshape(3, 3, order="F")                       # update_candidate.m:3-4
    t_W_C = S.est_trans[:, -1]
    loc = np.asarray(location_xy, dtype=np.float32).reshape(-1, 2)
    if S.C.shape[1] > 0:                                                    # :12-24
        keep = orc.candidate_filter(loc, S.C, S.P, hp.r_discard_redundant)
        new_valid = np.stack([loc[keep, 1], loc[keep, 0]]).astype(np.float64)
    else:                                                                   # :25-27
        new_valid = np.flipud(loc.T).astype(np.float64)
    S.C = np.hstack([S.C, new_valid])                                       # :55-57
    S.F = np.hstack([S.F, new_valid])
    pose = np.concatenate([R_W_C.reshape(-1, order="F"), t_W_C])
    S.T = np.hstack([S.T, np.tile(pose.reshape(12, 1), (1, loc.shape[0]))])
    return S


def VO_bundle_adjust(S, B, T_W_C, K, hp):
    if B.count_frame == B.keyframe_d:                                       # VO_bundle_adjust.m:3
        Obs_current = np.concatenate([[S.P.shape[1]], S.P.reshape(-1, order="F"), S.X[3]])
        if B.num_key < B.window_size:                                       # :4-15
            B.num_key += 1
            tw = orc.homog2twist(T_W_C)
            if B.tau.shape[1] < B.num_key:
                B.tau = np.hstack([B.tau, np.zeros((6, B.num_key - B.tau.shape[1]))])
            B.tau[:, B.num_key - 1] = tw
            B.observation[B.num_key - 1] = Obs_current
        else:
            o2 = B.observation[1]                                           # :18-25
            m_new_first_key = int(o2[0])
            lm_idx_new_first = o2[len(o2) - m_new_first_key:]
            last_landmark = lm_idx_new_first[-1]
            last_idx = int(np.flatnonzero(B.landmarks[3] == last_landmark)[0]) + 1
            landmarks1 = B.landmarks[:, :last_idx]
            landmarks2 = B.landmarks[:, last_idx:]
            remaining = matlab_find_cols(B.landmarks[3], lm_idx_new_first)
            B.landmarks = np.hstack([landmarks1[:, remaining], landmarks2])
            B.tau = np.hstack([B.tau[:, 1:], orc.homog2twist(T_W_C).reshape(6, 1)])     # :28
            B.observation = B.observation[1:] + [Obs_current]               # :29-30
            B.m = B.landmarks.shape[1]
            observation_loc = []
            for i in range(B.num_key):                                      # :35-41
                obs = B.observation[i].copy()
                n = int(obs[0])
                index_lm_new = matlab_find_cols(B.landmarks[3], obs[len(obs) - n:])
                obs[len(obs) - n:] = index_lm_new + 1
                observation_loc.append(obs)
            state = np.concatenate([B.tau.reshape(-1, order="F"), B.landmarks[:3].reshape(-1, order="F")])   # :43-46
            Obs = np.concatenate([[B.num_key, B.m]] + observation_loc)      # :47-50
            optimized_state, _ = orc.run_ba(state, Obs, K)                  # :51
            ncol = S.est_rot.shape[1]
            for i in range(1, B.num_key + 1):                               # :53-57
                T = orc.twist2homog(optimized_state[(i - 1) * 6:(i - 1) * 6 + 6])
                col = ncol - (B.keyframe_d + 1) * (B.num_key - i) - 1
                S.est_rot[:, col] = np.concatenate([T[:3, 0], T[:3, 1], T[:3, 2]])
                S.est_trans[:, col] = T[:3, 3]
            B.landmarks[:3] = optimized_state[B.num_key * 6:].reshape(3, B.m, order="F")   # :58
            T_C_W = np.linalg.inv(T_W_C)
            for i in range(S.X.shape[1]):                                   # :62-70
                idx = int(np.flatnonzero(B.landmarks[3] == S.X[3, i])[0])
                P_local = T_C_W @ np.append(B.landmarks[:3, idx], 1.0)
                if hp.min_depth < P_local[2] < 1.5 * hp.max_depth:
                    S.X[:3, i] = B.landmarks[:3, idx]
                else:
                    B.landmarks[:3, idx] = S.X[:3, i]
            for i in range(1, B.num_key):                                   # :71-86
                for j in range(1, B.keyframe_d + 1):
                    idx = B.keyframe_d * (i - 1) + j
                    T_f, kp_f, ids_f = B.normal_frame_refine[idx - 1]
                    index = matlab_find_cols(B.landmarks[3], ids_f)
                    T_r, _ = orc.pose_refine(T_f, kp_f, B.landmarks[:3, index], K)
                    col = ncol - (B.keyframe_d + 1) * (B.num_key - i) + j - 1
                    S.est_rot[:, col] = np.concatenate([T_r[:3, 0], T_r[:3, 1], T_r[:3, 2]])
                    S.est_trans[:, col] = T_r[:3, 3]
        B.count_frame = 0
    else:                                                                   # :89-98
        cur = (T_W_C.copy(), np.flipud(S.P).copy(), S.X[3].copy())
        if B.num_normal_frame < (B.window_size - 1) * B.keyframe_d:
            B.num_normal_frame += 1
            B.normal_frame_refine.append(cur)
        else:
            B.normal_frame_refine = B.normal_frame_refine[1:] + [cur]
        B.count_frame += 1
    return S, B


class OraclePipeline:
    def __init__(self, K, first_image, p_W_landmarks, keypoints_xy, T_init_WC, hp, window_size, keyframe_d, rng,
                 draw_triples, draw_octets, rng_map=None):
        self.K, self.hp, self.rng = np.asarray(K, dtype=np.float64), hp, rng
        self.rng_map = rng_map or rng.spawn(1)[0]    # the epipolar filter draws from its own stream
        self.draw_triples, self.draw_octets = draw_triples, draw_octets
        self.S, self.B = init_state(p_W_landmarks, keypoints_xy, np.asarray(T_init_WC, dtype=np.float64), window_size,
                                    keyframe_d)
        self.prev_img = np.ascontiguousarray(first_image)
        self.L_pts = np.fliplr(self.S.P.T)                                  # initialize(KLT_tracker_L, fliplr(S.P'), img)

    def process_frame(self, image):
        S, B, K, hp = self.S, self.B, self.K, self.hp
        image = np.ascontiguousarray(image)
        opts = dict(zip(hp.feature_extract_options[::2], hp.feature_extract_options[1::2]))
        ab, _, _ = orc.detect(image, hp.feature_extract, opts)                                # main_demo.m:377
        location = ab + np.float32(1.0)
        matched_points, validity = klt_step(self.prev_img, image, self.L_pts)               # :380
        matched_valid = np.fliplr(matched_points[validity])                                 # :381
        n = int(validity.sum())
        if n < 3:
            raise RuntimeError("fewer than 3 tracked landmarks")
        # :387-388 -- the RNG advances by exactly the iterations executed
        state = self.rng.bit_generator.state
        samples = self.draw_triples(self.rng, n, 14999)
        res = orc.ransac_localization(matched_valid, S.X[:3, validity].T, K, samples)
        self.rng.bit_generator.state = state
        if res["n_iter"] > 0:
            self.draw_triples(self.rng, n, res["n_iter"])
        if not res["found"]:
            raise RuntimeError("ransacLocalization found no pose")
        T_C_W = np.eye(4); T_C_W[:3, :3] = res["R_C_W"]; T_C_W[:3, 3] = res["t_C_W"]
        T_W_C = np.linalg.inv(T_C_W)                                                        # :391
        inl = res["mask"] > 0
        S.P = matched_valid[inl].T                                                          # :394-396
        S.X = S.X[:, validity][:, inl]
        if hp.is_refine_pose:
            T_W_C, _ = orc.pose_refine(T_W_C, np.flipud(S.P), S.X[:3], K)                   # :398-400
        S.est_trans = np.hstack([S.est_trans, T_W_C[:3, 3:4]])                              # :401-403
        S.est_rot = np.hstack([S.est_rot, T_W_C[:3, :3].reshape(9, 1, order="F")])
        if S.C.shape[1] > 0:                                                                # :406-408
            S, B = update_landmarks(S, B, self.prev_img, image, K, hp, self.draw_octets, self.rng_map)
        if hp.is_BA:                                                                        # :409-411
            S, B = VO_bundle_adjust(S, B, T_W_C, K, hp)
        S = update_candidate(S, location, hp)                                               # :413
        self.L_pts = np.fliplr(S.P.T)                                                       # :417-424
        self.prev_img = image
        S.num_X.append(S.X.shape[1]); S.num_C.append(S.C.shape[1])                          # :426-427
        self.S, self.B = S, B
        return T_W_C


# ---- bootstrap (main_demo.m:252-264) --------------------------------------------------------------

def matching_init(img0, img_seqs, hp):
    """Initialization/matching_init.m:1-45.  'KLT' branch: detect in img0, track through every
    image of img_seqs with a 31x31 / 5-level / 30-iteration tracker, forward-backward bound 3 px;
    a point lost in an intermediate image stays lost.  'Des_match' branch: SURF points and descriptors of
    the first and the last image, matched with matchFeatures' defaults."""
    img0 = np.ascontiguousarray(img0)
    opts = dict(zip(hp.feature_extract_options[::2], hp.feature_extract_options[1::2]))
    if getattr(hp, "init_matching_method", "KLT") == "Des_match":                          # :25-32 (SURF points only)
        img1 = np.ascontiguousarray(img_seqs[-1])
        sets = []
        for im in (img0, img1):
            ab, _, scale, _ = orc.detect_surf(im, opts.get("MetricThreshold", 1000.0), 16)
            loc = ab + np.float32(1.0)
            sets.append((loc, orc.surf_descriptors(im, loc, scale)[0]))
        pairs, _ = orc.match_features(sets[0][1], sets[1][1])
        return sets[0][0][pairs[:, 0]], sets[1][0][pairs[:, 1]]
    ab, _, _ = orc.detect(img0, hp.feature_extract, opts)                                  # :12
    init_points_ = ab + np.float32(1.0)                                                    # :13 (single)
    pts = init_points_.astype(np.float64)
    alive = np.ones(len(pts), dtype=bool)
    prev = img0
    for img in img_seqs:                                                                   # :16-21
        img = np.ascontiguousarray(img)
        out, valid, _ = orc.klt_track(prev, img, pts - 1.0, block=31, levels=5, max_iter=30, max_bidir_error=3.0)
        pts = out + 1.0
        alive &= valid
        prev = img
    return init_points_[alive], pts[alive]                                                 # :23-24


def pose_estimation_init(init_points, matched_points, K, draw_octets, rng, sfm_pose="fundamental"):
    """Initialization/pose_estimation_init.m:1-37.  'essential' (:19): the toolbox's estimateEssentialMatrix is
    closed source; it is defined here as the same RANSAC with that function's default settings (500 trials, Sampson
    distance 0.1 px^2) followed by the nearest essential matrix to K' F K."""
    K = np.asarray(K, dtype=np.float64)
    p1 = np.asarray(init_points, dtype=np.float64)
    p2 = np.asarray(matched_points, dtype=np.float64)
    if sfm_pose == "fundamental":
        res = orc.fmatrix_ransac(p1.T, p2.T, draw_octets(rng, p1.shape[0], 2000), 1e-2, 99.9)  # :7-12
        essMat = K.T @ res["F"] @ K                                                        # :15
    else:
        res = orc.fmatrix_ransac(p1.T, p2.T, draw_octets(rng, p1.shape[0], 500), 0.1, 99.9)    # :19
        U, _, Vt = np.linalg.svd(K.T @ res["F"] @ K)
        essMat = U @ np.diag([1.0, 1.0, 0.0]) @ Vt
    inl = res["mask"]
    init_valid, matched_valid = np.asarray(init_points)[inl], np.asarray(matched_points)[inl]   # :25-26
    R, loc, _ = orc.relative_pose(essMat, K, p1[inl].T, p2[inl].T)                         # :29-33
    T_W_C = np.eye(4)
    T_W_C[:3, :3] = R                                                                      # :35
    T_W_C[:3, 3] = loc
    return T_W_C, init_valid, matched_valid


def triangulation_init(init_valid, matched_valid, T_init_WC, K, hp):
    """Initialization/triangulation_init.m:1-34."""
    T_C_W = np.linalg.inv(T_init_WC)                                                       # :2-4
    K = np.asarray(K, dtype=np.float64)
    proj_mat0 = K @ np.hstack([np.eye(3), np.zeros((3, 1))])                               # :7-8, compProjMat.m:5
    proj_mat1 = K @ np.hstack([T_C_W[:3, :3], T_C_W[:3, 3:4]])
    p0 = np.asarray(init_valid, dtype=np.float64)
    p1 = np.asarray(matched_valid, dtype=np.float64)
    n = p0.shape[0]
    P = orc.linear_triangulation(np.vstack([p0.T, np.ones((1, n))]), np.vstack([p1.T, np.ones((1, n))]), proj_mat0,
                                 proj_mat1)                                                # :11-18
    pts3d = P[:3].T                                                                        # :20
    ok = (pts3d[:, 2] >= hp.min_depth) & (pts3d[:, 2] <= hp.max_depth)                     # :28
    return pts3d[ok], np.asarray(matched_valid)[ok]                                        # :29-31


def bootstrap(img0, img_seqs, K, hp, draw_octets, rng):
    """main_demo.m:252-264."""
    init_points, matched_points = matching_init(img0, img_seqs, hp)
    T_init_WC, init_valid, matched_valid = pose_estimation_init(init_points, matched_points, K, draw_octets, rng,
                                                                getattr(hp, "sfm_pose", "fundamental"))
    pts3d, matched_valid = triangulation_init(init_valid, matched_valid, T_init_WC, K, hp)
    p_W_landmarks = np.asarray(pts3d, dtype=np.float64)
    keypoints = np.asarray(matched_valid, dtype=np.float64)
    if hp.is_refine_pose:
        T_init_WC, _ = orc.pose_refine(T_init_WC, keypoints.T, p_W_landmarks.T, K)
    return p_W_landmarks, keypoints, T_init_WC


# ---- evaluation (utils/quantitative_eval.m) -------------------------------------------------------

def quantitative_eval(ground_truth, est_trans, bootstrap_frames):
    """utils/quantitative_eval.m:5-63 without the plotting: errs 3 x k."""
    gt = np.asarray(ground_truth, dtype=np.float64)
    est = np.asarray(est_trans, dtype=np.float64)
    T = est.shape[1]
    b2 = int(bootstrap_frames[1])
    GT = gt[b2:b2 + T].T                                                    # :5
    GT = np.vstack([GT[0], np.zeros(T), GT[1]])                             # :6
    d = np.linalg.norm(GT[:, 1:] - GT[:, :-1], axis=0)                      # :9
    cps = [c * 0.2 for c in (10, 40, 90, 160, 250, 360)]                    # :11-12
    errs = []
    cur, nxt = 0.0, cps.pop(0)                                              # :15-17
    for i in range(1, T):                                                   # :20, i as in MATLAB
        cur += d[i - 1]                                                     # :21
        if cur > nxt:                                                       # :22
            aligned, _, _ = orc.align_sim3(GT[:, :i], est[:, :i])           # :23-24
            errs.append(np.abs((aligned[:, -1] - aligned[:, 0]) - (GT[:, i - 1] - GT[:, 0])))   # :25-27
            if not cps:                                                     # :53-61
                break
            nxt = cps.pop(0)
    return np.array(errs).T.reshape(3, -1)
