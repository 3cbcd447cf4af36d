"""ctypes front-end of the CPU oracle (oracle/libvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by the smoke check and by the
cpu_baseline leg of bench.py.  The shipped package never imports this module.
All arrays follow MATLAB's column-major layout: a MATLAB r x c matrix is passed as a
numpy array of shape (c, r) (C order), i.e. one row per MATLAB column.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvo_oracle.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


def build():
    """Compile the oracle with gcc (incremental make)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.vo_ransac_localization.restype = C.c_int
        _lib.vo_ransac_num_iterations.restype = C.c_double
        _lib.vo_ransac_num_iterations.argtypes = [C.c_int, C.c_int]
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def solve_quartic(factors):
    f = _f64(factors)
    re = np.zeros(4)
    im = np.zeros(4)
    lib().vo_solve_quartic(_d(f), _d(re), _d(im))
    return re + 1j * im


def p3p(world_cols, bearing_cols):
    """world_cols, bearing_cols: (3 points, 3 coords).  Returns (4, 4, 3):
    poses[s, 0] = C, poses[s, 1:4] = columns of R_W_C."""
    w = _f64(world_cols)
    b = _f64(bearing_cols)
    out = np.zeros(48)
    lib().vo_p3p(_d(w), _d(b), _d(out))
    return out.reshape(4, 4, 3)


def project_points(p_cols, K):
    p = _f64(p_cols)
    Kc = _f64(np.asarray(K).T)  # column-major
    out = np.zeros((p.shape[0], 2))
    lib().vo_project_points(_d(p), C.c_int(p.shape[0]), _d(Kc), _d(out))
    return out


def ransac_hypothesis(kp_uv_cols, X_cols, K, idx3, want_mask=False):
    kp = _f64(kp_uv_cols)
    X = _f64(X_cols)
    Kc = _f64(np.asarray(K).T)
    n = kp.shape[0]
    idx = np.ascontiguousarray(idx3, dtype=np.int32)
    counts = np.zeros(4, dtype=np.int32)
    best = C.c_int(0)
    Rt = np.zeros(12)
    mask = np.zeros(n, dtype=np.uint8) if want_mask else None
    lib().vo_ransac_hypothesis(_d(kp), _d(X), C.c_int(n), _d(Kc), idx.ctypes.data_as(_ip),
                               counts.ctypes.data_as(_ip), C.byref(best), _d(Rt),
                               mask.ctypes.data_as(_bp) if want_mask else None)
    return counts, best.value, Rt[:9].reshape(3, 3).T.copy(), Rt[9:].copy(), mask


def ransac_localization(kp_rc_cols, X_cols, K, samples):
    """kp_rc_cols (N,2) rows = (row, col); X_cols (N,3); samples (H,3) 0-based int32.
    Returns dict with R_C_W (3x3 math layout), t_C_W, mask, found, n_iter,
    max_num_inliers_history, num_iteration_history, per_hyp_counts."""
    kp = _f64(kp_rc_cols)
    X = _f64(X_cols)
    Kc = _f64(np.asarray(K).T)
    s = np.ascontiguousarray(samples, dtype=np.int32)
    n, h = kp.shape[0], s.shape[0]
    R = np.zeros(9)
    t = np.zeros(3)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    found = C.c_int(0)
    mh = np.zeros(max(h, 1))
    ih = np.zeros(max(h, 1))
    pc = np.zeros(max(h, 1), dtype=np.int32)
    n_iter = lib().vo_ransac_localization(_d(kp), _d(X), C.c_int(n), _d(Kc),
                                          s.ctypes.data_as(_ip), C.c_int(h), _d(R), _d(t),
                                          mask.ctypes.data_as(_bp), C.byref(found), _d(mh), _d(ih),
                                          pc.ctypes.data_as(_ip))
    m = max(n_iter, 0) if n_iter >= 0 else h
    return dict(R_C_W=R.reshape(3, 3).T.copy(), t_C_W=t, mask=mask[:n], found=bool(found.value),
                n_iter=n_iter, max_num_inliers_history=mh[:m], num_iteration_history=ih[:m],
                per_hyp_counts=pc[:m])


def ransac_num_iterations(max_inliers, n):
    return lib().vo_ransac_num_iterations(int(max_inliers), int(n))


def estimate_pose_dlt(p_uv, P, K):
    """estimatePoseDLT.m: p_uv (n, 2) rows (u, v), P (n, 3) -> M 3x4 [R_tilde | alpha t]."""
    p, Pw = _f64(p_uv), _f64(P)
    Kc = _f64(np.asarray(K).T)
    M = np.zeros(12)
    lib().vo_estimate_pose_dlt.restype = C.c_int
    if lib().vo_estimate_pose_dlt(_d(p), _d(Pw), C.c_int(p.shape[0]), _d(Kc), _d(M)) != 0:
        raise ValueError("estimatePoseDLT needs at least 6 correspondences")
    return M.reshape(3, 4)


def ransac_localization_dlt(kp_rc_cols, X_cols, K, samples):
    """The use_p3p = false branch of ransacLocalization.m.  kp_rc_cols (N,2) rows = (row, col); X_cols (N,3);
    samples (H,6) 0-based int32.  Returns the dict of ransac_localization plus M_hyp (n_iter, 3, 4)."""
    kp = _f64(kp_rc_cols)
    X = _f64(X_cols)
    Kc = _f64(np.asarray(K).T)
    s = np.ascontiguousarray(samples, dtype=np.int32)
    n, h = kp.shape[0], s.shape[0]
    R, t = np.zeros(9), np.zeros(3)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    found = C.c_int(0)
    mh, ih = np.zeros(max(h, 1)), np.zeros(max(h, 1))
    pc = np.zeros(max(h, 1), dtype=np.int32)
    Mh = np.zeros((max(h, 1), 12))
    lib().vo_ransac_localization_dlt.restype = C.c_int
    n_iter = lib().vo_ransac_localization_dlt(_d(kp), _d(X), C.c_int(n), _d(Kc), s.ctypes.data_as(_ip), C.c_int(h), _d(R),
                                              _d(t), mask.ctypes.data_as(_bp), C.byref(found), _d(mh), _d(ih),
                                              pc.ctypes.data_as(_ip), _d(Mh))
    m = max(n_iter, 0) if n_iter >= 0 else h
    return dict(R_C_W=R.reshape(3, 3).T.copy(), t_C_W=t, mask=mask[:n], found=bool(found.value), n_iter=n_iter,
                max_num_inliers_history=mh[:m], num_iteration_history=ih[:m], per_hyp_counts=pc[:m],
                M_hyp=Mh[:m].reshape(-1, 3, 4))


# ---- geometry ---------------------------------------------------------------------------

def twist2homog(twist):
    t = _f64(twist).reshape(6)
    H = np.zeros(16)
    lib().vo_twist2homog(_d(t), _d(H))
    return H.reshape(4, 4).T.copy()


def homog2twist(H):
    Hc = _f64(np.asarray(H).T)
    t = np.zeros(6)
    lib().vo_homog2twist(_d(Hc), _d(t))
    return t


def _proj_cm(M):
    """3x4 (or n x 3 x 4) projection matrices -> column-major memory image(s)."""
    M = np.asarray(M, dtype=np.float64)
    if M.ndim == 2:
        return np.ascontiguousarray(M.T), 0
    return np.ascontiguousarray(np.transpose(M, (0, 2, 1))), 1


def linear_triangulation(p1, p2, M1, M2):
    """p1, p2: 3xn (math layout).  M: 3x4 or n x 3 x 4.  Returns 4xn."""
    a = _f64(np.asarray(p1).T)
    b = _f64(np.asarray(p2).T)
    n = a.shape[0]
    m1, f1 = _proj_cm(M1)
    m2, f2 = _proj_cm(M2)
    P = np.zeros((n, 4))
    lib().vo_linear_triangulation(_d(a), _d(b), _d(m1), _d(m2), C.c_int(f1), C.c_int(f2), C.c_int(n), _d(P))
    return P.T.copy()


def triangulation_refinement(P_init, p1, p2, M1, M2):
    Pi = _f64(np.asarray(P_init).T)
    a = _f64(np.asarray(p1).T)
    b = _f64(np.asarray(p2).T)
    n = Pi.shape[0]
    m1, _ = _proj_cm(M1)
    m2, _ = _proj_cm(M2)
    P = np.zeros((n, 4))
    lib().vo_triangulation_refinement(_d(Pi), _d(a), _d(b), C.c_int(a.shape[1]), _d(m1), _d(m2), C.c_int(n), _d(P))
    return P.T.copy()


def candidate_filter(new_xy, C_rc, P_rc, r):
    """new_xy (n_new, 2) float32 (x, y); C_rc 2xM, P_rc 2xN (row; col).  Returns bool keep[n_new]."""
    xy = np.ascontiguousarray(np.asarray(new_xy, dtype=np.float32).T)  # column-major n_new x 2
    n_new = xy.shape[1]
    Cc = _f64(np.asarray(C_rc).reshape(2, -1).T)
    Pc = _f64(np.asarray(P_rc).reshape(2, -1).T)
    keep = np.zeros(max(n_new, 1), dtype=np.uint8)
    lib().vo_candidate_filter(xy.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(n_new), _d(Cc), C.c_int(Cc.shape[0]),
                              _d(Pc), C.c_int(Pc.shape[0]), C.c_double(r), keep.ctypes.data_as(_bp))
    return keep[:n_new].astype(bool)


def candidate_triangulate(cur_rc, first_rc, T_first, T_W_C, K, angle_threshold, min_depth, max_depth, is_BA):
    """cur_rc, first_rc 2xn; T_first 12xn; T_W_C 4x4.  Returns status (n,), X (3xn), angles (n,)."""
    c = _f64(np.asarray(cur_rc).T)
    f = _f64(np.asarray(first_rc).T)
    Tf = _f64(np.asarray(T_first).T)
    n = c.shape[0]
    Tc = _f64(np.asarray(T_W_C).T)
    Kc = _f64(np.asarray(K).T)
    status = np.zeros(max(n, 1), dtype=np.uint8)
    X = np.zeros((max(n, 1), 3))
    ang = np.zeros(max(n, 1))
    lib().vo_candidate_triangulate(_d(c), _d(f), _d(Tf), C.c_int(n), _d(Tc), _d(Kc), C.c_double(angle_threshold),
                                   C.c_double(min_depth), C.c_double(max_depth), C.c_int(int(is_BA)),
                                   status.ctypes.data_as(_bp), _d(X), _d(ang))
    return status[:n], X[:n].T.copy(), ang[:n]


# ---- KLT ----------------------------------------------------------------------------------

def pyramid_level(img_ab, level):
    """img_ab: uint8 array of shape (B, A) (slow axis first).  Returns float32 (B_l, A_l)."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    out = np.zeros(A * B, dtype=np.float32)
    Ao, Bo = C.c_int(0), C.c_int(0)
    lib().vo_pyramid_level(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_int(level),
                           out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(Ao), C.byref(Bo))
    return out[:Ao.value * Bo.value].reshape(Bo.value, Ao.value).copy()


def set_threads(n=0):
    """Threads the tracker's loop over the points uses from now on (0: all the cores OpenMP finds); returns the number in force."""
    f = lib().vo_set_threads
    f.restype = C.c_int
    return int(f(C.c_int(int(n))))


def klt_track(prev_ab, cur_ab, pts_ab, block=21, levels=5, max_iter=20, max_bidir_error=6.0):
    """prev_ab, cur_ab uint8 (B, A); pts_ab (n, 2) 0-based (a, b) coordinates.
    Returns (out (n, 2), valid bool (n,), fb_err (n,))."""
    p = np.ascontiguousarray(prev_ab, dtype=np.uint8)
    q = np.ascontiguousarray(cur_ab, dtype=np.uint8)
    B, A = p.shape
    pts = np.ascontiguousarray(np.asarray(pts_ab, dtype=np.float64).T)  # 2 x n -> [a..., b...]
    n = pts.shape[1]
    out = np.zeros((2, max(n, 1)))
    valid = np.zeros(max(n, 1), dtype=np.uint8)
    fb = np.zeros(max(n, 1))
    if n:
        outc = np.zeros(2 * n)
        lib().vo_klt_track(p.ctypes.data_as(_bp), q.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), _d(pts), C.c_int(n),
                           C.c_int(block), C.c_int(levels), C.c_int(max_iter), C.c_double(max_bidir_error),
                           _d(outc), valid.ctypes.data_as(_bp), _d(fb))
        out = outc.reshape(2, n)
    return out[:, :n].T.copy(), valid[:n].astype(bool), fb[:n]


# ---- corner detection -----------------------------------------------------------------------

def corner_response(img_ab, filter_size=5, method=0):
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    R = np.zeros((B, A), dtype=np.float32)
    lib().vo_corner_response(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_int(filter_size), C.c_int(method),
                             R.ctypes.data_as(C.POINTER(C.c_float)))
    return R


def detect_corners(img_ab, min_quality=0.01, filter_size=5, border=16, method=0, cap=16384):
    """img_ab uint8 (B, A).  Returns ab (n, 2) float32 0-based sub-pixel (a, b), metric (n,), total found."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    ab = np.zeros(2 * cap, dtype=np.float32)
    metric = np.zeros(cap, dtype=np.float32)
    lib().vo_detect_corners.restype = C.c_int
    total = lib().vo_detect_corners(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_double(min_quality),
                                    C.c_int(filter_size), C.c_int(border), C.c_int(method),
                                    ab.ctypes.data_as(C.POINTER(C.c_float)), metric.ctypes.data_as(C.POINTER(C.c_float)),
                                    C.c_int(cap))
    n = min(total, cap)
    return np.stack([ab[:n], ab[cap:cap + n]], axis=1).copy(), metric[:n].copy(), total


def fast_score(img_ab, threshold):
    """Segment-test scores of every pixel (0 where no corner at `threshold`): int32 (B, A)."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    S = np.zeros((B, A), dtype=np.int32)
    lib().vo_fast_score(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_int(int(threshold)),
                        S.ctypes.data_as(C.POINTER(C.c_int)))
    return S


def detect_fast(img_ab, min_contrast=0.2, min_quality=0.1, border=16, cap=16384):
    """img_ab uint8 (B, A).  Returns ab (n, 2) float32 0-based (a, b), metric (n,), total found."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    ab = np.zeros(2 * cap, dtype=np.float32)
    metric = np.zeros(cap, dtype=np.float32)
    lib().vo_detect_fast.restype = C.c_int
    total = lib().vo_detect_fast(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_double(min_contrast),
                                 C.c_double(min_quality), C.c_int(border), ab.ctypes.data_as(C.POINTER(C.c_float)),
                                 metric.ctypes.data_as(C.POINTER(C.c_float)), C.c_int(cap))
    n = min(total, cap)
    return np.stack([ab[:n], ab[cap:cap + n]], axis=1).copy(), metric[:n].copy(), total


SURF_SIZES = ((9, 15, 21, 27), (15, 27, 39, 51), (27, 51, 75, 99))


def surf_response(img_ab, octave, level):
    """Determinant-of-Hessian map of (octave, level): float32 (ceil(B / s), ceil(A / s)), s = 2**octave."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    s = 1 << octave
    R = np.zeros(((B + s - 1) // s, (A + s - 1) // s), dtype=np.float32)
    lib().vo_surf_response(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_int(octave), C.c_int(level),
                           R.ctypes.data_as(C.POINTER(C.c_float)))
    return R


def detect_surf(img_ab, metric_threshold=1000.0, border=16, cap=16384):
    """img_ab uint8 (B, A).  Returns ab (n, 2) float32 0-based (a, b), metric (n,), scale (n,), total found."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    ab = np.zeros(2 * cap, dtype=np.float32)
    metric = np.zeros(cap, dtype=np.float32)
    scale = np.zeros(cap, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().vo_detect_surf.restype = C.c_int
    total = lib().vo_detect_surf(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_double(metric_threshold),
                                 C.c_int(border), ab.ctypes.data_as(fp), metric.ctypes.data_as(fp),
                                 scale.ctypes.data_as(fp), C.c_int(cap))
    n = min(total, cap)
    return np.stack([ab[:n], ab[cap:cap + n]], axis=1).copy(), metric[:n].copy(), scale[:n].copy(), total


def surf_descriptors(img, xy1, scale, upright=False):
    """img uint8 (rows, cols) in either storage order; xy1 (n, 2) 1-based (x, y) float32 (SURFPoints.Location); scale (n,).
    Returns features (n, 64) float32 and orientation (n,) float32."""
    img = np.asarray(img)
    swap = bool(img.flags.f_contiguous and not img.flags.c_contiguous)
    store = np.ascontiguousarray(img.T) if swap else np.ascontiguousarray(img)
    B, A = store.shape
    xy1 = np.asarray(xy1, dtype=np.float32).reshape(-1, 2)
    n = xy1.shape[0]
    cols = np.ascontiguousarray(np.concatenate([xy1[:, 0], xy1[:, 1]]), dtype=np.float32)
    sc = np.ascontiguousarray(scale, dtype=np.float32)
    feats = np.zeros((n, 64), dtype=np.float32)
    ori = np.zeros(n, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().vo_surf_descriptors(store.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_int(int(swap)), cols.ctypes.data_as(fp),
                              sc.ctypes.data_as(fp), C.c_int(n), C.c_int(int(upright)), feats.ctypes.data_as(fp),
                              ori.ctypes.data_as(fp))
    return feats, ori


def match_features(f0, f1, match_threshold=1.0, max_ratio=0.6):
    """matchFeatures(f0, f1) with the toolbox defaults.  Returns index pairs (m, 2) 0-based and their distances."""
    f0 = np.ascontiguousarray(f0, dtype=np.float32)
    f1 = np.ascontiguousarray(f1, dtype=np.float32)
    n0, dim = f0.shape
    n1 = f1.shape[0]
    pairs = np.zeros((max(n0, 1), 2), dtype=np.int32)
    metric = np.zeros(max(n0, 1), dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().vo_match_features.restype = C.c_int
    m = lib().vo_match_features(f0.ctypes.data_as(fp), C.c_int(n0), f1.ctypes.data_as(fp), C.c_int(n1), C.c_int(dim),
                                C.c_double(match_threshold), C.c_double(max_ratio),
                                pairs.ctypes.data_as(C.POINTER(C.c_int)), metric.ctypes.data_as(fp))
    return pairs[:m].copy(), metric[:m].copy()


def brisk_layer(img_ab, k):
    """Layer k (0..7, scales 1, 1.5, 2, 3, 4, 6, 8, 12) of the BRISK pyramid: uint8 (B_k, A_k)."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    out = np.zeros(A * B + 1, dtype=np.uint8)
    Ak, Bk = C.c_int(0), C.c_int(0)
    lib().vo_brisk_layer(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_int(k), out.ctypes.data_as(_bp),
                         C.byref(Ak), C.byref(Bk))
    return out[:Ak.value * Bk.value].reshape(Bk.value, Ak.value).copy()


def fast58_score(img_ab):
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    S = np.zeros((B, A), dtype=np.int32)
    lib().vo_fast58_score(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), S.ctypes.data_as(C.POINTER(C.c_int)))
    return S


def detect_brisk(img_ab, min_contrast=0.2, min_quality=0.1, num_octaves=4, border=16, cap=16384):
    """img_ab uint8 (B, A).  Returns ab (n, 2) float32 0-based (a, b), metric (n,), scale (n,), total found."""
    img = np.ascontiguousarray(img_ab, dtype=np.uint8)
    B, A = img.shape
    ab = np.zeros(2 * cap, dtype=np.float32)
    metric = np.zeros(cap, dtype=np.float32)
    scale = np.zeros(cap, dtype=np.float32)
    fp = C.POINTER(C.c_float)
    lib().vo_detect_brisk.restype = C.c_int
    total = lib().vo_detect_brisk(img.ctypes.data_as(_bp), C.c_int(A), C.c_int(B), C.c_double(min_contrast),
                                  C.c_double(min_quality), C.c_int(num_octaves), C.c_int(border), ab.ctypes.data_as(fp),
                                  metric.ctypes.data_as(fp), scale.ctypes.data_as(fp), C.c_int(cap))
    n = min(total, cap)
    return np.stack([ab[:n], ab[cap:cap + n]], axis=1).copy(), metric[:n].copy(), scale[:n].copy(), total


def detect(img_ab, method, options, border=16):
    """genKeypoints.m:1-33 for the detectors: options as the name-value dict of the reference."""
    if method == "SURF":
        ab, metric, _, total = detect_surf(img_ab, options.get("MetricThreshold", 1000.0), border)
        return ab, metric, total
    if method == "BRISK":
        ab, metric, _, total = detect_brisk(img_ab, options.get("MinContrast", 0.2), options.get("MinQuality", 0.1),
                                            options.get("NumOctaves", 4), border)
        return ab, metric, total
    if method == "FAST":
        return detect_fast(img_ab, options.get("MinContrast", 0.2), options.get("MinQuality", 0.1), border)
    return detect_corners(img_ab, min_quality=options.get("MinQuality", 0.01), filter_size=5, border=border,
                          method=0 if method == "Harris" else 1)


# ---- bundle adjustment / pose refinement ----------------------------------------------------

def ba_error(state, obs, K):
    """baError(hidden_state, observations, K): returns 2 x Nobs."""
    s = _f64(state).reshape(-1)
    o = _f64(obs).reshape(-1)
    Kc = _f64(np.asarray(K).T)
    r = np.zeros(2 * len(o))
    lib().vo_ba_error.restype = C.c_int
    n = lib().vo_ba_error(_d(s), _d(o), C.c_int(len(o)), _d(Kc), _d(r))
    if n < 0:
        raise ValueError("malformed observation vector")
    return r[:2 * n].reshape(n, 2).T.copy()


def run_ba(state, obs, K, max_iter=21, ftol=1e-6, xtol=1e-6):   # lsqnonlin's default tolerances (runBA.m:77-78)
    s = _f64(state).reshape(-1).copy()
    o = _f64(obs).reshape(-1)
    Kc = _f64(np.asarray(K).T)
    stats = np.zeros(4)
    lib().vo_run_ba.restype = C.c_int
    rc = lib().vo_run_ba(_d(s), _d(o), C.c_int(len(o)), _d(Kc), C.c_int(max_iter), C.c_double(ftol), C.c_double(xtol),
                         _d(stats))
    if rc < 0:
        raise ValueError("malformed observation vector")
    return s, dict(iterations=int(stats[0]), cost0=stats[1], cost=stats[2], lam=stats[3])


def pose_refine(T_W_C, kp_uv, X, K, max_iter=100, ftol=1e-6, xtol=1e-6):   # lsqnonlin's default tolerances (T_refinement.m:4)
    """T_refinement(T_W_C, keypoints (2xN (u,v)), landmarks (3xN), K)."""
    T = _f64(np.asarray(T_W_C).T).copy()
    kp = _f64(np.asarray(kp_uv).T)
    Xc = _f64(np.asarray(X).T)
    Kc = _f64(np.asarray(K).T)
    stats = np.zeros(3)
    lib().vo_pose_refine(_d(T), _d(kp), _d(Xc), C.c_int(kp.shape[0]), _d(Kc), C.c_int(max_iter), C.c_double(ftol),
                         C.c_double(xtol), _d(stats))
    return T.reshape(4, 4).T.copy(), dict(iterations=int(stats[0]), cost0=stats[1], cost=stats[2])


# ---- trajectory alignment (evaluation) ----------------------------------------------------------

def align_error(x, gt, est):
    """x (7,), gt / est 3xn.  Returns the 3n error vector of alignError.m."""
    g = _f64(np.asarray(gt).T)
    e = _f64(np.asarray(est).T)
    out = np.zeros(3 * g.shape[0])
    lib().vo_align_error(_d(_f64(np.asarray(x).reshape(-1))), _d(g), _d(e), C.c_int(g.shape[0]), _d(out))
    return out


def align_sim3(gt, est, max_iter=400, ftol=1e-6, xtol=1e-6):   # lsqnonlin's defaults (alignEstimateToGroundTruth.m:17-18)
    """gt, est 3xn.  Returns (aligned 3xn, x (7,), stats dict)."""
    g = _f64(np.asarray(gt).T)
    e = _f64(np.asarray(est).T)
    n = g.shape[0]
    x = np.zeros(7)
    out = np.zeros((n, 3))
    stats = np.zeros(3)
    lib().vo_align_sim3(_d(g), _d(e), C.c_int(n), C.c_int(max_iter), C.c_double(ftol), C.c_double(xtol), _d(x), _d(out),
                        _d(stats))
    return out.T.copy(), x, dict(iterations=int(stats[0]), cost0=stats[1], cost=stats[2])


# ---- relative pose (bootstrap) -----------------------------------------------------------------

def relative_pose(E, K, p1, p2):
    """E, K 3x3; p1, p2 2xn (x; y).  Returns (orientation 3x3, location (3,), valid fraction) or None."""
    a = _f64(np.asarray(p1).T)
    b = _f64(np.asarray(p2).T)
    R = np.zeros(9)
    loc = np.zeros(3)
    frac = C.c_double(0.0)
    lib().vo_relative_pose.restype = C.c_int
    ok = lib().vo_relative_pose(_d(_f64(np.asarray(E).T)), _d(_f64(np.asarray(K).T)), _d(a), _d(b), C.c_int(a.shape[0]),
                                _d(R), _d(loc), C.byref(frac))
    if not ok:
        return None
    return R.reshape(3, 3).T.copy(), loc, frac.value


# ---- fundamental-matrix RANSAC ---------------------------------------------------------------

def fmatrix_ransac(p1, p2, samples, threshold=5e-2, confidence=99.0):
    """p1, p2: 2xn; samples (T, 8) 0-based.  Returns dict(mask, best_count, trials, F, counts)."""
    a = _f64(np.asarray(p1).T)
    b = _f64(np.asarray(p2).T)
    s = np.ascontiguousarray(samples, dtype=np.int32)
    n, T = a.shape[0], s.shape[0]
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    best = C.c_int(0)
    F = np.zeros(9)
    counts = np.zeros(max(T, 1), dtype=np.int32)
    lib().vo_fmatrix_ransac.restype = C.c_int
    trials = lib().vo_fmatrix_ransac(_d(a), _d(b), C.c_int(n), s.ctypes.data_as(_ip), C.c_int(T), C.c_double(threshold),
                                     C.c_double(confidence),
                                     mask.ctypes.data_as(_bp), C.byref(best), _d(F), counts.ctypes.data_as(_ip))
    return dict(mask=mask[:n].astype(bool), best_count=best.value, trials=trials, F=F.reshape(3, 3), counts=counts[:trials])
