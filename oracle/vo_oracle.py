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
