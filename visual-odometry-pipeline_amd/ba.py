"""Host-side mirrors of the reference's bundle-adjustment interface (same names and argument
meaning): runBA (utils/exe9_BA/runBA.m:2), baError (utils/exe9_BA/baError.m:2) and T_refinement
(utils/T_refinement.m:1).  The arithmetic runs in libvohip.so on the GPU; no CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_f64, colmajor, dptr

BA_MAX_ITER = 21          # the reference's budget: lsqnonlin MaxIter 1 + MaxIter 20 (runBA.m:65-78)
POSE_MAX_ITER = 100       # T_refinement.m:4
# runBA.m:77-78 leaves the tolerances of its second, 20-iteration stage at lsqnonlin's defaults
# (FunctionTolerance = StepTolerance = 1e-6)
FTOL = 1e-6
XTOL = 1e-6
# T_refinement.m:4 leaves lsqnonlin's tolerances at their defaults (FunctionTolerance = StepTolerance = 1e-6)
POSE_FTOL = 1e-6
POSE_XTOL = 1e-6


def baError(hidden_state, observations, K, ctx=None):
    """error_terms = baError(hidden_state, observations, K): 2 x Nobs reprojection errors."""
    ctx = ctx or _lib.default_context()
    s = as_f64(np.asarray(hidden_state).reshape(-1))
    o = as_f64(np.asarray(observations).reshape(-1))
    r = np.zeros(max(2 * len(o), 2))
    n = C.c_int(0)
    ctx.check(_lib.load().vohip_ba_error(ctx.handle, dptr(s), len(s), dptr(o), len(o), dptr(colmajor(K)), dptr(r),
                                         C.byref(n)))
    return r[:2 * n.value].reshape(n.value, 2).T.copy()


def runBA(hidden_state, observations, K, max_iter=BA_MAX_ITER, ftol=FTOL, xtol=XTOL, return_stats=False, ctx=None):
    """hidden_state = runBA(hidden_state, observations, K)"""
    ctx = ctx or _lib.default_context()
    shape = np.asarray(hidden_state).shape
    s = as_f64(np.asarray(hidden_state).reshape(-1)).copy()
    o = as_f64(np.asarray(observations).reshape(-1))
    stats = np.zeros(4)
    ctx.check(_lib.load().vohip_ba_run(ctx.handle, dptr(s), len(s), dptr(o), len(o), dptr(colmajor(K)), int(max_iter),
                                       float(ftol), float(xtol), dptr(stats)))
    out = s.reshape(shape)
    if return_stats:
        return out, dict(iterations=int(stats[0]), cost0=stats[1], cost=stats[2], lam=stats[3])
    return out


def T_refinement(T_W_C, keypoints, landmarks, K, max_iter=POSE_MAX_ITER, ftol=POSE_FTOL, xtol=POSE_XTOL, return_stats=False,
                 ctx=None):
    """refined_T_W_C = T_refinement(T_W_C, keypoints, landmarks, K): keypoints 2xN (u, v),
    landmarks 3xN."""
    ctx = ctx or _lib.default_context()
    T = colmajor(T_W_C).copy()
    kp = colmajor(np.asarray(keypoints, dtype=np.float64).reshape(2, -1))
    X = colmajor(np.asarray(landmarks, dtype=np.float64).reshape(3, -1))
    if kp.shape[0] != X.shape[0]:
        raise ValueError("keypoints and landmarks differ in count")
    stats = np.zeros(3)
    ctx.check(_lib.load().vohip_pose_refine(ctx.handle, dptr(T), dptr(kp), dptr(X), kp.shape[0], dptr(colmajor(K)),
                                            int(max_iter), float(ftol), float(xtol), dptr(stats)))
    out = T.reshape(4, 4).T.copy()
    if return_stats:
        return out, dict(iterations=int(stats[0]), cost0=stats[1], cost=stats[2])
    return out


def T_refinement_batch(problems, K, max_iter=POSE_MAX_ITER, ftol=POSE_FTOL, xtol=POSE_XTOL, ctx=None):
    """[T_refinement(T, kp, X, K) for (T, kp, X) in problems] in one launch and one synchronisation (the
    re-refinement of the buffered normal frames, VO_bundle_adjust.m:71-86): the same results, bit for bit."""
    ctx = ctx or _lib.default_context()
    if not problems:
        return []
    T = np.concatenate([colmajor(p[0]).reshape(-1) for p in problems])
    kps = [colmajor(np.asarray(p[1], dtype=np.float64).reshape(2, -1)) for p in problems]
    Xs = [colmajor(np.asarray(p[2], dtype=np.float64).reshape(3, -1)) for p in problems]
    for kp, X in zip(kps, Xs):
        if kp.shape[0] != X.shape[0]:
            raise ValueError("keypoints and landmarks differ in count")
    n = np.array([kp.shape[0] for kp in kps], dtype=np.int32)
    kp_all = np.ascontiguousarray(np.concatenate(kps).reshape(-1)) if n.sum() else np.zeros(2)
    X_all = np.ascontiguousarray(np.concatenate(Xs).reshape(-1)) if n.sum() else np.zeros(3)
    ctx.check(_lib.load().vohip_pose_refine_batch(ctx.handle, len(problems), dptr(T), dptr(kp_all), dptr(X_all),
                                                  n.ctypes.data_as(_lib.c_int_p), dptr(colmajor(K)), int(max_iter),
                                                  float(ftol), float(xtol), None))
    return [T[16 * b:16 * b + 16].reshape(4, 4).T.copy() for b in range(len(problems))]
