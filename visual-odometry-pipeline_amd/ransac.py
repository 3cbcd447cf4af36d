"""Host-side mirror of the reference's RANSAC localisation interface.

Function names, argument meaning and return values follow
utils/exe7_Ransac/ransacLocalization.m:1-2 and utils/exe7_Ransac/p3p.m:41 of the reference;
the arithmetic runs in libvohip.so on the GPU (no CPU fallback).
Arrays use MATLAB's math layout: keypoints 2xN, landmarks 3xN, K 3x3.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VOHIP_NO_MODEL, VOHIP_OK, VOHIP_SAMPLES_EXHAUSTED, as_f64, colmajor, dptr

MAX_ITERATIONS = 15000  # cap of ransacLocalization.m:128


def triples_from_uniforms(r, n):
    """Maps a (count, 3) array of uniforms in [0, 1) to index triples, 3 distinct indices in [0, n)
    each."""
    r = np.asarray(r, dtype=np.float64)
    a = np.minimum((r[:, 0] * n).astype(np.int64), n - 1)
    b = np.minimum((r[:, 1] * (n - 1)).astype(np.int64), n - 2)
    c = np.minimum((r[:, 2] * (n - 2)).astype(np.int64), n - 3)
    b = b + (b >= a)
    lo = np.minimum(a, b)
    hi = np.maximum(a, b)
    c = c + (c >= lo)
    c = c + (c >= hi)
    return np.stack([a, b, c], axis=1).astype(np.int32)


def draw_triples(rng, n, count):
    """`count` index triples, each 3 distinct indices in [0, n) -- the stand-in for
    datasample(landmarks, 3, 2, 'Replace', false) at ransacLocalization.m:46-47.
    One call consumes exactly 3*count draws of `rng` in triple order, so drawing m triples and
    then k more equals drawing m+k at once."""
    return triples_from_uniforms(rng.random((count, 3)), n)


def sets_from_uniforms(r, n):
    """Maps a (count, k) array of uniforms in [0, 1) to index sets, k distinct indices in [0, n) each: draw j picks
    the floor(u_j (n - j))-th index not picked before (k = 3 gives triples_from_uniforms)."""
    r = np.asarray(r, dtype=np.float64)
    out = np.zeros(r.shape, dtype=np.int64)
    for row in range(r.shape[0]):
        chosen = []
        for j in range(r.shape[1]):
            idx = min(int(r[row, j] * (n - j)), n - j - 1)
            for c in sorted(chosen):
                idx += idx >= c
            chosen.append(idx)
            out[row, j] = idx
    return out.astype(np.int32)


def draw_sextets(rng, n, count):
    """`count` sets of 6 distinct indices in [0, n) -- the stand-in for datasample(landmarks, 6, 2, 'Replace', false)
    at ransacLocalization.m:46-47 with k = 6 (:23).  Consumes exactly 6*count draws, in set order."""
    return sets_from_uniforms(rng.random((count, 6)), n)


def estimatePoseDLT(p, P, K, ctx=None):
    """M_tilde = estimatePoseDLT(p, P, K) (estimatePoseDLT.m:1-60): p n x 2 undistorted (u, v), P n x 3, K 3 x 3 ->
    3 x 4 [R_tilde | alpha t]."""
    ctx = ctx or _lib.default_context()
    p = np.asarray(p, dtype=np.float64).reshape(-1, 2)
    P = np.asarray(P, dtype=np.float64).reshape(-1, 3)
    if p.shape[0] != P.shape[0]:
        raise ValueError("p and P differ in count")
    M = np.zeros(12)
    ctx.check(_lib.load().vohip_estimate_pose_dlt(ctx.handle, dptr(colmajor(p)), dptr(colmajor(P)), p.shape[0],
                                                  dptr(colmajor(K)), dptr(M)))
    return M.reshape(4, 3).T.copy()


def ransacLocalization(matched_query_keypoints, corresponding_landmarks, K, samples=None, rng=None,
                       ctx=None, use_p3p=True):
    """[R_C_W, t_C_W, best_inlier_mask, max_num_inliers_history, num_iteration_history] =
    ransacLocalization(matched_query_keypoints, corresponding_landmarks, K)

    use_p3p: the switch at ransacLocalization.m:9 (true in the reference).  False selects the DLT branch: six
    samples per hypothesis (`samples` is then H x 6), estimatePoseDLT per hypothesis and over the inliers at the end.

    matched_query_keypoints 2xN in (row, col) order, corresponding_landmarks 3xN, K 3x3.
    `samples` (H x 3, 0-based) fixes the hypothesis set; otherwise triples are drawn from `rng`
    (numpy Generator), whose state afterwards is as if exactly n_iter triples had been drawn --
    the reference consumes one datasample call per executed iteration.
    Returns R_C_W = t_C_W = empty arrays when no model reaches 30 inliers
    (ransacLocalization.m:138-140) and best_inlier_mask as a 1xN row of 0/1.
    """
    ctx = ctx or _lib.default_context()
    lib = _lib.load()
    kp = colmajor(matched_query_keypoints)      # (N, 2) rows = (row, col)
    X = colmajor(corresponding_landmarks)       # (N, 3)
    Kc = colmajor(K)
    n = kp.shape[0]
    if X.shape[0] != n:
        raise ValueError("keypoints and landmarks differ in count")
    R = np.zeros(9)
    t = np.zeros(3)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    n_iter = C.c_int(0)
    k = 3 if use_p3p else 6                                                             # :19, :23
    draw = draw_triples if use_p3p else draw_sextets
    entry = lib.vohip_ransac_p3p if use_p3p else lib.vohip_ransac_dlt

    def lib_call(smp, mh, ih, n_iter):
        return entry(ctx.handle, dptr(kp), dptr(X), n, dptr(Kc), smp.ctypes.data_as(_lib.c_int32_p),
                     smp.shape[0], dptr(R), dptr(t), mask.ctypes.data_as(_lib.c_uint8_p), dptr(mh),
                     dptr(ih), C.byref(n_iter))

    if samples is not None:
        samples = np.ascontiguousarray(samples, dtype=np.int32)
        if samples.ndim != 2 or samples.shape[1] != k:
            raise ValueError("samples must be H x %d" % k)
        max_hist = np.zeros(samples.shape[0])
        iter_hist = np.zeros(samples.shape[0])
        st = lib_call(samples, max_hist, iter_hist, n_iter)
        ctx.check(st, allow=(VOHIP_NO_MODEL,))
    else:
        # Draw only as many triples as the adaptive loop turns out to need: first the 104 of the
        # library's first chunk, then 8x more whenever they run out.  The prefix property of
        # draw_triples makes every attempt see the same leading triples.
        if rng is None:
            rng = np.random.default_rng()
        rng_state = rng.bit_generator.state
        have = 104
        while True:
            rng.bit_generator.state = rng_state
            samples = draw(rng, n, have) if n >= k else np.zeros((1, k), np.int32)
            max_hist = np.zeros(samples.shape[0])
            iter_hist = np.zeros(samples.shape[0])
            st = lib_call(samples, max_hist, iter_hist, n_iter)
            if st == VOHIP_SAMPLES_EXHAUSTED and have < MAX_ITERATIONS - 1:
                have = min(8 * have, MAX_ITERATIONS - 1)
                continue
            ctx.check(st, allow=(VOHIP_NO_MODEL,))
            break
        # leave the generator where n_iter single draws would have left it
        rng.bit_generator.state = rng_state
        if n_iter.value > 0:
            rng.random((n_iter.value, k))
    k = n_iter.value
    if st == VOHIP_NO_MODEL:
        return (np.zeros((0, 0)), np.zeros((0, 0)), np.zeros((1, n)), max_hist[:k].copy(),
                iter_hist[:k].copy())
    return (R.reshape(3, 3).T.copy(), t.reshape(3, 1).copy(), mask[:n].reshape(1, n).astype(np.float64),
            max_hist[:k].copy(), iter_hist[:k].copy())


def p3p(worldPoints, imageVectors, ctx=None):
    """poses = p3p(worldPoints, imageVectors): 3x3 world points and 3x3 unit bearing vectors
    (columns) -> 3x16 [C1 R1 C2 R2 C3 R3 C4 R4] with R_i rotating camera to world (p3p.m:35-39).
    Real part only, as every caller of the reference takes it."""
    ctx = ctx or _lib.default_context()
    w = colmajor(worldPoints).reshape(1, 9)
    b = colmajor(imageVectors).reshape(1, 9)
    out = np.zeros(48)
    ctx.check(_lib.load().vohip_p3p(ctx.handle, dptr(as_f64(w)), dptr(as_f64(b)), 1, dptr(out)))
    return out.reshape(16, 3).T.copy()
