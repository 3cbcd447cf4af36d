"""Host-side mirror of estimateFundamentalMatrix as update_landmarks.m:13-14 calls it
('Method','RANSAC'); the arithmetic runs in libvohip.so on the GPU, no CPU fallback."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VOHIP_NO_MODEL, dptr


def octets_from_uniforms(r, n):
    """Maps a (count, 8) array of uniforms in [0, 1) to `count` sets of 8 distinct indices in
    [0, n): draw k picks the floor(r_k (n - k))-th index not picked before (a partial Fisher-Yates
    shuffle written without the array), so exactly 8 uniforms are consumed per set."""
    r = np.asarray(r, dtype=np.float64)
    count = r.shape[0]
    chosen = np.zeros((count, 0), dtype=np.int64)      # kept sorted along axis 1
    out = np.zeros((count, 8), dtype=np.int64)
    for k in range(8):
        idx = np.minimum((r[:, k] * (n - k)).astype(np.int64), n - k - 1)
        for m in range(k):
            idx = idx + (idx >= chosen[:, m])
        out[:, k] = idx
        chosen = np.sort(np.concatenate([chosen, idx[:, None]], axis=1), axis=1)
    return out.astype(np.int32)


def draw_octets(rng, n, count):
    """`count` sets of 8 distinct indices in [0, n): the stand-in for the toolbox's internal
    sampling.  Consumes exactly 8 * count draws of `rng`."""
    if n < 8:
        raise ValueError("need at least 8 points")
    return octets_from_uniforms(rng.random((count, 8)), n)


def estimateFundamentalMatrix(matchedPoints1, matchedPoints2, NumTrials=500, DistanceThreshold=0.01,
                              Confidence=99.0, Method="RANSAC", samples=None, rng=None, ctx=None):
    """[F, inliersIndex] = estimateFundamentalMatrix(matchedPoints1, matchedPoints2, 'Method',
    'RANSAC', 'NumTrials', T, 'DistanceThreshold', d, 'Confidence', c).  Points are N-by-2.  `samples` (T x 8,
    0-based) fixes the trials; otherwise they are drawn from `rng` (numpy Generator)."""
    if Method != "RANSAC":
        raise NotImplementedError("only Method = 'RANSAC' is built (the one the pipeline uses)")
    ctx = ctx or _lib.default_context()
    p1 = np.ascontiguousarray(np.asarray(matchedPoints1, dtype=np.float64).reshape(-1, 2))
    p2 = np.ascontiguousarray(np.asarray(matchedPoints2, dtype=np.float64).reshape(-1, 2))
    n = p1.shape[0]
    if p2.shape[0] != n:
        raise ValueError("point sets differ in count")
    if samples is None:
        samples = draw_octets(rng or np.random.default_rng(), n, int(NumTrials)) if n >= 8 else np.zeros((1, 8))
    s = np.ascontiguousarray(samples, dtype=np.int32)
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    best, trials = C.c_int(0), C.c_int(0)
    F = np.zeros(9)
    st = _lib.load().vohip_fmatrix_ransac(ctx.handle, dptr(p1), dptr(p2), n, s.ctypes.data_as(_lib.c_int32_p),
                                          s.shape[0], float(DistanceThreshold), float(Confidence), mask.ctypes.data_as(_lib.c_uint8_p),
                                          C.byref(best), C.byref(trials), dptr(F))
    ctx.check(st, allow=(VOHIP_NO_MODEL,))
    return F.reshape(3, 3), mask[:n].astype(bool)
