"""Pins the corner-detector part of the CPU oracle against a direct numpy/scipy evaluation of
its definition and the golden vectors.  detectHarrisFeatures itself is closed source: parity with
MATLAB's detector is unpinned."""
import os

import numpy as np
from scipy.ndimage import correlate1d, maximum_filter

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _response_numpy(img, method=0):
    I = np.pad(img.astype(np.float64) / 255.0, 3, mode="edge")
    gx = I[:, 2:] - I[:, :-2]
    gy = I[2:, :] - I[:-2, :]
    gx = gx[1:-1, :]
    gy = gy[:, 1:-1]
    sigma = 5 / 3.0
    w = np.exp(-np.arange(-2, 3) ** 2 / (2 * sigma ** 2))
    w /= w.sum()

    def g(p):
        return correlate1d(correlate1d(p, w, axis=1, mode="nearest"), w, axis=0, mode="nearest")[2:-2, 2:-2]
    A, B, C = g(gx * gx), g(gy * gy), g(gx * gy)
    if method == 0:
        return (A * B - C * C) - 0.04 * (A + B) ** 2
    return ((A + B) - np.sqrt((A - B) ** 2 + 4 * C * C)) / 2


def test_response_matches_numpy(oracle):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    for method in (0, 1):
        R = oracle.corner_response(g["img"], method=method)
        ref = _response_numpy(g["img"], method)
        assert R.shape == ref.shape
        scale = np.abs(ref).max()
        assert np.abs(R - ref).max() < 2e-5 * scale


def test_detections_are_thresholded_strict_local_maxima(oracle):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    img = g["img"]
    R = oracle.corner_response(img)
    ab, metric, total = oracle.detect_corners(img, min_quality=2e-4, border=16)
    assert total == len(ab) and 50 < total < 5000
    ia = np.rint(ab).astype(int)
    assert np.all(np.abs(ab - ia) <= 0.5)
    # every detection is a strict 3x3 maximum above the threshold, inside the border
    mx = maximum_filter(R, size=3, mode="nearest")
    thr = np.float32(2e-4) * R.max()
    for (a, b), m in zip(ia, metric):
        assert R[b, a] == m == mx[b, a] and m > thr
        assert 16 <= a < img.shape[1] - 16 and 16 <= b < img.shape[0] - 16
    # and nothing is missed
    cand = (R == mx) & (R > thr)
    cand[:16] = cand[-16:] = False
    cand[:, :16] = cand[:, -16:] = False
    # strictness: drop plateaus
    n_strict = 0
    for b, a in zip(*np.nonzero(cand)):
        nb = R[b - 1:b + 2, a - 1:a + 2].copy()
        nb[1, 1] = -np.inf
        n_strict += R[b, a] > nb.max()
    assert n_strict == total
    # storage order: ascending b * A + a
    key = ia[:, 1] * img.shape[1] + ia[:, 0]
    assert np.all(np.diff(key) > 0)


def test_planted_corner_is_found_with_subpixel_position(oracle):
    img = np.full((120, 160), 40, np.uint8)
    img[50:, 70:] = 200   # one corner at (x, y) = (69.5, 49.5) in 0-based pixel-centre coordinates
    ab, metric, total = oracle.detect_corners(img, min_quality=0.01, border=8)
    assert total >= 1
    best = ab[np.argmax(metric)]
    assert abs(best[0] - 69.5) < 1.5 and abs(best[1] - 49.5) < 1.5


def test_capacity_and_flat_image(oracle):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    ab, metric, total = oracle.detect_corners(g["img"], min_quality=2e-4, cap=10)
    assert len(ab) == 10 and total > 10
    ab, metric, total = oracle.detect_corners(np.full((64, 64), 7, np.uint8))
    assert total == 0


def test_harris_oracle_matches_golden(oracle):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    ab, metric, total = oracle.detect_corners(g["img"], min_quality=2e-4, border=16)
    np.testing.assert_array_equal(ab + 1.0, g["xy"])
    np.testing.assert_array_equal(metric, g["metric"])
    ab, metric, _ = oracle.detect_corners(g["img"], min_quality=1e-2, border=16, method=1)
    np.testing.assert_array_equal(ab + 1.0, g["xy_mineigen"])


# ---- FAST --------------------------------------------------------------------------------------
# detectFASTFeatures is closed source as well: what is pinned here is the oracle against a second,
# vectorised statement of the published segment test and against images whose corners are known.

_CIRCLE = [(0, -3), (1, -3), (2, -2), (3, -1), (3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0),
           (-3, -1), (-2, -2), (-1, -3)]


def _fast_score_numpy(img, threshold):
    """score = (largest t for which 9 contiguous circle pixels are all brighter than p + t or all darker
    than p - t), 0 where that is below `threshold` or closer than 3 px to an edge."""
    I = img.astype(np.int64)
    H, W = I.shape
    p = I[3:H - 3, 3:W - 3]
    d = np.stack([I[3 + db:H - 3 + db, 3 + da:W - 3 + da] - p for da, db in _CIRCLE])      # 16 x h x w
    d2 = np.concatenate([d, d[:8]])                                                          # wrap around
    bright = np.max([d2[s:s + 9].min(0) for s in range(16)], axis=0)
    dark = np.max([(-d2[s:s + 9]).min(0) for s in range(16)], axis=0)
    score = np.maximum(bright, dark) - 1
    out = np.zeros((H, W), dtype=np.int64)
    out[3:H - 3, 3:W - 3] = np.where(score >= threshold, score, 0)
    return out


def test_fast_score_matches_numpy(oracle):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (61, 83), dtype=np.uint8)
    for img in (g["img"], noise, np.ascontiguousarray(g["img"].T)):
        for t in (1, 28, 51):
            np.testing.assert_array_equal(oracle.fast_score(img, t), _fast_score_numpy(img, t))


def test_fast_finds_the_corners_of_a_square_and_not_its_edges(oracle):
    img = np.full((64, 80), 50, np.uint8)
    img[20:41, 30:61] = 200                      # rows 20..40, columns 30..60
    S = oracle.fast_score(img, 28)
    # a convex corner pixel sees 11 of the 16 circle pixels outside the square: a corner with the full contrast
    for b, a in ((20, 30), (20, 60), (40, 30), (40, 60)):
        assert S[b, a] == 200 - 50 - 1
    # the middle of an edge sees 7 pixels outside, the interior none
    assert S[20, 45] == 0 and S[30, 30] == 0 and S[30, 45] == 0
    # on the ideal square the scores around a corner form a plateau and strict suppression keeps nothing;
    # with the corner pixels a little brighter each corner is one detection, exactly there
    ab, metric, total = oracle.detect_fast(img, min_contrast=0.11, min_quality=0.1, border=4)
    assert total == 0
    corners = ((20, 30), (20, 60), (40, 30), (40, 60))
    for b, a in corners:
        img[b, a] = 210
    ab, metric, total = oracle.detect_fast(img, min_contrast=0.11, min_quality=0.1, border=4)
    assert total == 4
    np.testing.assert_array_equal(ab, np.array([[a, b] for b, a in corners], np.float32))
    np.testing.assert_array_equal(metric, np.full(4, 210 - 50 - 1, np.float32))
    # a dark square on a bright ground gives the same corners (the test is symmetric in the polarity)
    ab2, metric2, _ = oracle.detect_fast(255 - img, min_contrast=0.11, min_quality=0.1, border=4)
    np.testing.assert_array_equal(ab, ab2)
    np.testing.assert_array_equal(metric, metric2)
    # and a contrast threshold above the square's contrast finds nothing
    assert oracle.detect_fast(img, min_contrast=0.7, min_quality=0.1, border=4)[2] == 0


def test_fast_detections_are_thresholded_strict_maxima(oracle):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    img = g["img"]
    t = int(np.floor(0.11 * 255 + 0.5))
    S = _fast_score_numpy(img, t)
    ab, metric, total = oracle.detect_fast(img, min_contrast=0.11, min_quality=0.1, border=16)
    assert total == len(ab) and total > 20
    ia = ab.astype(int)
    assert np.array_equal(ia, ab)                                  # integer positions
    thr = np.float32(0.1) * np.float32(S.max())
    keep = np.zeros_like(S, dtype=bool)
    H, W = S.shape
    for b in range(16, H - 16):
        for a in range(16, W - 16):
            r = S[b, a]
            nb = S[b - 1:b + 2, a - 1:a + 2].copy()
            nb[1, 1] = -1
            keep[b, a] = np.float32(r) > thr and r > nb.max()
    bs, as_ = np.nonzero(keep)
    np.testing.assert_array_equal(ia, np.stack([as_, bs], axis=1))   # ascending b * A + a
    np.testing.assert_array_equal(metric, S[bs, as_].astype(np.float32))
    # a higher contrast threshold keeps a subset with unchanged metrics
    ab_hi, metric_hi, _ = oracle.detect_fast(img, min_contrast=0.3, min_quality=0.1, border=16)
    assert 0 < len(ab_hi) < len(ab)
    assert metric_hi.min() >= np.floor(0.3 * 255 + 0.5)
