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
