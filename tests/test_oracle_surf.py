"""Pins the SURF part of the CPU oracle against a direct numpy evaluation of its definition (box filters
as explicit kernels correlated with the image, no integral image) and against what the detector must do on
images with known blobs.  detectSURFFeatures itself is closed source: parity with MATLAB's detector is
unpinned."""
import numpy as np
from scipy.ndimage import correlate, maximum_filter

SIZES = ((9, 15, 21, 27), (15, 27, 39, 51), (27, 51, 75, 99))


def _kernels(L):
    """The three box filters of side L as dense (L, L) arrays indexed [b, a], normalised per box area."""
    l, c = L // 3, (L - 1) // 2
    dxx = np.zeros((L, L))
    dxx[c - l + 1:c + l, :] = 1.0
    dxx[c - l + 1:c + l, l:2 * l] = -2.0
    dxx /= l * (2 * l - 1)
    dxy = np.zeros((L, L))
    dxy[c - l:c, c - l:c] = 1.0
    dxy[c + 1:c + 1 + l, c + 1:c + 1 + l] = 1.0
    dxy[c - l:c, c + 1:c + 1 + l] = -1.0
    dxy[c + 1:c + 1 + l, c - l:c] = -1.0
    dxy /= l * l
    return dxx, dxx.T.copy(), dxy


def _response_numpy(img, octave, level):
    L, s = SIZES[octave][level], 1 << octave
    h = (L - 1) // 2
    kxx, kyy, kxy = _kernels(L)
    I = img.astype(np.float64)
    dxx, dyy, dxy = (correlate(I, k, mode="constant") for k in (kxx, kyy, kxy))
    det = dxx * dyy - 0.81 * dxy * dxy
    inside = np.zeros_like(det, dtype=bool)
    inside[h:img.shape[0] - h, h:img.shape[1] - h] = True
    det[~inside] = 0.0
    return det[::s, ::s]


def _blobs(shape=(200, 300), seed=0):
    rng = np.random.default_rng(seed)
    B, A = shape
    yy, xx = np.mgrid[0:B, 0:A]
    img = np.full(shape, 100.0)
    blobs = [(80, 60, 3.0, 120), (200, 120, 5.0, -80), (150, 50, 8.0, 100), (250, 160, 4.0, 90)]
    for cx, cy, sg, amp in blobs:
        img += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
    return np.clip(img + rng.normal(0, 1, shape), 0, 255).astype(np.uint8), blobs


def test_response_matches_numpy(oracle, vo):
    img = vo.synth.to_uint8(vo.synth.textured_image(120, 150, seed=5))
    for octave in range(3):
        for level in range(4):
            R = oracle.surf_response(img, octave, level)
            ref = _response_numpy(img, octave, level)
            assert R.shape == ref.shape
            assert np.abs(R - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1.0)
    assert not oracle.surf_response(img[:90], 2, 3).any()    # (a 99 x 99 filter does not fit 90 rows)
    # box filters have no response to a constant or to a linear ramp along either axis (zero mean, even)
    ramp = np.tile(np.arange(150, dtype=np.uint8), (120, 1))
    for im in (np.full((120, 150), 93, np.uint8), ramp, np.ascontiguousarray(ramp.T)):
        for octave in range(3):
            assert not oracle.surf_response(im, octave, 1).any()


def test_blobs_are_found_where_they_are(oracle):
    img, blobs = _blobs()
    ab, metric, scale, total = oracle.detect_surf(img, 1000.0, 16)
    assert total == len(ab) and 4 <= total <= 8
    for cx, cy, sg, _ in blobs:
        d = np.hypot(ab[:, 0] - cx, ab[:, 1] - cy)
        k = int(np.argmin(d))
        assert d[k] < 0.5
        # the scale of the strongest response grows with the blob (filter side ~ 3.4 sigma .. 6 sigma)
        assert 0.5 * sg < scale[k] < 1.2 * sg
    assert np.all(metric > 1000.0)
    # the polarity of a blob does not matter (the determinant is even in the image), nor does a constant offset
    ab2, metric2, scale2, total2 = oracle.detect_surf(255 - img, 1000.0, 16)
    assert total2 == total
    np.testing.assert_array_equal(ab2, ab)
    np.testing.assert_array_equal(metric2, metric)
    # the threshold only removes points
    ab_hi, metric_hi, _, total_hi = oracle.detect_surf(img, 5000.0, 16)
    keep = metric > 5000.0
    assert total_hi == keep.sum() and 0 < total_hi < total
    np.testing.assert_array_equal(ab_hi, ab[keep])
    assert oracle.detect_surf(np.full((100, 100), 50, np.uint8), 0.0, 1)[3] == 0


def test_points_are_strict_maxima_of_their_block_and_transposition_exchanges_the_axes(oracle, vo):
    img = vo.synth.to_uint8(vo.synth.textured_image(180, 260, seed=9))
    ab, metric, scale, total = oracle.detect_surf(img, 200.0, 16)
    assert total == len(ab) and total > 30
    maps = {(o, i): oracle.surf_response(img, o, i) for o in range(3) for i in range(4)}
    found = 0
    for (a, b), m, sc in zip(ab, metric, scale):
        # which (octave, level, sample) made it: the one whose response equals the metric near the position
        hits = []
        for o in range(3):
            s = 1 << o
            for lev in (1, 2):
                R = maps[(o, lev)]
                ga, gb = int(round(a / s)), int(round(b / s))
                for gb2 in range(max(gb - 1, 1), min(gb + 2, R.shape[0] - 1)):
                    for ga2 in range(max(ga - 1, 1), min(ga + 2, R.shape[1] - 1)):
                        if R[gb2, ga2] == m:
                            hits.append((o, lev, gb2, ga2))
        assert hits
        o, lev, gb, ga = hits[0]
        block = np.stack([maps[(o, lev + d)][gb - 1:gb + 2, ga - 1:ga + 2] for d in (-1, 0, 1)])
        assert block[1, 1, 1] == m and (block < m).sum() == 26 and m > np.float32(200.0)
        s = 1 << o
        assert abs(a - ga * s) <= s and abs(b - gb * s) <= s
        L, dL = SIZES[o][lev], SIZES[o][lev] - SIZES[o][lev - 1]
        assert (1.2 / 9) * (L - dL) - 1e-6 <= sc <= (1.2 / 9) * (L + dL) + 1e-6
        assert 16 <= ga * s < img.shape[1] - 16 and 16 <= gb * s < img.shape[0] - 16
        found += 1
    assert found == total
    # nothing missed at the coarsest level of detail the check can afford: count the strict block maxima
    # above the threshold of octave 0, level 1 and compare with the detections of that level (those the
    # refinement keeps are a subset)
    R0, R1, R2 = (maps[(0, i)] for i in (0, 1, 2))
    stack = np.stack([R0, R1, R2])
    mx = maximum_filter(stack, size=3, mode="constant")[1]
    cand = (R1 == mx) & (R1 > np.float32(200.0))
    htop = (SIZES[0][2] - 1) // 2
    lo = max(htop + 1, 16)
    inner = np.zeros_like(cand)
    inner[lo:img.shape[0] - lo, lo:img.shape[1] - lo] = True
    n_cand = int((cand & inner).sum())
    n_lvl = sum(1 for m in metric if np.any((R1 == m) & cand & inner))
    assert 0 < n_lvl <= n_cand
    # MATLAB's column-major storage is the transposed problem: positions exchange their axes, the order is
    # that of the transposed image
    abT, metricT, scaleT, totalT = oracle.detect_surf(np.ascontiguousarray(img.T), 200.0, 16)
    assert totalT == total
    key = lambda p, m: sorted(zip(np.round(p[:, 0], 3).tolist(), np.round(p[:, 1], 3).tolist(), m.tolist()))
    assert key(abT[:, ::-1], metricT) == key(ab, metric)
