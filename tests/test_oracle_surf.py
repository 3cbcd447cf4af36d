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


# ---- descriptors and matching --------------------------------------------------------------------------

def _descriptor_numpy(img, x1, y1, sc, upright):
    """The descriptor definition written with plain pixel sums (no integral image), one point."""
    I = img.astype(np.int64)
    H, W = I.shape

    def box(x0, y0, x1_, y1_):
        x0, y0, x1_, y1_ = max(x0, 0), max(y0, 0), min(x1_, W), min(y1_, H)
        return int(I[y0:y1_, x0:x1_].sum()) if x1_ > x0 and y1_ > y0 else 0

    def haar_x(x, y, z):
        return float(box(x, y - z // 2, x + z // 2, y - z // 2 + z) - box(x - z // 2, y - z // 2, x, y - z // 2 + z))

    def haar_y(x, y, z):
        return float(box(x - z // 2, y, x - z // 2 + z, y + z // 2) - box(x - z // 2, y - z // 2, x - z // 2 + z, y))

    x, y = float(x1) - 1.0, float(y1) - 1.0
    S, X, Y = int(np.floor(sc + 0.5)), int(np.floor(x + 0.5)), int(np.floor(y + 0.5))
    co, si = 1.0, 0.0
    if not upright:
        r = []
        for j in range(-6, 7):
            for i in range(-6, 7):
                if i * i + j * j < 36:
                    w = np.exp(-(i * i + j * j) / 8.0)
                    r.append((w * haar_x(X + i * S, Y + j * S, 4 * S), w * haar_y(X + i * S, Y + j * S, 4 * S)))
        r = np.array(r)
        best, bx, by = 0.0, 0.0, 0.0
        for k in range(42):
            u, v = (np.cos(0.15 * k), np.sin(0.15 * k)), (np.cos(0.15 * k + np.pi / 3), np.sin(0.15 * k + np.pi / 3))
            inside = (u[0] * r[:, 1] - u[1] * r[:, 0] >= 0) & (r[:, 0] * v[1] - r[:, 1] * v[0] >= 0)
            sx, sy = r[inside, 0].sum(), r[inside, 1].sum()
            if sx * sx + sy * sy > best * (1 + 1e-12):
                best, bx, by = sx * sx + sy * sy, sx, sy
        if best > 0:
            co, si = bx / np.sqrt(best), by / np.sqrt(best)
    d = np.zeros((4, 4, 4))
    for j in range(-10, 10):
        for i in range(-10, 10):
            u, v = (i + 0.5) * sc, (j + 0.5) * sc
            xs, ys = int(np.floor(x + co * u - si * v + 0.5)), int(np.floor(y + si * u + co * v + 0.5))
            w = np.exp(-((i + 0.5) ** 2 + (j + 0.5) ** 2) / (2 * 3.3 ** 2))
            hx, hy = w * haar_x(xs, ys, 2 * S), w * haar_y(xs, ys, 2 * S)
            dx, dy = co * hx + si * hy, co * hy - si * hx
            d[(j + 10) // 5, (i + 10) // 5] += (dx, dy, abs(dx), abs(dy))
    d = d.reshape(-1)
    n = np.linalg.norm(d)
    return (d / n if n > 0 else d), np.arctan2(si, co)


def test_descriptors_match_a_plain_pixel_sum_statement(oracle, vo):
    img = vo.synth.to_uint8(vo.synth.textured_image(150, 200, seed=4))
    ab, metric, scale, total = oracle.detect_surf(img, 300.0, 16)
    assert total > 20
    loc = ab + np.float32(1.0)
    pick = np.linspace(0, total - 1, 12).astype(int)
    # plus points whose sampling squares leave the image
    extra = np.array([[2.0, 2.0], [199.0, 149.0], [100.5, 3.25]], dtype=np.float32)
    pts = np.vstack([loc[pick], extra])
    scs = np.concatenate([scale[pick], np.array([2.0, 3.1, 6.4], dtype=np.float32)])
    for upright in (True, False):
        feats, ori = oracle.surf_descriptors(img, pts, scs, upright=upright)
        assert feats.shape == (len(pts), 64)
        for k in range(len(pts)):
            want, want_ori = _descriptor_numpy(img, pts[k, 0], pts[k, 1], float(scs[k]), upright)
            assert np.abs(feats[k] - want).max() < 2e-6, (k, upright)
            assert abs(ori[k] - want_ori) < 1e-5
    # the storage order of the image does not matter (MATLAB's arrays are column-major)
    fF, oF = oracle.surf_descriptors(np.asfortranarray(img), pts, scs)
    fC, oC = oracle.surf_descriptors(img, pts, scs)
    np.testing.assert_array_equal(fF, fC)
    np.testing.assert_array_equal(oF, oC)


def test_descriptors_follow_a_quarter_turn_and_matching_finds_it(oracle, vo):
    """A quarter turn of the image maps the sampling pattern onto itself up to pixel rounding: the points of
    the turned image match the original's where they belong, and their directions turn with them."""
    img = vo.synth.to_uint8(vo.synth.textured_image(300, 400, seed=9))
    ab, _, scale, total = oracle.detect_surf(img, 500.0, 30)
    f0, o0 = oracle.surf_descriptors(img, ab + np.float32(1.0), scale)
    turned = np.ascontiguousarray(np.rot90(img))             # (x, y) -> (y, W - 1 - x)
    ab3, _, scale3, total3 = oracle.detect_surf(turned, 500.0, 30)
    f3, o3 = oracle.surf_descriptors(turned, ab3 + np.float32(1.0), scale3)
    pairs, dist = oracle.match_features(f0, f3)
    assert len(pairs) > 0.4 * total
    W = img.shape[1]
    want = np.stack([ab[pairs[:, 0], 1], W - 1 - ab[pairs[:, 0], 0]], axis=1)
    ok = np.abs(ab3[pairs[:, 1]] - want).max(axis=1) < 0.6
    assert ok.mean() > 0.9
    turn = np.angle(np.exp(1j * (o3[pairs[ok, 1]] - o0[pairs[ok, 0]])))
    assert np.median(np.abs(turn + np.pi / 2)) < 0.05       # a quarter turn, counter-clockwise on the screen = -pi/2 with y down


def test_matching_is_the_stated_rule(oracle):
    rng = np.random.default_rng(8)
    a = rng.normal(size=(300, 64))
    b = np.vstack([a[:200] + 0.03 * rng.normal(size=(200, 64)), rng.normal(size=(150, 64))])[rng.permutation(350)]
    a = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    b = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    d = ((a[:, None, :].astype(np.float64) - b[None].astype(np.float64)) ** 2).sum(axis=2)
    order = np.argsort(d, axis=1, kind="stable")
    for mt, mr in ((1.0, 0.6), (5.0, 0.8), (100.0, 1.0)):
        pairs, dist = oracle.match_features(a, b, mt, mr)
        d1, d2 = d[np.arange(300), order[:, 0]], d[np.arange(300), order[:, 1]]
        keep = (d1 <= 4 * mt / 100) & (d1 <= mr * mr * d2)
        # (float32 sums against float64 ones: leave out the pairs within rounding of a bound)
        sure = (np.abs(d1 - 4 * mt / 100) > 1e-5) & (np.abs(d1 - mr * mr * d2) > 1e-5)
        got = np.zeros(300, dtype=bool)
        got[pairs[:, 0]] = True
        np.testing.assert_array_equal(got[sure], keep[sure])
        np.testing.assert_array_equal(pairs[:, 1], order[pairs[:, 0], 0])
        np.testing.assert_allclose(dist, d1[pairs[:, 0]], atol=1e-5)
        assert np.all(np.diff(pairs[:, 0]) > 0)
    pairs, _ = oracle.match_features(a, b)
    assert 150 < len(pairs) <= 200          # the 200 noisy copies, less those another random vector comes close to
    # a single candidate has no second neighbour: the ratio test passes
    assert len(oracle.match_features(a[:5], a[:1], 100.0, 0.1)[0]) == 5
