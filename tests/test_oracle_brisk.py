"""Pins the BRISK part of the CPU oracle against direct numpy statements of its definition (resampling, FAST 5-8,
scale-space maxima written as loops over the pinned FAST score maps) and against what the detector must do on
corners of known place and size.  detectBRISKFeatures itself is closed source: parity with MATLAB's detector is
unpinned."""
import numpy as np

SIGMA = (1.0, 1.5, 2.0, 3.0, 4.0, 6.0, 8.0, 12.0)


def _half(I):
    I = I.astype(np.int64)
    H, W = I.shape[0] // 2, I.shape[1] // 2
    I = I[:2 * H, :2 * W]
    return ((I[0::2, 0::2] + I[0::2, 1::2] + I[1::2, 0::2] + I[1::2, 1::2] + 2) // 4).astype(np.uint8)


def _twothirds(I):
    """Area-weighted resampling by 2/3, written with an explicit 3x upsampling: every source pixel becomes 2 x 2
    sub-pixels, an output pixel is the mean of its 3 x 3 sub-pixels (= 1.5 x 1.5 source pixels)."""
    H, W = I.shape[0] // 3, I.shape[1] // 3
    up = np.repeat(np.repeat(I[:3 * H, :3 * W].astype(np.int64), 2, axis=0), 2, axis=1)
    s = up.reshape(2 * H, 3, 2 * W, 3).sum(axis=(1, 3))
    return ((s + 4) // 9).astype(np.uint8)


def _layer(img, k):
    cur = img
    if k & 1:
        cur = _twothirds(cur)
    for _ in range(k // 2):
        cur = _half(cur)
    return cur


def test_layers_match_numpy(oracle, vo):
    img = vo.synth.to_uint8(vo.synth.textured_image(131, 200, seed=2))
    for k in range(8):
        got, want = oracle.brisk_layer(img, k), _layer(img, k)
        assert got.shape == want.shape
        np.testing.assert_array_equal(got, want)
    # resampling a constant keeps it; the layer sizes are those of the definition
    flat = np.full((100, 160), 77, np.uint8)
    assert [oracle.brisk_layer(flat, k).shape for k in range(4)] == [(100, 160), (66, 106), (50, 80), (33, 53)]
    assert all((oracle.brisk_layer(flat, k) == 77).all() for k in range(8))


def test_fast58_matches_a_brute_force_statement(oracle):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    S = oracle.fast58_score(img)
    ring = [(0, -1), (1, -1), (1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1)]
    I = img.astype(int)
    for b in range(40):
        for a in range(50):
            if a in (0, 49) or b in (0, 39):
                assert S[b, a] == 0
                continue
            d = [I[b + db, a + da] - I[b, a] for da, db in ring]
            # the largest t for which 5 contiguous neighbours are all > p + t or all < p - t
            best = 0
            for t in range(1, 256):
                ok = any(all(d[(s + j) % 8] > t for j in range(5)) or all(d[(s + j) % 8] < -t for j in range(5))
                         for s in range(8))
                if ok:
                    best = t
            assert S[b, a] == best, (a, b)


def _detect_python(oracle, img, c, q, n_oct, border):
    """The scale-space test written as loops over the FAST score maps (oracle.fast_score is pinned on its own)."""
    t = int(np.floor(c * 255 + 0.5))
    nl = max(1, 2 * n_oct)
    L = [_layer(img, k) for k in range(nl)]
    S = [oracle.fast_score(l, t) if min(l.shape) >= 7 else np.zeros(l.shape, np.int32) for l in L]
    S58 = oracle.fast58_score(img)
    smax = max(int(s.max()) if s.size else 0 for s in S)
    out = []
    for k in range(nl):
        Bk, Ak = S[k].shape
        for b in range(1, Bk - 1):
            for a in range(1, Ak - 1):
                s = int(S[k][b, a])
                if s <= 0 or not np.float32(s) > np.float32(q) * np.float32(smax):
                    continue
                patch = S[k][b - 1:b + 2, a - 1:a + 2]
                if (patch >= s).sum() != 1:
                    continue
                Xa, Xb = int(np.floor((a + 0.5) * SIGMA[k])), int(np.floor((b + 0.5) * SIGMA[k]))
                if not (border <= Xa < img.shape[1] - border and border <= Xb < img.shape[0] - border):
                    continue
                ms = {}
                ok = True
                for k2 in (k - 1, k + 1):
                    if k2 < 0 or k2 >= nl:
                        continue
                    r = SIGMA[k] / SIGMA[k2]
                    # pixels of layer k2 whose centres lie within [a - 1, a + 2) x [b - 1, b + 2) of layer k, closed
                    B2, A2 = S[k2].shape
                    m = 0
                    for y in range(B2):
                        cy = (y + 0.5) / r - 0.5          # the centre of pixel y of layer k2 in layer k's coordinates
                        if not (b - 1 - 1e-9 <= cy + 0.5 - 0.5 <= b + 1 + 1e-9):
                            continue
                        for x in range(A2):
                            cx = (x + 0.5) / r - 0.5
                            if a - 1 - 1e-9 <= cx <= a + 1 + 1e-9:
                                m = max(m, int(S[k2][y, x]))
                    ms[k2] = m
                    ok = ok and s > m
                if not ok:
                    continue
                out.append((k, a, b, s))
    return out


def test_scale_space_maxima_match_a_loop_statement(oracle, vo):
    img = vo.synth.to_uint8(vo.synth.textured_image(90, 120, seed=7))
    for c, q, n_oct in ((0.1, 0.1, 4), (0.05, 0.5, 2), (0.1, 0.1, 0)):
        ab, metric, scale, total = oracle.detect_brisk(img, c, q, n_oct, 4)
        want = _detect_python(oracle, img, c, q, n_oct, 4)
        assert total == len(want) and (total > 10 or n_oct == 0)
        for (k, a, b, s), p, m, sc in zip(want, ab, metric, scale):
            assert m == s
            # within half a layer pixel of the pixel's centre in the image
            assert abs(p[0] - ((a + 0.5) * SIGMA[k] - 0.5)) <= 0.5 * SIGMA[k] + 1e-5
            assert abs(p[1] - ((b + 0.5) * SIGMA[k] - 0.5)) <= 0.5 * SIGMA[k] + 1e-5
            lo = SIGMA[k - 1] if k > 0 else 2.0 / 3.0
            hi = SIGMA[k + 1] if k + 1 < max(1, 2 * n_oct) else SIGMA[k]
            assert lo - 1e-6 <= sc <= hi + 1e-6


def test_a_corner_is_found_at_the_scale_it_was_drawn_at(oracle):
    """A bright square on a dark ground with blurred edges: the wider the blur, the coarser the layer whose FAST test
    its corners pass best, and they are found at the square's corners to within that layer's circle."""
    from scipy.ndimage import gaussian_filter
    found_scales = []
    for blur in (0.6, 1.5, 5.0):
        img = np.full((240, 320), 40.0)
        img[70:170, 100:220] = 210.0
        img = np.clip(gaussian_filter(img, blur), 0, 255).astype(np.uint8)
        ab, metric, scale, total = oracle.detect_brisk(img, 0.2, 0.1, 4, 16)
        assert total >= 4
        corners = np.array([[100, 70], [219, 70], [100, 169], [219, 169]], dtype=float)
        best = []
        for cpt in corners:
            d = np.hypot(ab[:, 0] - cpt[0], ab[:, 1] - cpt[1])
            i = int(np.argmin(d))
            assert d[i] < 2.0 + 2.5 * scale[i], (blur, cpt, d[i], scale[i])   # (the circle of the layer has radius 3 sigma)
            best.append(scale[i])
        found_scales.append(np.median(best))
    assert found_scales[0] < found_scales[1] < found_scales[2]
    # the storage order does not matter beyond exchanging the axes
    img = np.clip(gaussian_filter(np.pad(np.full((60, 90), 200.0), 40, constant_values=30.0), 1.0), 0, 255).astype(np.uint8)
    ab, metric, scale, total = oracle.detect_brisk(img, 0.2, 0.1, 4, 8)
    abT, metricT, scaleT, totalT = oracle.detect_brisk(np.ascontiguousarray(img.T), 0.2, 0.1, 4, 8)
    assert total == totalT > 0
    key = lambda p, m: sorted(zip(np.round(p[:, 0], 3).tolist(), np.round(p[:, 1], 3).tolist(), m.tolist()))
    assert key(abT[:, ::-1], metricT) == key(ab, metric)
