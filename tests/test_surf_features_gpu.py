"""Parity of the HIP SURF descriptors and feature matching with the CPU oracle through extractFeatures /
matchFeatures / the 'Des_match' bootstrap branch and the C ABI.
Bar: descriptors, index pairs and distances bit-exact (integer box sums, fp64 / fp32 sums in a fixed order, every
transcendental from host-made tables); the reported Orientation within 1e-6 rad (an atan2 on either side)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _points(vo, ctx, img, thr=300.0):
    _, kp = vo.genKeypoints(img, "SURF", ["MetricThreshold", thr], ctx=ctx)
    return kp


@pytest.mark.parametrize("shape,seed", [((376, 1241), 1), ((97, 131), 2), ((300, 400), 9)])
def test_descriptors_match_oracle(vo, ctx, oracle, shape, seed):
    img = vo.synth.to_uint8(vo.synth.textured_image(shape[0], shape[1], seed=seed))
    kp = _points(vo, ctx, img)
    assert kp.Count > 20
    for upright in (False, True):
        want, ori = oracle.surf_descriptors(img, kp.Location, kp.Scale, upright=upright)
        got, valid = vo.extractFeatures(img, kp, Upright=upright, ctx=ctx)
        assert got.shape == (kp.Count, 64) and got.dtype == np.float32 and valid.Count == kp.Count
        np.testing.assert_array_equal(got, want)
        np.testing.assert_allclose(valid.Orientation, ori, atol=1e-6)
        np.testing.assert_allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=1e-6)
        if upright:
            assert not valid.Orientation.any()
    # MATLAB's column-major storage of the same image: the same points, the same descriptors
    kpF = _points(vo, ctx, np.asfortranarray(img))
    gotF, _ = vo.extractFeatures(np.asfortranarray(img), kp, ctx=ctx)
    got, _ = vo.extractFeatures(img, kp, ctx=ctx)
    np.testing.assert_array_equal(gotF, got)
    assert kpF.Count == kp.Count
    # genKeypoints(..., with_features=True) is genKeypoints.m:31-32
    feats, valid = vo.genKeypoints(img, "SURF", ["MetricThreshold", 300.0], ctx=ctx, with_features=True)
    np.testing.assert_array_equal(feats, got)
    np.testing.assert_array_equal(valid.Location, kp.Location)


def test_points_at_the_image_edge_and_errors(vo, ctx, oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (120, 160), dtype=np.uint8)
    # hand-made points whose sampling squares leave the image on every side, and a large scale
    loc = np.array([[1.0, 1.0], [160.0, 120.0], [3.5, 60.25], [80.0, 2.0], [80.5, 60.5], [155.0, 5.0]], dtype=np.float32)
    scale = np.array([1.6, 2.0, 3.3, 5.0, 9.7, 2.4], dtype=np.float32)
    pts = vo.SURFPoints(loc, np.ones(len(loc), np.float32), scale)
    want, ori = oracle.surf_descriptors(img, loc, scale)
    got, valid = vo.extractFeatures(img, pts, ctx=ctx)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_allclose(valid.Orientation, ori, atol=1e-6)
    # a flat image has no gradient: zero descriptors and the direction (1, 0) where the sampling square lies inside
    # it (boxes clipped at an edge see the edge)
    flat = np.full((120, 160), 9, np.uint8)
    inner = vo.SURFPoints(loc[4:5], np.ones(1, np.float32), np.array([2.0], np.float32))
    got, valid = vo.extractFeatures(flat, inner, ctx=ctx)
    assert not got.any() and not valid.Orientation.any()
    got, valid = vo.extractFeatures(flat, pts, ctx=ctx)
    np.testing.assert_array_equal(got, oracle.surf_descriptors(flat, loc, scale)[0])
    empty = vo.SURFPoints(np.zeros((0, 2), np.float32), np.zeros(0, np.float32), np.zeros(0, np.float32))
    assert vo.extractFeatures(img, empty, ctx=ctx)[0].shape == (0, 64)
    with pytest.raises(NotImplementedError):
        vo.extractFeatures(img, vo.cornerPoints(loc, np.ones(len(loc), np.float32)), ctx=ctx)
    for bad_scale in (0.0, -1.0, np.nan, 1e9):
        with pytest.raises(vo.VohipError):
            vo.extractFeatures(img, vo.SURFPoints(loc[:1], np.ones(1, np.float32), np.array([bad_scale], np.float32)), ctx=ctx)
    with pytest.raises(vo.VohipError):
        vo.extractFeatures(img, vo.SURFPoints(np.array([[np.inf, 3.0]], np.float32), np.ones(1, np.float32),
                                              np.array([2.0], np.float32)), ctx=ctx)


def test_matching_matches_oracle_and_finds_a_known_shift(vo, ctx, oracle):
    img = vo.synth.to_uint8(vo.synth.textured_image(300, 400, seed=9))
    img2 = np.ascontiguousarray(np.roll(img, (3, 5), axis=(0, 1)))
    f0, k0 = vo.genKeypoints(img, "SURF", ["MetricThreshold", 500.0, "Border", 30], ctx=ctx, with_features=True)
    f1, k1 = vo.genKeypoints(img2, "SURF", ["MetricThreshold", 500.0, "Border", 30], ctx=ctx, with_features=True)
    pairs, dist = vo.matchFeatures(f0, f1, ctx=ctx)
    want_pairs, want_dist = oracle.match_features(f0, f1)
    np.testing.assert_array_equal(pairs, want_pairs)
    np.testing.assert_array_equal(dist, want_dist)
    assert len(pairs) > 0.6 * k0.Count and np.all(np.diff(pairs[:, 0]) > 0)
    disp = k1.Location[pairs[:, 1]] - k0.Location[pairs[:, 0]]
    np.testing.assert_array_equal(np.median(disp, axis=0), [5.0, 3.0])
    assert np.mean(np.abs(disp - [5.0, 3.0]).max(axis=1) < 0.01) > 0.7
    # other thresholds, a set against itself (every feature its own nearest neighbour at distance 0), one-element sets
    for mt, mr in ((10.0, 0.8), (0.2, 0.3), (100.0, 1.0)):
        p, d = vo.matchFeatures(f0, f1, MatchThreshold=mt, MaxRatio=mr, ctx=ctx)
        wp, wd = oracle.match_features(f0, f1, mt, mr)
        np.testing.assert_array_equal(p, wp)
        np.testing.assert_array_equal(d, wd)
    p, d = vo.matchFeatures(f0, f0, ctx=ctx)
    wp, wd = oracle.match_features(f0, f0)
    np.testing.assert_array_equal(p, wp)
    assert np.all(p[:, 0] == p[:, 1]) and not d.any()
    p, d = vo.matchFeatures(f0[:1], f1[:1], MatchThreshold=100.0, ctx=ctx)
    assert p.tolist() == [[0, 0]]
    rng = np.random.default_rng(5)   # random vectors of another length, 3 000 against 70
    a, b = rng.normal(size=(3000, 17)).astype(np.float32), rng.normal(size=(70, 17)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    p, d = vo.matchFeatures(a, b, MatchThreshold=30.0, MaxRatio=0.9, ctx=ctx)
    wp, wd = oracle.match_features(a, b, 30.0, 0.9)
    assert len(wp) > 20
    np.testing.assert_array_equal(p, wp)
    np.testing.assert_array_equal(d, wd)
    a, b = rng.normal(size=(130, 80)).astype(np.float32), rng.normal(size=(257, 80)).astype(np.float32)   # (longer than 64: the untiled kernel)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    p, d = vo.matchFeatures(a, b, MatchThreshold=60.0, MaxRatio=0.95, ctx=ctx)
    wp, wd = oracle.match_features(a, b, 60.0, 0.95)
    assert len(wp) > 5
    np.testing.assert_array_equal(p, wp)
    np.testing.assert_array_equal(d, wd)
    assert vo.matchFeatures(np.zeros((0, 64), np.float32), f1, ctx=ctx)[0].shape == (0, 2)
    with pytest.raises(vo.VohipError):
        vo.matchFeatures(f0, f1, MaxRatio=1.5, ctx=ctx)
