"""The scale-space detectors (SURF, BRISK) on images of awkward shapes -- the smallest the library accepts, sizes just
below and above the filters' and layers' reach, long thin strips, both storage orders -- against the CPU oracle.
Bar: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIXED = [(8, 8), (8, 300), (300, 8), (9, 27), (26, 26), (27, 27), (28, 29), (33, 64), (64, 65), (65, 129), (50, 51), (99, 99),
         (100, 101), (13, 400), (127, 193)]


def _shapes():
    rng = np.random.default_rng(2024)
    return FIXED + [(int(r), int(c)) for r, c in zip(rng.integers(8, 220, 20), rng.integers(8, 260, 20))]


def test_surf_and_brisk_on_awkward_shapes(vo, ctx, oracle):
    rng = np.random.default_rng(77)
    n_surf = n_brisk = 0
    for rows, cols in _shapes():
        # noise smoothed a little: blobs and corners at several scales even in small images
        img = rng.integers(0, 256, (rows, cols)).astype(np.float64)
        img = (img + np.roll(img, 1, 0) + np.roll(img, 1, 1) + np.roll(img, (1, 1), (0, 1))) / 4
        img = img.astype(np.uint8)
        for order in ("C", "F"):
            im = np.ascontiguousarray(img) if order == "C" else np.asfortranarray(img)
            store = img if order == "C" else np.ascontiguousarray(img.T)
            for border in (1, 5):
                ab, metric, scale, total = oracle.detect_surf(store, 20.0, border)
                _, kp = vo.genKeypoints(im, "SURF", ["MetricThreshold", 20.0, "Border", border], ctx=ctx)
                assert kp.Count == total, (rows, cols, order, border)
                want = ab if order == "C" else ab[:, ::-1]
                np.testing.assert_array_equal(kp.Location, want + 1.0, err_msg=str((rows, cols, order, border)))
                np.testing.assert_array_equal(kp.Metric, metric)
                np.testing.assert_array_equal(kp.Scale, scale)
                n_surf += total
                ab, metric, scale, total = oracle.detect_brisk(store, 0.04, 0.02, 4, border)
                _, kp = vo.genKeypoints(im, "BRISK", ["MinContrast", 0.04, "MinQuality", 0.02, "Border", border], ctx=ctx)
                assert kp.Count == total, (rows, cols, order, border)
                want = ab if order == "C" else ab[:, ::-1]
                np.testing.assert_array_equal(kp.Location, want + 1.0, err_msg=str((rows, cols, order, border)))
                np.testing.assert_array_equal(kp.Metric, metric)
                np.testing.assert_array_equal(kp.Scale, scale)
                n_brisk += total
            if order == "C" and rows >= 40 and cols >= 40:
                _, kp = vo.genKeypoints(im, "SURF", ["MetricThreshold", 20.0, "Border", 1], ctx=ctx)
                if kp.Count:
                    got, _ = vo.extractFeatures(im, kp, ctx=ctx)
                    np.testing.assert_array_equal(got, oracle.surf_descriptors(im, kp.Location, kp.Scale)[0])
    assert n_surf > 500 and n_brisk > 500
    with pytest.raises(vo.VohipError):
        vo.genKeypoints(np.zeros((7, 40), np.uint8), "SURF", None, ctx=ctx)
    with pytest.raises(vo.VohipError):
        vo.genKeypoints(np.zeros((40, 7), np.uint8), "BRISK", None, ctx=ctx)
