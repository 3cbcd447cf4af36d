"""Parity of the HIP corner detector with the CPU oracle through genKeypoints and the C ABI.
Bar: bit-exact (same fp32 formula evaluated in the same order, index work)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_harris_matches_golden(vo, ctx):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    feats, kp = vo.genKeypoints(g["img"], "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert feats is None and kp.Location.dtype == np.float32
    assert kp.Count == len(g["xy"])
    np.testing.assert_array_equal(kp.Location, g["xy"])
    np.testing.assert_array_equal(kp.Metric, g["metric"])
    _, kp = vo.genKeypoints(g["img"], "MinEigen", {"MinQuality": 1e-2}, ctx=ctx)
    np.testing.assert_array_equal(kp.Location, g["xy_mineigen"])
    np.testing.assert_array_equal(kp.Metric, g["metric_mineigen"])


@pytest.mark.parametrize("shape,seed", [((376, 1241), 1), ((97, 131), 2), ((64, 64), 3), ((600, 800), 4)])
def test_harris_matches_oracle(vo, ctx, oracle, shape, seed):
    img = vo.synth.to_uint8(vo.synth.textured_image(shape[0], shape[1], seed=seed))
    ab, metric, total = oracle.detect_corners(img, min_quality=2e-4, border=16)
    _, kp = vo.genKeypoints(img, "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert kp.Count == total
    np.testing.assert_array_equal(kp.Location, ab + 1.0)
    np.testing.assert_array_equal(kp.Metric, metric)
    # MATLAB storage: the column-major image is the transposed problem -> (x, y) swapped roles
    abT, metricT, totalT = oracle.detect_corners(np.ascontiguousarray(img.T), min_quality=2e-4, border=16)
    _, kpF = vo.genKeypoints(np.asfortranarray(img), "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert kpF.Count == totalT
    np.testing.assert_array_equal(kpF.Location, abT[:, ::-1] + 1.0)
    # column-major order = MATLAB's find() order: x ascending, then y
    xi, yi = np.rint(kpF.Location[:, 0]), np.rint(kpF.Location[:, 1])
    assert np.all(np.diff(xi * 10000 + yi) > 0)


def test_capacity_flat_and_errors(vo, ctx):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    with pytest.raises(vo.VohipError) as ei:
        vo.genKeypoints(g["img"], "Harris", ["MinQuality", 2e-4, "Capacity", 10], ctx=ctx)
    assert ei.value.status == vo._lib.VOHIP_CAPACITY_EXCEEDED
    _, kp = vo.genKeypoints(np.full((64, 64), 7, np.uint8), "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert kp.Count == 0 and kp.Location.shape == (0, 2)
    with pytest.raises(NotImplementedError):
        vo.genKeypoints(g["img"], "SURF", ["MetricThreshold", 200], ctx=ctx)
    with pytest.raises(ValueError):
        vo.genKeypoints(g["img"], "Harris", ["Bogus", 1], ctx=ctx)
