"""Pins the KLT part of the CPU oracle: pyramid against a direct numpy convolution, tracks
against known (Fourier-exact) image shifts, axis symmetry, golden vectors.  vision.PointTracker
itself is closed source and cannot be run, so parity with MATLAB's tracker is unpinned."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pyr_down_numpy(img):
    k = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16
    pad = np.pad(img.astype(np.float64), 2, mode="edge")
    h = sum(k[j] * pad[:, j:j + img.shape[1]] for j in range(5))
    v = sum(k[j] * h[j:j + img.shape[0], :] for j in range(5))
    return v[::2, ::2]


def test_pyramid_levels_match_numpy(oracle, vo):
    img = vo.synth.to_uint8(vo.synth.textured_image(75, 121, seed=2))  # odd sizes
    ref = img.astype(np.float64)
    for l in range(1, 5):
        ref = _pyr_down_numpy(ref)
        got = oracle.pyramid_level(img, l)
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-4)
    assert np.array_equal(oracle.pyramid_level(img, 0), img.astype(np.float32))


def test_tracks_known_subpixel_shift(oracle, vo):
    base = vo.synth.textured_image(240, 320, seed=1)
    I0 = vo.synth.to_uint8(base)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(40, 280, 150), rng.uniform(40, 200, 150)], axis=1)
    for dx, dy in [(0.0, 0.0), (3.3, -1.7), (-7.25, 4.5), (14.6, 7.2)]:
        I1 = vo.synth.to_uint8(vo.synth.shift_image(base, dx, dy))
        out, valid, fb = oracle.klt_track(I0, I1, pts)
        assert valid.mean() > 0.95
        d = out[valid] - pts[valid]
        # uint8 quantisation + the 0.01 px stopping rule limit the accuracy
        assert np.abs(d[:, 0] - dx).max() < 0.05 and np.abs(d[:, 1] - dy).max() < 0.05
        assert fb[valid].max() < 0.05


def test_tracker_is_symmetric_in_the_axes(oracle, vo):
    """A column-major image is the transposed row-major image: results must be the swapped ones."""
    base = vo.synth.textured_image(120, 160, seed=3)
    I0 = vo.synth.to_uint8(base)
    I1 = vo.synth.to_uint8(vo.synth.shift_image(base, 2.2, -3.1))
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(30, 130, 40), rng.uniform(30, 90, 40)], axis=1)
    out, valid, _ = oracle.klt_track(I0, I1, pts)
    outT, validT, _ = oracle.klt_track(np.ascontiguousarray(I0.T), np.ascontiguousarray(I1.T), pts[:, ::-1])
    # bilinear interpolation runs along the fast axis first, so only rounding differs
    np.testing.assert_allclose(outT[:, ::-1], out, rtol=0, atol=1e-4)
    np.testing.assert_array_equal(validT, valid)


def test_invalid_points(oracle, vo):
    g = np.load(os.path.join(GOLD, "klt_320x240.npz"))
    assert not g["valid"][0]          # flat patch
    assert g["valid"][3:].mean() > 0.9
    # large forward-backward error threshold 0 -> (almost) nothing is valid
    out, valid, fb = oracle.klt_track(g["I0"], g["I1"], g["pts"] - 1.0, max_bidir_error=0.0)
    assert valid.sum() <= 2


def test_klt_oracle_matches_golden(oracle):
    g = np.load(os.path.join(GOLD, "klt_320x240.npz"))
    out, valid, fb = oracle.klt_track(g["I0"], g["I1"], g["pts"] - 1.0)
    np.testing.assert_array_equal(out + 1.0, g["out"])
    np.testing.assert_array_equal(valid, g["valid"])
    np.testing.assert_array_equal(oracle.pyramid_level(g["I0"], 3), g["level3"])
