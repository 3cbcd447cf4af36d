"""Parity of the HIP pyramid + KLT kernels with the CPU oracle, through the C ABI and the
PointTracker mirror.  Bar: pyramid levels bit-exact (same fp32 formula); tracks within 0.05 px
(observed ~1e-9: fp64 sums differ only in summation order); validity identical except points
whose forward-backward error or eigenvalue sits on a threshold."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_PX = 0.05


def test_pyramid_levels_bit_exact(vo, ctx, oracle):
    g = np.load(os.path.join(GOLD, "klt_320x240.npz"))
    pyr = vo.Pyramid(g["I0"], levels=5, ctx=ctx)
    np.testing.assert_array_equal(pyr.level(0), g["I0"].astype(np.float32))
    np.testing.assert_array_equal(pyr.level(1), g["level1"])
    np.testing.assert_array_equal(pyr.level(3), g["level3"])
    np.testing.assert_array_equal(pyr.level(4), g["level4"])
    pyr.release()
    # odd sizes and a column-major (MATLAB) image
    img = vo.synth.to_uint8(vo.synth.textured_image(75, 121, seed=2))
    pyr = vo.Pyramid(img, levels=4, ctx=ctx)
    pyrF = vo.Pyramid(np.asfortranarray(img), levels=4, ctx=ctx)
    for l in range(4):
        want = oracle.pyramid_level(img, l)
        np.testing.assert_array_equal(pyr.level(l), want)
        np.testing.assert_array_equal(pyrF.level(l), oracle.pyramid_level(np.ascontiguousarray(img.T), l))
    pyr.release()
    pyrF.release()


def _check_tracks(out, valid, want_out, want_valid, fb_want, max_bidir=6.0):
    d = np.abs(out - want_out)
    both = valid & want_valid
    assert d[both].max() < TOL_PX, d[both].max()
    flips = valid != want_valid
    # a validity flip is only acceptable on a threshold
    assert np.all(np.abs(fb_want[flips] - max_bidir) < 1e-6), np.flatnonzero(flips)
    return d[both].max(), flips.sum()


def test_tracks_match_golden(vo, ctx):
    g = np.load(os.path.join(GOLD, "klt_320x240.npz"))
    trk = vo.PointTracker(BlockSize=[21, 21], NumPyramidLevels=5, MaxIterations=20, MaxBidirectionalError=6, ctx=ctx)
    trk.initialize(g["pts"], g["I0"])
    out, valid = trk(g["I1"])
    worst, nflip = _check_tracks(out, valid, g["out"], g["valid"], g["fb_err"])
    assert worst < 1e-6 and nflip == 0
    assert not valid[0]
    trk.release()


@pytest.mark.parametrize("shape,n,shift,seed", [((376, 1241), 2000, (6.4, -2.3), 1), ((240, 320), 500, (-11.7, 5.1), 2),
                                                ((131, 97), 64, (0.4, 0.6), 3)])
def test_tracks_match_oracle(vo, ctx, oracle, shape, n, shift, seed):
    h, w = shape
    base = vo.synth.textured_image(h, w, seed=seed)
    I0 = vo.synth.to_uint8(base)
    I1 = vo.synth.to_uint8(vo.synth.shift_image(base, *shift))
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(1, w, n), rng.uniform(1, h, n)], axis=1)
    want_out, want_valid, fb = oracle.klt_track(I0, I1, pts - 1.0)
    trk = vo.PointTracker(BlockSize=[21, 21], NumPyramidLevels=5, MaxIterations=20, MaxBidirectionalError=6, ctx=ctx)
    trk.initialize(pts, I0)
    out, valid = trk(I1)
    _check_tracks(out, valid, want_out + 1.0, want_valid, fb)
    # and against the truth, for the points well inside the image
    inner = valid & (pts[:, 0] > 40) & (pts[:, 0] < w - 40) & (pts[:, 1] > 40) & (pts[:, 1] < h - 40)
    if inner.sum() > 10:
        d = out[inner] - pts[inner]
        assert np.abs(d[:, 0] - shift[0]).max() < 0.06 and np.abs(d[:, 1] - shift[1]).max() < 0.06
    # MATLAB column-major images give the same tracks
    trk2 = vo.PointTracker(BlockSize=[21, 21], NumPyramidLevels=5, MaxIterations=20, MaxBidirectionalError=6, ctx=ctx)
    trk2.initialize(pts, np.asfortranarray(I0))
    out2, valid2 = trk2(np.asfortranarray(I1))
    # (bilinear interpolation runs along the fast axis first, so rounding differs slightly)
    both = valid & valid2
    np.testing.assert_allclose(out2[both], out[both], rtol=0, atol=1e-3)
    assert (valid2 != valid).mean() < 0.005
    trk.release()
    trk2.release()


def test_block_size_31_and_protocol(vo, ctx, oracle):
    base = vo.synth.textured_image(200, 260, seed=5)
    I0 = vo.synth.to_uint8(base)
    I1 = vo.synth.to_uint8(vo.synth.shift_image(base, 2.5, 1.5))
    I2 = vo.synth.to_uint8(vo.synth.shift_image(base, 5.0, 3.0))
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(40, 220, 80), rng.uniform(40, 160, 80)], axis=1)
    want, wv, fb = oracle.klt_track(I0, I1, pts - 1.0, block=31, levels=3, max_iter=30, max_bidir_error=3.0)
    trk = vo.PointTracker(NumPyramidLevels=3, MaxBidirectionalError=3, ctx=ctx)  # defaults: 31x31, 30 iterations
    with pytest.raises(RuntimeError):
        trk(I1)
    trk.initialize(pts, I0)
    out, valid = trk(I1)
    _check_tracks(out, valid, want + 1.0, wv, fb, 3.0)
    # stepping again continues from the tracked points on the new previous image
    out2, valid2 = trk(I2)
    d = out2[valid2] - pts[valid2]
    assert np.abs(d[:, 0] - 5.0).max() < 0.1 and np.abs(d[:, 1] - 3.0).max() < 0.1
    # a shared Pyramid object can seed several trackers
    pyr0, pyr1 = vo.Pyramid(I0, 3, ctx=ctx), vo.Pyramid(I1, 3, ctx=ctx)
    t2 = vo.PointTracker(NumPyramidLevels=3, MaxBidirectionalError=3, ctx=ctx)
    t2.initialize(pts, pyr0)
    o3, v3 = t2(pyr1)
    np.testing.assert_array_equal(o3, out)
    np.testing.assert_array_equal(v3, valid)
    for t in (trk, t2):
        t.release()
    pyr0.release()
    pyr1.release()
    with pytest.raises(ValueError):
        vo.PointTracker(BlockSize=[7, 7], ctx=ctx)
    # zero points
    t3 = vo.PointTracker(BlockSize=[21, 21], NumPyramidLevels=3, ctx=ctx)
    t3.initialize(np.zeros((0, 2)), I0)
    o, v = t3(I1)
    assert o.shape == (0, 2) and v.shape == (0,)
    t3.release()
