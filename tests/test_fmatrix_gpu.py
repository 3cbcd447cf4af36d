"""Parity of the HIP fundamental-matrix RANSAC with the CPU oracle.  Bar: inlier masks bit-exact
except pairs whose Sampson distance is within 1e-9 (relative) of the threshold -- the Hartley
normalisation is a tree sum on the GPU, which moves F by a few ulp."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
THR = 5e-2


def _sampson(F, p1, p2):
    x1 = np.hstack([p1, np.ones((len(p1), 1))])
    x2 = np.hstack([p2, np.ones((len(p2), 1))])
    a = x1 @ F.T
    b = x2 @ F
    e = np.sum(x2 * a, axis=1)
    return e * e / (a[:, 0] ** 2 + a[:, 1] ** 2 + b[:, 0] ** 2 + b[:, 1] ** 2)


def _check(mask, want_mask, F_want, p1, p2):
    diff = mask != want_mask
    d = _sampson(F_want, p1, p2)
    assert np.all(np.abs(d[diff] - THR) < 1e-9 * THR + 1e-12), np.flatnonzero(diff)
    return diff.sum()


def test_fmatrix_matches_golden(vo, ctx):
    g = np.load(os.path.join(GOLD, "fmatrix_n1500.npz"))
    F, inl = vo.estimateFundamentalMatrix(g["p1"], g["p2"], NumTrials=500, DistanceThreshold=THR,
                                          samples=g["samples"], ctx=ctx)
    assert _check(inl, g["mask"], g["F"], g["p1"], g["p2"]) == 0
    Fg = g["F"] / np.linalg.norm(g["F"])
    Fo = F / np.linalg.norm(F)
    assert min(np.abs(Fg - Fo).max(), np.abs(Fg + Fo).max()) < 1e-9


@pytest.mark.parametrize("n,ratio,seed", [(4000, 0.3, 1), (300, 0.5, 2), (8, 0.0, 3), (65, 0.1, 4)])
def test_fmatrix_matches_oracle(vo, ctx, oracle, n, ratio, seed):
    p1, p2, _ = vo.synth.two_view_matches(n, outlier_ratio=ratio, noise_px=0.1, seed=seed)
    samples = vo.draw_octets(np.random.default_rng(seed), n, 500)
    want = oracle.fmatrix_ransac(p1.T, p2.T, samples, THR)
    F, inl = vo.estimateFundamentalMatrix(p1, p2, NumTrials=500, DistanceThreshold=THR, samples=samples, ctx=ctx)
    _check(inl, want["mask"], want["F"], p1, p2)


def test_fmatrix_edge_cases(vo, ctx):
    p1, p2, _ = vo.synth.two_view_matches(20, outlier_ratio=0.0, noise_px=0.0, seed=5)
    with pytest.raises(vo.VohipError) as ei:
        vo.estimateFundamentalMatrix(p1[:7], p2[:7], samples=np.zeros((1, 8), np.int32), ctx=ctx)
    assert ei.value.status == vo._lib.VOHIP_TOO_FEW_POINTS
    # a degenerate sample (repeated point) scores zero and cannot win
    s = np.array([[0, 0, 0, 0, 0, 0, 0, 0], [0, 1, 2, 3, 4, 5, 6, 7]], np.int32)
    F, inl = vo.estimateFundamentalMatrix(p1, p2, DistanceThreshold=THR, samples=s, ctx=ctx)
    assert inl.sum() >= 8
    # rng-driven call works and is reproducible
    a = vo.estimateFundamentalMatrix(p1, p2, DistanceThreshold=THR, rng=np.random.default_rng(3), ctx=ctx)[1]
    b = vo.estimateFundamentalMatrix(p1, p2, DistanceThreshold=THR, rng=np.random.default_rng(3), ctx=ctx)[1]
    np.testing.assert_array_equal(a, b)
