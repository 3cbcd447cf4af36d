"""Pins the fundamental-matrix RANSAC of the CPU oracle: exact data gives an exact rank-2 F with
every pair an inlier; numpy's SVD gives the same 8-point solution; budget formula; golden vectors.
estimateFundamentalMatrix is closed source: parity with MATLAB's trials is unpinned."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_exact_data_all_inliers_and_rank2(oracle, vo):
    p1, p2, is_in = vo.synth.two_view_matches(400, outlier_ratio=0.0, noise_px=0.0, seed=1)
    samples = vo.draw_octets(np.random.default_rng(1), 400, 20)
    res = oracle.fmatrix_ransac(p1.T, p2.T, samples, 5e-2)
    assert res["best_count"] == 400 and res["mask"].all() and res["trials"] == 1
    F = res["F"]
    assert np.linalg.svd(F)[1][2] < 1e-12 * np.linalg.svd(F)[1][0]
    x1 = np.hstack([p1, np.ones((400, 1))])
    x2 = np.hstack([p2, np.ones((400, 1))])
    e = np.einsum("ni,ij,nj->n", x2, F, x1)
    assert np.abs(e).max() < 1e-8 * np.abs(F).max() * 1e6


def test_eight_point_matches_numpy_svd(oracle, vo):
    p1, p2, _ = vo.synth.two_view_matches(50, outlier_ratio=0.0, noise_px=0.3, seed=2)
    idx = np.arange(8)
    samples = idx[None].astype(np.int32)
    F = oracle.fmatrix_ransac(p1.T, p2.T, samples, 1e9)["F"]

    def hart(p):
        c = p.mean(0)
        s = np.sqrt(2) / np.linalg.norm(p - c, axis=1).mean()
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
    T1, T2 = hart(p1), hart(p2)
    a = (T1 @ np.hstack([p1[idx], np.ones((8, 1))]).T).T
    b = (T2 @ np.hstack([p2[idx], np.ones((8, 1))]).T).T
    A = np.stack([np.kron(b[i], a[i]) for i in range(8)])
    f = np.linalg.svd(A)[2][8].reshape(3, 3)
    U, S, Vt = np.linalg.svd(f)
    Fn = T2.T @ (U @ np.diag([S[0], S[1], 0]) @ Vt) @ T1
    Fn /= np.linalg.norm(Fn)
    Fo = F / np.linalg.norm(F)
    assert min(np.abs(Fn - Fo).max(), np.abs(Fn + Fo).max()) < 1e-9


def test_budget_and_outlier_rejection(oracle, vo):
    assert oracle.lib().vo_fmatrix_trial_budget is not None
    p1, p2, is_in = vo.synth.two_view_matches(1000, outlier_ratio=0.3, noise_px=0.05, seed=3)
    samples = vo.draw_octets(np.random.default_rng(3), 1000, 500)
    res = oracle.fmatrix_ransac(p1.T, p2.T, samples, 5e-2)
    assert res["trials"] <= 500 and res["best_count"] == res["mask"].sum()
    # the gross outliers (tens of pixels) are rejected, most true inliers kept
    assert res["mask"][~is_in].mean() < 0.02 and res["mask"][is_in].mean() > 0.8
    # the winner is the first trial with the maximum count inside the executed trials
    assert res["counts"].max() == res["best_count"]


def test_fmatrix_oracle_matches_golden(oracle):
    g = np.load(os.path.join(GOLD, "fmatrix_n1500.npz"))
    res = oracle.fmatrix_ransac(g["p1"].T, g["p2"].T, g["samples"], 5e-2)
    np.testing.assert_array_equal(res["mask"], g["mask"])
    assert res["trials"] == int(g["trials"]) and res["best_count"] == int(g["best_count"])


def test_draw_octets_distinct(vo):
    s = vo.draw_octets(np.random.default_rng(0), 9, 300)
    assert s.shape == (300, 8) and s.min() >= 0 and s.max() <= 8
    assert all(len(set(r)) == 8 for r in s)
