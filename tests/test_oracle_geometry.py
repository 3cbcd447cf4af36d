"""Pins the geometry part of the CPU oracle against independent implementations
(scipy.linalg.expm/logm, numpy.linalg.svd, scipy.optimize.least_squares, brute-force numpy)
and the committed golden vectors.  Parity against MATLAB's own expm/logm/svd/lsqnonlin is
unpinned (closed source, not runnable here)."""
import os

import numpy as np
import pytest
from scipy.linalg import expm, logm
from scipy.optimize import least_squares

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _se(tw):
    W = np.array([[0, -tw[5], tw[4]], [tw[5], 0, -tw[3]], [-tw[4], tw[3], 0]])
    se = np.zeros((4, 4))
    se[:3, :3] = W
    se[:3, 3] = tw[:3]
    return se


def test_twist_exp_matches_scipy_expm(oracle):
    rng = np.random.default_rng(0)
    for i in range(240):
        sc = [0.0, 1e-9, 1e-5, 1e-3, 0.05, 0.1, 1.0, 3.0][i % 8]
        tw = np.concatenate([rng.normal(0, 2, 3), rng.normal(0, 1, 3) * sc])
        nw = np.linalg.norm(tw[3:])
        if nw > 3.1:
            tw[3:] *= 3.1 / nw
        H = oracle.twist2homog(tw)
        np.testing.assert_allclose(H, expm(_se(tw)), rtol=0, atol=2e-14)
        back = oracle.homog2twist(H)
        np.testing.assert_allclose(back, tw, rtol=0, atol=1e-12)
        if nw > 1e-3:
            lg = logm(H).real
            np.testing.assert_allclose(back[:3], lg[:3, 3], atol=1e-10)
            np.testing.assert_allclose(back[3:], [-lg[1, 2], lg[0, 2], -lg[0, 1]], atol=1e-10)


def test_twist_log_near_pi(oracle):
    ax = np.array([0.3, -0.5, 0.8])
    ax /= np.linalg.norm(ax)
    for th in (np.pi - 1e-9, np.pi - 1e-4, np.pi):
        tw = np.concatenate([[1.0, 2.0, 3.0], ax * th])
        H = oracle.twist2homog(tw)
        back = oracle.homog2twist(H)
        assert abs(np.linalg.norm(back[3:]) - th) < 1e-7
        np.testing.assert_allclose(oracle.twist2homog(back), H, atol=1e-9)


def test_linear_triangulation_exact_and_vs_numpy_svd(oracle, vo):
    p1, p2, M1, M2, X = vo.synth.two_view_problem(500, noise_px=0.0, seed=1)
    P = oracle.linear_triangulation(p1, p2, M1, M2)
    assert np.max(np.abs(P[:3] - X) / np.abs(X).max()) < 1e-10 and np.all(P[3] == 1.0)
    p1, p2, M1, M2, X = vo.synth.two_view_problem(500, noise_px=0.5, seed=2)
    P = oracle.linear_triangulation(p1, p2, M1, M2)

    def cx(p):
        return np.array([[0, -p[2], p[1]], [p[2], 0, -p[0]], [-p[1], p[0], 0]])
    for j in range(0, 500, 7):
        A = np.vstack([cx(p1[:, j]) @ M1, cx(p2[:, j]) @ M2])
        v = np.linalg.svd(A)[2][3]
        v = v / v[3]
        assert np.max(np.abs(P[:, j] - v) / np.abs(v).max()) < 1e-9
    # per-point projection matrices (the way update_landmarks calls it)
    M2s = np.stack([M2 * (1 + 0.0 * j) for j in range(500)])
    np.testing.assert_array_equal(oracle.linear_triangulation(p1, p2, M1, M2s), P)


def test_triangulation_refinement_reaches_least_squares_minimum(oracle, vo):
    p1, p2, M1, M2, X = vo.synth.two_view_problem(60, noise_px=0.5, seed=3)
    P0 = oracle.linear_triangulation(p1, p2, M1, M2)
    P = oracle.triangulation_refinement(P0, p1, p2, M1, M2)

    def res(x, j):
        out = []
        for M, p in ((M1, p1), (M2, p2)):
            h = M @ np.append(x, 1.0)
            out += list(h[:2] / h[2] - p[:2, j])
        return np.array(out)
    for j in range(60):
        s = least_squares(res, P0[:3, j], args=(j,), xtol=1e-15, ftol=1e-15, gtol=1e-15)
        assert np.max(np.abs(s.x - P[:3, j])) / np.abs(s.x).max() < 1e-6
        assert np.sum(res(P[:3, j], j) ** 2) <= np.sum(res(P0[:3, j], j) ** 2) + 1e-15
    assert np.all(P[3] == P0[3])


def test_candidate_filter_matches_bruteforce(oracle):
    rng = np.random.default_rng(4)
    Cp = np.stack([rng.uniform(0, 376, 700), rng.uniform(0, 1241, 700)])
    Pp = np.stack([rng.uniform(0, 376, 900), rng.uniform(0, 1241, 900)])
    new = np.stack([rng.uniform(0, 1241, 1300), rng.uniform(0, 376, 1300)], axis=1).astype(np.float32)
    # plant exact boundary cases: distance exactly r in x, inside in y -> rejected (closed box)
    new[0] = (np.float32(Cp[1, 0]) + np.float32(10.0), np.float32(Cp[0, 0]))
    keep = oracle.candidate_filter(new, Cp, Pp, 10.0)
    allp = np.concatenate([Cp, Pp], axis=1).astype(np.float32)
    r = np.float32(10.0)
    want = np.array([not np.any((allp[1] >= x - r) & (allp[1] <= x + r) & (allp[0] >= y - r) & (allp[0] <= y + r))
                     for x, y in new])
    np.testing.assert_array_equal(keep, want)
    assert not keep[0] and 0.1 < keep.mean() < 0.9
    # S.C empty -> keep everything, S.P is not even consulted (update_candidate.m:25-27)
    assert oracle.candidate_filter(new, np.zeros((2, 0)), Pp, 10.0).all()


def test_candidate_triangulate_statuses(oracle, vo):
    cur, first, T_first, T_W_C, X = vo.synth.candidate_problem(800, seed=5)
    K = vo.synth.K_KITTI
    st, Xo, ang = oracle.candidate_triangulate(cur, first, T_first, T_W_C, K, 1.0, 1.0, 80.0, False)
    assert set(np.unique(st)) == {0, 1, 2}
    assert np.all(ang[st == 0] <= 1.0) and np.all(ang[st > 0] > 1.0)
    new = st == 1
    # accepted landmarks are close to the truth and inside the depth range in the current camera
    depth = (np.linalg.inv(T_W_C) @ np.vstack([Xo[:, new], np.ones(new.sum())]))[2]
    assert np.all((depth > 1.0) & (depth < 80.0))
    rel = np.linalg.norm(Xo[:, new] - X[:, new], axis=0) / np.linalg.norm(X[:, new], axis=0)
    assert np.median(rel) < 0.05
    # the BA variant only ever removes landmarks (extra reprojection filter)
    st2, _, _ = oracle.candidate_triangulate(cur, first, T_first, T_W_C, K, 1.0, 1.0, 80.0, True)
    assert np.all((st2 == 1) <= (st == 1)) and np.array_equal(st2 == 0, st == 0)


def test_geometry_oracle_matches_golden(oracle):
    g = np.load(os.path.join(GOLD, "geometry.npz"))
    for i in range(len(g["twists"])):
        np.testing.assert_array_equal(oracle.twist2homog(g["twists"][i]), g["H"][i])
        np.testing.assert_array_equal(oracle.homog2twist(g["H"][i]), g["twists_back"][i])
    np.testing.assert_array_equal(oracle.linear_triangulation(g["p1"], g["p2"], g["M1"], g["M2"]), g["P_lin"])
    np.testing.assert_array_equal(oracle.triangulation_refinement(g["P_lin"], g["p1"], g["p2"], g["M1"], g["M2"]),
                                  g["P_ref"])
    np.testing.assert_array_equal(oracle.candidate_filter(g["new_xy"], g["C_rc"], g["P_rc"], 10.0), g["keep"])
    st, Xo, ang = oracle.candidate_triangulate(g["cur_rc"], g["first_rc"], g["T_first"], g["T_W_C"], g["K"],
                                               1.0, 1.0, 80.0, False)
    np.testing.assert_array_equal(st, g["status"])
    np.testing.assert_array_equal(Xo, g["X_new"])
