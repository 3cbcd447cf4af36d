"""Pins the bundle-adjustment part of the CPU oracle: residuals against a direct numpy
evaluation of baError.m, analytic Jacobians against finite differences (through the cost's
behaviour), convergence against scipy.optimize.least_squares, known-answer recovery, golden
vectors.  lsqnonlin is closed source: parity with MATLAB's iterates is unpinned."""
import os

import numpy as np
from scipy.linalg import expm
from scipy.optimize import least_squares

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ba_error_numpy(state, obs, K):
    F, m = int(obs[0]), int(obs[1])
    X = state[6 * F:].reshape(m, 3).T
    out = []
    pos = 2
    for f in range(F):
        tw = state[6 * f:6 * f + 6]
        se = np.zeros((4, 4))
        se[:3, :3] = np.array([[0, -tw[5], tw[4]], [tw[5], 0, -tw[3]], [-tw[4], tw[3], 0]])
        se[:3, 3] = tw[:3]
        T_C_W = np.linalg.inv(expm(se))
        n = int(obs[pos])
        kp = obs[pos + 1:pos + 1 + 2 * n].reshape(n, 2)[:, ::-1].T   # flipud: (row, col) -> (u, v)
        ids = obs[pos + 1 + 2 * n:pos + 1 + 3 * n].astype(int) - 1
        pc = T_C_W[:3, :3] @ X[:, ids] + T_C_W[:3, 3:4]
        h = K @ pc
        out.append(kp - h[:2] / h[2])
        pos += 1 + 3 * n
    return np.hstack(out)


def test_ba_error_matches_numpy_restatement(oracle, vo):
    s0, obs, st = vo.synth.ba_problem(F=4, m=120, seed=1)
    r = oracle.ba_error(s0, obs, vo.synth.K_KITTI)
    np.testing.assert_allclose(r, _ba_error_numpy(s0, obs, vo.synth.K_KITTI), rtol=0, atol=1e-9)
    # at the ground truth the residual is the injected pixel noise
    rt = oracle.ba_error(st, obs, vo.synth.K_KITTI)
    assert 0.3 < rt.std() < 0.7


def test_run_ba_converges_like_scipy(oracle, vo):
    K = vo.synth.K_KITTI
    s0, obs, st = vo.synth.ba_problem(F=5, m=150, seed=2)
    s1, info = oracle.run_ba(s0, obs, K, ftol=1e-10, xtol=1e-10)   # (to the minimum scipy finds, not to the default tolerances)
    assert info["cost"] < 0.05 * info["cost0"] and 2 <= info["iterations"] <= 21  # fixed landmarks keep their residuals
    F, m = 5, 150
    cnt = np.zeros(m, int)
    pos = 2
    for f in range(F):
        n = int(obs[pos])
        np.add.at(cnt, obs[pos + 1 + 2 * n:pos + 1 + 3 * n].astype(int) - 1, 1)
        pos += 1 + 3 * n
    free = np.concatenate([np.ones(6 * F, bool), np.repeat(cnt >= 2, 3)])
    # landmarks seen fewer than twice stay put (runBA.m:45-49)
    assert np.array_equal(s1[~free], s0[~free])

    def fun(xf):
        x = s0.copy()
        x[free] = xf
        return oracle.ba_error(x, obs, K).T.reshape(-1)
    sol = least_squares(fun, s0[free], method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=300)
    # same minimum value (the minimiser itself is only defined up to the 7-dof gauge)
    assert abs(2 * sol.cost - info["cost"]) < 1e-4 * info["cost"]
    # gradient at our solution vanishes: a tiny step in any direction does not lower the cost
    x1 = s1[free]
    f1 = np.sum(fun(x1) ** 2)
    rng = np.random.default_rng(0)
    for _ in range(20):
        d = rng.normal(size=x1.size)
        d *= 1e-6 / np.linalg.norm(d)
        assert np.sum(fun(x1 + d) ** 2) > f1 - 1e-7 * f1


def test_run_ba_known_answer_without_noise(oracle, vo):
    K = vo.synth.K_KITTI
    s0, obs, st = vo.synth.ba_problem(F=5, m=200, visible=1.0, noise_px=0.0, seed=3)
    s1, info = oracle.run_ba(s0, obs, K, max_iter=50, ftol=1e-10, xtol=1e-10)
    assert info["cost"] < 1e-12 * info["cost0"] + 1e-14


def test_pose_refine_recovers_pose(oracle, vo):
    K = vo.synth.K_KITTI
    kp_rc, X, R, t, _ = vo.synth.pnp_problem(300, inlier_ratio=1.0, noise_px=0.0, seed=5)
    T_C_W = np.eye(4)
    T_C_W[:3, :3] = R
    T_C_W[:3, 3] = t
    T_true = np.linalg.inv(T_C_W)
    T0 = T_true @ vo.synth.pose_matrix([0.02, -0.01, 0.03], [0.2, 0.1, -0.3])
    T1, info = oracle.pose_refine(T0, kp_rc[::-1], X, K)
    np.testing.assert_allclose(T1, T_true, rtol=0, atol=1e-8)
    assert info["cost"] < 1e-15 and info["iterations"] <= 100


def test_ba_oracle_matches_golden(oracle):
    g = np.load(os.path.join(GOLD, "ba_f5_m300.npz"))
    np.testing.assert_array_equal(oracle.ba_error(g["state0"], g["obs"], g["K"]), g["r0"])
    s1, info = oracle.run_ba(g["state0"], g["obs"], g["K"], ftol=1e-10, xtol=1e-10)   # the tolerances the fixture was made with
    np.testing.assert_array_equal(s1, g["state_opt"])
    assert info["iterations"] == int(g["iterations"])
    T1, pinfo = oracle.pose_refine(g["pr_T0"], g["pr_kp_uv"], g["pr_X"], g["K"])
    np.testing.assert_array_equal(T1, g["pr_T1"])
