"""The CPU restatement of estimatePoseDLT.m and of the DLT branch of ransacLocalization.m (oracle/dlt_oracle.c),
pinned from outside: exact correspondences of a known pose, and an independent numpy statement of
estimatePoseDLT.m:14-58 over numpy.linalg.svd (LAPACK, what MATLAB's svd calls).  The reference holds no
fixture for this (dead) branch: parity with MATLAB is unpinned."""
import numpy as np
import pytest

K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]])


def numpy_estimate_pose_dlt(p, P, K):
    """estimatePoseDLT.m:14-58, line by line, with numpy.linalg.svd."""
    pn = np.linalg.solve(K, np.hstack([p, np.ones((len(p), 1))]).T).T          # :14
    Q = np.zeros((2 * len(p), 12))                                            # :17-18
    for i in range(len(p)):                                                   # :20-32
        u, v = pn[i, 0], pn[i, 1]
        Q[2 * i, 0:3] = P[i]
        Q[2 * i, 3] = 1
        Q[2 * i, 8:12] = -u * np.append(P[i], 1)
        Q[2 * i + 1, 4:7] = P[i]
        Q[2 * i + 1, 7] = 1
        Q[2 * i + 1, 8:12] = -v * np.append(P[i], 1)
    M = np.linalg.svd(Q)[2][-1].reshape(3, 4)                                 # :34-37
    if np.linalg.det(M[:, :3]) < 0:                                           # :41-43
        M = -M
    R = M[:, :3]
    U, _, Vt = np.linalg.svd(R)                                               # :50-51
    Rt = U @ Vt
    alpha = np.linalg.norm(Rt) / np.linalg.norm(R)                            # :55
    return np.hstack([Rt, alpha * M[:, 3:4]])                                 # :58


def rotation(w):
    th = np.linalg.norm(w)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def scene(n, seed, noise=0.0, outliers=0.0):
    rng = np.random.default_rng(seed)
    R, t = rotation(rng.normal(size=3) * 0.2), rng.normal(size=3) * 0.5
    P = rng.uniform(-8, 8, (n, 3)) + np.array([0, 0, 25.0])
    pc = P @ R.T + t
    uv = pc[:, :2] / pc[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + K[:2, 2]
    uv = uv + rng.normal(0, noise, uv.shape)
    bad = rng.random(n) < outliers
    uv[bad] = rng.uniform([0, 0], [1241, 376], (int(bad.sum()), 2))
    return uv, P, R, t, bad


@pytest.mark.parametrize("n", [6, 7, 12, 100, 2000])
def test_exact_correspondences_give_the_pose_back(oracle, n):
    uv, P, R, t, _ = scene(n, n)
    M = oracle.estimate_pose_dlt(uv, P, K)
    np.testing.assert_allclose(M[:, :3], R, atol=1e-9)
    np.testing.assert_allclose(M[:, 3], t, atol=1e-8)
    assert abs(np.linalg.det(M[:, :3]) - 1) < 1e-12


@pytest.mark.parametrize("n,seed", [(6, 0), (6, 1), (9, 2), (40, 3), (700, 4)])
def test_matches_numpy_svd_statement(oracle, n, seed):
    uv, P, _, _, _ = scene(n, seed, noise=0.7)
    np.testing.assert_allclose(oracle.estimate_pose_dlt(uv, P, K), numpy_estimate_pose_dlt(uv, P, K), rtol=0, atol=1e-9)


def test_too_few_correspondences(oracle):
    uv, P, _, _, _ = scene(5, 0)
    with pytest.raises(ValueError):
        oracle.estimate_pose_dlt(uv, P, K)


def sextets(n, count, seed):
    rng = np.random.default_rng(seed)
    return np.array([rng.permutation(n)[:6] for _ in range(count)], dtype=np.int32)


def test_dlt_branch_of_ransac_localization(oracle):
    """Every step of the branch against straight numpy: hypothesis poses, counts, the running maximum and the
    adaptive bound with exponent 6 (:118-129), the stop index, and the refit over the best hypothesis' inliers."""
    n = 400
    uv, P, R, t, bad = scene(n, 11, noise=0.4, outliers=0.25)
    kp_rc = uv[:, ::-1].copy()
    S = sextets(n, 3000, 5)
    out = oracle.ransac_localization_dlt(kp_rc, P, K, S)
    assert out["found"] and 99 <= out["n_iter"] < 3000
    mx, best, num_it = 0, None, np.inf
    i = 1
    while num_it > i:
        M = numpy_estimate_pose_dlt(uv[S[i - 1]], P[S[i - 1]], K)
        np.testing.assert_allclose(out["M_hyp"][i - 1], M, atol=1e-7)
        pc = P @ M[:, :3].T + M[:, 3]
        proj = pc[:, :2] / pc[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + K[:2, 2]
        inl = ((uv - proj) ** 2).sum(1) < 100.0
        assert inl.sum() == out["per_hyp_counts"][i - 1], i
        if inl.sum() > mx and inl.sum() >= 30:
            mx, best = int(inl.sum()), inl.copy()
        o = min(0.95, 1 - mx / n)
        num_it = min(15000, max(100, np.log(1 - 0.9999) / np.log(1 - (1 - o) ** 6)))
        assert out["max_num_inliers_history"][i - 1] == mx and out["num_iteration_history"][i - 1] == pytest.approx(num_it)
        i += 1
    assert out["n_iter"] == i - 1
    np.testing.assert_array_equal(out["mask"].astype(bool), best)
    M = numpy_estimate_pose_dlt(uv[best], P[best], K)
    np.testing.assert_allclose(out["R_C_W"], M[:, :3], atol=1e-9)
    np.testing.assert_allclose(out["t_C_W"], M[:, 3], atol=1e-9)
    assert np.abs(out["R_C_W"] - R).max() < 5e-3 and np.abs(out["t_C_W"] - t).max() < 0.2


def test_no_model_and_samples_exhausted(oracle):
    n = 200
    uv, P, _, _, _ = scene(n, 3, outliers=1.0)            # nothing fits
    out = oracle.ransac_localization_dlt(uv[:, ::-1].copy(), P, K, sextets(n, 15000, 1))
    assert not out["found"] and out["n_iter"] == 14999 and not out["mask"].any()
    uv, P, _, _, _ = scene(n, 3, noise=0.3)
    out = oracle.ransac_localization_dlt(uv[:, ::-1].copy(), P, K, sextets(n, 50, 1))   # the loop wants at least 99
    assert out["n_iter"] == -1 and out["found"]
