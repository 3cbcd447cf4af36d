"""The HIP DLT pose and the DLT branch of the RANSAC localisation (csrc/dlt.hip) against the CPU restatement
(oracle/dlt_oracle.c), same six-samples.
Bar: poses within 1e-9 (the two sides order their Jacobi rotations differently, and the refit goes through a
Householder factor on the GPU); per-hypothesis counts, histories, stop index and the inlier mask bit-exact,
except correspondences whose squared error lies within 1e-6 px^2 of the 100 px^2 threshold under the oracle's
pose (asserted absent from the fixtures used here)."""
import numpy as np
import pytest

from test_oracle_dlt import K, numpy_estimate_pose_dlt, scene, sextets

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed,noise", [(6, 0, 0.0), (6, 1, 0.6), (7, 2, 0.6), (13, 3, 0.6), (257, 4, 0.5), (2000, 5, 0.5),
                                          (5000, 6, 0.5)])
def test_estimate_pose_dlt_matches_oracle_and_numpy(vo, ctx, oracle, n, seed, noise):
    uv, P, R, t, _ = scene(n, seed, noise=noise)
    got = vo.estimatePoseDLT(uv, P, K, ctx=ctx)
    np.testing.assert_allclose(got, oracle.estimate_pose_dlt(uv, P, K), rtol=0, atol=1e-9)
    np.testing.assert_allclose(got, numpy_estimate_pose_dlt(uv, P, K), rtol=0, atol=1e-9)
    assert abs(np.linalg.det(got[:, :3]) - 1) < 1e-12
    if noise == 0.0:
        np.testing.assert_allclose(got[:, :3], R, atol=1e-9)
        np.testing.assert_allclose(got[:, 3], t, atol=1e-8)


def test_estimate_pose_dlt_needs_six(vo, ctx):
    uv, P, _, _, _ = scene(5, 0)
    with pytest.raises(vo.VohipError):
        vo.estimatePoseDLT(uv, P, K, ctx=ctx)


def _tie_free(uv, P, out):
    """No correspondence within 1e-6 px^2 of the threshold under any executed hypothesis of the oracle run."""
    for M in out["M_hyp"]:
        pc = P @ M[:, :3].T + M[:, 3]
        with np.errstate(all="ignore"):
            proj = pc[:, :2] / pc[:, 2:3] * np.array([K[0, 0], K[1, 1]]) + K[:2, 2]
            e = ((uv - proj) ** 2).sum(1)
        if np.any(np.abs(e - 100.0) < 1e-6):
            return False
    return True


@pytest.mark.parametrize("n,seed,outliers", [(400, 11, 0.25), (2000, 12, 0.3), (150, 13, 0.1), (8192, 14, 0.3)])
def test_ransac_dlt_matches_oracle(vo, ctx, oracle, n, seed, outliers):
    uv, P, R, t, _ = scene(n, seed, noise=0.4, outliers=outliers)
    kp_rc = uv[:, ::-1].copy()
    S = sextets(n, 1200, seed)
    want = oracle.ransac_localization_dlt(kp_rc, P, K, S)
    assert want["found"] and want["n_iter"] >= 99 and _tie_free(uv, P, want)
    Rg, tg, mask, mh, ih = vo.ransacLocalization(kp_rc.T, P.T, K, samples=S, ctx=ctx, use_p3p=False)
    assert len(mh) == want["n_iter"]
    np.testing.assert_array_equal(mh, want["max_num_inliers_history"])
    np.testing.assert_array_equal(ih, want["num_iteration_history"])
    np.testing.assert_array_equal(mask.reshape(-1).astype(np.uint8), want["mask"])
    np.testing.assert_allclose(Rg, want["R_C_W"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(tg.reshape(-1), want["t_C_W"], rtol=0, atol=1e-9)
    assert np.abs(Rg - R).max() < 5e-3


def test_ransac_dlt_draws_its_own_samples_and_leaves_the_generator_after_n_iter(vo, ctx, oracle):
    n = 600
    uv, P, _, _, _ = scene(n, 21, noise=0.4, outliers=0.55)     # many outliers: the first 104 sets do not end the loop
    kp_rc = uv[:, ::-1].copy()
    rng = np.random.default_rng(9)
    Rg, tg, mask, mh, ih = vo.ransacLocalization(kp_rc.T, P.T, K, rng=rng, ctx=ctx, use_p3p=False)
    k = len(mh)
    assert k > 104
    ref = np.random.default_rng(9)
    S = vo.draw_sextets(ref, n, k)
    assert rng.bit_generator.state == ref.bit_generator.state
    assert all(len(set(row)) == 6 and min(row) >= 0 and max(row) < n for row in S[:50].tolist())
    want = oracle.ransac_localization_dlt(kp_rc, P, K, S)
    assert want["n_iter"] == k
    np.testing.assert_array_equal(mask.reshape(-1).astype(np.uint8), want["mask"])
    np.testing.assert_allclose(Rg, want["R_C_W"], rtol=0, atol=1e-9)


def test_ransac_dlt_without_a_model_and_with_bad_indices(vo, ctx):
    n = 200
    uv, P, _, _, _ = scene(n, 3, outliers=1.0)
    S = sextets(n, 300, 1)
    Rg, tg, mask, mh, ih = vo.ransacLocalization(uv[:, ::-1].T, P.T, K, samples=np.vstack([S] * 50), ctx=ctx, use_p3p=False)
    assert Rg.size == 0 and tg.size == 0 and not mask.any() and len(mh) == 14999      # ransacLocalization.m:138-140
    # a sample set with an index outside the list counts nothing (NaN pose), as a degenerate P3P triple does
    uv, P, _, _, _ = scene(n, 4, noise=0.3)
    S = sextets(n, 200, 2)
    S[0, 3] = n + 5
    S[1, 0] = -1
    Rg, tg, mask, mh, ih = vo.ransacLocalization(uv[:, ::-1].T, P.T, K, samples=S, ctx=ctx, use_p3p=False)
    assert mh[0] == 0 and mh[1] == 0 and mh[-1] >= 30 and Rg.shape == (3, 3)
    with pytest.raises(vo.VohipError):
        vo.ransacLocalization(uv[:5, ::-1].T, P[:5].T, K, samples=S[:, :], ctx=ctx, use_p3p=False)
