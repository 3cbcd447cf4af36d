"""Parity of the HIP bundle adjustment / pose refinement with the CPU oracle.
Bars: residuals 1e-9 px; BA-refined poses within 1e-5 rad / 1e-5 m and landmarks within 1e-6
relative of the oracle's (same algorithm, fp64; only summation order differs); identical
iteration counts."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pose_diff(vo, s_a, s_b, F, ctx):
    rot, trans = 0.0, 0.0
    for f in range(F):
        Ta = vo.twist2HomogMatrix(s_a[6 * f:6 * f + 6], ctx=ctx)
        Tb = vo.twist2HomogMatrix(s_b[6 * f:6 * f + 6], ctx=ctx)
        dR = Ta[:3, :3].T @ Tb[:3, :3]
        rot = max(rot, np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
        trans = max(trans, np.linalg.norm(Ta[:3, 3] - Tb[:3, 3]))
    return rot, trans


def test_ba_error_matches_golden(vo, ctx):
    g = np.load(os.path.join(GOLD, "ba_f5_m300.npz"))
    r = vo.baError(g["state0"], g["obs"], g["K"], ctx=ctx)
    assert r.shape == g["r0"].shape
    np.testing.assert_allclose(r, g["r0"], rtol=0, atol=1e-9)


def test_run_ba_matches_golden(vo, ctx):
    g = np.load(os.path.join(GOLD, "ba_f5_m300.npz"))
    s1, info = vo.runBA(g["state0"], g["obs"], g["K"], ftol=1e-10, xtol=1e-10, return_stats=True, ctx=ctx)   # the tolerances the fixture was made with
    assert info["iterations"] == int(g["iterations"])
    assert abs(info["cost"] - float(g["cost"])) < 1e-9 * float(g["cost"])
    rot, trans = _pose_diff(vo, s1, g["state_opt"], 5, ctx)
    assert rot < 1e-5 and trans < 1e-5, (rot, trans)
    Xa, Xb = s1[30:].reshape(-1, 3), g["state_opt"][30:].reshape(-1, 3)
    assert np.max(np.linalg.norm(Xa - Xb, axis=1) / np.linalg.norm(Xb, axis=1)) < 1e-6
    assert s1.shape == g["state0"].shape


@pytest.mark.parametrize("F,m,seed", [(5, 1000, 21), (3, 80, 22), (8, 400, 23), (2, 17, 24), (5, 16, 25), (1, 60, 26), (4, 120, 27),
                                      (6, 200, 28), (7, 150, 29)])
def test_run_ba_matches_oracle(vo, ctx, oracle, F, m, seed):
    K = vo.synth.K_KITTI
    s0, obs, st = vo.synth.ba_problem(F=F, m=m, seed=seed)
    want, winfo = oracle.run_ba(s0, obs, K)
    got, info = vo.runBA(s0.reshape(-1, 1), obs.reshape(-1, 1), K, return_stats=True, ctx=ctx)
    assert got.shape == (len(s0), 1)
    got = got[:, 0]
    assert info["iterations"] == winfo["iterations"]
    assert abs(info["cost"] - winfo["cost"]) < 1e-8 * (1 + winfo["cost"])
    rot, trans = _pose_diff(vo, got, want, F, ctx)
    # 16 landmarks seen 60 % of the time barely constrain 5 free poses (and nothing fixes the
    # gauge): rounding differences are amplified along the weak directions, so that case gets a
    # looser bar; the well-constrained windows meet the 1e-5 rad / 1e-5 m bar.
    tol_pose, tol_lm = (1e-5, 1e-6) if m >= 17 else (1e-3, 1e-4)
    assert rot < tol_pose and trans < tol_pose, (rot, trans)
    Xa, Xb = got[6 * F:].reshape(-1, 3), want[6 * F:].reshape(-1, 3)
    assert np.max(np.linalg.norm(Xa - Xb, axis=1) / np.linalg.norm(Xb, axis=1)) < tol_lm


def test_run_ba_edge_cases(vo, ctx):
    K = vo.synth.K_KITTI
    s0, obs, st = vo.synth.ba_problem(F=3, m=40, seed=31)
    # max_iter = 0 leaves the state untouched
    np.testing.assert_array_equal(vo.runBA(s0, obs, K, max_iter=0, ctx=ctx), s0)
    # malformed observation vectors are rejected, not read out of bounds
    with pytest.raises(vo.VohipError):
        vo.runBA(s0, obs[:-1], K, ctx=ctx)
    with pytest.raises(vo.VohipError):
        vo.runBA(s0[:-3], obs, K, ctx=ctx)
    bad = obs.copy()
    bad[-1] = 41  # landmark id out of range
    with pytest.raises(vo.VohipError):
        vo.runBA(s0, bad, K, ctx=ctx)
    # no observations at all
    empty = np.array([2.0, 3.0, 0.0, 0.0])
    s = np.zeros(12 + 9)
    np.testing.assert_array_equal(vo.runBA(s, empty, K, ctx=ctx), s)


def test_pose_refinement_matches_golden_and_oracle(vo, ctx, oracle):
    g = np.load(os.path.join(GOLD, "ba_f5_m300.npz"))
    T1, info = vo.T_refinement(g["pr_T0"], g["pr_kp_uv"], g["pr_X"], g["K"], return_stats=True, ctx=ctx)
    np.testing.assert_allclose(T1, g["pr_T1"], rtol=0, atol=1e-9)
    assert info["iterations"] == int(g["pr_iterations"])
    K = vo.synth.K_KITTI
    for n, seed in [(2000, 41), (64, 42), (6, 43), (1500, 44)]:
        kp_rc, X, R, t, _ = vo.synth.pnp_problem(n, inlier_ratio=1.0, seed=seed)
        T_C_W = np.eye(4)
        T_C_W[:3, :3] = R
        T_C_W[:3, 3] = t
        T0 = np.linalg.inv(T_C_W) @ vo.synth.pose_matrix([0.01, 0.02, -0.01], [0.1, 0.1, 0.1])
        want, winfo = oracle.pose_refine(T0, kp_rc[::-1], X, K)
        got = vo.T_refinement(T0, kp_rc[::-1], X, K, ctx=ctx)
        dR = want[:3, :3].T @ got[:3, :3]
        assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-5
        assert np.linalg.norm(want[:3, 3] - got[:3, 3]) < 1e-5
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-8)


def test_batched_pose_refinement_equals_single_calls(vo, ctx):
    """VO_bundle_adjust.m:71-86 re-refines the buffered normal frames one T_refinement call after the other;
    here they go out as one launch.  Bar: bit-identical to the single calls, whatever the mix of sizes
    (an empty frame keeps its pose)."""
    K = vo.synth.K_KITTI
    problems = []
    for n, seed in [(900, 51), (64, 52), (0, 53), (1500, 54), (7, 55)]:
        kp_rc, X, R, t, _ = vo.synth.pnp_problem(max(n, 6), inlier_ratio=1.0, seed=seed)
        T_C_W = np.eye(4)
        T_C_W[:3, :3] = R
        T_C_W[:3, 3] = t
        T0 = np.linalg.inv(T_C_W) @ vo.synth.pose_matrix([0.01, -0.02, 0.01], [0.1, -0.1, 0.05])
        problems.append((T0, kp_rc[::-1][:, :n], X[:, :n]))
    single = [vo.T_refinement(T, kp, X, K, ctx=ctx) if kp.shape[1] else np.array(T) for T, kp, X in problems]
    batch = vo.T_refinement_batch(problems, K, ctx=ctx)
    assert len(batch) == len(problems)
    for a, b in zip(single, batch):
        np.testing.assert_array_equal(a, b)
    assert vo.T_refinement_batch([], K, ctx=ctx) == []
    # twist2HomogMatrix for a window of twists in one call
    tw = np.random.default_rng(5).normal(size=(5, 6)) * 0.3
    many = vo.geometry.twist2HomogMatrix_many(tw, ctx=ctx)
    for k in range(5):
        np.testing.assert_array_equal(many[k], vo.twist2HomogMatrix(tw[k], ctx=ctx))
