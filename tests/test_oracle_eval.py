"""Pins the CPU restatement of the evaluation (oracle/align_oracle.c, pipeline_oracle.quantitative_eval)
with the closed-form similarity alignment (Umeyama) and scipy's least squares, and checks the host
side that needs no GPU (the result file).  lsqnonlin's iterates are unpinned; the minimum is not."""
import numpy as np
import pytest
from scipy.optimize import least_squares


def umeyama(gt, est):
    """Closed-form least-squares similarity: gt ~ s R est + t."""
    mg, me = gt.mean(axis=1, keepdims=True), est.mean(axis=1, keepdims=True)
    G, E = gt - mg, est - me
    U, D, Vt = np.linalg.svd(G @ E.T / gt.shape[1])
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = np.trace(np.diag(D) @ S) / (E ** 2).sum() * gt.shape[1]
    return s, R, mg - s * R @ me


def trajectory(vo, n, seed, noise=0.02):
    rng = np.random.default_rng(seed)
    est = np.cumsum(np.vstack([rng.normal(0, 0.05, n), rng.normal(0, 0.01, n), np.full(n, 0.5)]), axis=1)
    R = vo.synth.rotation_from_rotvec(rng.normal(size=3) * 0.3)
    s, t = 1.7 + 0.2 * rng.normal(), rng.normal(size=(3, 1)) * 2
    gt = s * R @ est + t + rng.normal(size=(3, n)) * noise
    return gt, est


@pytest.mark.parametrize("n,seed", [(3, 0), (40, 1), (540, 2)])
def test_alignment_reaches_the_closed_form_minimum(oracle, vo, n, seed):
    gt, est = trajectory(vo, n, seed)
    aligned, x, st = oracle.align_sim3(gt, est, ftol=1e-10, xtol=1e-10)   # (tighter than the reference's 1e-6: the minimum itself)
    s, R, t = umeyama(gt, est)
    want = s * R @ est + t
    # three nearly collinear points leave the rotation about their line almost free
    np.testing.assert_allclose(aligned, want, atol=1e-5 if n == 3 else 1e-6)
    assert abs(x[6] - s) < (1e-5 if n == 3 else 1e-7)
    assert st["cost"] <= st["cost0"] and st["iterations"] < 40
    # the returned points are the transform of alignEstimateToGroundTruth.m:20-25 applied to the estimate
    T = oracle.twist2homog(x[:6])
    np.testing.assert_allclose(aligned, x[6] * T[:3, :3] @ est + T[:3, 3:4], atol=1e-12)


def test_alignment_matches_scipy_least_squares(oracle, vo):
    gt, est = trajectory(vo, 100, 5, noise=0.1)
    _, x, st = oracle.align_sim3(gt, est, ftol=1e-10, xtol=1e-10)
    r = least_squares(lambda v: oracle.align_error(v, gt, est), np.array([0, 0, 0, 0, 0, 0, 1.0]), method="lm",
                      xtol=1e-14, ftol=1e-14)
    assert abs(2 * r.cost - st["cost"]) < 1e-8 * (1 + st["cost"])
    np.testing.assert_allclose(x, r.x, atol=1e-6)


def test_align_error_layout(oracle, vo):
    gt, est = trajectory(vo, 5, 7)
    e = oracle.align_error(np.array([0, 0, 0, 0, 0, 0, 1.0]), gt, est)
    np.testing.assert_allclose(e, (gt - est).reshape(-1, order="F"), atol=1e-15)


def test_quantitative_eval_on_a_scaled_trajectory(oracle, vo):
    """An estimate that is an exact similarity of the ground truth has zero error at every
    checkpoint; all six checkpoints (2, 8, 18, 32, 50, 72 m) are passed on a 100 m run."""
    from oracle import pipeline_oracle as po
    n = 210
    z = np.arange(n) * 0.5
    gt_xz = np.stack([0.05 * z * np.sin(z / 15), z], axis=1)          # N x 2 table (x, z)
    b = (0, 2)
    GT3 = np.vstack([gt_xz[2:, 0], np.zeros(n - 2), gt_xz[2:, 1]])
    R = vo.synth.rotation_from_rotvec(np.array([0.02, 0.4, -0.01]))
    est = (R.T @ (GT3 - GT3[:, :1])) / 2.5
    errs = po.quantitative_eval(gt_xz, est, b)
    assert errs.shape == (3, 6)
    assert np.abs(errs).max() < 1e-6
    # a drifting estimate has growing error
    drift = est + np.vstack([np.zeros(n - 2), np.zeros(n - 2), 0.002 * np.arange(n - 2) ** 1.5 / 2.5])
    errs2 = po.quantitative_eval(gt_xz, drift, b)
    assert errs2.shape == (3, 6) and np.linalg.norm(errs2[:, -1]) > np.linalg.norm(errs2[:, 0])


def test_result_file_has_the_reference_key_set(vo, tmp_path):
    """main_demo.m:473-476: errs, S, total_time, frame_init, frame_ct, toc_param, toc_bootstrap, toc_ct,
    fps_ct -- read back with scipy.io.loadmat the way the evaluation notebook does."""
    from types import SimpleNamespace
    from scipy.io import loadmat
    S = SimpleNamespace(P=np.ones((2, 5)), X=np.ones((4, 5)), C=np.zeros((2, 3)), F=np.zeros((2, 3)), T=np.zeros((12, 3)),
                        est_trans=np.arange(12.0).reshape(3, 4), est_rot=np.zeros((9, 4)), num_X=[5, 5, 5, 5],
                        num_C=[0, 1, 2, 3], num_new=[0, 0, 0, 0])
    hp = vo.default_hyper_paras("kitti")
    import datetime
    name = vo.experiment_name("kitti", hp, now=datetime.datetime(2022, 1, 9, 13, 5, 7))
    assert name == "dataset@kitti_feat@Harris_time@13_05_07"
    hp.is_refine_pose = False
    assert vo.experiment_name("kitti", hp, now=datetime.datetime(2022, 1, 9, 13, 5, 7)).endswith("_NoRefined")
    path = vo.save_results(str(tmp_path / name), np.ones((3, 6)), S, 12.5, 3, 3, 0.1, 1.2, 11.2, 3 / 11.2)
    assert path.endswith("_res.mat")
    raw = loadmat(path)
    assert {"errs", "S", "total_time", "frame_init", "frame_ct", "toc_param", "toc_bootstrap", "toc_ct",
            "fps_ct"} <= set(raw)
    assert raw["errs"].shape == (3, 6) and float(raw["fps_ct"][0, 0]) == pytest.approx(3 / 11.2)
    back = vo.load_results(path)
    np.testing.assert_array_equal(back["S"]["est_trans"], S.est_trans)
    assert back["S"]["num_C"].shape == (4, 1) and back["frame_ct"] == 3.0
