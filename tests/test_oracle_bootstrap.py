"""Pins the CPU statement of the bootstrap's relative-pose step (oracle/bootstrap_oracle.c) with
analytic known answers and an independent numpy implementation (LAPACK SVD for the decomposition
and for the DLT).  The toolbox function it stands in for (relativeCameraPose) is closed source:
parity against MATLAB itself is unpinned."""
import numpy as np
import pytest

K = np.array([[718.856, 0, 607.1928], [0, 718.856, 185.2157], [0, 0, 1]])


def skew(t):
    return np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])


def two_views(vo, n, seed, noise_px=0.0, outliers=0):
    """Points in front of both cameras; camera 2 = R (x - C).  Returns p1, p2 (2xn), R, t."""
    rng = np.random.default_rng(seed)
    R = vo.synth.rotation_from_rotvec(rng.normal(size=3) * 0.05)
    C2 = np.array([0.3, -0.1, 1.5]) + rng.normal(size=3) * 0.1
    t = -R @ C2
    X = np.stack([rng.uniform(-8, 8, n), rng.uniform(-3, 3, n), rng.uniform(6, 40, n)])
    x1 = K @ X
    x2 = K @ (R @ X + t[:, None])
    p1 = x1[:2] / x1[2] + rng.normal(size=(2, n)) * noise_px
    p2 = x2[:2] / x2[2] + rng.normal(size=(2, n)) * noise_px
    if outliers:
        p2[:, :outliers] = rng.uniform(0, 300, size=(2, outliers))
    return p1, p2, R, t, C2


def numpy_relative_pose(E, p1, p2):
    """Independent implementation: LAPACK SVD, DLT by LAPACK SVD, fewest points behind either camera."""
    U, _, Vt = np.linalg.svd(E)
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    cands = []
    for Rc in (U @ W @ Vt, U @ W.T @ Vt):
        if np.linalg.det(Rc) < 0:
            Rc = -Rc
        for sg in (-1, 1):
            cands.append((Rc, sg * U[:, 2]))
    M1 = K @ np.eye(3, 4)
    best = None
    for Rc, tc in cands:
        M2 = K @ np.hstack([Rc, tc[:, None]])
        neg = 0
        for i in range(p1.shape[1]):
            a, b = np.append(p1[:, i], 1), np.append(p2[:, i], 1)
            A = np.vstack([skew(a) @ M1, skew(b) @ M2])
            P = np.linalg.svd(A)[2][-1]
            P = P / P[3]
            neg += (P[2] < 0) or ((Rc @ P[:3] + tc)[2] < 0)
        if best is None or neg < best[0]:
            best = (neg, Rc, tc)
    neg, Rc, tc = best
    return Rc, -Rc.T @ tc, 1 - neg / p1.shape[1]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_exact_correspondences_recover_the_pose(oracle, vo, seed):
    p1, p2, R, t, C2 = two_views(vo, 200, seed)
    E = skew(t) @ R
    ori, loc, frac = oracle.relative_pose(E * (-3.7 if seed % 2 else 1.0), K, p1, p2)   # scale and sign of E are free
    assert frac == 1.0
    np.testing.assert_allclose(ori, R, atol=1e-9)
    np.testing.assert_allclose(loc, C2 / np.linalg.norm(C2), atol=1e-9)
    assert abs(np.linalg.norm(loc) - 1) < 1e-12


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_matches_numpy_on_noisy_data_with_outliers(oracle, vo, seed):
    p1, p2, R, t, _ = two_views(vo, 300, seed, noise_px=0.3, outliers=40)
    E = skew(t) @ R + np.random.default_rng(seed).normal(size=(3, 3)) * 1e-3   # not exactly essential
    ori, loc, frac = oracle.relative_pose(E, K, p1, p2)
    Rn, locn, fracn = numpy_relative_pose(E, p1, p2)
    np.testing.assert_allclose(ori, Rn, atol=1e-9)
    np.testing.assert_allclose(loc, locn, atol=1e-9)
    assert abs(frac - fracn) < 1e-12
    assert abs(np.linalg.det(ori) - 1) < 1e-12


def test_degenerate_matrix_is_reported(oracle, vo):
    p1, p2, _, _, _ = two_views(vo, 20, 0)
    assert oracle.relative_pose(np.zeros((3, 3)), K, p1, p2) is None
    rank1 = np.outer([1.0, 2, 3], [0.5, -1, 2])
    assert oracle.relative_pose(rank1, K, p1, p2) is None


def test_bootstrap_restatement_recovers_the_rendered_motion(oracle, vo):
    """main_demo.m:252-264 on three rendered corridor frames: the pose between frames 0 and 2 comes
    back with a unit baseline, and the triangulated landmarks agree with the scene geometry up to
    that scale."""
    from oracle import pipeline_oracle as po
    Ks = np.array([[400.0, 0, 320], [0, 400.0, 120], [0, 0, 1]])
    scene = vo.synth.CorridorScene(K=Ks, width=640, height=240)
    poses = vo.synth.corridor_trajectory(3, step=0.5)
    frames = [scene.render(T) for T in poses]
    hp = vo.default_hyper_paras("kitti")
    init_points, matched_points = po.matching_init(frames[0], frames[1:], hp)
    assert len(init_points) > 1000
    rel = np.linalg.inv(poses[0]) @ poses[2]                 # camera 2 in camera-0 coordinates
    base = np.linalg.norm(rel[:3, 3])
    T0, init_valid, _ = po.pose_estimation_init(init_points, matched_points, Ks, vo.draw_octets,
                                                np.random.default_rng(3))
    assert len(init_valid) > 0.3 * len(init_points)
    assert np.linalg.norm(T0[:3, 3] - rel[:3, 3] / base) < 0.03          # direction within ~2 degrees
    assert np.degrees(np.arccos((np.trace(T0[:3, :3].T @ rel[:3, :3].T) - 1) / 2)) < 0.2
    # the whole bootstrap, including the final pose refinement against the fresh landmarks (which
    # moves the pose noticeably under forward motion, in the reference as well)
    X, kp, T = po.bootstrap(frames[0], frames[1:], Ks, hp, vo.draw_octets, np.random.default_rng(3))
    assert X.shape[0] == kp.shape[0] > 500
    assert np.linalg.norm(T[:3, 3] - rel[:3, 3] / base) < 0.35
    truth = scene.backproject(poses[2], kp)                  # world points under the keypoints of frame 2
    truth0 = (np.linalg.inv(poses[0]) @ np.vstack([truth.T, np.ones(len(truth))]))[:3].T / base
    err = np.linalg.norm(X - truth0, axis=1) / np.linalg.norm(truth0, axis=1)
    print("median relative landmark error", np.median(err))
    assert np.median(err) < 0.15
