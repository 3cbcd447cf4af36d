"""End-to-end parity: the GPU frame loop (VOPipeline and the stage functions) against the CPU
restatement of the reference loop on a rendered corridor sequence, same RNG seed.
Bar: identical landmark / candidate counts per frame and trajectories within 1e-5 m / 1e-5 rad
(every stage underneath meets its own, tighter bar; a flip of one borderline decision would change
a count and fail this test loudly)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 640, 240
K = np.array([[400.0, 0, 320], [0, 400.0, 120], [0, 0, 1]])


@pytest.fixture(scope="module")
def sequence(vo):
    scene = vo.synth.CorridorScene(K=K, width=W, height=H)
    poses = vo.synth.corridor_trajectory(9, step=0.5)
    return scene, poses, [scene.render(T) for T in poses]


def _run_both(vo, ctx, oracle, sequence, is_BA, window, kd, n_frames):
    from oracle import pipeline_oracle as po
    scene, poses, frames = sequence
    ab, _, _ = oracle.detect_corners(frames[0], min_quality=2e-4, border=16)
    xy = (ab + 1.0).astype(np.float64)
    X0 = scene.backproject(poses[0], xy)
    hp = vo.default_hyper_paras("kitti", is_BA=is_BA)
    ref = po.OraclePipeline(K, frames[0], X0, xy, poses[0], hp, window, kd, np.random.default_rng(1), vo.draw_triples,
                            vo.draw_octets)
    gpu = vo.VOPipeline(K, frames[0], X0, xy, poses[0], hyper_paras=hp, window_size=window, keyframe_d=kd,
                        rng=np.random.default_rng(1), ctx=ctx)
    for i in range(1, n_frames):
        Tr = ref.process_frame(frames[i])
        Tg = gpu.process_frame(frames[i])
        assert gpu.S.X.shape == ref.S.X.shape and gpu.S.C.shape == ref.S.C.shape, i
        assert gpu.S.num_new[-1] == ref.S.num_new[-1], i
        assert np.linalg.norm(Tr[:3, 3] - Tg[:3, 3]) < 1e-5, i
        dR = Tr[:3, :3].T @ Tg[:3, :3]
        assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-5, i
        # and the estimate is close to the ground truth of the rendered sequence
        assert np.linalg.norm(Tg[:3, 3] - poses[i][:3, 3]) < 0.25, i
    gpu.close()
    return ref, gpu


def test_frame_loop_matches_cpu_restatement(vo, ctx, oracle, sequence):
    ref, gpu = _run_both(vo, ctx, oracle, sequence, False, 3, 2, 9)
    et, er = gpu.trajectory()
    np.testing.assert_allclose(et, ref.S.est_trans, rtol=0, atol=1e-5)
    np.testing.assert_allclose(er, ref.S.est_rot, rtol=0, atol=1e-5)
    assert gpu.S.num_X == ref.S.num_X and gpu.S.num_C == ref.S.num_C
    np.testing.assert_array_equal(gpu.S.X[3], ref.S.X[3])           # landmark ids
    np.testing.assert_allclose(gpu.S.X[:3], ref.S.X[:3], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(gpu.S.P, ref.S.P, rtol=0, atol=0.05)  # tracked keypoints within 0.05 px


def test_frame_loop_with_bundle_adjustment(vo, ctx, oracle, sequence):
    """main_BA.m configuration scaled down: window of 3 keyframes, every 2nd frame a keyframe."""
    ref, gpu = _run_both(vo, ctx, oracle, sequence, True, 3, 1, 9)
    et, er = gpu.trajectory()
    np.testing.assert_allclose(et, ref.S.est_trans, rtol=0, atol=1e-5)
    np.testing.assert_allclose(er, ref.S.est_rot, rtol=0, atol=1e-5)
    assert gpu.B.num_key == ref.B.num_key == 3 and gpu.B.m == ref.B.m
    np.testing.assert_array_equal(gpu.B.landmarks[3], ref.B.landmarks[3])
    np.testing.assert_allclose(gpu.B.landmarks[:3], ref.B.landmarks[:3], rtol=1e-6, atol=1e-6)
