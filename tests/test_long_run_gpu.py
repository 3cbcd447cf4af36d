"""A reference-length run: main_eval.m:370-462 processes KITTI 05 frames 3..540.  540 frames of the rendered
KITTI-shape corridor at the benchmark's headline settings (1241 x 376, thinning cell 10 px for the first
frame's corners, candidate exclusion radius 5.5 px: about 2 000 landmarks + 2 000 candidates per frame).  The
camera advances 0.7 m per frame: the rendered corridor is 400 m long, and at 1 m per frame the camera reaches its
end wall in frame 399, where the restatement and the device loop both stop ("no pose", as the reference would).

(1) The device-resident loop against the CPU restatement of the loop, every frame: identical landmark /
    candidate / new-landmark counts, identical landmark ids, the same number of RANSAC iterations consumed
    (generator positions equal), poses within 1e-5 m / 1e-5 rad.  The loop feeds its own output back (landmarks are
    triangulated from its poses, poses localised against its landmarks), so a free-running comparison
    measures how the PIPELINE amplifies a rounding difference (observed: 3e-10 m after 50 frames, 1e-5 m after
    175, both sides equally right), not the kernels; the restatement is therefore handed the device's state
    every 30 frames (S.P, S.X, S.C, S.F, S.T: main_demo.m:272-279), which bounds that growth and still compares
    every one of the 539 frames.  The free-running stretch before the first hand-over must agree as well.
(2) The same sequence through main_eval (Harris and FAST): errs 3 x 6 fully populated (all six distance
    checkpoints of quantitative_eval.m:11-12 passed), 538 frames per run inside the record's capacity."""
import importlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES = 540
RESYNC = 30


@pytest.fixture(scope="module")
def long_sequence(vo, tmp_path_factory):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    cache = os.environ.get("VOHIP_CACHE") or str(tmp_path_factory.mktemp("cache"))
    frames, poses = vo.synth.corridor_sequence(N_FRAMES + 1, seed=5678, step=0.7, cache_dir=cache)
    return bench, vo.synth.CorridorScene(seed=5678), frames, poses


def test_540_frames_against_the_cpu_restatement(vo, ctx, oracle, long_sequence):
    from oracle import pipeline_oracle as po
    bench, scene, frames, poses = long_sequence
    hp = bench.kitti_hyper_paras(vo)
    assert (hp.r_discard_redundant, bench.KITTI_THIN_CELL) == (5.5, 10.0)
    X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
    K = vo.synth.K_KITTI
    ref = po.OraclePipeline(K, frames[0], X0, xy0, poses[0], hp, 3, 2, np.random.default_rng(100), vo.draw_triples,
                            vo.draw_octets)
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy0, poses[0], hyper_paras=hp, rng=np.random.default_rng(100), ctx=ctx)
    worst_t = worst_r = 0.0
    for i in range(1, N_FRAMES):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i], next_image=frames[i + 1])
        assert (dev.num_X[-1], dev.num_C[-1], dev.num_new[-1]) == (ref.S.num_X[-1], ref.S.num_C[-1], ref.S.num_new[-1]), i
        assert dev.rng.bit_generator.state == ref.rng.bit_generator.state, i
        dt = float(np.linalg.norm(Tr[:3, 3] - Td[:3, 3]))
        dR = Tr[:3, :3].T @ Td[:3, :3]
        dr = float(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)))
        worst_t, worst_r = max(worst_t, dt), max(worst_r, dr)
        assert dt < 1e-5 and dr < 1e-5, (i, dt, dr)
        if i % RESYNC == 0:
            S = dev.S
            np.testing.assert_array_equal(S.X[3], ref.S.X[3], err_msg="landmark ids after frame %d" % i)
            np.testing.assert_allclose(S.P, ref.S.P, rtol=0, atol=1e-4)
            np.testing.assert_allclose(S.C, ref.S.C, rtol=0, atol=1e-4)
            # hand the device's state to the restatement (its S.T keeps one column per detection, update_candidate.m:57:
            # only the first size(S.C, 2) are ever read)
            ref.S.P, ref.S.X, ref.S.C, ref.S.F = S.P.copy(), S.X.copy(), S.C.copy(), S.F.copy()
            ref.S.T = S.T.copy()
            ref.L_pts = np.fliplr(ref.S.P.T)
    np.testing.assert_array_equal(dev.S.X[3], ref.S.X[3])
    assert 1000 < dev.num_X[-1] < 8192 and 1000 < dev.num_C[-1] < 8192
    assert len(dev.num_X) == N_FRAMES
    print("worst |dt| %.3g m, worst angle %.3g rad over %d frames" % (worst_t, worst_r, N_FRAMES - 1))
    dev.close()


def test_540_frames_through_main_eval(vo, ctx, long_sequence, tmp_path):
    bench, scene, frames, poses = long_sequence
    frames, poses = frames[:N_FRAMES], poses[:N_FRAMES]
    seq = dict(name="kitti", images=frames, K=vo.synth.K_KITTI, hyper_paras=bench.kitti_hyper_paras(vo),
               ground_truth=np.array([[T[0, 3], T[2, 3]] for T in poses]))
    alone = vo.main_eval([seq], feats=("Harris", "FAST"), repetition=1, seed=3, ctx=ctx, save_dir=str(tmp_path / "a"))
    grouped = vo.main_eval([seq], feats=("Harris", "FAST"), repetition=1, seed=3, ctx=ctx, group=2)
    for a, g in zip(alone, grouped):
        assert not a["failed"] and a["frame_ct"] == N_FRAMES - 3 and a["frames"] == N_FRAMES - 2 <= vo.batch.T_MAX
        assert a["errs"].shape == (3, 6) and np.all(np.isfinite(a["errs"]))      # 376 m travelled: 2 ... 72 m checkpoints
        assert np.all(a["errs"][:, 0] < 0.5)
        for key in ("est_trans", "est_rot", "num_X", "num_C", "num_new", "errs"):
            np.testing.assert_array_equal(a[key], g[key], err_msg=key)
    back = vo.load_results(os.path.join(tmp_path / "a", sorted(os.listdir(tmp_path / "a"))[0]))
    assert back["S"]["est_trans"].shape == (3, N_FRAMES - 2) and back["errs"].shape == (3, 6)


def test_a_group_of_loops_over_200_frames_is_its_single_loops(vo, ctx, long_sequence):
    """Five sequences of 120-200 frames each (windows of the long sequence starting 0, 40, ... frames into it,
    own start sets and seeds) as one group of loops in lockstep against their own single loops: poses, counts, ids,
    final state and generator positions bit for bit, through hundreds of frames in which the landmark and
    candidate counts of every sequence move independently across the kernels' block sizes and the shorter
    sequences drop out of the group."""
    bench, scene, frames, poses = long_sequence
    hp = bench.kitti_hyper_paras(vo)
    K = vo.synth.K_KITTI
    specs = []
    for s, (first, length) in enumerate(((0, 200), (40, 160), (80, 200), (120, 120), (160, 180))):
        fr = frames[first:first + length + 1]
        X0, xy0 = bench.init_points(vo, ctx, scene, fr[0], poses[first])
        specs.append((fr, X0, xy0, poses[first], 500 + s))
    refs = []
    for fr, X0, xy0, T0, seed in specs:
        dev = vo.VOPipelineDevice(K, fr[0], X0, xy0, T0, hyper_paras=hp, rng=np.random.default_rng(seed), ctx=ctx)
        Ts = [dev.process_frame(fr[i], next_image=fr[i + 1] if i + 1 < len(fr) else None) for i in range(1, len(fr))]
        refs.append((np.array(Ts), list(dev.num_X), list(dev.num_C), list(dev.num_new), dev.S, dev.rng.bit_generator.state))
        dev.close()
    group = vo.VOGroupDevice(K, frames[0].shape, len(specs), max(len(sp[0]) for sp in specs), hyper_paras=hp, ctx=ctx)
    for s, (fr, X0, xy0, T0, seed) in enumerate(specs):
        group.add_sequence(s, fr, X0, xy0, T0, rng=np.random.default_rng(seed))
    assert group.run() == sum(len(sp[0]) - 1 for sp in specs)
    for s, (T, nX, nC, nN, S, state) in enumerate(refs):
        q = group.seqs[s]
        assert q.error is None, s
        et, er = group.trajectory(s)
        np.testing.assert_array_equal(et[:, 1:], T[:, :3, 3].T, err_msg="translations of sequence %d" % s)
        np.testing.assert_array_equal(er[:, 1:], np.array([t[:3, :3].reshape(-1, order="F") for t in T]).T)
        assert (q.num_X, q.num_C, q.num_new) == (nX, nC, nN), s
        assert q.rng.bit_generator.state == state, s
        Sg = group.S(s)
        for key in ("P", "X", "C", "F", "T"):
            np.testing.assert_array_equal(getattr(Sg, key), getattr(S, key), err_msg="%s of sequence %d" % (key, s))
    group.close()
