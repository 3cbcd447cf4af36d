"""Parity of the HIP trajectory alignment and of the evaluation built on it with the CPU oracle, and
the stand-alone run (bootstrap -> continuous operation -> evaluation -> result file).
Bar: aligned points and checkpoint errors within 1e-6 m (the GPU uses analytic Jacobians, the
oracle central differences; both stop at the same minimum to that accuracy)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _trajectory(vo, n, seed, noise=0.02):
    rng = np.random.default_rng(seed)
    est = np.cumsum(np.vstack([rng.normal(0, 0.05, n), rng.normal(0, 0.01, n), np.full(n, 0.5)]), axis=1)
    R = vo.synth.rotation_from_rotvec(rng.normal(size=3) * 0.3)
    s, t = 1.7 + 0.2 * rng.normal(), rng.normal(size=(3, 1)) * 2
    return s * R @ est + t + rng.normal(size=(3, n)) * noise, est


@pytest.mark.parametrize("n,seed", [(4, 0), (40, 1), (255, 2), (540, 3), (3000, 4)])
def test_alignment_matches_oracle(vo, ctx, oracle, n, seed):
    gt, est = _trajectory(vo, n, seed)
    # at lsqnonlin's default tolerances (alignEstimateToGroundTruth.m:17-18, what both sides default to) and at 1e-10
    for tol in ({}, dict(ftol=1e-10, xtol=1e-10)):
        want, xw, sw = oracle.align_sim3(gt, est, **tol)
        got, xg, sg = vo.alignEstimateToGroundTruth(gt, est, return_x=True, ctx=ctx, **tol)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
        np.testing.assert_allclose(xg, xw, rtol=0, atol=1e-6)
        assert abs(sg["cost"] - sw["cost"]) < 1e-9 * (1 + sw["cost"])
        assert sg["cost0"] == pytest.approx(sw["cost0"], rel=1e-12)
        assert sg["iterations"] == sw["iterations"]
    # alignError of the mirror agrees with the oracle's
    np.testing.assert_allclose(vo.alignError(xg, gt, est, ctx=ctx), oracle.align_error(xg, gt, est), atol=1e-12)


def test_prefix_alignments_equal_single_calls(vo, ctx):
    """vohip_align_sim3_prefixes (the checkpoints of quantitative_eval.m:20-27 side by side in one launch) gives each
    prefix what vohip_align_sim3 gives it alone, bit for bit."""
    import ctypes as C
    lib = vo._lib.load()
    gt, est = _trajectory(vo, 540, 7)
    lengths = [21, 83, 84, 256, 257, 540, 1]
    k = len(lengths)
    g, e = vo._lib.colmajor(gt), vo._lib.colmajor(est)
    x, fl, st = np.zeros((k, 7)), np.zeros((k, 6)), np.zeros((k, 3))
    ctx.check(lib.vohip_align_sim3_prefixes(ctx.handle, vo._lib.dptr(g), vo._lib.dptr(e), 540, k, (C.c_int * k)(*lengths), 400,
                                            1e-6, 1e-6, vo._lib.dptr(x), vo._lib.dptr(fl), vo._lib.dptr(st)))
    for b, n in enumerate(lengths):
        al, xs, ss = vo.alignEstimateToGroundTruth(gt[:, :n], est[:, :n], return_x=True, ctx=ctx)
        np.testing.assert_array_equal(x[b], xs)
        np.testing.assert_array_equal(fl[b, :3], al[:, 0])
        np.testing.assert_array_equal(fl[b, 3:], al[:, -1])
        assert (st[b, 0], st[b, 2]) == (ss["iterations"], ss["cost"])


def test_quantitative_eval_matches_oracle(vo, ctx, oracle):
    from types import SimpleNamespace
    from oracle import pipeline_oracle as po
    n = 210
    z = np.arange(n) * 0.5
    gt_xz = np.stack([0.05 * z * np.sin(z / 15), z], axis=1)
    GT3 = np.vstack([gt_xz[2:, 0], np.zeros(n - 2), gt_xz[2:, 1]])
    rng = np.random.default_rng(0)
    est = (GT3 - GT3[:, :1]) / 2.5 + np.cumsum(rng.normal(size=(3, n - 2)) * 0.004, axis=1)
    want = po.quantitative_eval(gt_xz, est, (0, 2))
    got = vo.quantitative_eval(gt_xz, SimpleNamespace(est_trans=est), (0, 2), ctx=ctx)
    assert got.shape == want.shape == (3, 6)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    # a short run passes fewer checkpoints
    short = vo.quantitative_eval(gt_xz, SimpleNamespace(est_trans=est[:, :30]), (0, 2), ctx=ctx)
    assert short.shape == (3, 2)


def test_run_sequence_writes_the_result_file(vo, ctx, tmp_path):
    K = np.array([[400.0, 0, 320], [0, 400.0, 120], [0, 0, 1]])
    scene = vo.synth.CorridorScene(K=K, width=640, height=240)
    poses = vo.synth.corridor_trajectory(14, step=0.5)
    frames = [scene.render(T) for T in poses]
    gt_xz = np.array([[T[0, 3], T[2, 3]] for T in poses])
    hp = vo.default_hyper_paras("kitti")
    out = vo.run_sequence(frames, K, hp, ground_truth=gt_xz, bootstrap_frames=(0, 2), dataset="corridor",
                          save_dir=str(tmp_path), rng=np.random.default_rng(5), ctx=ctx)
    assert out["frame_ct"] == 11 and out["frame_init"] == 3
    assert out["S"].est_trans.shape == (3, 12)          # bootstrap pose + 11 frames
    assert out["errs"].shape[0] == 3 and out["errs"].shape[1] >= 1      # 5.5 m travelled: the 2 m checkpoint
    assert np.all(out["errs"] < 0.5)
    back = vo.load_results(out["path"])
    np.testing.assert_array_equal(back["S"]["est_trans"], out["S"].est_trans)
    np.testing.assert_array_equal(back["errs"], out["errs"])
    assert back["fps_ct"] == pytest.approx(out["fps_ct"]) and back["frame_ct"] == 11.0


def test_main_eval_grid_is_independent_of_world_size(vo, ctx, tmp_path):
    """main_eval.m:22-23 over two sequences x two feature variants, the unit that is sharded over GPUs: the
    grid run by one process, and then the shares of rank 0 and rank 1 of a two-process job run in turn on
    this GPU.  Every run's record must be bit-identical (its random numbers depend on the item only), and
    every run leaves its own '<experiment>_res.mat' with the reference's key set."""
    import glob
    K = np.array([[400.0, 0, 320], [0, 400.0, 120], [0, 0, 1]])
    sequences = []
    for k in range(2):
        scene = vo.synth.CorridorScene(seed=100 + k, K=K, width=640, height=240)
        poses = vo.synth.corridor_trajectory(12, step=0.5)
        sequences.append(dict(name="corridor%d" % k, images=[scene.render(T) for T in poses], K=K,
                              ground_truth=np.array([[T[0, 3], T[2, 3]] for T in poses]),
                              hyper_paras=vo.default_hyper_paras("kitti")))
    feats = ("Harris", "FAST")                                               # two of main_eval.m:19's variants
    whole = vo.main_eval(sequences, feats=feats, repetition=1, save_dir=str(tmp_path / "w1"), seed=9, ctx=ctx)
    assert len(whole) == 4 and all(r is not None for r in whole)
    parts = [vo.main_eval(sequences, feats=feats, repetition=1, save_dir=str(tmp_path / "w2"), seed=9, ctx=ctx, rank=r,
                          world=2) for r in range(2)]
    for k in range(4):
        assert (parts[0][k] is None) != (parts[1][k] is None)            # exactly one rank ran item k
        assert (parts[k % 2][k] is not None)                             # round-robin
        a, b = whole[k], parts[k % 2][k]
        for key in ("est_trans", "est_rot", "num_X", "num_C", "num_new", "errs"):
            np.testing.assert_array_equal(a[key], b[key], err_msg="%d %s" % (k, key))
        assert a["frames"] == 10 and a["frame_ct"] == 9 and a["frame_init"] == 3
        assert a["fps_ct"] > 0 and a["toc_ct"] > 0 and a["errs"].shape[0] == 3 and a["errs"].shape[1] >= 1
        assert (a["sequence"], a["feat"]) == ("corridor%d" % (k // 2), feats[k % 2])
    # ... and of how many runs a process keeps in flight on its GPU (a thread and a context per run)
    together = vo.main_eval(sequences, feats=feats, repetition=1, save_dir=str(tmp_path / "w3"), seed=9, ctx=ctx, concurrent=3)
    for k in range(4):
        for key in ("est_trans", "est_rot", "num_X", "num_C", "num_new", "errs"):
            np.testing.assert_array_equal(whole[k][key], together[k][key], err_msg="concurrent %d %s" % (k, key))
    # ... and of whether its loop advances alone or in a group of loops in lockstep (three feature variants of
    # main_eval.m:19 -- with FAST above that is its whole list --, three runs to a group)
    feats3 = ("Harris", "SURF", "BRISK")
    alone = vo.main_eval(sequences, feats=feats3, repetition=1, save_dir=str(tmp_path / "w4"), seed=9, ctx=ctx)
    grouped = vo.main_eval(sequences, feats=feats3, repetition=1, save_dir=str(tmp_path / "w5"), seed=9, ctx=ctx, group=3)
    for k in range(6):
        for key in ("est_trans", "est_rot", "num_X", "num_C", "num_new", "errs", "frame_ct", "frame_init"):
            np.testing.assert_array_equal(alone[k][key], grouped[k][key], err_msg="grouped %d %s" % (k, key))
        assert grouped[k]["fps_ct"] > 0 and grouped[k]["toc_ct"] > 0 and grouped[k]["feat"] == feats3[k % 3]
    assert len(glob.glob(str(tmp_path / "w5" / "*_res.mat"))) == 6
    # ... and of whether the groups run one after the other or two at a time (a thread and a context per group)
    two = vo.main_eval(sequences, feats=feats3, repetition=1, save_dir=str(tmp_path / "w6"), seed=9, ctx=ctx, group=2,
                       concurrent=2)
    for k in range(6):
        for key in ("est_trans", "est_rot", "num_X", "num_C", "num_new", "errs", "frame_ct", "frame_init"):
            np.testing.assert_array_equal(alone[k][key], two[k][key], err_msg="two groups in flight %d %s" % (k, key))
    assert len(glob.glob(str(tmp_path / "w6" / "*_res.mat"))) == 6
    # the two feature variants are different runs
    assert not np.array_equal(whole[0]["num_C"], whole[1]["num_C"])
    for d in ("w1", "w2", "w3"):
        files = sorted(glob.glob(str(tmp_path / d / "*_res.mat")))
        assert len(files) == 4
        back = vo.load_results(files[0])
        assert {"errs", "S", "total_time", "frame_init", "frame_ct", "toc_param", "toc_bootstrap", "toc_ct",
                "fps_ct"} <= set(back)


def test_main_eval_records_a_failed_run(vo, ctx, tmp_path):
    """A run in which the scene disappears (blank frames: every track is lost) ends where the reference's script
    stops with an error (main_demo.m:389-391).  on_error="raise" passes that on; on_error="record" marks the run
    failed, goes on with the grid, and the failed run has given its device state and pyramid slots back."""
    K = np.array([[400.0, 0, 320], [0, 400.0, 120], [0, 0, 1]])
    scene = vo.synth.CorridorScene(seed=100, K=K, width=640, height=240)
    poses = vo.synth.corridor_trajectory(10, step=0.5)
    good = [scene.render(T) for T in poses]
    bad = [f.copy() for f in good]
    for f in bad[6:]:
        f[:] = 128
    gt = np.array([[T[0, 3], T[2, 3]] for T in poses])
    sequences = [dict(name="blank", images=bad, K=K, ground_truth=gt, hyper_paras=vo.default_hyper_paras("kitti")),
                 dict(name="corridor", images=good, K=K, ground_truth=gt, hyper_paras=vo.default_hyper_paras("kitti"))]
    with pytest.raises((RuntimeError, vo.VohipError)):
        vo.main_eval(sequences, feats=("Harris",), repetition=1, seed=9, ctx=ctx)
    for _ in range(3):   # (more failed runs than the context has spare pyramid slots for, were they not released)
        recs = vo.main_eval(sequences, feats=("Harris",), repetition=2, seed=9, ctx=ctx, on_error="record",
                            save_dir=str(tmp_path / "runs"))
        assert [r["failed"] for r in recs] == [True, True, False, False]
        assert recs[0]["frames"] == 0 and recs[0]["est_trans"].shape == (3, 0) and recs[0]["sequence"] == "blank"
        assert recs[2]["frames"] == 8 and recs[2]["fps_ct"] > 0 and recs[2]["errs"].shape[0] == 3
    # the same grid with the four loops in one group: the failed runs stop, the others' records do not change
    grouped = vo.main_eval(sequences, feats=("Harris",), repetition=2, seed=9, ctx=ctx, on_error="record", group=4)
    assert [r["failed"] for r in grouped] == [True, True, False, False]
    for k in (2, 3):
        for key in ("est_trans", "est_rot", "num_X", "num_C", "num_new", "errs"):
            np.testing.assert_array_equal(recs[k][key], grouped[k][key], err_msg="%d %s" % (k, key))
    with pytest.raises((RuntimeError, vo.VohipError)):
        vo.main_eval(sequences, feats=("Harris",), repetition=2, seed=9, ctx=ctx, group=4)


def test_run_dataset_reads_a_parking_layout_and_runs_main_demo(vo, ctx, tmp_path):
    """main_demo.m with ds = 2 on a directory in that dataset's layout (images/img_%05d.png in colour, K.txt,
    poses.txt; main_demo.m:100-116, :199-214, :360-362), written here from the rendered corridor: the run is the one
    run_sequence makes from the same frames, with the scripts' SURF setting of that branch."""
    import os
    K = np.array([[400.0, 0, 320], [0, 400.0, 120], [0, 0, 1]])
    scene = vo.synth.CorridorScene(K=K, width=640, height=240)
    poses = vo.synth.corridor_trajectory(14, step=0.5)
    frames = [scene.render(T) for T in poses]
    root = str(tmp_path / "data")
    os.makedirs(os.path.join(root, "parking", "images"))
    for i, f in enumerate(frames):
        vo.datasets.write_png(os.path.join(root, "parking", "images", "img_%05d.png" % i), np.dstack([f, f, f]))
    np.savetxt(os.path.join(root, "parking", "K.txt"), K)
    np.savetxt(os.path.join(root, "parking", "poses.txt"), np.array([T[:3].reshape(-1) for T in poses]))
    out = vo.run_dataset(2, root, last_frame=13, save_dir=str(tmp_path / "res"), rng=np.random.default_rng(5), ctx=ctx)
    assert out["frame_ct"] == 11 and out["frame_init"] == 3 and out["S"].est_trans.shape == (3, 12)
    assert os.path.basename(out["path"]).startswith("dataset@parking_feat@SURF") and out["path"].endswith("_res.mat")
    assert out["errs"].shape[0] == 3 and np.all(out["errs"] < 0.5)
    hp = vo.dataset_config(2).hyper_paras
    assert (hp.feature_extract, hp.feature_extract_options) == ("SURF", ["MetricThreshold", 100])
    gt_xz = np.array([[T[0, 3], T[2, 3]] for T in poses])
    ref = vo.run_sequence(frames, K, hp, ground_truth=gt_xz, bootstrap_frames=(0, 2), dataset="parking",
                          rng=np.random.default_rng(5), ctx=ctx)
    np.testing.assert_array_equal(out["S"].est_trans, ref["S"].est_trans)
    np.testing.assert_array_equal(out["errs"], ref["errs"])
    # the scripts' commented alternative (main_demo.m:27-28) as an override
    alt = vo.run_dataset("parking", root, last_frame=13, rng=np.random.default_rng(5), ctx=ctx,
                         hyper_paras={"feature_extract": "Harris", "feature_extract_options": ["MinQuality", 1e-4]})
    assert alt["frame_ct"] == 11 and np.all(alt["errs"] < 0.5)
    # the same directory as an entry of main_eval's sequence list (main_eval.m:16-23)
    item = vo.eval_sequence(2, root, last_frame=13)
    recs = vo.main_eval([item], feats=("Harris", "FAST"), repetition=1, seed=3, ctx=ctx)
    assert len(recs) == 2 and all(not r.get("failed", False) for r in recs)
