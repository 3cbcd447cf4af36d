"""The device-resident frame loop (vohip_frame_* behind VOPipelineDevice) against the stage-by-stage
loop (VOPipeline) and the CPU restatement, same sequence and RNG seed.
Bar: identical landmark / candidate counts and landmark ids every frame; poses, landmarks and
keypoints within 1e-8 of the stage-by-stage GPU loop (same kernels, the only difference is the
closed-form inverse of the RANSAC pose before refinement) and within the end-to-end bar
(1e-5 m / rad) of the CPU restatement."""
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


def _start(vo, oracle, sequence):
    scene, poses, frames = sequence
    ab, _, _ = oracle.detect_corners(frames[0], min_quality=2e-4, border=16)
    xy = (ab + 1.0).astype(np.float64)
    return scene.backproject(poses[0], xy), xy


def test_uniform_to_index_maps(vo):
    """The two maps the device applies to the caller's uniforms, against brute-force definitions."""
    rng = np.random.default_rng(0)
    for n in (8, 9, 31, 500):
        u = rng.random((300, 8))
        got = vo.octets_from_uniforms(u, n)
        for row_u, row in zip(u, got):
            pool = list(range(n))
            want = [pool.pop(min(int(x * len(pool)), len(pool) - 1)) for x in row_u]
            assert list(row) == want
    for n in (3, 4, 50):
        u = rng.random((300, 3))
        got = vo.triples_from_uniforms(u, n)
        for row_u, row in zip(u, got):
            pool = list(range(n))
            want = [pool.pop(min(int(x * len(pool)), len(pool) - 1)) for x in row_u]
            assert list(row) == want


@pytest.mark.parametrize("n_start,spoil", [(37, 0), (61, 17), (130, 48), (529, 200)])
def test_device_loop_with_few_landmarks_and_lost_tracks(vo, ctx, oracle, sequence, n_start, spoil):
    """The frame's first hypotheses are solved in the launch that also selects the valid tracks; the solve
    workgroups find the rows of their samples from the valid flags themselves (16 flags per mask).  Sizes
    that are no multiples of 16, fewer landmarks than a mask group holds lanes, and a run of `spoil`
    landmarks whose keypoints are wrong by 12 px (lost by the tracker's forward-backward test or rejected as
    outliers: holes in the flags): poses, counts and generator state as the stage-by-stage loop gives them."""
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    pick = np.sort(np.random.default_rng(n_start).permutation(len(xy))[:n_start])
    X0, xy = X0[pick], xy[pick].copy()
    xy[5:5 + spoil] += 12.0
    hp = vo.default_hyper_paras("kitti")
    ref = vo.VOPipeline(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(2), ctx=ctx)
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(2), ctx=ctx)
    for i in range(1, 5):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i], next_image=frames[i + 1])
        assert dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1], i
        assert dev.last_pose.n_tracked <= (n_start if i == 1 else 10 ** 6)
        np.testing.assert_allclose(Td, Tr, rtol=0, atol=1e-8)
    assert dev.rng.bit_generator.state == ref.rng.bit_generator.state
    np.testing.assert_array_equal(dev.S.X[3], ref.S.X[3])
    ref.close()
    dev.close()


@pytest.mark.parametrize("feat,on_device,ahead", [("Harris", False, True), ("Harris", True, True), ("Harris", False, False),
                                                  ("MinEigen", True, True), ("FAST", False, True)])
def test_prepared_sequence_changes_nothing(vo, ctx, oracle, sequence, feat, on_device, ahead):
    """vohip_frame_prepare: the stored sequence's images on the GPU once, all pyramids and corner lists from
    batched launches (FAST: the pyramids only).  Poses, counts and the final state are those of the loop that
    transfers, builds and detects frame by frame, bit for bit; so is a frame that was not prepared."""
    import torch
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    hp.feature_extract = feat
    hp.feature_extract_options = {"Harris": ["MinQuality", 2e-4], "MinEigen": ["MinQuality", 2e-3],
                                  "FAST": ["MinContrast", 0.11]}[feat]
    imgs = [torch.from_numpy(f).cuda() for f in frames] if on_device else frames
    runs = []
    for prepared in (False, True):
        dev = vo.VOPipelineDevice(K, imgs[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(5), ctx=ctx)
        if prepared:
            dev.prepare(imgs[1:-1])            # (the last frame is left to the unprepared path)
        Ts = []
        for i in range(1, len(imgs)):
            nxt = imgs[i + 1] if ahead and i + 1 < len(imgs) else None
            Ts.append(dev.process_frame(imgs[i], next_image=nxt))
        runs.append((np.array(Ts), list(dev.num_X), list(dev.num_C), list(dev.num_new), dev.S,
                     dev.last_counts.n_detected, dev.rng.bit_generator.state))
        dev.close()
    a, b = runs
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1:4] == b[1:4] and a[5] == b[5] and a[6] == b[6]
    for key in ("P", "X", "C", "F", "T"):
        np.testing.assert_array_equal(getattr(a[4], key), getattr(b[4], key), err_msg=key)


@pytest.mark.parametrize("on_device,ahead", [(False, True), (True, True), (True, False)])
def test_prepare_in_chunks_and_release_mid_run(vo, ctx, oracle, sequence, on_device, ahead):
    """A long stored sequence is prepared a chunk at a time, and the prepared data may be released while the loop
    runs: the pyramid slots of the loop still show pyramids inside the block that a new prepare (or prepare([]))
    frees, so the library moves those pyramids into the slots' own buffers first.  Bit for bit the unprepared run."""
    import torch
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    imgs = [torch.from_numpy(f).cuda() for f in frames] if on_device else frames
    n = len(imgs)

    def run(plan):
        """plan: {frame index: list of images to prepare right before that frame is begun}"""
        dev = vo.VOPipelineDevice(K, imgs[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(5), ctx=ctx)
        Ts = []
        for i in range(1, n):
            if i in plan:
                dev.prepare(plan[i])
            Ts.append(dev.process_frame(imgs[i], next_image=imgs[i + 1] if ahead and i + 1 < n else None))
        out = (np.array(Ts), list(dev.num_X), list(dev.num_C), list(dev.num_new), dev.S, dev.rng.bit_generator.state)
        dev.close()
        return out

    plain = run({})
    # two consecutive chunks (the second prepared while the look-ahead of frame 3 already shows image 4 of the
    # first), then the rest unprepared after a release in the middle of the run
    chunks = run({1: imgs[1:5], 4: imgs[4:7], 6: []})
    # released right after the first prepared frames, prepared again later
    again = run({1: imgs[1:n], 3: [], 5: imgs[5:n]})
    for other in (chunks, again):
        np.testing.assert_array_equal(plain[0], other[0])
        assert plain[1:4] == other[1:4] and plain[5] == other[5]
        for key in ("P", "X", "C", "F", "T"):
            np.testing.assert_array_equal(getattr(plain[4], key), getattr(other[4], key), err_msg=key)


def test_prepare_finds_images_it_had_to_convert_and_warns_when_it_is_never_hit(vo, ctx, oracle, sequence):
    """The library recognises a prepared image by its address.  Images in the other memory order are copied by
    prepare(): later calls with the caller's own objects must find those copies (same results as the unprepared
    loop).  A loop that is fed copies of the prepared images never hits them: that earns a warning."""
    import warnings
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    fortran = [np.asfortranarray(f) for f in frames]      # (the loop below is row-major: every one is converted)

    def run(images, prepared, feed):
        dev = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(5), ctx=ctx)
        if prepared:
            dev.prepare(images[1:])
        fed = [feed(im) for im in images]
        Ts = [dev.process_frame(fed[i], next_image=fed[i + 1] if i + 1 < len(fed) else None) for i in range(1, 6)]
        hits = getattr(dev, "_prepared_hits", 0)
        dev.close()
        return np.array(Ts), hits

    plain, _ = run(frames, False, lambda im: im)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        conv, hits = run(fortran, True, lambda im: im)
    np.testing.assert_array_equal(plain, conv)
    assert hits >= 5
    with pytest.warns(RuntimeWarning, match="none of the"):
        copies, hits = run(frames, True, lambda im: im.copy())
    np.testing.assert_array_equal(plain, copies)
    assert hits == 0


def test_device_loop_matches_stage_by_stage_loop(vo, ctx, oracle, sequence):
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    ref = vo.VOPipeline(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(1), ctx=ctx)
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(1), ctx=ctx)
    for i in range(1, len(frames)):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i])
        assert dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1], i
        assert dev.num_new == ref.S.num_new, i
        np.testing.assert_allclose(Td, Tr, rtol=0, atol=1e-8)
        assert np.linalg.norm(Td[:3, 3] - poses[i][:3, 3]) < 0.25, i
    S, R = dev.S, ref.S
    np.testing.assert_array_equal(S.X[3], R.X[3])
    np.testing.assert_allclose(S.X[:3], R.X[:3], rtol=1e-6, atol=1e-6)   # low-parallax triangulation amplifies the 1e-10 pose difference
    np.testing.assert_allclose(S.P, R.P, rtol=0, atol=1e-9)
    np.testing.assert_allclose(S.C, R.C, rtol=0, atol=1e-9)
    np.testing.assert_allclose(S.F, R.F, rtol=0, atol=1e-9)
    np.testing.assert_allclose(S.T, R.T[:, :R.C.shape[1]], rtol=0, atol=1e-8)
    np.testing.assert_allclose(S.est_trans, R.est_trans, rtol=0, atol=1e-8)
    np.testing.assert_allclose(S.est_rot, R.est_rot, rtol=0, atol=1e-8)
    # both generators consumed the same number of draws
    assert dev.rng.bit_generator.state == ref.rng.bit_generator.state
    ref.close()
    dev.close()


def test_device_loop_matches_cpu_restatement(vo, ctx, oracle, sequence):
    from oracle import pipeline_oracle as po
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    ref = po.OraclePipeline(K, frames[0], X0, xy, poses[0], hp, 3, 2, np.random.default_rng(7), vo.draw_triples,
                            vo.draw_octets)
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    for i in range(1, 7):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i])
        assert dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1], i
        assert np.linalg.norm(Tr[:3, 3] - Td[:3, 3]) < 1e-5, i
        dR = Tr[:3, :3].T @ Td[:3, :3]
        assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-5, i
    np.testing.assert_array_equal(dev.S.X[3], ref.S.X[3])
    dev.close()


def test_device_loop_with_fast_detector_matches_cpu_restatement(vo, ctx, oracle, sequence):
    """The 'FAST' feature variant of main_eval.m:172-173 ({'MinContrast', 0.11}) through the whole loop:
    bootstrap points and every frame's candidates come from the FAST detector.  Bar as for Harris: the same
    counts and landmark ids every frame, poses within 1e-5 m / rad of the CPU restatement; and the look-ahead
    form bit-identical to the plain one."""
    from oracle import pipeline_oracle as po
    scene, poses, frames = sequence
    hp = vo.default_hyper_paras("kitti")
    hp.feature_extract = "FAST"
    hp.feature_extract_options = ["MinContrast", 0.11]
    ab, _, total = oracle.detect_fast(frames[0], min_contrast=0.11, min_quality=0.1, border=16)
    assert total > 200
    xy = (ab + 1.0).astype(np.float64)
    X0 = scene.backproject(poses[0], xy)
    ref = po.OraclePipeline(K, frames[0], X0, xy, poses[0], hp, 3, 2, np.random.default_rng(7), vo.draw_triples,
                            vo.draw_octets)
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    ahead = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    for i in range(1, 7):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i])
        Ta = ahead.process_frame(frames[i], next_image=frames[i + 1])
        np.testing.assert_array_equal(Ta, Td)
        assert dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1], i
        assert dev.last_counts.n_detected == ahead.last_counts.n_detected > 100
        assert np.linalg.norm(Tr[:3, 3] - Td[:3, 3]) < 1e-5, i
        dR = Tr[:3, :3].T @ Td[:3, :3]
        assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-5, i
        assert np.linalg.norm(Td[:3, 3] - poses[i][:3, 3]) < 0.25, i
    np.testing.assert_array_equal(dev.S.X[3], ref.S.X[3])
    np.testing.assert_array_equal(dev.S.C, ahead.S.C)
    assert dev.num_C[-1] > 0
    dev.close()
    ahead.close()


def test_device_loop_with_surf_detector_matches_cpu_restatement(vo, ctx, oracle, sequence):
    """The 'SURF' feature variant, the reference's default (main_demo.m:25; main_eval.m:166-167:
    {'MetricThreshold', 200}), through the whole loop: the starting points and every frame's candidates come from
    the SURF detector (sub-pixel, several octaves).  Bar as for Harris: the same counts and landmark ids every
    frame, poses within 1e-5 m / rad of the CPU restatement; the look-ahead and the prepared form bit-identical
    to the plain one."""
    from oracle import pipeline_oracle as po
    scene, poses, frames = sequence
    hp = vo.default_hyper_paras("kitti")
    hp.feature_extract = "SURF"
    hp.feature_extract_options = ["MetricThreshold", 200]
    ab, _, _, total = oracle.detect_surf(frames[0], 200.0, 16)
    assert total > 200
    xy = (ab + 1.0).astype(np.float64)
    X0 = scene.backproject(poses[0], xy)
    ref = po.OraclePipeline(K, frames[0], X0, xy, poses[0], hp, 3, 2, np.random.default_rng(7), vo.draw_triples,
                            vo.draw_octets)
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    ahead = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    stored = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    stored.prepare(frames[1:8])
    for i in range(1, 7):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i])
        Ta = ahead.process_frame(frames[i], next_image=frames[i + 1])
        Ts = stored.process_frame(frames[i], next_image=frames[i + 1])
        np.testing.assert_array_equal(Ta, Td)
        np.testing.assert_array_equal(Ts, Td)
        assert dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1], i
        assert dev.last_counts.n_detected == ahead.last_counts.n_detected == stored.last_counts.n_detected > 100
        assert np.linalg.norm(Tr[:3, 3] - Td[:3, 3]) < 1e-5, i
        dR = Tr[:3, :3].T @ Td[:3, :3]
        assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-5, i
        assert np.linalg.norm(Td[:3, 3] - poses[i][:3, 3]) < 0.25, i
    np.testing.assert_array_equal(dev.S.X[3], ref.S.X[3])
    np.testing.assert_array_equal(dev.S.C, ahead.S.C)
    np.testing.assert_array_equal(dev.S.C, stored.S.C)
    assert dev.num_C[-1] > 0
    for d in (dev, ahead, stored):
        d.close()


def test_device_loop_with_brisk_detector_matches_cpu_restatement(vo, ctx, oracle, sequence):
    """The 'BRISK' feature variant of main_eval.m:170-171 ({'MinQuality', 2e-3}) through the whole loop: keypoints of
    eight scale-space layers as starting points and candidates.  Bar as for the other detectors; the look-ahead and
    the prepared form bit-identical to the plain one."""
    from oracle import pipeline_oracle as po
    scene, poses, frames = sequence
    hp = vo.default_hyper_paras("kitti")
    hp.feature_extract = "BRISK"
    hp.feature_extract_options = ["MinQuality", 2e-3]
    ab, _, _, total = oracle.detect_brisk(frames[0], 0.2, 2e-3, 4, 16)
    assert total > 200
    xy = (ab + 1.0).astype(np.float64)
    X0 = scene.backproject(poses[0], xy)
    ref = po.OraclePipeline(K, frames[0], X0, xy, poses[0], hp, 3, 2, np.random.default_rng(7), vo.draw_triples,
                            vo.draw_octets)
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    ahead = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    stored = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(7), ctx=ctx)
    stored.prepare(frames[1:8])
    for i in range(1, 7):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i])
        Ta = ahead.process_frame(frames[i], next_image=frames[i + 1])
        Ts = stored.process_frame(frames[i], next_image=frames[i + 1])
        np.testing.assert_array_equal(Ta, Td)
        np.testing.assert_array_equal(Ts, Td)
        assert dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1], i
        assert dev.last_counts.n_detected == ahead.last_counts.n_detected == stored.last_counts.n_detected > 100
        assert np.linalg.norm(Tr[:3, 3] - Td[:3, 3]) < 1e-5, i
        dR = Tr[:3, :3].T @ Td[:3, :3]
        assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-5, i
        assert np.linalg.norm(Td[:3, 3] - poses[i][:3, 3]) < 0.25, i
    np.testing.assert_array_equal(dev.S.X[3], ref.S.X[3])
    np.testing.assert_array_equal(dev.S.C, ahead.S.C)
    np.testing.assert_array_equal(dev.S.C, stored.S.C)
    assert dev.num_C[-1] > 0
    for d in (dev, ahead, stored):
        d.close()


def test_device_loop_accepts_device_images_and_reports_errors(vo, ctx, oracle, sequence):
    import torch
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    a = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(3), ctx=ctx)
    d0 = torch.from_numpy(frames[0]).cuda()
    b = vo.VOPipelineDevice(K, d0, X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(3), ctx=ctx)
    for i in range(1, 4):
        Ta = a.process_frame(frames[i])
        Tb = b.process_frame(torch.from_numpy(frames[i]).cuda())
        np.testing.assert_array_equal(Ta, Tb)
    a.close()
    b.close()
    # a blank image loses every track: the reference stops at main_demo.m:389-391, so does this
    c = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(3), ctx=ctx)
    with pytest.raises((RuntimeError, vo.VohipError)):
        c.process_frame(np.full_like(frames[0], 127))
    c.close()
    # a frame prepared ahead must be the next one processed
    d = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(3), ctx=ctx)
    d.process_frame(frames[1], next_image=frames[2])
    with pytest.raises(ValueError):
        d.process_frame(frames[3])
    d.close()
    # capacity is checked
    with pytest.raises(vo.VohipError):
        vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, ctx=ctx, cap_landmarks=64)


def test_overflow_stops_the_loop(vo, ctx, oracle, sequence):
    """A frame that finds more candidates than cap_candidates holds drops the surplus, which the reference
    never does (update_candidate.m grows S.C): the call reports VOHIP_CAPACITY_EXCEEDED and the loop then
    refuses every further frame."""
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    e = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(3), ctx=ctx,
                            cap_candidates=64)
    with pytest.raises(vo.VohipError) as err:
        e.process_frame(frames[1])
    assert err.value.status == vo._lib.VOHIP_CAPACITY_EXCEEDED
    with pytest.raises(vo.VohipError) as err:
        e.process_frame(frames[2])
    assert "vohip_frame_reset" in str(err.value)
    e.close()
    # a detector's options are checked when the loop is made, not at the first frame that reaches the detector
    import copy
    for feat, opts in (("BRISK", ["NumOctaves", 7]), ("FAST", ["MinContrast", 1.5]), ("SURF", ["MetricThreshold", -3.0]),
                       ("Harris", ["MinQuality", 2.0])):
        bad = copy.copy(hp)
        bad.feature_extract, bad.feature_extract_options = feat, opts
        with pytest.raises(vo.VohipError):
            vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=bad, rng=np.random.default_rng(3), ctx=ctx)


@pytest.mark.parametrize("on_device", [False, True, "pinned"])
def test_preparing_the_next_frame_ahead_changes_nothing(vo, ctx, oracle, sequence, on_device):
    """vohip_frame_step_ahead: with the next image in hand its pyramid and corners are prepared beside
    the current frame's localisation.  Bar: every pose, count and state array bit-identical to the
    plain loop; also when the lookahead starts or stops in mid-sequence."""
    import torch
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    if on_device == "pinned":   # host images in pinned memory are fetched by a kernel, pageable ones by a copy
        keep = [torch.from_numpy(f).pin_memory() for f in frames]
        imgs = [t.numpy() for t in keep]
    else:
        imgs = [torch.from_numpy(f).cuda() for f in frames] if on_device else frames
    plain = vo.VOPipelineDevice(K, imgs[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(1), ctx=ctx)
    ahead = vo.VOPipelineDevice(K, imgs[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(1), ctx=ctx)
    last = len(frames) - 1
    for i in range(1, len(frames)):
        Tp = plain.process_frame(imgs[i])
        # frame 4 is processed without lookahead: frame 5 then starts from scratch again
        nxt = imgs[i + 1] if i < last and i != 4 else None
        Ta = ahead.process_frame(imgs[i], next_image=nxt)
        np.testing.assert_array_equal(Ta, Tp)
        assert (ahead.num_X[-1], ahead.num_C[-1]) == (plain.num_X[-1], plain.num_C[-1]), i
        for k in ("n_detected", "n_kept", "n_candidates_tracked", "n_candidates_inliers", "num_new"):
            assert getattr(ahead.last_counts, k) == getattr(plain.last_counts, k), (i, k)
        assert ahead.last_pose.n_tracked == plain.last_pose.n_tracked and ahead.last_pose.n_iter == plain.last_pose.n_iter
    A, P = ahead.S, plain.S
    for k in ("X", "P", "C", "F", "T", "est_trans", "est_rot"):
        np.testing.assert_array_equal(getattr(A, k), getattr(P, k), err_msg=k)
    assert ahead.rng.bit_generator.state == plain.rng.bit_generator.state
    plain.close()
    ahead.close()


def test_one_call_entry_points_match_the_two_halves(vo, ctx, oracle, sequence):
    """vohip_frame_step / vohip_frame_step_ahead (one call per frame) against vohip_frame_begin +
    vohip_frame_finish, which VOPipelineDevice uses: same draws in, identical poses and counts out; and
    the state machine refuses a finish without a begin."""
    import ctypes as C
    lib = vo._lib.load()
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    a = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(5), ctx=ctx)
    b = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(5), ctx=ctx)
    pose, counts = vo._lib.FramePose(), vo._lib.FrameCounts()
    assert lib.vohip_frame_finish(b._h, C.byref(pose), C.byref(counts)) != 0      # nothing was begun
    n_C = 0
    for i in range(1, 6):
        Ta = a.process_frame(frames[i])
        # the same frame through the one-call entry points, with the draws process_frame makes
        u8 = b.rng_map.random((b.params.fm_trials, 8)) if n_C > 0 else None
        state = b.rng.bit_generator.state
        u3 = b.rng.random((104, 3))
        img = np.ascontiguousarray(frames[i])
        args = (vo._lib.dptr(u3), 104, vo._lib.dptr(u8) if u8 is not None else None, C.byref(pose), C.byref(counts))
        if i % 2:
            st = lib.vohip_frame_step(b._h, C.c_void_p(img.ctypes.data), 0, *args)
        else:
            st = lib.vohip_frame_step_ahead(b._h, C.c_void_p(img.ctypes.data), 0, None, 0, *args)
        assert st == 0, (i, st, vo._lib.load().vohip_last_error(ctx.handle))
        b.rng.bit_generator.state = state
        b.rng.random((pose.n_iter, 3))
        n_C = counts.n_candidates
        np.testing.assert_array_equal(np.array(pose.T_W_C).reshape(4, 4).T, Ta)
        assert (counts.n_landmarks, counts.n_candidates) == (a.num_X[-1], a.num_C[-1])
    a.close()
    b.close()


@pytest.mark.parametrize("stress", [False, True])
def test_device_loop_matches_cpu_restatement_at_kitti_shape(vo, ctx, oracle, stress):
    """The benchmark's configurations -- 1241x376 frames, KITTI camera, the following frame prepared
    ahead; about 2k tracked points, or (stress) the strongest 8192 corners at MinQuality 2e-5 with 4096
    hypotheses solved per frame -- against the CPU restatement of the loop on the same frames and
    draws.  Bar: identical counts (landmarks, candidates, RANSAC iterations) and landmark ids every
    frame; poses within 1e-5 m / 1e-5 rad."""
    from oracle import pipeline_oracle as po
    scene = vo.synth.CorridorScene()
    poses = vo.synth.corridor_trajectory(7)
    frames = [scene.render(T) for T in poses]
    assert frames[0].shape == (376, 1241)
    hp = vo.default_hyper_paras("kitti")
    if stress:
        hp.feature_extract_options = ["MinQuality", 2e-5]
        hp.r_discard_redundant = 5.0
    ab, metric, _ = oracle.detect_corners(frames[0], min_quality=hp.feature_extract_options[1], border=16)
    xy = (ab + 1.0).astype(np.float64)
    if stress:
        xy = xy[np.sort(np.argsort(-metric, kind="stable")[:8192])]
        assert 6000 < len(xy) <= 8192
    else:
        # strongest corner per 14 px cell: about 2k starting points, as the benchmark thins them
        key = (np.floor(xy[:, 1] / 14.0).astype(np.int64) << 20) + np.floor(xy[:, 0] / 14.0).astype(np.int64)
        order = np.lexsort((-metric, key))
        first = np.ones(len(order), dtype=bool)
        first[1:] = key[order][1:] != key[order][:-1]
        xy = xy[np.sort(order[first])]
        assert 1500 < len(xy) < 3000
    X0 = scene.backproject(poses[0], xy)
    ref = po.OraclePipeline(scene.K, frames[0], X0, xy, poses[0], hp, 3, 2, np.random.default_rng(11), vo.draw_triples,
                            vo.draw_octets)
    dev = vo.VOPipelineDevice(scene.K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(11), ctx=ctx)
    if stress:
        dev.first_hypotheses = 4096
    for i in range(1, 4 if stress else 6):
        Tr = ref.process_frame(frames[i])
        Td = dev.process_frame(frames[i], next_image=frames[i + 1])
        assert dev.num_X[-1] == ref.S.num_X[-1] and dev.num_C[-1] == ref.S.num_C[-1], i
        assert np.linalg.norm(Tr[:3, 3] - Td[:3, 3]) < 1e-5, i
        dR = Tr[:3, :3].T @ Td[:3, :3]
        assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-5, i
    np.testing.assert_array_equal(dev.S.X[3], ref.S.X[3])
    assert dev.rng.bit_generator.state == ref.rng.bit_generator.state
    dev.close()


@pytest.mark.parametrize("lookahead", [False, True])
def test_more_hypotheses_on_demand(vo, ctx, oracle, sequence, lookahead):
    """vohip_frame_retry: a frame that was handed fewer hypotheses than the adaptive bound wants (here 16,
    so every frame) is localised again with more of the same stream.  Bar: poses, counts and state
    bit-identical to frames that had enough from the start; with the look-ahead the following frame's
    trackers, enqueued behind the first attempt, are run again from the final lists."""
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    a = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(2), ctx=ctx)
    b = vo.VOPipelineDevice(K, frames[0], X0, xy, poses[0], hyper_paras=hp, rng=np.random.default_rng(2), ctx=ctx)
    b.first_hypotheses = 16
    last = len(frames) - 1
    for i in range(1, len(frames)):
        Ta = a.process_frame(frames[i])
        Tb = b.process_frame(frames[i], next_image=frames[i + 1] if lookahead and i < last else None)
        np.testing.assert_array_equal(Ta, Tb)
        assert b.last_pose.n_iter == a.last_pose.n_iter >= 99   # ransacLocalization.m:42,128: at least 99 iterations
        assert (a.num_X[-1], a.num_C[-1]) == (b.num_X[-1], b.num_C[-1]), i
    for k in ("X", "P", "C", "F", "T"):
        np.testing.assert_array_equal(getattr(a.S, k), getattr(b.S, k), err_msg=k)
    assert a.rng.bit_generator.state == b.rng.bit_generator.state
    a.close()
    b.close()


def test_two_loops_on_one_context_with_frames_in_flight_together(vo, ctx, oracle, sequence):
    """Two loops share a context and have their frames in flight at the same time (begin A, begin B, finish
    A, finish B), one of them preparing the next image ahead on its own streams: each loop owns the scratch
    its kernels use (candidate poses of the hypotheses, the corner detector's records), so both must give
    exactly what they give when run alone."""
    scene, poses, frames = sequence
    X0, xy = _start(vo, oracle, sequence)
    hp = vo.default_hyper_paras("kitti")
    flipped = [np.ascontiguousarray(f[:, ::-1]) for f in frames]     # another sequence: other corners, other poses

    def make(seed, imgs):
        return vo.VOPipelineDevice(K, imgs[0], X0 if imgs is frames else Xf, xy if imgs is frames else xyf, poses[0],
                                   hyper_paras=hp, rng=np.random.default_rng(seed), ctx=ctx)

    xyf = xy.copy()
    xyf[:, 0] = W + 1 - xy[:, 0]                                       # the same points in the mirrored image
    Xf = X0 * np.array([-1.0, 1.0, 1.0])
    alone_a, alone_b = make(1, frames), make(2, flipped)
    Ta = [alone_a.process_frame(frames[i], next_image=frames[i + 1]) for i in range(1, 7)]
    Tb = [alone_b.process_frame(flipped[i]) for i in range(1, 7)]
    a, b = make(1, frames), make(2, flipped)
    for i in range(1, 7):
        a.begin_frame(frames[i], next_image=frames[i + 1])
        b.begin_frame(flipped[i])
        np.testing.assert_array_equal(a.finish_frame(), Ta[i - 1])
        np.testing.assert_array_equal(b.finish_frame(), Tb[i - 1])
    for k in ("X", "P", "C", "F", "T"):
        np.testing.assert_array_equal(getattr(a.S, k), getattr(alone_a.S, k), err_msg=k)
        np.testing.assert_array_equal(getattr(b.S, k), getattr(alone_b.S, k), err_msg=k)
    assert a.num_X == alone_a.num_X and b.num_C == alone_b.num_C
    for p in (a, b, alone_a, alone_b):
        p.close()
