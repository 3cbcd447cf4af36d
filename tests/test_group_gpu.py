"""A group of loops advanced in lockstep (vohip_group_* behind VOGroupDevice) against each sequence's own
device-resident loop (VOPipelineDevice): same images, same generators.
Bar: bit-identical -- counts, ids, poses, final state, generator positions.  A sequence of a group reads
nothing of another, and every kernel it runs is the single loop's kernel with a sequence index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 640, 240
K = np.array([[400.0, 0, 320], [0, 400.0, 120], [0, 0, 1]])


@pytest.fixture(scope="module")
def worlds(vo):
    """Three different corridors / trajectories (different seeds, step lengths), rendered once."""
    out = []
    for seed, step, n in ((1234, 0.5, 10), (77, 0.4, 8), (5, 0.6, 7)):
        scene = vo.synth.CorridorScene(K=K, width=W, height=H, seed=seed)
        poses = vo.synth.corridor_trajectory(n, step=step)
        out.append((scene, poses, [scene.render(T) for T in poses]))
    return out


def _start(oracle, scene, poses, frames, thin=None, seed=0):
    ab, _, _ = oracle.detect_corners(frames[0], min_quality=2e-4, border=16)
    xy = (ab + 1.0).astype(np.float64)
    if thin:
        pick = np.sort(np.random.default_rng(seed).permutation(len(xy))[:thin])
        xy = xy[pick]
    return scene.backproject(poses[0], xy), xy


def _single(vo, ctx, hp, frames, X0, xy, T0, seed, first_hypotheses=None, feat=None):
    import copy
    hp = copy.copy(hp)
    if feat:
        hp.feature_extract, hp.feature_extract_options = feat
    dev = vo.VOPipelineDevice(K, frames[0], X0, xy, T0, hyper_paras=hp, rng=np.random.default_rng(seed), ctx=ctx)
    if first_hypotheses:
        dev.first_hypotheses = first_hypotheses
    Ts, err = [], None
    try:
        for i in range(1, len(frames)):
            Ts.append(dev.process_frame(frames[i], next_image=frames[i + 1] if i + 1 < len(frames) else None))
    except Exception as e:   # noqa: BLE001  (a run that stops where the reference stops)
        err = e
    S = dev.S if err is None else None
    out = dict(T=np.array(Ts), num_X=list(dev.num_X), num_C=list(dev.num_C), num_new=list(dev.num_new), S=S,
               rng=dev.rng.bit_generator.state, rng_map=dev.rng_map.bit_generator.state, err=err,
               n_iter=dev.last_pose.n_iter if dev.last_pose else None)
    dev.close()
    return out


def _compare(group, i, ref):
    q = group.seqs[i]
    et, er = group.trajectory(i)
    T = ref["T"]
    assert et.shape[1] == len(T) + 1, (i, et.shape, len(T))
    np.testing.assert_array_equal(et[:, 1:], T[:, :3, 3].T, err_msg="translation of sequence %d" % i)
    np.testing.assert_array_equal(er[:, 1:], np.array([t[:3, :3].reshape(-1, order="F") for t in T]).T)
    assert q.num_X == ref["num_X"] and q.num_C == ref["num_C"] and q.num_new == ref["num_new"], i
    assert q.rng.bit_generator.state == ref["rng"], i
    if ref["S"] is not None:
        # (the single loop has drawn the following frame's trial uniforms ahead, so its second generator is
        # one block further; that both consumed the same draws shows in the identical candidates below)
        S = group.S(i)
        for key in ("P", "X", "C", "F", "T"):
            np.testing.assert_array_equal(getattr(S, key), getattr(ref["S"], key), err_msg="%s of sequence %d" % (key, i))


@pytest.mark.parametrize("n_seq", [2, 5])
def test_group_is_bit_identical_to_single_loops(vo, ctx, oracle, worlds, n_seq):
    """S sequences with different images, lengths, start sets and seeds; a group slot more than needed stays idle."""
    hp = vo.default_hyper_paras("kitti")
    specs = []
    for s in range(n_seq):
        scene, poses, frames = worlds[s % len(worlds)]
        frames = frames[:len(frames) - (s // len(worlds))]          # (different lengths)
        X0, xy = _start(oracle, scene, poses, frames, thin=None if s % 2 == 0 else 300 + 50 * s, seed=s)
        specs.append((frames, X0, xy, poses[0], 100 + s))
    refs = [_single(vo, ctx, hp, *sp) for sp in specs]
    group = vo.VOGroupDevice(K, (H, W), n_seq + 1, max(len(sp[0]) for sp in specs), hyper_paras=hp, ctx=ctx)
    for s, (frames, X0, xy, T0, seed) in enumerate(specs):
        group.add_sequence(s, frames, X0, xy, T0, rng=np.random.default_rng(seed))
    pairs = group.run()
    assert pairs == sum(len(sp[0]) - 1 for sp in specs)
    assert group.frames == max(len(sp[0]) for sp in specs) - 1
    for s in range(n_seq):
        assert group.seqs[s].error is None
        _compare(group, s, refs[s])
    # the same group object once more, fewer sequences, other seeds: a new run starts with the first reset
    for s in (1, 0):
        frames, X0, xy, T0, seed = specs[s]
        group.add_sequence(s, frames, X0, xy, T0, rng=np.random.default_rng(seed))
    group.run()
    for s in (0, 1):
        _compare(group, s, refs[s])
    group.close()


def test_group_retries_one_sequence_alone(vo, ctx, oracle, worlds):
    """Too few hypotheses for the adaptive bound (8 supplied, at least 99 wanted): every sequence answers
    VOHIP_SAMPLES_EXHAUSTED in every frame and is retried on its own, after the others' look-ahead trackers
    were enqueued; results as the single loops' (which retry the same way)."""
    hp = vo.default_hyper_paras("kitti")
    specs = []
    for s in range(3):
        scene, poses, frames = worlds[s]
        frames = frames[:6]
        X0, xy = _start(oracle, scene, poses, frames, thin=400, seed=10 + s)
        specs.append((frames, X0, xy, poses[0], 40 + s))
    refs = [_single(vo, ctx, hp, *sp, first_hypotheses=8) for sp in specs]
    group = vo.VOGroupDevice(K, (H, W), 3, 6, hyper_paras=hp, ctx=ctx)
    group.first_hypotheses = 8
    for s, (frames, X0, xy, T0, seed) in enumerate(specs):
        group.add_sequence(s, frames, X0, xy, T0, rng=np.random.default_rng(seed))
    group.run()
    for s in range(3):
        assert group.seqs[s].error is None
        assert group.seqs[s].last_pose.n_iter == refs[s]["n_iter"] >= 99
        _compare(group, s, refs[s])
    group.close()


def test_group_feature_variants_and_a_sequence_that_stops(vo, ctx, oracle, worlds):
    """Harris, MinEigen, FAST, SURF and BRISK sequences in one group (main_eval.m:165-174: the loop only reads the corner lists),
    and a sequence whose localisation finds no pose (landmarks that fit no camera): it stops with the single
    loop's error while the others run on unchanged."""
    hp = vo.default_hyper_paras("kitti")
    feats = [("Harris", ["MinQuality", 2e-4]), ("MinEigen", ["MinQuality", 2e-3]), ("Harris", ["MinQuality", 2e-4]),
             ("FAST", ["MinContrast", 0.11]), ("SURF", ["MetricThreshold", 200]), ("BRISK", ["MinQuality", 2e-3])]
    specs = []
    for s in range(6):
        scene, poses, frames = worlds[s % 3]
        frames = frames[:6]
        X0, xy = _start(oracle, scene, poses, frames, thin=500, seed=20 + s)
        if s == 2:
            X0 = np.random.default_rng(3).uniform(-5, 5, X0.shape) + np.array([0, 0, 20.0])
        specs.append((frames, X0, xy, poses[0], 60 + s))
    refs = [_single(vo, ctx, hp, *sp, feat=feats[s]) for s, sp in enumerate(specs)]
    assert refs[2]["err"] is not None and refs[0]["err"] is None and refs[1]["err"] is None
    assert refs[3]["err"] is None and refs[4]["err"] is None and refs[5]["err"] is None
    group = vo.VOGroupDevice(K, (H, W), 6, 6, hyper_paras=hp, ctx=ctx)
    for s, (frames, X0, xy, T0, seed) in enumerate(specs):
        fo = dict(zip(feats[s][1][::2], feats[s][1][1::2]))
        group.add_sequence(s, frames, X0, xy, T0, rng=np.random.default_rng(seed), feature_extract=feats[s][0],
                           min_quality=fo.get("MinQuality"), min_contrast=fo.get("MinContrast"),
                           metric_threshold=fo.get("MetricThreshold"))
    group.run()
    assert group.seqs[2].error is not None and type(group.seqs[2].error) is type(refs[2]["err"])
    assert len(group.seqs[2].est_trans) == len(refs[2]["T"]) + 1
    for s in (0, 1, 3, 4, 5):
        assert group.seqs[s].error is None
        _compare(group, s, refs[s])
    group.close()


def test_group_slots_across_the_64_bit_mask(vo, ctx, oracle, worlds):
    """A group of 40 slots of which only slots 0, 31, 32 and 39 hold a sequence: a launch serves the set bits of
    a 64-bit mask, and a workgroup finds its sequence as the position of the blockIdx.y-th set bit -- bits on both
    sides of the 32-bit boundary, idle slots in between.  Every sequence as its own single loop, bit for bit."""
    hp = vo.default_hyper_paras("kitti")
    used = (0, 31, 32, 39)
    specs = {}
    for k, slot in enumerate(used):
        scene, poses, frames = worlds[k % 3]
        frames = frames[:5 + (k % 2)]
        X0, xy = _start(oracle, scene, poses, frames, thin=350, seed=30 + k)
        specs[slot] = (frames, X0, xy, poses[0], 80 + k)
    refs = {slot: _single(vo, ctx, hp, *sp) for slot, sp in specs.items()}
    group = vo.VOGroupDevice(K, (H, W), 40, 6, hyper_paras=hp, ctx=ctx)
    for slot, (frames, X0, xy, T0, seed) in specs.items():
        group.add_sequence(slot, frames, X0, xy, T0, rng=np.random.default_rng(seed))
    group.run()
    for slot in used:
        assert group.seqs[slot].error is None
        _compare(group, slot, refs[slot])
    assert all(group.seqs[s] is None for s in range(40) if s not in used)
    group.close()
