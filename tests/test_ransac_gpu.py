"""Parity of the HIP RANSAC path against the CPU oracle, through the C ABI.

Bar: inlier counts and masks bit-exact for identical sample triples, except threshold-boundary
ties (a point whose squared reprojection error is within TIE_BAND of 100 px^2 may flip, because
the kernel tests (x-uz)^2+(y-vz)^2 < 100 z^2 with FMAs instead of dividing first).  Poses:
1e-9 relative (they differ only through the last bits of pow/atan2/sincos)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_p3p  # noqa: E402

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TIE_BAND = 1e-6  # px^2 around the 100 px^2 threshold


def _sq_errors(kp_rc, X, K, R, t):
    pc = R @ X + t.reshape(3, 1)
    h = K @ pc
    with np.errstate(all="ignore"):
        uv = h[:2] / h[2]
    d = kp_rc[::-1] - uv
    return (d * d).sum(0)


def _score_dev(vo, ctx, kp_rc, X, K, samples):
    lib = vo._lib.load()
    dev = torch.device("cuda:0")
    d_kp = torch.from_numpy(np.ascontiguousarray(kp_rc.T)).to(dev)
    d_X = torch.from_numpy(np.ascontiguousarray(X.T)).to(dev)
    d_s = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.int32)).to(dev)
    h = samples.shape[0]
    d_c = torch.full((h, 4), -1, dtype=torch.int32, device=dev)
    Kc = vo._lib.colmajor(K)
    torch.cuda.synchronize()
    ctx.check(lib.vohip_ransac_p3p_score_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), kp_rc.shape[1],
                                              vo._lib.dptr(Kc), d_s.data_ptr(), h, d_c.data_ptr()))
    ctx.synchronize()
    Rt = np.zeros((h, 4, 12))
    ctx.check(lib.vohip_ransac_p3p_poses(ctx.handle, 0, h, vo._lib.dptr(Rt)))
    return d_c.cpu().numpy(), Rt, (d_kp, d_X)


def _assert_counts_match(counts, Rt, g_counts, kp_rc, X, K):
    """Counts equal, or explained by points inside the tie band."""
    bad = np.argwhere(counts != g_counts)
    for h, s in bad:
        R = Rt[h, s, :9].reshape(3, 3).T
        e = _sq_errors(kp_rc, X, K, R, Rt[h, s, 9:])
        near = np.sum(np.abs(e - 100.0) < TIE_BAND)
        assert abs(int(counts[h, s]) - int(g_counts[h, s])) <= near, (h, s, counts[h, s], g_counts[h, s])
    return len(bad)


def test_score_counts_match_golden(vo, ctx):
    g = np.load(os.path.join(GOLD, "ransac_n2000_h256.npz"))
    counts, Rt, _ = _score_dev(vo, ctx, g["kp_rc"], g["X"], g["K"], g["samples"])
    nbad = _assert_counts_match(counts, Rt, g["counts"], g["kp_rc"], g["X"], g["K"])
    assert nbad == 0, "no threshold ties expected on this fixture"
    # best-solution poses agree with the oracle's
    best = g["best_sol"]
    got = Rt[np.arange(len(best)), best]
    want = g["best_pose"]
    ok = np.isfinite(want).all(1)
    scale = np.maximum(1.0, np.abs(want[ok]))
    assert np.max(np.abs(got[ok] - want[ok]) / scale) < 1e-9


def test_score_counts_match_independent_p3p(vo, ctx):
    """HIP counts and poses of the 256 golden hypotheses against tests/ref_p3p.py, which shares neither
    the Ferrari closed form nor the hand-rolled complex helpers nor the division-free inlier test with
    the kernel: per hypothesis the four counts agree as a set (numpy.roots orders the roots differently),
    outside the tie band, and every finite HIP pose has a reference pose next to it."""
    g = np.load(os.path.join(GOLD, "ransac_n2000_h256.npz"))
    counts, Rt, _ = _score_dev(vo, ctx, g["kp_rc"], g["X"], g["K"], g["samples"])
    kp_uv = g["kp_rc"][::-1].T.copy()
    X = g["X"].T.copy()
    worst = 0.0
    for i, s in enumerate(g["samples"]):
        want, errs, poses = ref_p3p.hypothesis_counts(kp_uv, X, g["K"], s)
        want = sorted(want) if want else [0, 0, 0, 0]
        got = sorted(counts[i].tolist())
        near = max([int(np.sum(np.abs(e - 100.0) < TIE_BAND)) for e in errs] + [0])
        assert max(abs(a - b) for a, b in zip(got, want)) <= near, (i, got, want)
        for sol in range(4):
            if counts[i, sol] < 30:
                continue   # (poses of root pairs that are complex in exact arithmetic are ill-conditioned)
            R = Rt[i, sol, :9].reshape(3, 3).T
            t = Rt[i, sol, 9:]
            dist = min(np.abs(R - Rr).max() + np.abs(t - tr).max() / max(1.0, np.abs(tr).max()) for Rr, tr in poses)
            worst = max(worst, dist)
    assert worst < 1e-6, worst   # observed 1.6e-7: companion-matrix roots and the Ferrari closed form differ in conditioning


def test_p3p_matches_golden(vo, ctx):
    p = np.load(os.path.join(GOLD, "p3p_64.npz"))
    lib = vo._lib.load()
    out = np.zeros((64, 48))
    w = np.ascontiguousarray(p["world"].reshape(64, 9))
    b = np.ascontiguousarray(p["bearings"].reshape(64, 9))
    ctx.check(lib.vohip_p3p(ctx.handle, vo._lib.dptr(w), vo._lib.dptr(b), 64, vo._lib.dptr(out)))
    want = p["poses"]
    assert np.all(out[5] == 0)  # collinear world points, p3p.m:56-58
    both_nan = np.isnan(out) & np.isnan(want)
    scale = np.maximum(1.0, np.abs(want))
    err = np.where(both_nan, 0.0, np.abs(out - want) / scale)
    assert np.nanmax(err) < 1e-9 and not np.isnan(err).any()
    # the mirror of p3p.m's own signature
    poses = vo.p3p(p["world"][0].T, p["bearings"][0].T, ctx=ctx)
    np.testing.assert_allclose(poses.T.reshape(48), want[0], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("n,h,seed", [(2000, 512, 1), (8192, 256, 2), (777, 333, 3), (64, 100, 4), (3, 8, 5),
                                      (1025, 17, 6)])
def test_score_counts_match_oracle(vo, ctx, oracle, n, h, seed):
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(n, seed=seed)
    samples = vo.synth.sample_triples(n, h, seed=seed)
    counts, Rt, _ = _score_dev(vo, ctx, kp_rc, X, vo.synth.K_KITTI, samples)
    kp_uv = kp_rc[::-1].T.copy()
    want = np.zeros((h, 4), np.int32)
    for i in range(h):
        want[i] = oracle.ransac_hypothesis(kp_uv, X.T, vo.synth.K_KITTI, samples[i])[0]
    _assert_counts_match(counts, Rt, want, kp_rc, X, vo.synth.K_KITTI)


@pytest.mark.parametrize("n,seed,ratio", [(2000, 11, 0.7), (500, 12, 0.4), (300, 13, 0.2), (8192, 14, 0.7)])
def test_ransac_localization_matches_oracle(vo, ctx, oracle, n, seed, ratio):
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(n, inlier_ratio=ratio, seed=seed)
    samples = vo.synth.sample_triples(n, 3000, seed=seed)
    want = oracle.ransac_localization(kp_rc.T, X.T, vo.synth.K_KITTI, samples)
    R, t, mask, mh, ih = vo.ransacLocalization(kp_rc, X, vo.synth.K_KITTI, samples=samples, ctx=ctx)
    assert want["found"] and want["n_iter"] > 0
    assert len(mh) == want["n_iter"]
    np.testing.assert_array_equal(mh, want["max_num_inliers_history"])
    np.testing.assert_array_equal(ih, want["num_iteration_history"])
    e = _sq_errors(kp_rc, X, vo.synth.K_KITTI, want["R_C_W"], want["t_C_W"])
    diff = mask[0].astype(bool) != want["mask"].astype(bool)
    assert np.all(np.abs(e[diff] - 100.0) < TIE_BAND)
    assert diff.sum() == 0
    np.testing.assert_allclose(R, want["R_C_W"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(t[:, 0], want["t_C_W"], rtol=1e-9, atol=1e-9)


def test_ransac_golden_end_to_end(vo, ctx):
    g = np.load(os.path.join(GOLD, "ransac_n2000_h256.npz"))
    R, t, mask, mh, ih = vo.ransacLocalization(g["kp_rc"], g["X"], g["K"], samples=g["samples"], ctx=ctx)
    assert len(mh) == int(g["n_iter"])
    np.testing.assert_array_equal(mask[0].astype(np.uint8), g["mask"])
    np.testing.assert_array_equal(mh, g["max_num_inliers_history"])
    np.testing.assert_array_equal(ih, g["num_iteration_history"])
    np.testing.assert_allclose(R, g["R_C_W"], rtol=0, atol=1e-9)


def test_no_model_returns_empty_like_reference(vo, ctx):
    rng = np.random.default_rng(5)
    n = 200
    kp_rc = np.stack([rng.uniform(0, 376, n), rng.uniform(0, 1241, n)])
    X = rng.uniform(-20, 20, (3, n)) + np.array([[0], [0], [40.0]])
    # enough triples to reach the 14999-iteration cap: pure outliers never give 30 inliers
    samples = vo.synth.sample_triples(n, 14999, seed=2)
    R, t, mask, mh, ih = vo.ransacLocalization(kp_rc, X, vo.synth.K_KITTI, samples=samples, ctx=ctx)
    assert R.size == 0 and t.size == 0 and mask.shape == (1, n) and mask.sum() == 0
    assert len(mh) == 14999 and np.all(ih == 15000.0)
    # too few triples for the adaptive stop: explicit status instead of silently truncating
    with pytest.raises(vo.VohipError) as ei:
        vo.ransacLocalization(kp_rc, X, vo.synth.K_KITTI, samples=samples[:150], ctx=ctx)
    assert ei.value.status == vo._lib.VOHIP_SAMPLES_EXHAUSTED
    # fewer than 3 correspondences
    with pytest.raises(vo.VohipError) as ei:
        vo.ransacLocalization(kp_rc[:, :2], X[:, :2], vo.synth.K_KITTI, samples=samples[:150] * 0, ctx=ctx)
    assert ei.value.status == vo._lib.VOHIP_TOO_FEW_POINTS


def test_degenerate_samples_score_zero(vo, ctx, oracle):
    """Collinear world points -> p3p returns zeros -> projection NaN -> 0 inliers (p3p.m:56-58)."""
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(100, seed=9)
    X[:, 1] = X[:, 0] + np.array([1.0, 0, 0])
    X[:, 2] = X[:, 0] + np.array([2.0, 0, 0])
    samples = np.array([[0, 1, 2], [2, 1, 0], [5, 6, 7]], np.int32)
    counts, Rt, _ = _score_dev(vo, ctx, kp_rc, X, vo.synth.K_KITTI, samples)
    assert np.all(counts[:2] == 0)
    want = oracle.ransac_hypothesis(kp_rc[::-1].T.copy(), X.T, vo.synth.K_KITTI, samples[2])[0]
    assert np.array_equal(counts[2], want)


def test_rng_stream_advances_by_executed_iterations(vo, ctx):
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(1000, seed=21)
    a = np.random.default_rng(77)
    b = np.random.default_rng(77)
    _, _, _, mh, _ = vo.ransacLocalization(kp_rc, X, vo.synth.K_KITTI, rng=a, ctx=ctx)
    vo.draw_triples(b, 1000, len(mh))
    assert a.bit_generator.state == b.bit_generator.state


def test_inlier_mask_kernel_consistent_with_counts(vo, ctx):
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(5000, seed=31)
    samples = vo.synth.sample_triples(5000, 64, seed=31)
    counts, Rt, (d_kp, d_X) = _score_dev(vo, ctx, kp_rc, X, vo.synth.K_KITTI, samples)
    lib = vo._lib.load()
    Kc = vo._lib.colmajor(vo.synth.K_KITTI)
    d_mask = torch.zeros(5000, dtype=torch.uint8, device="cuda:0")
    for h in (0, 7, 63):
        for s in range(4):
            if not np.isfinite(Rt[h, s]).all():
                continue
            pose = np.ascontiguousarray(Rt[h, s])
            ctx.check(lib.vohip_ransac_inlier_mask_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), 5000,
                                                       vo._lib.dptr(Kc), vo._lib.dptr(pose), d_mask.data_ptr()))
            ctx.synchronize()
            assert int(d_mask.sum().item()) == counts[h, s]


def test_full_size_property_permutation_invariance(vo, ctx):
    """At the stress size (8192 points, 4096 hypotheses) the oracle is too slow to run in full;
    the inlier count of a pose does not depend on the order of the points, so scoring a permuted
    copy with remapped sample indices must give identical counts."""
    n, h = 8192, 4096
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(n, seed=41)
    samples = vo.synth.sample_triples(n, h, seed=41)
    c1, _, _ = _score_dev(vo, ctx, kp_rc, X, vo.synth.K_KITTI, samples)
    perm = np.random.default_rng(0).permutation(n)
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)
    c2, _, _ = _score_dev(vo, ctx, kp_rc[:, perm], X[:, perm], vo.synth.K_KITTI, inv[samples].astype(np.int32))
    np.testing.assert_array_equal(c1, c2)
    assert c1.max() > 0.6 * n * 0.7
