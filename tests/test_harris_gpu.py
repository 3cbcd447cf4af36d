"""Parity of the HIP corner detector with the CPU oracle through genKeypoints and the C ABI.
Bar: bit-exact (same fp32 formula evaluated in the same order, index work)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_harris_matches_golden(vo, ctx):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    feats, kp = vo.genKeypoints(g["img"], "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert feats is None and kp.Location.dtype == np.float32
    assert kp.Count == len(g["xy"])
    np.testing.assert_array_equal(kp.Location, g["xy"])
    np.testing.assert_array_equal(kp.Metric, g["metric"])
    _, kp = vo.genKeypoints(g["img"], "MinEigen", {"MinQuality": 1e-2}, ctx=ctx)
    np.testing.assert_array_equal(kp.Location, g["xy_mineigen"])
    np.testing.assert_array_equal(kp.Metric, g["metric_mineigen"])


@pytest.mark.parametrize("shape,seed", [((376, 1241), 1), ((97, 131), 2), ((64, 64), 3), ((600, 800), 4)])
def test_harris_matches_oracle(vo, ctx, oracle, shape, seed):
    img = vo.synth.to_uint8(vo.synth.textured_image(shape[0], shape[1], seed=seed))
    ab, metric, total = oracle.detect_corners(img, min_quality=2e-4, border=16)
    _, kp = vo.genKeypoints(img, "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert kp.Count == total
    np.testing.assert_array_equal(kp.Location, ab + 1.0)
    np.testing.assert_array_equal(kp.Metric, metric)
    # MATLAB storage: the column-major image is the transposed problem -> (x, y) swapped roles
    abT, metricT, totalT = oracle.detect_corners(np.ascontiguousarray(img.T), min_quality=2e-4, border=16)
    _, kpF = vo.genKeypoints(np.asfortranarray(img), "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert kpF.Count == totalT
    np.testing.assert_array_equal(kpF.Location, abT[:, ::-1] + 1.0)
    # column-major order = MATLAB's find() order: x ascending, then y
    xi, yi = np.rint(kpF.Location[:, 0]), np.rint(kpF.Location[:, 1])
    assert np.all(np.diff(xi * 10000 + yi) > 0)


def test_capacity_flat_and_errors(vo, ctx):
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    with pytest.raises(vo.VohipError) as ei:
        vo.genKeypoints(g["img"], "Harris", ["MinQuality", 2e-4, "Capacity", 10], ctx=ctx)
    assert ei.value.status == vo._lib.VOHIP_CAPACITY_EXCEEDED
    _, kp = vo.genKeypoints(np.full((64, 64), 7, np.uint8), "Harris", ["MinQuality", 2e-4], ctx=ctx)
    assert kp.Count == 0 and kp.Location.shape == (0, 2)
    with pytest.raises(NotImplementedError):
        vo.genKeypoints(g["img"], "ORB", ["ScaleFactor", 1.2], ctx=ctx)
    with pytest.raises(ValueError):
        vo.genKeypoints(g["img"], "Harris", ["Bogus", 1], ctx=ctx)


@pytest.mark.parametrize("shape,seed", [((376, 1241), 1), ((97, 131), 2), ((64, 64), 3), ((600, 800), 4)])
def test_fast_matches_oracle(vo, ctx, oracle, shape, seed):
    """genKeypoints(image, 'FAST', {'MinContrast', 0.11}) (main_eval.m:172-173) against the CPU statement of
    the segment test.  Bar: bit-exact (integer arithmetic), both storage orders, host and device images."""
    import torch
    img = vo.synth.to_uint8(vo.synth.textured_image(shape[0], shape[1], seed=seed))
    for contrast, quality in ((0.11, 0.1), (0.04, 0.3)):
        ab, metric, total = oracle.detect_fast(img, min_contrast=contrast, min_quality=quality, border=16)
        assert total > 10
        opts = ["MinContrast", contrast, "MinQuality", quality]
        _, kp = vo.genKeypoints(img, "FAST", opts, ctx=ctx)
        assert kp.Count == total
        np.testing.assert_array_equal(kp.Location, ab + 1.0)
        np.testing.assert_array_equal(kp.Metric, metric)
        _, kpd = vo.genKeypoints(torch.from_numpy(img).cuda(), "FAST", opts, ctx=ctx)
        np.testing.assert_array_equal(kpd.Location, kp.Location)
        np.testing.assert_array_equal(kpd.Metric, kp.Metric)
    abT, metricT, totalT = oracle.detect_fast(np.ascontiguousarray(img.T), min_contrast=0.11, min_quality=0.1, border=16)
    _, kpF = vo.genKeypoints(np.asfortranarray(img), "FAST", ["MinContrast", 0.11], ctx=ctx)
    assert kpF.Count == totalT
    np.testing.assert_array_equal(kpF.Location, abT[:, ::-1] + 1.0)
    np.testing.assert_array_equal(kpF.Metric, metricT)


def test_fast_edge_cases(vo, ctx, oracle):
    # noise: nearly every pixel passes the segment test somewhere; flat: nothing does; detections near the
    # image edge (border 3 = the circle's own reach)
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (70, 90), dtype=np.uint8)
    ab, metric, total = oracle.detect_fast(noise, min_contrast=0.05, min_quality=0.05, border=3)
    _, kp = vo.genKeypoints(noise, "FAST", ["MinContrast", 0.05, "MinQuality", 0.05, "Border", 3], ctx=ctx)
    assert kp.Count == total and total > 50
    np.testing.assert_array_equal(kp.Location, ab + 1.0)
    np.testing.assert_array_equal(kp.Metric, metric)
    assert kp.Location.min() >= 4.0
    _, kp = vo.genKeypoints(np.full((64, 64), 7, np.uint8), "FAST", None, ctx=ctx)
    assert kp.Count == 0 and kp.Location.shape == (0, 2)
    with pytest.raises(vo.VohipError):
        vo.genKeypoints(noise, "FAST", ["MinContrast", 1.5], ctx=ctx)
    with pytest.raises(ValueError):
        vo.genKeypoints(noise, "FAST", ["FilterSize", 5], ctx=ctx)
    # the two detectors share scratch memory and the alternating maximum slots: interleaved calls stay exact
    g = np.load(os.path.join(GOLD, "harris_320x240.npz"))
    for _ in range(3):
        _, kh = vo.genKeypoints(g["img"], "Harris", ["MinQuality", 2e-4], ctx=ctx)
        np.testing.assert_array_equal(kh.Location, g["xy"])
        abf, mf, _ = oracle.detect_fast(g["img"], min_contrast=0.11, min_quality=0.1, border=16)
        _, kf = vo.genKeypoints(g["img"], "FAST", ["MinContrast", 0.11], ctx=ctx)
        np.testing.assert_array_equal(kf.Location, abf + 1.0)
        np.testing.assert_array_equal(kf.Metric, mf)


@pytest.mark.parametrize("method", [0, 1])
def test_batched_detection_and_pyramids_equal_single_calls(vo, ctx, method):
    """vohip_detect_corners_batch_dev / vohip_pyramid_batch_dev over a stored sequence (image after image in
    device memory): every image's corner list, count and pyramid levels are those of the single-image calls,
    bit for bit."""
    import ctypes as C
    import torch
    lib = vo._lib.load()
    scene = vo.synth.CorridorScene(seed=11, width=640, height=232)
    poses = vo.synth.corridor_trajectory(5, step=0.7)
    frames = np.stack([scene.render(T) for T in poses])
    n, rows, cols = frames.shape
    dev = torch.device("cuda", 0)
    d_imgs = torch.from_numpy(frames).to(dev)
    cap = 4096
    d_xy = torch.zeros((n, 2, cap), dtype=torch.float32, device=dev)
    d_metric = torch.zeros((n, cap), dtype=torch.float32, device=dev)
    d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
    stream = C.c_void_p(lib.vohip_stream(ctx.handle))
    torch.cuda.synchronize()
    for _ in range(2):   # (a second batch on the same scratch)
        ctx.check(lib.vohip_detect_corners_batch_dev(ctx.handle, d_imgs.data_ptr(), n, rows, cols, 0, 2e-4, 5, 16, method,
                                                     d_xy.data_ptr(), d_metric.data_ptr(), cap, d_counts.data_ptr()))
    ctx.synchronize()
    counts = d_counts.cpu().numpy()
    assert counts.min() > 200
    s_xy = torch.zeros((2, cap), dtype=torch.float32, device=dev)
    s_metric = torch.zeros(cap, dtype=torch.float32, device=dev)
    s_count = torch.zeros(1, dtype=torch.int32, device=dev)
    for k in range(n):
        ctx.check(lib.vohip_detect_corners_dev(ctx.handle, d_imgs[k].data_ptr(), rows, cols, 0, 2e-4, 5, 16, method,
                                               s_xy.data_ptr(), s_metric.data_ptr(), cap, s_count.data_ptr()))
        ctx.synchronize()
        c = int(s_count.item())
        assert c == counts[k]
        assert torch.equal(s_xy[:, :c], d_xy[k][:, :c]) and torch.equal(s_metric[:c], d_metric[k][:c])
    # pyramids
    levels = 5
    per = lib.vohip_pyramid_floats(rows, cols, 0, levels)
    d_pyr = torch.zeros((n, per), dtype=torch.float32, device=dev)
    ctx.check(lib.vohip_pyramid_batch_dev(ctx.handle, d_imgs.data_ptr(), n, rows, cols, 0, levels, d_pyr.data_ptr()))
    ctx.synchronize()
    got = d_pyr.cpu().numpy()
    for k in (0, n - 1):
        pyr = vo.Pyramid(d_imgs[k], levels, ctx=ctx)
        off, A, B = 0, cols, rows
        for l in range(levels):
            want = pyr.level(l)
            assert want.shape == (B, A)
            np.testing.assert_array_equal(got[k, off:off + A * B].reshape(B, A), want)
            off += (A * B + 63) // 64 * 64
            A, B = (A + 1) // 2, (B + 1) // 2
        assert off == per
        pyr.release()
    del stream


def test_batched_fast_equals_single_calls(vo, ctx):
    """vohip_detect_fast_batch_dev over a stored sequence: every image's list is that of a single call, bit for bit;
    the corner batch call takes Harris and MinEigen only."""
    import torch
    lib = vo._lib.load()
    scene = vo.synth.CorridorScene(seed=12, width=640, height=232)
    poses = vo.synth.corridor_trajectory(5, step=0.7)
    frames = np.stack([scene.render(T) for T in poses])
    n, rows, cols = frames.shape
    dev = torch.device("cuda", 0)
    d_imgs = torch.from_numpy(frames).to(dev)
    cap = 4096
    d_xy = torch.zeros((n, 2, cap), dtype=torch.float32, device=dev)
    d_metric = torch.zeros((n, cap), dtype=torch.float32, device=dev)
    d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(2):   # (a second batch on the same scratch)
        ctx.check(lib.vohip_detect_fast_batch_dev(ctx.handle, d_imgs.data_ptr(), n, rows, cols, 0, 0.11, 0.1, 16,
                                                  d_xy.data_ptr(), d_metric.data_ptr(), cap, d_counts.data_ptr()))
    ctx.synchronize()
    counts = d_counts.cpu().numpy()
    assert counts.min() > 100
    s_xy = torch.zeros((2, cap), dtype=torch.float32, device=dev)
    s_metric = torch.zeros(cap, dtype=torch.float32, device=dev)
    s_count = torch.zeros(1, dtype=torch.int32, device=dev)
    for k in range(n):
        ctx.check(lib.vohip_detect_fast_dev(ctx.handle, d_imgs[k].data_ptr(), rows, cols, 0, 0.11, 0.1, 16, s_xy.data_ptr(),
                                            s_metric.data_ptr(), cap, s_count.data_ptr()))
        ctx.synchronize()
        c = int(s_count.item())
        assert c == counts[k]
        assert torch.equal(s_xy[:, :c], d_xy[k][:, :c]) and torch.equal(s_metric[:c], d_metric[k][:c])
    for bad_method in (2, 3, 4):
        assert lib.vohip_detect_corners_batch_dev(ctx.handle, d_imgs.data_ptr(), n, rows, cols, 0, 2e-4, 5, 16, bad_method,
                                                  d_xy.data_ptr(), d_metric.data_ptr(), cap, d_counts.data_ptr()) != 0
    assert lib.vohip_detect_fast_batch_dev(ctx.handle, d_imgs.data_ptr(), n, rows, cols, 0, 0.0, 0.1, 16, d_xy.data_ptr(),
                                           d_metric.data_ptr(), cap, d_counts.data_ptr()) != 0
