"""Parity of the HIP SURF detector with the CPU oracle through genKeypoints and the C ABI.
Bar: bit-exact (integer box sums; the fp64 determinant and refinement evaluated in the same order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,seed", [((376, 1241), 1), ((97, 131), 2), ((64, 64), 3), ((600, 800), 4), ((233, 517), 5)])
def test_surf_matches_oracle(vo, ctx, oracle, shape, seed):
    """genKeypoints(image, 'SURF', {'MetricThreshold', t}) (genKeypoints.m:16; main_eval.m:151,167), host and
    device images, both storage orders."""
    import torch
    img = vo.synth.to_uint8(vo.synth.textured_image(shape[0], shape[1], seed=seed))
    for thr in (200.0, 10.0):
        ab, metric, scale, total = oracle.detect_surf(img, thr, 16)
        if min(shape) > 90:
            assert total > 10
        _, kp = vo.genKeypoints(img, "SURF", ["MetricThreshold", thr], ctx=ctx)
        assert isinstance(kp, vo.SURFPoints) and kp.Location.dtype == np.float32
        assert kp.Count == total
        np.testing.assert_array_equal(kp.Location, ab + 1.0)
        np.testing.assert_array_equal(kp.Metric, metric)
        np.testing.assert_array_equal(kp.Scale, scale)
        _, kpd = vo.genKeypoints(torch.from_numpy(img).cuda(), "SURF", ["MetricThreshold", thr], ctx=ctx)
        np.testing.assert_array_equal(kpd.Location, kp.Location)
        np.testing.assert_array_equal(kpd.Metric, kp.Metric)
        np.testing.assert_array_equal(kpd.Scale, kp.Scale)
    # MATLAB storage: the column-major image is the transposed problem -> (x, y) exchange their roles
    abT, metricT, scaleT, totalT = oracle.detect_surf(np.ascontiguousarray(img.T), 200.0, 16)
    _, kpF = vo.genKeypoints(np.asfortranarray(img), "SURF", ["MetricThreshold", 200.0], ctx=ctx)
    assert kpF.Count == totalT
    np.testing.assert_array_equal(kpF.Location, abT[:, ::-1] + 1.0)
    np.testing.assert_array_equal(kpF.Metric, metricT)


def test_surf_response_maps_and_edge_cases(vo, ctx, oracle):
    # blobs of known place and size
    yy, xx = np.mgrid[0:200, 0:300]
    img = np.full((200, 300), 100.0)
    blobs = [(80, 60, 3.0, 120), (200, 120, 5.0, -80), (150, 50, 8.0, 100)]
    for cx, cy, sg, amp in blobs:
        img += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
    img = np.clip(img, 0, 255).astype(np.uint8)
    _, kp = vo.genKeypoints(img, "SURF", None, ctx=ctx)       # the toolbox default MetricThreshold 1000
    _, kp_default = vo.genKeypoints(img, ctx=ctx)             # genKeypoints.m:3-5: the method defaults to SURF
    np.testing.assert_array_equal(kp_default.Location, kp.Location)
    ab, metric, scale, total = oracle.detect_surf(img, 1000.0, 16)
    assert kp.Count == total >= 3
    np.testing.assert_array_equal(kp.Location, ab + 1.0)
    for cx, cy, sg, _ in blobs:
        d = np.hypot(kp.Location[:, 0] - 1 - cx, kp.Location[:, 1] - 1 - cy)
        assert d.min() < 0.5
    # noise at threshold 0 with the smallest border: many points, some next to the margin the filters leave
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (90, 140), dtype=np.uint8)
    ab, metric, scale, total = oracle.detect_surf(noise, 0.0, 1)
    _, kp = vo.genKeypoints(noise, "SURF", ["MetricThreshold", 0.0, "Border", 1], ctx=ctx)
    assert kp.Count == total and total > 50
    np.testing.assert_array_equal(kp.Location, ab + 1.0)
    np.testing.assert_array_equal(kp.Metric, metric)
    np.testing.assert_array_equal(kp.Scale, scale)
    # flat image, an image too small for any filter triple, capacity, bad options
    _, kp = vo.genKeypoints(np.full((64, 64), 7, np.uint8), "SURF", None, ctx=ctx)
    assert kp.Count == 0 and kp.Location.shape == (0, 2)
    _, kp = vo.genKeypoints(np.ascontiguousarray(noise[:20, :20]), "SURF", ["MetricThreshold", 0.0], ctx=ctx)
    assert kp.Count == 0
    with pytest.raises(vo.VohipError) as ei:
        vo.genKeypoints(noise, "SURF", ["MetricThreshold", 0.0, "Border", 1, "Capacity", 10], ctx=ctx)
    assert ei.value.status == vo._lib.VOHIP_CAPACITY_EXCEEDED
    with pytest.raises(vo.VohipError):
        vo.genKeypoints(noise, "SURF", ["MetricThreshold", -1.0], ctx=ctx)
    with pytest.raises(ValueError):
        vo.genKeypoints(noise, "SURF", ["MinQuality", 0.1], ctx=ctx)
    # the other detectors keep their results when SURF runs between their calls
    _, kh = vo.genKeypoints(img, "Harris", ["MinQuality", 2e-4], ctx=ctx)
    vo.genKeypoints(img, "SURF", None, ctx=ctx)
    _, kh2 = vo.genKeypoints(img, "Harris", ["MinQuality", 2e-4], ctx=ctx)
    np.testing.assert_array_equal(kh.Location, kh2.Location)


def test_batched_surf_equals_single_calls(vo, ctx):
    """vohip_detect_surf_batch_dev over a stored sequence: every image's list is that of a single call."""
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
    torch.cuda.synchronize()
    for _ in range(2):   # (a second batch on the same scratch)
        ctx.check(lib.vohip_detect_surf_batch_dev(ctx.handle, d_imgs.data_ptr(), n, rows, cols, 0, 50.0, 16,
                                                  d_xy.data_ptr(), d_metric.data_ptr(), cap, d_counts.data_ptr()))
    ctx.synchronize()
    counts = d_counts.cpu().numpy()
    assert counts.min() > 20
    s_xy = torch.zeros((2, cap), dtype=torch.float32, device=dev)
    s_metric = torch.zeros(cap, dtype=torch.float32, device=dev)
    s_count = torch.zeros(1, dtype=torch.int32, device=dev)
    for k in range(n):
        ctx.check(lib.vohip_detect_surf_dev(ctx.handle, d_imgs[k].data_ptr(), rows, cols, 0, 50.0, 16, s_xy.data_ptr(),
                                            s_metric.data_ptr(), None, cap, s_count.data_ptr()))
        ctx.synchronize()
        c = int(s_count.item())
        assert c == counts[k]
        assert torch.equal(s_xy[:, :c], d_xy[k][:, :c]) and torch.equal(s_metric[:c], d_metric[k][:c])
