"""Parity of the HIP BRISK detector with the CPU oracle through genKeypoints and the C ABI.
Bar: bit-exact (integer pyramids, scores and tests; the refinement in fp32 / fp64 evaluated in the same order)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,seed", [((376, 1241), 1), ((97, 131), 2), ((64, 64), 3), ((600, 800), 4), ((233, 517), 5)])
def test_brisk_matches_oracle(vo, ctx, oracle, shape, seed):
    """genKeypoints(image, 'BRISK', {'MinQuality', q}) (genKeypoints.m:20; main_eval.m:170-171), host and device
    images, both storage orders."""
    import torch
    img = vo.synth.to_uint8(vo.synth.textured_image(shape[0], shape[1], seed=seed))
    for opts, (c, q, n) in ((["MinQuality", 2e-3], (0.2, 2e-3, 4)), (["MinContrast", 0.08, "MinQuality", 0.4], (0.08, 0.4, 4)),
                            (["MinContrast", 0.1, "NumOctaves", 2], (0.1, 0.1, 2)), (["NumOctaves", 0], (0.2, 0.1, 0))):
        ab, metric, scale, total = oracle.detect_brisk(img, c, q, n, 16)
        _, kp = vo.genKeypoints(img, "BRISK", opts, ctx=ctx)
        assert isinstance(kp, vo.BRISKPoints) and kp.Location.dtype == np.float32
        assert kp.Count == total
        np.testing.assert_array_equal(kp.Location, ab + 1.0)
        np.testing.assert_array_equal(kp.Metric, metric)
        np.testing.assert_array_equal(kp.Scale, scale)
        _, kpd = vo.genKeypoints(torch.from_numpy(img).cuda(), "BRISK", opts, ctx=ctx)
        np.testing.assert_array_equal(kpd.Location, kp.Location)
        np.testing.assert_array_equal(kpd.Metric, kp.Metric)
        np.testing.assert_array_equal(kpd.Scale, kp.Scale)
    if min(shape) > 90:
        assert oracle.detect_brisk(img, 0.2, 2e-3, 4, 16)[3] > 10
    # MATLAB storage: the column-major image is the transposed problem -> (x, y) exchange their roles
    abT, metricT, scaleT, totalT = oracle.detect_brisk(np.ascontiguousarray(img.T), 0.2, 0.1, 4, 16)
    _, kpF = vo.genKeypoints(np.asfortranarray(img), "BRISK", None, ctx=ctx)
    assert kpF.Count == totalT
    np.testing.assert_array_equal(kpF.Location, abT[:, ::-1] + 1.0)
    np.testing.assert_array_equal(kpF.Metric, metricT)
    np.testing.assert_array_equal(kpF.Scale, scaleT)


def test_brisk_edge_cases(vo, ctx, oracle):
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, (90, 140), dtype=np.uint8)
    ab, metric, scale, total = oracle.detect_brisk(noise, 0.05, 0.05, 4, 1)
    _, kp = vo.genKeypoints(noise, "BRISK", ["MinContrast", 0.05, "MinQuality", 0.05, "Border", 1], ctx=ctx)
    assert kp.Count == total and total > 50
    np.testing.assert_array_equal(kp.Location, ab + 1.0)
    np.testing.assert_array_equal(kp.Metric, metric)
    np.testing.assert_array_equal(kp.Scale, scale)
    assert len(np.unique(np.round(kp.Scale))) >= 3         # keypoints of several layers
    # a flat image, an image whose coarse layers are smaller than the FAST circle, capacity, bad options
    _, kp = vo.genKeypoints(np.full((64, 64), 7, np.uint8), "BRISK", None, ctx=ctx)
    assert kp.Count == 0 and kp.Location.shape == (0, 2)
    small = np.ascontiguousarray(noise[:20, :26])
    ab, metric, scale, total = oracle.detect_brisk(small, 0.05, 0.05, 4, 1)
    _, kp = vo.genKeypoints(small, "BRISK", ["MinContrast", 0.05, "MinQuality", 0.05, "Border", 1], ctx=ctx)
    assert kp.Count == total
    np.testing.assert_array_equal(kp.Location, ab + 1.0)
    with pytest.raises(vo.VohipError) as ei:
        vo.genKeypoints(noise, "BRISK", ["MinContrast", 0.05, "Border", 1, "Capacity", 10], ctx=ctx)
    assert ei.value.status == vo._lib.VOHIP_CAPACITY_EXCEEDED
    with pytest.raises(vo.VohipError):
        vo.genKeypoints(noise, "BRISK", ["NumOctaves", 9], ctx=ctx)
    with pytest.raises(ValueError):
        vo.genKeypoints(noise, "BRISK", ["MetricThreshold", 10.0], ctx=ctx)
    with pytest.raises(NotImplementedError):
        vo.genKeypoints(noise, "BRISK", None, ctx=ctx, with_features=True)     # (binary BRISK descriptors are not built)
    with pytest.raises(NotImplementedError):
        vo.genKeypoints(noise, "ORB", None, ctx=ctx)


def test_batched_brisk_equals_single_calls(vo, ctx):
    """vohip_detect_brisk_batch_dev over a stored sequence: every image's list is that of a single call."""
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
        ctx.check(lib.vohip_detect_brisk_batch_dev(ctx.handle, d_imgs.data_ptr(), n, rows, cols, 0, 0.2, 2e-3, 4, 16,
                                                   d_xy.data_ptr(), d_metric.data_ptr(), cap, d_counts.data_ptr()))
    ctx.synchronize()
    counts = d_counts.cpu().numpy()
    assert counts.min() > 20
    s_xy = torch.zeros((2, cap), dtype=torch.float32, device=dev)
    s_metric = torch.zeros(cap, dtype=torch.float32, device=dev)
    s_count = torch.zeros(1, dtype=torch.int32, device=dev)
    for k in range(n):
        ctx.check(lib.vohip_detect_brisk_dev(ctx.handle, d_imgs[k].data_ptr(), rows, cols, 0, 0.2, 2e-3, 4, 16, s_xy.data_ptr(),
                                             s_metric.data_ptr(), None, cap, s_count.data_ptr()))
        ctx.synchronize()
        c = int(s_count.item())
        assert c == counts[k]
        assert torch.equal(s_xy[:, :c], d_xy[k][:, :c]) and torch.equal(s_metric[:c], d_metric[k][:c])
