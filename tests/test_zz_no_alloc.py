"""The allocation-free entry point vohip_ransac_p3p_score_noalloc against the entry point it is a
variant of, vohip_ransac_p3p_score_dev, through the C ABI: inlier counts bit-exact (integer work),
on two calls in a row, the second with a smaller input (the static scratch keeps the first call's
contents), and a size beyond the scratch's maximum is rejected with a clear error."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(vo, n, h, seed):
    dev = torch.device("cuda:0")
    kp_rc, X, _, _, _ = vo.synth.pnp_problem(n, seed=seed)
    samples = vo.synth.sample_triples(n, h, seed=seed)
    return (torch.from_numpy(np.ascontiguousarray(kp_rc.T)).to(dev), torch.from_numpy(np.ascontiguousarray(X.T)).to(dev),
            torch.from_numpy(np.ascontiguousarray(samples, dtype=np.int32)).to(dev))


def _counts(vo, ctx, fn, d_kp, d_X, d_s, n, h):
    d_c = torch.full((h, 4), -1, dtype=torch.int32, device="cuda:0")
    Kc = vo._lib.colmajor(vo.synth.K_KITTI)
    torch.cuda.synchronize()
    ctx.check(fn(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, vo._lib.dptr(Kc), d_s.data_ptr(), h, d_c.data_ptr()))
    ctx.synchronize()
    return d_c.cpu().numpy()


def test_noalloc_matches_score_dev_two_calls_in_a_row(vo, ctx):
    lib = vo._lib.load()
    # first call: large (the one-wavefront-per-hypothesis geometry with 16 hypotheses per workgroup);
    # second call: smaller in both points and hypotheses, into the same scratch
    for n, h, seed in ((8192, 4096, 3), (777, 333, 4), (64, 5, 5)):
        d_kp, d_X, d_s = _inputs(vo, n, h, seed)
        want = _counts(vo, ctx, lib.vohip_ransac_p3p_score_dev, d_kp, d_X, d_s, n, h)
        got = _counts(vo, ctx, lib.vohip_ransac_p3p_score_noalloc, d_kp, d_X, d_s, n, h)
        np.testing.assert_array_equal(got, want)
        assert want.max() >= 30, "the fixture must contain real models"


def test_noalloc_rejects_sizes_beyond_the_scratch(vo, ctx):
    lib = vo._lib.load()
    d_kp, d_X, d_s = _inputs(vo, 100, 4097, 6)
    d_c = torch.zeros((4097, 4), dtype=torch.int32, device="cuda:0")
    Kc = vo._lib.colmajor(vo.synth.K_KITTI)
    st = lib.vohip_ransac_p3p_score_noalloc(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), 100, vo._lib.dptr(Kc),
                                            d_s.data_ptr(), 4097, d_c.data_ptr())
    assert st == vo._lib.VOHIP_ERR_INVALID_ARGUMENT
    assert b"4096" in lib.vohip_last_error(ctx.handle)
