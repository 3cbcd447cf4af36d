"""vohip_window_* (the struct B of main_demo.m:288-303 and VO_bundle_adjust.m:1-98 kept by the library)
through the C ABI, on host arrays, against the host mirror pipeline.VO_bundle_adjust: the stage-by-stage
loop runs with bundle adjustment while every call of VO_bundle_adjust is replayed through
vohip_window_step with the same inputs.  Both use the same GPU solvers, so the bar is tight: identical
landmark columns and ids, landmarks and rewritten poses within 1e-12."""
import ctypes as C
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make_window(vo, ctx, K, hp, window_size, keyframe_d, X0, xy, T0):
    lib = vo._lib.load()
    wp = vo._lib.WindowParams()
    lib.vohip_window_default_params(C.byref(wp))
    assert (wp.window_size, wp.keyframe_d, wp.ba_max_iter, wp.refine_max_iter) == (5, 1, 21, 100)   # main_BA.m:197-198
    wp.window_size, wp.keyframe_d = window_size, keyframe_d
    wp.min_depth, wp.max_depth = hp.min_depth, hp.max_depth
    w = C.c_void_p()
    ctx.check(lib.vohip_window_create(ctx.handle, vo._lib.dptr(vo._lib.colmajor(K)), C.byref(wp), C.byref(w)))
    X = np.ascontiguousarray(X0, dtype=np.float64)
    P_rc = np.ascontiguousarray(xy[:, ::-1], dtype=np.float64)
    ctx.check(lib.vohip_window_reset(w, vo._lib.dptr(X), vo._lib.dptr(P_rc), len(X), vo._lib.dptr(vo._lib.colmajor(T0))))
    return w


def test_window_step_replays_VO_bundle_adjust(vo, ctx, oracle, monkeypatch):
    lib = vo._lib.load()
    dptr, colmajor = vo._lib.dptr, vo._lib.colmajor
    K, W, H = vo.synth.K_PARKING, vo.synth.PARKING_W, vo.synth.PARKING_H
    scene = vo.synth.CorridorScene(K=K, width=W, height=H, half_width=5.0, roof=4.0)
    poses = vo.synth.corridor_trajectory(14, step=0.3)
    frames = [scene.render(T) for T in poses]
    ab, _, _ = oracle.detect_corners(frames[0], min_quality=2e-4, border=16)
    xy = (ab + 1.0).astype(np.float64)
    X0 = scene.backproject(poses[0], xy)
    hp = vo.default_hyper_paras("parking", is_BA=True)
    window_size, keyframe_d = 3, 1
    w = _make_window(vo, ctx, K, hp, window_size, keyframe_d, X0, xy, poses[0])
    cap = window_size * (keyframe_d + 1)
    offsets, out_poses = (C.c_int * cap)(), np.zeros((cap, 16))
    pl = sys.modules[vo.VOPipeline.__module__]
    mirror = pl.VO_bundle_adjust
    seen = dict(runs=0, frames=0)

    def both(S, B, T_W_C, K_, hyper_paras, ctx=None):
        P = np.ascontiguousarray(S.P.T)
        X = np.ascontiguousarray(S.X[:3].T)
        ids = np.ascontiguousarray(S.X[3])
        et, er = S.est_trans.copy(), S.est_rot.copy()
        S, B = mirror(S, B, T_W_C, K_, hyper_paras, ctx=ctx)
        opt, npose = C.c_int(0), C.c_int(0)
        ctx.check(lib.vohip_window_step(w, dptr(P), dptr(X), dptr(ids), len(ids), dptr(colmajor(T_W_C)), C.byref(opt),
                                        C.byref(npose), offsets, dptr(out_poses), cap))
        seen["frames"] += 1
        np.testing.assert_allclose(X.T, S.X[:3], rtol=1e-12, atol=1e-12)
        nk, m = C.c_int(0), C.c_int(0)
        ctx.check(lib.vohip_window_state(w, C.byref(nk), C.byref(m), None, None))
        assert (nk.value, m.value) == (B.num_key, B.landmarks.shape[1])
        tau, lm = np.zeros((nk.value, 6)), np.zeros((m.value, 4))
        ctx.check(lib.vohip_window_state(w, None, None, dptr(tau), dptr(lm)))
        np.testing.assert_array_equal(lm[:, 3], B.landmarks[3])
        np.testing.assert_allclose(lm[:, :3].T, B.landmarks[:3], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(tau.T, B.tau[:, :nk.value])
        changed = np.flatnonzero(np.any(S.est_trans != et, axis=0) | np.any(S.est_rot != er, axis=0))
        if opt.value:
            seen["runs"] += 1
            assert npose.value == window_size + (window_size - 1) * keyframe_d
            last = S.est_trans.shape[1] - 1
            cols = [last + offsets[k] for k in range(npose.value)]
            assert set(changed) <= set(cols) and len(set(cols)) == npose.value
            for k, col in enumerate(cols):
                T = out_poses[k].reshape(4, 4).T
                np.testing.assert_allclose(T[:3, 3], S.est_trans[:, col], rtol=0, atol=1e-12)
                np.testing.assert_allclose(T[:3, :3].reshape(-1, order="F"), S.est_rot[:, col], rtol=0, atol=1e-12)
        else:
            assert npose.value == 0 and changed.size == 0
        return S, B

    monkeypatch.setattr(pl, "VO_bundle_adjust", both)
    stg = vo.VOPipeline(K, frames[0], X0, xy, poses[0], hyper_paras=hp, window_size=window_size, keyframe_d=keyframe_d,
                        rng=np.random.default_rng(4), ctx=ctx)
    for i in range(1, 13):
        stg.process_frame(frames[i])
    stg.close()
    assert seen["frames"] == 12 and seen["runs"] >= 3
    lib.vohip_window_destroy(w)


def test_window_rejects_bad_input(vo, ctx):
    lib = vo._lib.load()
    dptr, colmajor = vo._lib.dptr, vo._lib.colmajor
    hp = vo.default_hyper_paras("parking", is_BA=True)
    K = vo.synth.K_PARKING
    wp = vo._lib.WindowParams()
    lib.vohip_window_default_params(C.byref(wp))
    w = C.c_void_p()
    wp.window_size = 1
    assert lib.vohip_window_create(ctx.handle, dptr(colmajor(K)), C.byref(wp), C.byref(w)) == vo._lib.VOHIP_ERR_INVALID_ARGUMENT
    X0 = np.array([[0.0, 0.0, 5.0], [1.0, 0.0, 6.0], [0.0, 1.0, 7.0]])
    xy = np.array([[10.0, 20.0], [30.0, 40.0], [50.0, 60.0]])
    w = _make_window(vo, ctx, K, hp, 2, 0, X0, xy, np.eye(4))
    opt, npose = C.c_int(0), C.c_int(0)
    P = np.ascontiguousarray(xy[:, ::-1])
    X = X0.copy()
    T = dptr(colmajor(np.eye(4)))
    # the step before vohip_window_reset is refused; so are ids that do not ascend
    bad = np.array([1.0, 3.0, 2.0])
    assert lib.vohip_window_step(w, dptr(P), dptr(X), dptr(bad), 3, T, C.byref(opt), C.byref(npose), None, None, 0) \
        == vo._lib.VOHIP_ERR_INVALID_ARGUMENT
    assert b"ascend" in lib.vohip_last_error(ctx.handle)
    # keyframe_d = 0: every frame is a keyframe; the second one fills the window of 2 ...
    ids = np.array([1.0, 2.0, 3.0])
    ctx.check(lib.vohip_window_step(w, dptr(P), dptr(X), dptr(ids), 3, T, C.byref(opt), C.byref(npose), None, None, 0))
    assert opt.value == 0
    # ... and the third one optimises it, for which room for the poses is required
    assert lib.vohip_window_step(w, dptr(P), dptr(X), dptr(ids), 3, T, C.byref(opt), C.byref(npose), None, None, 0) \
        == vo._lib.VOHIP_ERR_INVALID_ARGUMENT
    lib.vohip_window_destroy(w)
