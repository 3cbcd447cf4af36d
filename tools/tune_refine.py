"""Times pose_refine_kernel against the iteration cap (per-iteration cost of the in-kernel LM)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vo = importlib.import_module("visual_odometry_pipeline_amd")
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = vo.Context(0, stream=stream.cuda_stream)
    lib = vo._lib.load()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
    kp_rc, X, R, t, is_in = vo.synth.pnp_problem(n, inlier_ratio=1.0, noise_px=0.3, seed=3)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    T_W_C = np.linalg.inv(T)
    Tp = T_W_C.copy(); Tp[:3, 3] += np.array([0.05, -0.02, 0.08])
    d_kp = torch.from_numpy(np.ascontiguousarray(kp_rc[::-1].T)).cuda()
    d_X = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    d_stats = torch.zeros(8, dtype=torch.float64).cuda()
    Kp = vo._lib.dptr(vo._lib.colmajor(vo.synth.K_KITTI))
    for max_iter in (1, 2, 3, 5, 100):
        Ts = torch.from_numpy(np.ascontiguousarray(Tp.T)).cuda()
        d_T = Ts.clone()
        def run():
            d_T.copy_(Ts)
            ctx.check(lib.vohip_pose_refine_dev(ctx.handle, d_T.data_ptr(), d_kp.data_ptr(), d_X.data_ptr(), n, Kp, max_iter, 1e-10, 1e-10, d_stats.data_ptr()))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        print("n=%d max_iter=%3d: %.1f us per call (incl. a %d-byte copy), iterations %d cost %.3f -> %.3f" % (n, max_iter, 1e3 * e0.elapsed_time(e1) / 50, 128, d_stats[0].item(), d_stats[1].item(), d_stats[2].item()), flush=True)
