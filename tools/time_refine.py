"""Pose refinement alone (T_refinement on device-resident points): kernel time for a few point counts."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
lib = vo._lib.load()
K = vo.synth.K_KITTI
Kc = vo._lib.colmajor(K)
for n in [int(a) for a in sys.argv[1:]] or (300, 1300, 3000, 8000):
    kp_rc, X, R, t, inl = vo.synth.pnp_problem(n, inlier_ratio=1.0, seed=3)
    T_C_W = np.eye(4)
    T_C_W[:3, :3], T_C_W[:3, 3] = R, t
    T0 = np.linalg.inv(T_C_W) @ vo.synth.pose_matrix(np.array([0.01, -0.02, 0.015]), np.array([0.05, -0.03, 0.08]))
    d_kp = torch.from_numpy(np.ascontiguousarray(kp_rc[::-1].T)).cuda()
    d_X = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
    d_stats = torch.zeros(3, dtype=torch.float64).cuda()
    T0c = torch.from_numpy(vo._lib.colmajor(T0).reshape(-1)).cuda()
    d_T = T0c.clone()

    def run():
        d_T.copy_(T0c)
        ctx.check(lib.vohip_pose_refine_dev(ctx.handle, d_T.data_ptr(), d_kp.data_ptr(), d_X.data_ptr(), n, vo._lib.dptr(Kc),
                                            100, 1e-6, 1e-6, d_stats.data_ptr()))
    st = torch.cuda.ExternalStream(lib.vohip_stream(ctx.handle))
    with torch.cuda.stream(st):
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
    want = vo.T_refinement(T0, kp_rc[::-1], X, K, ctx=ctx)
    got = d_T.cpu().numpy().reshape(4, 4).T
    print("n %5d: %.1f us per call (incl. a 128-byte copy), %d LM steps, cost %.3g -> %.3g, |dev - host| %.2e"
          % (n, 1e3 * e0.elapsed_time(e1) / 50, d_stats[0].item(), d_stats[1].item(), d_stats[2].item(), np.abs(got - want).max()))
