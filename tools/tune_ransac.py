"""Times the RANSAC scoring kernel variants (interleaved rounds in one process)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vo = importlib.import_module("visual_odometry_pipeline_amd")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = vo.Context(0, stream=s.cuda_stream)
    lib = vo._lib.load()
    for (n, h) in [(8192, 4096), (2000, 4096), (2000, 104)]:
        kp_rc, X, _, _, _ = vo.synth.pnp_problem(n, seed=1)
        samples = vo.synth.sample_triples(n, h, seed=1)
        d_kp = torch.from_numpy(np.ascontiguousarray(kp_rc.T)).cuda()
        d_X = torch.from_numpy(np.ascontiguousarray(X.T)).cuda()
        d_s = torch.from_numpy(samples).cuda()
        d_c = torch.zeros((h, 4), dtype=torch.int32, device="cuda")
        Kc = vo._lib.colmajor(vo.synth.K_KITTI)
        ctx.check(lib.vohip_ransac_p3p_solve_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, vo._lib.dptr(Kc), d_s.data_ptr(), h))
        res = {}
        ref = None
        for rnd in range(5):
            for v in range(6):
                ctx.set_option("ransac_score_variant", v)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                lib.vohip_ransac_p3p_count_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, h, d_c.data_ptr())
                e0.record(s)
                for _ in range(150):
                    lib.vohip_ransac_p3p_count_dev(ctx.handle, d_kp.data_ptr(), d_X.data_ptr(), n, h, d_c.data_ptr())
                e1.record(s)
                torch.cuda.synchronize()
                res.setdefault(v, []).append(e0.elapsed_time(e1) / 150 * 1e3)
                c = d_c.cpu().numpy()
                if ref is None: ref = c
                assert np.array_equal(ref, c), v
        for v in range(6):
            us = np.median(res[v])
            print("n=%d h=%d variant %d: %.1f us  (min %.1f)  %.1f TF" % (n, h, v, us, min(res[v]), 116.0 * n * h / us / 1e6))
