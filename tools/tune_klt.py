"""Times the KLT kernel for different iteration caps / pyramid depths (where does the time go)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
vo = importlib.import_module("visual_odometry_pipeline_amd")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx = vo.Context(0, stream=s.cuda_stream)
    lib = vo._lib.load()
    frames, poses = vo.synth.corridor_sequence(3, cache_dir="/tmp/vohip_cache")
    rng = np.random.default_rng(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    pts = np.stack([rng.uniform(20, 1220, n), rng.uniform(20, 356, n)])
    d_pts = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    d_out = torch.zeros_like(d_pts); d_valid = torch.zeros(n, dtype=torch.uint8, device="cuda"); d_fb = torch.zeros(n, dtype=torch.float64, device="cuda")
    for levels in (5,):
        p0 = vo.Pyramid(frames[0], levels, ctx=ctx); p1 = vo.Pyramid(frames[1], levels, ctx=ctx)
        for max_iter in (1, 5, 20):
            def run():
                ctx.check(lib.vohip_klt_track_dev(ctx.handle, p0.slot, p1.slot, d_pts.data_ptr(), n, 21, max_iter, 6.0, d_out.data_ptr(), d_valid.data_ptr(), d_fb.data_ptr()))
            run(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(s)
            for _ in range(10): run()
            e1.record(s); torch.cuda.synchronize()
            print("n=%d levels=%d max_iter=%2d: %.1f us  valid %.2f" % (n, levels, max_iter, e0.elapsed_time(e1) / 10 * 1e3, d_valid.float().mean().item()))
        p0.release(); p1.release()
