import importlib, os, sys, ctypes as C
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
    n = int(sys.argv[1])
    pts = np.stack([rng.uniform(20, 1220, n), rng.uniform(20, 356, n)])
    d_pts = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    d_out = torch.zeros_like(d_pts); d_valid = torch.zeros(n, dtype=torch.uint8, device="cuda")
    p0 = vo.Pyramid(frames[0], 5, ctx=ctx); p1 = vo.Pyramid(frames[1], 5, ctx=ctx)
    ctx.set_option("klt_stats", 1)
    tr, it = C.c_longlong(0), C.c_longlong(0)
    def run():
        ctx.check(lib.vohip_klt_track_dev(ctx.handle, p0.slot, p1.slot, d_pts.data_ptr(), n, 21, 20, 6.0, d_out.data_ptr(), d_valid.data_ptr(), None))
    for _ in range(5): run()
    torch.cuda.synchronize()
    ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(s); run(); e1.record(s); torch.cuda.synchronize()
    print("kernel %.1f us" % (e0.elapsed_time(e1) * 1e3))
    ctx.check(lib.vohip_klt_stats(ctx.handle, C.byref(tr), C.byref(it), 1))
