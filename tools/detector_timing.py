"""Launch sequences of the detectors of genKeypoints.m:16-19 on one KITTI-shape frame resident on the GPU, for a
kernel trace:  rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/detector_timing.py [--reps 200]
Prints the wall time per call of each detector (events on the library's stream) and per frame of the batched calls."""
import argparse
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--batch", type=int, default=64)
    args = ap.parse_args()
    import torch
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    lib = vo._lib.load()
    scene = vo.synth.CorridorScene(seed=3)
    frame = scene.render(vo.synth.corridor_trajectory(2, step=0.7)[1])
    rows, cols = frame.shape
    dev = torch.device("cuda", 0)
    d_frame = torch.from_numpy(frame).to(dev)
    cap = 16384
    d_xy = torch.zeros(2 * cap, dtype=torch.float32, device=dev)
    d_metric = torch.zeros(cap, dtype=torch.float32, device=dev)
    d_count = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.ExternalStream(lib.vohip_stream(ctx.handle))
    calls = {
        "Harris": lambda: lib.vohip_detect_corners_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 2e-4, 5, 16, 0,
                                                       d_xy.data_ptr(), d_metric.data_ptr(), cap, d_count.data_ptr()),
        "FAST": lambda: lib.vohip_detect_fast_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 0.11, 0.1, 16,
                                                  d_xy.data_ptr(), d_metric.data_ptr(), cap, d_count.data_ptr()),
        "BRISK": lambda: lib.vohip_detect_brisk_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 0.2, 2e-3, 4, 16, d_xy.data_ptr(),
                                                    d_metric.data_ptr(), None, cap, d_count.data_ptr()),
        "SURF": lambda: lib.vohip_detect_surf_dev(ctx.handle, d_frame.data_ptr(), rows, cols, 0, 200.0, 16, d_xy.data_ptr(),
                                                  d_metric.data_ptr(), None, cap, d_count.data_ptr()),
    }
    out = {"shape": [rows, cols]}
    for name, fn in calls.items():
        ctx.check(fn())
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.reps):
            ctx.check(fn())
        e1.record(stream)
        ctx.synchronize()
        out[name] = {"us_per_call": 1e3 * e0.elapsed_time(e1) / args.reps, "points": int(d_count.item())}
    nb = args.batch
    d_batch = d_frame.repeat(nb, 1, 1)
    b_xy = torch.zeros((nb, 2 * cap), dtype=torch.float32, device=dev)
    b_metric = torch.zeros((nb, cap), dtype=torch.float32, device=dev)
    b_counts = torch.zeros(nb, dtype=torch.int32, device=dev)
    args_b = (b_xy.data_ptr(), b_metric.data_ptr(), cap, b_counts.data_ptr())
    batch_calls = {
        "Harris_batch": lambda: lib.vohip_detect_corners_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 2e-4, 5, 16,
                                                                   0, *args_b),
        "FAST_batch": lambda: lib.vohip_detect_fast_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 0.11, 0.1, 16,
                                                              *args_b),
        "BRISK_batch": lambda: lib.vohip_detect_brisk_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 0.2, 2e-3, 4,
                                                                16, *args_b),
        "SURF_batch": lambda: lib.vohip_detect_surf_batch_dev(ctx.handle, d_batch.data_ptr(), nb, rows, cols, 0, 200.0, 16,
                                                              *args_b),
    }
    for name, fn in batch_calls.items():
        ctx.check(fn())
        ctx.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(5):
            ctx.check(fn())
        e1.record(stream)
        ctx.synchronize()
        out[name] = {"frames": nb, "us_per_frame": 1e3 * e0.elapsed_time(e1) / 5 / nb}
    # the 'Des_match' bootstrap's two stages on this frame's SURF points (host arrays in and out, as the bootstrap calls them)
    import time
    feats, pts = vo.genKeypoints(frame, "SURF", ["MetricThreshold", 200], ctx=ctx, with_features=True)
    frame2 = scene.render(vo.synth.corridor_trajectory(4, step=0.7)[3])
    feats2, pts2 = vo.genKeypoints(frame2, "SURF", ["MetricThreshold", 200], ctx=ctx, with_features=True)
    for name, fn in (("extractFeatures", lambda: vo.extractFeatures(frame, pts, ctx=ctx)),
                     ("matchFeatures", lambda: vo.matchFeatures(feats, feats2, ctx=ctx))):
        fn()
        t0 = time.perf_counter()
        for _ in range(20):
            r = fn()
        out[name] = {"ms_per_call": 1e3 * (time.perf_counter() - t0) / 20, "points": [pts.Count, pts2.Count],
                     "pairs": int(len(r[0])) if name == "matchFeatures" else None}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
