"""Does the order of the tracks in a launch matter?  Workgroups are dealt to the 8 XCDs round-robin
(workgroup i runs on XCD i % 8, each with an L2 of its own), so tracks that are neighbours in the
list run on different XCDs and every L2 ends up fetching both pyramids.  Times the tracker kernel for
the same points in different orders: detection (raster) order, shuffled, and orders in which the tracks
with i % 8 == k all lie in one band of the image."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
lib = vo._lib.load()


def time_order(ctx, p0, p1, pts, reps=30):
    dev = torch.device("cuda", 0)
    n = len(pts)
    d_pts = torch.from_numpy(np.ascontiguousarray(pts.T)).to(dev)
    d_out = torch.zeros_like(d_pts)
    d_valid = torch.zeros(n, dtype=torch.uint8, device=dev)

    def run():
        ctx.check(lib.vohip_klt_track_dev(ctx.handle, p0.slot, p1.slot, d_pts.data_ptr(), n, 21, 20, 6.0,
                                          d_out.data_ptr(), d_valid.data_ptr(), None))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        run()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e3, d_out.cpu().numpy(), d_valid.cpu().numpy()


def banded(pts, key):
    """Order in which the tracks with i % 8 == k are the k-th eighth of the points sorted by key."""
    n = len(pts)
    by_key = np.argsort(key, kind="stable")
    m = (n + 7) // 8
    order = []
    for j in range(m):
        for k in range(8):
            q = k * m + j
            if q < n:
                order.append(by_key[q])
    return np.array(order)


stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = vo.Context(0, stream=stream.cuda_stream)
    scene, frames, poses = bench.make_sequence(vo, 4)
    dev = torch.device("cuda", 0)
    d_frames = [torch.from_numpy(f).to(dev) for f in frames[:2]]
    _, kp = vo.genKeypoints(frames[0], "Harris", ["MinQuality", 1e-6], ctx=ctx)
    xy_all = kp.Location.astype(np.float64)
    p0, p1 = vo.Pyramid(d_frames[0], 5, ctx=ctx), vo.Pyramid(d_frames[1], 5, ctx=ctx)
    rng = np.random.default_rng(3)
    for n in (500, 1800, 4084, 8192):
        reps_xy = int(np.ceil(n / len(xy_all)))
        pts = np.concatenate([xy_all + 0.37 * k for k in range(reps_xy)])
        pts = pts[np.sort(rng.permutation(len(pts))[:n])]
        orders = {
            "raster": np.arange(n),
            "shuffled": rng.permutation(n),
            "bands of y": banded(pts, pts[:, 1]),
            "bands of x": banded(pts, pts[:, 0]),
            "4 x 2 cells": banded(pts, np.floor(pts[:, 1] / 188.0) * 4 + np.floor(pts[:, 0] / 311.0)),
        }
        ref = None
        for name, o in orders.items():
            us, out, valid = time_order(ctx, p0, p1, pts[o])
            inv = np.empty(n, dtype=np.int64)
            inv[o] = np.arange(n)
            res = (out[:, inv], valid[inv])
            if ref is None:
                ref = res
            same = np.array_equal(ref[0], res[0]) and np.array_equal(ref[1], res[1])
            print("n %5d  %-12s %7.1f us  identical %s" % (n, name, us, same), flush=True)
