"""Wall time of the two pose branches of ransacLocalization (P3P and DLT, ransacLocalization.m:9) and of
estimatePoseDLT alone on synthetic correspondences (host arrays in and out, as the MATLAB shadows call them).
python tools/dlt_timing.py [--points 1300] [--outliers 0.2] [--reps 50]"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1300)
    ap.add_argument("--outliers", type=float, default=0.2)
    ap.add_argument("--reps", type=int, default=50)
    args = ap.parse_args()
    import torch
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    K = vo.synth.K_KITTI
    rng = np.random.default_rng(5)
    n = args.points
    X = np.vstack([rng.uniform(-10, 10, n), rng.uniform(-3, 3, n), rng.uniform(8, 40, n)])
    w = np.array([0.02, -0.03, 0.01])
    R = np.asarray(vo.twist2HomogMatrix(np.concatenate([np.zeros(3), w])))[:3, :3]
    t = np.array([0.3, -0.1, 0.5])
    p = K @ (R @ X + t[:, None])
    xy = p[:2] / p[2] + rng.normal(0, 0.3, (2, n))
    bad = rng.permutation(n)[: int(args.outliers * n)]
    xy[:, bad] += rng.uniform(-80, 80, (2, len(bad)))
    kp_rc = xy[::-1].copy()
    out = {}
    for name, kw in (("p3p", dict(use_p3p=True)), ("dlt", dict(use_p3p=False))):
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.reps):
                res = vo.ransacLocalization(kp_rc, X, K, rng=np.random.default_rng(k), ctx=ctx, **kw)
            dt = (time.perf_counter() - t0) / args.reps
        out["ransacLocalization_" + name] = dict(ms_per_call=1e3 * dt, inliers=int(np.sum(res[2])), iterations=len(res[3]))
    for m in (6, 100, n, 4 * n):
        Xm = np.tile(X, (1, 4))[:, :m]
        pm = np.tile(p[:2] / p[2], (1, 4))[:, :m]
        for rep in range(3):
            t0 = time.perf_counter()
            for k in range(args.reps):
                M = vo.estimatePoseDLT(pm.T, Xm.T, K, ctx=ctx)
            dt = (time.perf_counter() - t0) / args.reps
        out["estimatePoseDLT_n%d" % m] = dict(ms_per_call=1e3 * dt, max_abs_R_error=float(np.abs(np.asarray(M)[:3, :3] - R).max()))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
