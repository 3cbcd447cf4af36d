"""Wall time of relativeCameraPose (pose_estimation_init.m:29-33: essential matrix -> the one of four candidate
poses with the fewest points behind a camera) on exact synthetic pairs, host arrays in and out.
python tools/relative_pose_timing.py [--reps 100]"""
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
    ap.add_argument("--reps", type=int, default=100)
    args = ap.parse_args()
    vo = importlib.import_module("visual_odometry_pipeline_amd")
    ctx = vo.Context(0)
    K = vo.synth.K_KITTI
    rng = np.random.default_rng(3)
    w = np.array([0.01, 0.03, -0.02])
    R = np.asarray(vo.twist2HomogMatrix(np.concatenate([np.zeros(3), w])))[:3, :3]
    t = np.array([0.1, -0.05, 1.0])
    t /= np.linalg.norm(t)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    out = {}
    for n in (64, 540, 2000, 8000):
        X = np.vstack([rng.uniform(-10, 10, n), rng.uniform(-3, 3, n), rng.uniform(6, 40, n)])
        a = K @ X
        b = K @ (R @ X + t[:, None])
        p1, p2 = (a[:2] / a[2]).T.copy(), (b[:2] / b[2]).T.copy()
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(args.reps):
                Ro, loc, frac = vo.relativeCameraPose(E, K, p1, p2, ctx=ctx)
            dt = (time.perf_counter() - t0) / args.reps
        out["n=%d" % n] = dict(us_per_call=1e6 * dt, valid_fraction=frac, max_abs_R_error=float(np.abs(Ro - R).max()),
                               location_error=float(np.abs(loc.ravel() + R.T @ t).max()))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
