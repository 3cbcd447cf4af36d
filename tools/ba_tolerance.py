"""The loop with bundle adjustment under two stopping tolerances of runBA: iterations per window, time per frame and
how far the trajectories differ."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
pl = importlib.import_module("visual_odometry_pipeline_amd.pipeline")
ctx = vo.Context(0)
scene, frames, poses = bench.make_sequence(vo, 47)
d_frames = [torch.from_numpy(f).cuda() for f in frames]
X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
real = vo.ba.runBA
out = {}
for tol in (1e-10, 1e-6):
    its = []

    def run(state, Obs, K, ctx=None, tol=tol, its=its):
        s, info = real(state, Obs, K, ftol=tol, xtol=tol, return_stats=True, ctx=ctx)
        its.append(info["iterations"])
        return s

    pl.runBA = run
    hp = vo.default_hyper_paras("kitti", is_BA=True)
    hp.r_discard_redundant = bench.KITTI_EXCLUSION_RADIUS
    pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp,
                               rng=np.random.default_rng(1), ctx=ctx, window_size=5, keyframe_d=1)
    traj = []
    t0 = time.perf_counter()
    for i in range(1, 45):
        pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
    dt = time.perf_counter() - t0
    out[tol] = (its, dt, pipe.est_trans)
    print("tol", tol, "ms/frame", 1e3 * dt / 44, "iterations per window", its)
a, b = out[1e-10][2], out[1e-6][2]
print("max trajectory difference", max(np.abs(x - y).max() for x, y in zip(a, b)))
gt = [np.asarray(p)[:3, 3] for p in poses[:45]]   # est_trans starts with the first frame's pose
for tol in out:
    e = np.array([np.linalg.norm(x - g) for x, g in zip(out[tol][2], gt)])
    print("tol", tol, "position error vs the rendered trajectory: rms", np.sqrt((e ** 2).mean()), "last", e[-1])
