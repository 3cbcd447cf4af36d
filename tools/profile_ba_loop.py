"""cProfile of the device-resident loop with bundle adjustment (main_BA.m's window): where the host time goes."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
scene, frames, poses = bench.make_sequence(vo, 47)
d_frames = [torch.from_numpy(f).cuda() for f in frames]
X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
hp = vo.default_hyper_paras("kitti", is_BA=True)
hp.r_discard_redundant = bench.KITTI_EXCLUSION_RADIUS
pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp, rng=np.random.default_rng(1),
                           ctx=ctx, window_size=5, keyframe_d=1)
for i in range(1, 12):
    pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(12, 45):
    pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
pr.disable()
print("ms/frame", 1e3 * (time.perf_counter() - t0) / 33, "BA runs", pipe.ba_runs, "landmarks in window", pipe.B.m)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
