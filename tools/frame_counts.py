"""Per-frame counts of the benchmark's KITTI-shape loop (what the trackers and the mapping stage handle)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
ctx = vo.Context(0)
scene, frames, poses = bench.make_sequence(vo, 47)
d_frames = [torch.from_numpy(f).cuda() for f in frames]
X0, xy0 = bench.init_points(vo, ctx, scene, frames[0], poses[0])
hp = vo.default_hyper_paras("kitti")
hp.r_discard_redundant = bench.KITTI_EXCLUSION_RADIUS
pipe = vo.VOPipelineDevice(vo.synth.K_KITTI, d_frames[0], X0, xy0, poses[0], hyper_paras=hp, rng=np.random.default_rng(1), ctx=ctx)
for i in range(1, 45):
    pipe.process_frame(d_frames[i], next_image=d_frames[i + 1])
    c, p = pipe.last_counts, pipe.last_pose
    if i >= 30:
        print("frame %d: landmarks tracked %d inliers %d -> %d (+%d new) | candidates tracked %d F-inliers %d removed %d (dropped %d) left %d | detected %d kept %d"
              % (i, p.n_tracked, p.n_inliers, c.n_landmarks, c.n_landmarks - p.n_inliers, c.n_candidates_tracked,
                 c.n_candidates_inliers, c.num_new, c.num_new - (c.n_landmarks - p.n_inliers), c.n_candidates - c.n_kept,
                 c.n_detected, c.n_kept))
