"""The benchmark's KITTI-shape frame loop alone (40 frames), for profiling under rocprofv3."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

vo = importlib.import_module("visual_odometry_pipeline_amd")
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    ctx = vo.Context(0, stream=stream.cuda_stream)
    scene, frames, poses = bench.make_sequence(vo, 47)
    sync = lambda: torch.cuda.synchronize()   # noqa: E731
    el, info = bench.bench_frames(vo, torch, ctx, scene, frames, poses, 40, 5, 100, sync, lambda p, s: [])
    print("frames/s %.0f" % (40 / el), info)
